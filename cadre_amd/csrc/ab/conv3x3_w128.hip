// conv3x3_w128.hip — 3x3 / stride 1 / pad 1 convolution of the bf16 model as a window kernel with a 128-position x 128-channel
// WAVE tile (round 5; VERDICT r4 item 2, the form DESIGN.md 3.3's LDS-bandwidth analysis asks for).
// A/B BUILD ONLY (CADRE_BUILD_AB=1 python -m cadre_amd.build; include/cadre_hip_ab.h): parity-green, and a TIE with the 8-wave
// ping-pong kernel within box-to-box variance — 2048 frames, same box: layer4 0.585 / 0.608 ms (no residual / residual) vs 0.613 /
// 0.621, layer3 0.615 / 0.660 vs 0.611 / 0.631, layer2 0.683 / 0.798 vs 0.662 / 0.719 (profiles/r05_w128_wave_tile_vs_ping_pong.txt).
// Its MFMA-only schedule runs 1.70 PFLOP/s (ping-pong: 1.43) — the board's power limit, not the pipe — and every operand byte
// costs on top of that (weight stream 13 %, pixel fragments 8-10 %, exposed epilogue 5-12 %: profiles/r05_w128_ablation.txt).
// Reference layers: carla_perception/Networks/danet_blocks/resnet.py:26-55 (BasicBlock conv1 / conv2 of layer2 .. layer4),
// danet.py:21-41 (conv5a / conv5c / conv51 / conv52).
//
//   * 4 waves per workgroup, ONE PER SIMD, 512 registers each: 256 accumulators (4 x 4 blocks of 32 x 32) pinned to the AGPR file by
//     inline-asm v_mfma_f32_32x32x16_bf16 — per k-step of 16 input channels a wave issues 16 MFMAs on 4 pixel + 4 weight fragments:
//     half a fragment per MFMA, where the 8-wave ping-pong kernel (64 x 64 wave tiles) needs one;
//   * the PIXEL fragments come from the window in LDS (one 64-channel chunk of 128 MW + 2 W + 2 consecutive pixels, staged by
//     LDS-DMA once per chunk and read at nine tap offsets; two buffers: the next chunk's window lands while this one is read);
//     the WEIGHT fragments never touch LDS: the host stores the weights in fragment order (cadre_amd/encoder.py _w128_w), a wave
//     streams its 128 channels with four 1 KB buffer loads per k-step straight into registers, W1_D k-steps ahead (a ring of
//     W1_D + 1 register sets).  LDS traffic per MFMA: 0.25 KB read (ping-pong kernel: 1 KB) and no weight writes;
//   * no barrier inside a chunk (36 k-steps = 576 MFMAs per wave): the only shared data is the window.  One barrier per chunk, in
//     front of its LAST k-step — behind it every wave's pieces of the next window have landed and nobody reads the current one
//     any more (the last fragments were read a k-step earlier), so the first fragments of the next chunk are read under the last
//     16 MFMAs of this one;
//   * every memory instruction sits BETWEEN MFMAs (two MFMAs, one request: winograd_c64.hip's finding that a burst of requests in
//     front of an MFMA block holds all four waves in the address path); the file is built without the machine scheduler
//     (cadre_amd/build.py EXTRA_FLAGS) so the source order is the issue order;
//   * the folded-BN SHIFT is the accumulators' initial value, put there BY THE MATRIX CORES: one extra k-step per item multiplies a
//     "bias" weight fragment (k elements 0 .. 2 of a channel = the three bf16 pieces hi + mid + lo of its fp32 shift: their sum is
//     the shift exactly) with a constant pixel fragment of ones — 16 MFMAs per item instead of 256 accumulator writes or 256
//     additions in the epilogue.  The folded-BN SCALE is in the weights (scale == NULL is part of the contract);
//   * epilogue without LDS (nothing runs under it — one wave per SIMD — so every instruction counts): lane-half exchange
//     (v_permlane32_swap) leaves a lane with eight consecutive channels of one position -> (+ residual, requested a chunk ahead for
//     the first half of the tile) -> bf16 pairs -> ReLU as v_pk_max_i16 on the pairs (a negative bf16 is a negative int16) -> 16-byte
//     stores.
// Tile shapes: MW = 2: 256 positions x 256 channels per workgroup (N % 256 == 0: layer3 / layer4);
//              MW = 4: 512 positions x 128 channels (N % 128 == 0: layer2, the head convs).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include <utility>
#include "../../../include/cadre_hip_ab.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

int cadre_fail(const char* msg);

#ifndef W1_ABL          // ablation builds (tools/w128_ablate.py): 1 no MFMAs, 2 no pixel-fragment reads, 4 no window DMA, 8 no weight
#define W1_ABL 0        // loads, 16 no stores, 32 no epilogue arithmetic, 64 no residual loads, 128 zero bias (no table read), 256 no bias MFMAs
#endif
#ifndef W1_D
#define W1_D 5          // weight prefetch depth in k-steps (3 or 5: 36 k-steps per chunk must be a multiple of W1_D + 1)
#endif

struct w128_args {
  const void* x;         // [M][Cin] bf16
  const void* w;         // [N/128][Cin/64][9][4 k-steps][4 blocks][64 lanes][8] bf16 (fragment order)
  const float* shift;    // [N] or null
  const void* resid;     // [M][N] bf16 or null
  void* out;             // [M][N] bf16
  int M, H, W, Cin, N, NC;
  int ntiles;            // N / BN
  int items, ipw, act;
};

template <class F, int... I>
__device__ __forceinline__ void w1_static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void w1_static_for(F&& f) { w1_static_for_impl(f, std::make_integer_sequence<int, N>{}); }

template <int N>
__device__ __forceinline__ void w1_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// acc (AGPRs) += W fragment (32 channels x 16 k) x pixel fragment (16 k x 32 positions); w1_mfma0: acc = product
__device__ __forceinline__ void w1_mfma(f32x16& acc, const f32x4& wf, const f32x4& pf) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr ((W1_ABL & 1) != 0) { asm volatile("" : "+a"(acc) : "v"(wf), "v"(pf)); }
  else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(wf), "v"(pf));
#endif
}
__device__ __forceinline__ void w1_mfma0(f32x16& acc, const f32x4& wf, const f32x4& pf) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(acc) : "v"(wf), "v"(pf));
#endif
}

template <int MW, int NPW, bool RES>
__global__ __launch_bounds__(256, 1) void conv3x3_w128_kernel(w128_args a) {
  constexpr int NWN = 4 / MW, BM = 128 * MW, BN = 128 * NWN;
  constexpr int WIN_B = 4 * NPW * 1024;                    // one window buffer: 4 waves x NPW pieces of 8 pixels x 128 B
  constexpr int NSL = W1_D + 1;                            // weight register sets
  static_assert(36 % NSL == 0, "the register set of a k-step must not depend on the chunk");
  constexpr int NRQ = 8;                                   // residual pieces in registers (a quarter of the wave's 32; 16 spill)
  constexpr unsigned OOB = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int wm = wave / NWN, wn = wave % NWN;              // 128-position block, 128-channel block of the workgroup tile
  char* win0 = smem;
  char* zrow = smem + 2 * WIN_B;                           // 1 KB of zeros: the halo taps' row
  float* sh_lds = reinterpret_cast<float*>(zrow + 1024);   // folded-BN shift of every channel
  const int i_begin = blockIdx.x * a.ipw, i_end = min(a.items, i_begin + a.ipw);
  const int nitems = i_end - i_begin;
  if (nitems <= 0) return;
  // (Items take the same time on every CU, so all workgroups reach their epilogues together.  Spreading the items evenly over all
  //  256 CUs and starting the workgroups with the smaller share late - their store bursts then fall into the others' main loops -
  //  gained 1-3 % over the even spread without the delay, but ceil(items / CUs) items on FEWER workgroups was faster than either:
  //  profiles/r05_w128_wave_tile_vs_ping_pong.txt.)
  const int cin_b = a.Cin * 2;
  const int KB = a.NC * 36 * 4096;                         // bytes of one 128-channel group's weight stream
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.M * cin_b, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (a.N / 128) * KB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, 0, 0x00020000);      // zero records: reads zeros
  const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, a.M * a.N * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void*)(RES ? a.resid : a.x), 0, RES ? a.M * a.N * 2 : 0, 0x00020000);
  for (int i = tid; i < 256; i += 256) reinterpret_cast<unsigned*>(zrow)[i] = 0u;
  for (int i = tid; i < a.N; i += 256) sh_lds[i] = a.shift ? a.shift[i] : 0.f;
  auto swz = [](int idx) constexpr -> int { return (idx >> 1) & 7; };

  // ---- window DMA.  Piece j = 4 n + wave: LDS rows 8 j .. 8 j + 7 of the buffer; this lane brings row 8 j + (lane >> 3), LDS chunk
  // (lane & 7) <- source chunk (lane & 7) ^ swz(row), swz(row) = (lane >> 4) ^ 4 (j & 1), j & 1 = wave & 1.  Row r holds position
  // mt * BM - W - 1 + r: before the tensor the offset is negative (as unsigned: past num_records), past its end likewise — zeros.
  // Every piece is requested (the buffers are 4 NPW pieces long; rows past the window are never read).
  const int a_lane = (lane >> 3) * cin_b + ((((lane & 7) ^ (lane >> 4) ^ (4 * (wave & 1)))) << 4);
  const int a_wave = 8 * wave * cin_b;
  auto send_a = [&](int sbase, int buf, int n) __attribute__((always_inline)) {
    // sbase = (mt * BM - W - 1) * cin_b + 128 c of the window's chunk (scalar)
    int al = a_lane;
    asm volatile("" : "+v"(al));                           // (one add per request here instead of NPW hoisted offsets held in registers)
    unsigned voff = (unsigned)((sbase + a_wave + n * 32 * cin_b) + al);
    char* dst = win0 + buf * WIN_B + (4 * n + wave) * 1024;
    if constexpr ((W1_ABL & 4) != 0) { voff = OOB; dst = zrow; }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (__attribute__((address_space(3))) void*)dst, 16, (int)voff, 0, 0, 0);
  };
  // ---- weight stream: k-step q of a group = 4 blocks x 1 KB at byte q * 4096; lane: 16 bytes at lane * 16 of each block
  const int b_lane = lane * 16;
  auto load_b = [&](const __amdgpu_buffer_rsrc_t rs, int soff, int cb) __attribute__((always_inline)) -> f32x4 {
    if constexpr ((W1_ABL & 8) != 0) return f32x4{0.01f * lane, -2.5f + cb, 0.125f, 1.f};
    int bl = b_lane;
    asm volatile("" : "+v"(bl));                           // (the block's 1 KB step goes into the instruction's offset field)
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, bl + cb * 1024, soff, 0));
  };

  // ---- positions of this wave's four 32-position blocks: (row, column) inside the frame, advanced by float-reciprocal carries
  const float inv_w = 1.0f / (float)a.W, inv_h = 1.0f / (float)a.H;
  int ph[4], pw[4];
  int mt = i_begin / a.ntiles, nt = i_begin - mt * a.ntiles;
  {
    const int HW = a.H * a.W;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      const int m = mt * BM + 128 * wm + 32 * mb + l31;
      const int rem = m % HW;
      ph[mb] = rem / a.W;
      pw[mb] = rem - ph[mb] * a.W;
    }
  }
  auto advance_mtile = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      const int x = pw[mb] + BM;
      const int q1 = (int)(((float)x + 0.5f) * inv_w);
      pw[mb] = x - q1 * a.W;
      const int y = ph[mb] + q1;
      const int q2 = (int)(((float)y + 0.5f) * inv_h);
      ph[mb] = y - q2 * a.H;
    }
  };
  auto masks_of = [&](int mt_i, unsigned* mk) __attribute__((always_inline)) {
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      const int m = mt_i * BM + 128 * wm + 32 * mb + l31;
      unsigned colm = 0, v = 0;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) colm |= ((unsigned)(pw[mb] - 1 + kw) < (unsigned)a.W) ? (1u << kw) : 0u;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) v |= ((unsigned)(ph[mb] - 1 + kh) < (unsigned)a.H) ? (colm << (3 * kh)) : 0u;
      mk[mb] = m < a.M ? v : 0u;
    }
  };
  // fragment address (byte offset in LDS) of block mb at tap `tap`, k-step 0, this lane's k half; k-step s: ^ (s << 5)
  const unsigned zrow_off = (unsigned)(zrow - smem);
  const unsigned lhb = (unsigned)lh << 4;
  const int base_idx = 128 * wm + l31;
  auto frag_addr = [&](int tap, unsigned win_off, const unsigned* mk, unsigned* out) __attribute__((always_inline)) {
    const int toff = (tap / 3) * a.W + (tap % 3);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      const int idx = base_idx + 32 * mb + toff;
      unsigned m = mk[mb];
      asm volatile("" : "+v"(m));                          // (the tap's bit is tested here: hoisted, the 36 tests of an item occupy 36 registers)
      const unsigned row = ((m >> tap) & 1u) ? win_off + (unsigned)(idx << 7) : zrow_off;
      out[mb] = row ^ (unsigned)(swz(idx) << 4) ^ lhb;
    }
  };
  auto lds_read = [&](unsigned off) __attribute__((always_inline)) -> f32x4 {
    if constexpr ((W1_ABL & 2) != 0) return f32x4{(float)(lane * 3), 1.5f, -0.75f * lane, 0.3f + (float)off};
    return *reinterpret_cast<const f32x4*>(smem + off);
  };

  // ---- epilogue piece p = (mb = p >> 3, cb = (p >> 1) & 3, h = p & 1): position 128 wm + 32 mb + l31 of the M tile, channels
  // 128 (nt NWN + wn) + 32 cb + 16 h + 8 lh .. + 7 (after the lane-half exchange)
  f32x16 acc[4][4];
  u32x4 rq[NRQ];
  const unsigned ifloor = (a.act & 15) == 1 ? 0u : 0x80008000u;      // ReLU floor of a bf16 pair read as two int16 (none: the minimum)
  const int e_lane = (l31 * a.N + 8 * lh) * 2;
  auto ebyte = [&](int mt_e, int nt_e, int p) __attribute__((always_inline)) -> int {
    const int mb = p >> 3, cb = (p >> 1) & 3, h = p & 1;
    // scalar part + lane part, added where it is used (shared between a piece's residual request and its store the compiler
    // keeps 32 addresses in registers from one to the other)
    const int sc = ((mt_e * BM + 128 * wm + 32 * mb) * a.N + (nt_e * NWN + wn) * 128 + 32 * cb + 16 * h) * 2;
    int el = e_lane;
    asm volatile("" : "+v"(el));
    return sc + el;                                        // (pos >= M lies past num_records; N % 128 == 0)
  };
  auto rq_load = [&](int mt_e, int nt_e, int p) __attribute__((always_inline)) {
    if constexpr (RES && (W1_ABL & 64) == 0)
      rq[p % NRQ] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsR, ebyte(mt_e, nt_e, p), 0, 0));
  };
  auto epi_piece = [&](int mt_e, int nt_e, auto p_c) __attribute__((always_inline)) {
    constexpr int p = decltype(p_c)::value;
    constexpr int mb = p >> 3, cb = (p >> 1) & 3, h = p & 1;
    if constexpr ((W1_ABL & 32) != 0) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float t = acc[mb][cb][8 * h + e]; asm volatile("" :: "v"(t)); }
    } else {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = acc[mb][cb][8 * h + e];
      asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %4\n\tv_permlane32_swap_b32 %1, %5\n\tv_permlane32_swap_b32 %2, %6\n\t"
          "v_permlane32_swap_b32 %3, %7\n\ts_nop 1"
          : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
      if constexpr (RES) {
        const u32x4 t = rq[p % NRQ];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const unsigned rbits = (e & 1) ? (t[e >> 1] & 0xffff0000u) : (t[e >> 1] << 16);
          v[e] += __builtin_bit_cast(float, rbits);
        }
      }
      const int bo = ebyte(mt_e, nt_e, p);
      if constexpr ((W1_ABL & 16) != 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) asm volatile("" :: "v"(v[e]), "v"(bo));
      } else {
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bf16x2 pr = {(__bf16)v[2 * e], (__bf16)v[2 * e + 1]};
          unsigned u = __builtin_bit_cast(unsigned, pr);
          // ReLU on the pair: as int16 a negative bf16 (and -0) is negative, a positive one keeps its order
          asm("v_pk_max_i16 %0, %1, %2" : "=v"(u) : "v"(u), "v"(ifloor));
          o[e] = u;
        }
        __builtin_amdgcn_raw_buffer_store_b128(o, rsC, bo, 0, 0);
      }
    }
  };

  // ---- prologue: the first window (all pieces), the weights of k-steps 0 .. W1_D - 1, the first pixel fragments
  f32x4 breg[NSL][4], afr[2][4];
  int gb = (nt * NWN + wn) * KB;                           // this wave's weight stream of the current item
  {
    const int sbase = (mt * BM - a.W - 1) * cin_b;
#pragma unroll
    for (int n = 0; n < NPW; ++n) send_a(sbase, 0, n);
#pragma unroll
    for (int q = 0; q < W1_D; ++q)
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) breg[q][cb] = load_b(rsW, gb + q * 4096, cb);
  }
  w1_wait_vm<0>();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  unsigned mk[4], mkn[4];
  masks_of(mt, mk);
  unsigned cur[4], nxt[4];                                 // fragment addresses of the current / the next tap
  frag_addr(0, 0u, mk, cur);
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) afr[0][mb] = lds_read(cur[mb]);
  int wb = 0;                                              // window buffer of the current chunk
  const f32x4 ones = __builtin_bit_cast(f32x4, u32x4{lh ? 0u : 0x3f803f80u, lh ? 0u : 0x00003f80u, 0u, 0u});      // k = 0, 1, 2: 1.0

  using std::integral_constant;
  // one chunk = 36 k-steps (9 taps x 4 k-steps of 16 channels), fully unrolled.  LAST: the item's last chunk (residual requests,
  // masks of the next item's M tile, the epilogue follows)
  auto chunk = [&](auto last_c, int c, int mt_n, int c_n, int wnext, const __amdgpu_buffer_rsrc_t rsWn,
                   int mt_e, int nt_e) __attribute__((always_inline)) {
    constexpr bool LAST = decltype(last_c)::value;
    const int wcur = gb + c * (36 * 4096);
    const int sbase_n = (mt_n * BM - a.W - 1) * cin_b + c_n * 128;
    const unsigned wcur_off = (unsigned)(wb * WIN_B), wnxt_off = (unsigned)((wb ^ 1) * WIN_B);
    w1_static_for<36>([&](auto q_c) __attribute__((always_inline)) {
      constexpr int q = decltype(q_c)::value;
      constexpr int tap = q / 4, s = q % 4;
      constexpr int qa = q + 1, qb = q + W1_D;             // k-steps whose pixel / weight fragments are requested here
      constexpr int sa = qa & 1, sb = qb % NSL, sq = q % NSL;
      if constexpr (q == 35) {
        // the chunk's barrier: my pieces of the next window have landed (issued before k-step NPW <= 20: in-order completion,
        // only the requests of the last two k-steps may still be in flight), my reads of this window are done
        w1_wait_vm<(RES && LAST) ? 10 : 8>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      if constexpr (s == 0) {
        // addresses of the next tap's fragments (tap 8: tap 0 of the next chunk — the other buffer, the next item's masks
        // behind an item's last chunk)
        if constexpr (tap < 8) frag_addr(tap + 1, wcur_off, mk, nxt);
        else frag_addr(0, wnxt_off, LAST ? mkn : mk, nxt);
      }
      unsigned ra[4];
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) ra[mb] = (s < 3) ? (cur[mb] ^ (unsigned)((s + 1) << 5)) : nxt[mb];
      const int soff = qb < 36 ? wcur + qb * 4096 : wnext + (qb - 36) * 4096;
      // 16 MFMAs, a request behind every second one: four pixel-fragment reads (k-step q + 1), four weight loads (k-step q + W1_D)
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
          w1_mfma(acc[mb][cb], breg[sq][cb], afr[q & 1][mb]);
          if (mb == 1 || mb == 3) {
            const int r = 2 * cb + (mb >> 1);              // request slot 0 .. 7
            if (r < 4) afr[sa][r] = lds_read(ra[r]);
            else if (qb < 36) breg[sb][r - 4] = load_b(rsW, soff, r - 4);
            else breg[sb][r - 4] = load_b(rsWn, soff, r - 4);
          }
        }
      }
      if constexpr (q < NPW) send_a(sbase_n, wb ^ 1, q);
      if constexpr (RES && LAST && q >= 36 - NRQ) rq_load(mt_e, nt_e, q - (36 - NRQ));
      if constexpr (s == 3) {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) cur[mb] = nxt[mb];
      }
    });
    wb ^= 1;
  };

  for (int li = 0; li < nitems; ++li) {
    int mt1 = mt, nt1 = nt + 1;
    if (nt1 == a.ntiles) { nt1 = 0; ++mt1; }
    const bool more = li + 1 < nitems;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) mkn[mb] = mk[mb];
    if (mt1 != mt) { advance_mtile(); masks_of(mt1, mkn); }
    const int gb1 = (nt1 * NWN + wn) * KB;
    const __amdgpu_buffer_rsrc_t rsWn = more ? rsW : rsW0;
    // the accumulators start from the folded-BN shift: bias fragment (hi, mid, lo of the channel's shift at k = 0, 1, 2) x ones
    {
      f32x4 bf[4];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        const float sv = (W1_ABL & 128) ? 0.f : sh_lds[(nt * NWN + wn) * 128 + 32 * cb + l31];
        const __bf16 hi = (__bf16)sv;
        const float r1 = sv - (float)hi;
        const __bf16 mid = (__bf16)r1;
        const __bf16 lo = (__bf16)(r1 - (float)mid);
        const unsigned d0 = (unsigned)__builtin_bit_cast(unsigned short, hi) | ((unsigned)__builtin_bit_cast(unsigned short, mid) << 16);
        const unsigned d1 = (unsigned)__builtin_bit_cast(unsigned short, lo);
        bf[cb] = __builtin_bit_cast(f32x4, u32x4{lh ? 0u : d0, lh ? 0u : d1, 0u, 0u});
      }
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
          if constexpr ((W1_ABL & 256) != 0) asm volatile("" : "=a"(acc[mb][cb]));
          else w1_mfma0(acc[mb][cb], bf[cb], ones);
        }
    }
    for (int c = 0; c + 1 < a.NC; ++c)
      chunk(integral_constant<bool, false>{}, c, mt, c + 1, gb + (c + 1) * (36 * 4096), rsW, mt, nt);
    chunk(integral_constant<bool, true>{}, a.NC - 1, mt1, 0, gb1, rsWn, mt, nt);
    // ---- epilogue: 32 pieces; the residual of piece p + 8 is requested as the registers of piece p come free
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");     // (the last MFMA's result is readable 18 wait states later at most)
    w1_static_for<32>([&](auto p_c) __attribute__((always_inline)) {
      constexpr int p = decltype(p_c)::value;
      if constexpr ((p & 7) == 0) {
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) asm volatile("" : "+a"(acc[p >> 3][cb]));      // (keeps the accumulator reads of later blocks behind this point)
      }
      epi_piece(mt, nt, p_c);
      if constexpr (p < 32 - NRQ) rq_load(mt, nt, p + NRQ);
    });
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) mk[mb] = mkn[mb];
    mt = mt1; nt = nt1; gb = gb1;
  }
}

// ---------------------------------------------------------------------------------------------------------------
static int w128_pick(int W, int N, int* mw, int* npw) {
  // MW = 2 (256 x 256) where N allows it; pieces per wave: 4 * NPW * 8 pixels >= BM + 2 W + 2
  const int m = (N % 256 == 0) ? 2 : 4;
  const int need = (128 * m + 2 * W + 2 + 31) / 32;
  int p;
  if (m == 2) p = need <= 9 ? 9 : (need <= 10 ? 10 : (need <= 11 ? 11 : 0));
  else p = need <= 17 ? 17 : (need <= 19 ? 19 : 0);
  *mw = m; *npw = p;
  return p != 0;
}

static int w128_capable(int F, int H, int W, int Cin, int N) {
  if (F < 1 || H < 1 || W < 2) return 0;
  if (Cin % 64 != 0 || Cin < 64 || N % 128 != 0) return 0;
  int mw, npw;
  if (!w128_pick(W, N, &mw, &npw)) return 0;
  const long long M = (long long)F * H * W, lim = 1ll << 31;
  if (M * Cin * 2 >= lim || M * N * 2 >= lim || (long long)N * 9 * Cin * 2 >= lim || M >= (1 << 22)) return 0;
  if ((size_t)2 * 4 * npw * 1024 + 1024 + (size_t)N * 4 > 160 * 1024) return 0;
  return 1;
}

extern "C" int cadre_conv3x3_w128_supported(int32_t F, int32_t H, int32_t W, int32_t Cin, int32_t N) {
  return w128_capable(F, H, W, Cin, N);
}

extern "C" int cadre_conv3x3_w128(const void* x, const void* w, const float* scale, const float* shift, const void* resid, void* out,
                                  int32_t F, int32_t H, int32_t W, int32_t Cin, int32_t N, int32_t act, void* stream) {
  if (!x || !w || !out) return cadre_fail("cadre_conv3x3_w128: null operand");
  if (scale) return cadre_fail("cadre_conv3x3_w128: fold the BN scale into the weights (scale must be NULL)");
  if (!w128_capable(F, H, W, Cin, N))
    return cadre_fail("cadre_conv3x3_w128: unsupported geometry (Cin % 64 == 0, N % 128 == 0, W <= 46 (N % 256 == 0) / 47, every tensor < 2 GiB)");
  if ((act & ~1) != 0) return cadre_fail("cadre_conv3x3_w128: act 0 (none) and 1 (ReLU) only");
  if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)out | (uintptr_t)resid) & 15) return cadre_fail("cadre_conv3x3_w128: operands must be 16-byte aligned");
  int mw, npw;
  w128_pick(W, N, &mw, &npw);
  w128_args a;
  a.x = x; a.w = w; a.shift = shift; a.resid = resid; a.out = out;
  a.M = F * H * W; a.H = H; a.W = W; a.Cin = Cin; a.N = N; a.NC = Cin / 64; a.act = act;
  const int bm = 128 * mw, bn = 128 * (4 / mw);
  const int mtiles = (a.M + bm - 1) / bm;
  a.ntiles = N / bn;
  a.items = mtiles * a.ntiles;
  static const int n_cu = [] { int dev = 0, n = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
  const int wgs = a.items < n_cu ? a.items : n_cu;
  a.ipw = (a.items + wgs - 1) / wgs;
  const int grid = (a.items + a.ipw - 1) / a.ipw;
  const size_t lds = (size_t)2 * 4 * npw * 1024 + 1024 + (size_t)N * 4;
  hipStream_t st = (hipStream_t)stream;
#define W1_LAUNCH(MW_, NPW_)                                                                                                        \
  do {                                                                                                                               \
    if (resid) {                                                                                                                     \
      (void)hipFuncSetAttribute((const void*)conv3x3_w128_kernel<MW_, NPW_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);  \
      hipLaunchKernelGGL((conv3x3_w128_kernel<MW_, NPW_, true>), dim3(grid), dim3(256), lds, st, a);                                 \
    } else {                                                                                                                         \
      (void)hipFuncSetAttribute((const void*)conv3x3_w128_kernel<MW_, NPW_, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      hipLaunchKernelGGL((conv3x3_w128_kernel<MW_, NPW_, false>), dim3(grid), dim3(256), lds, st, a);                                \
    }                                                                                                                                \
  } while (0)
  if (mw == 2) {
    if (npw == 9) W1_LAUNCH(2, 9); else if (npw == 10) W1_LAUNCH(2, 10); else W1_LAUNCH(2, 11);
  } else {
    if (npw == 17) W1_LAUNCH(4, 17); else W1_LAUNCH(4, 19);
  }
#undef W1_LAUNCH
  return (int)hipGetLastError();
}
