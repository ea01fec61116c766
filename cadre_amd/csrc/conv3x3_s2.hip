// conv3x3_s2.hip — 3x3 / STRIDE 2 / pad 1 convolution on dense bf16 NHWC for gfx950: the first conv of the three
// down-sampling BasicBlocks of the DANet trunk (carla_perception/Networks/danet_blocks/resnet.py:26-55 with stride = 2,
// layer2.0 / layer3.0 / layer4.0: resnet.py:152-158), bf16 model (BASELINE config C3).
//
// Until round 5 these three layers ran as an implicit GEMM on the tile kernel (gemm_bf16.hip, a_mode 2): every input pixel
// gathered once per tap it serves (2.25 times on average) from L2 through VGPRs into LDS under a workgroup barrier per
// k-tile — 534 / 814 / 924 TFLOP/s, the kernels furthest below either roof.  Here they get what the
// stride-1 convs have had since round 2 (conv3x3_ring.hip), adapted to the stride:
//   * PIXEL PLANES.  With H = 2 Ho, W = 2 Wo the input splits into four planes by (row parity, column parity); in plane
//     coordinates every tap is a CONSTANT offset of the flattened output position p = (f Ho + ho) Wo + wo:
//         plane (1,1): taps (0,0) (0,2) (2,0) (2,2)  ->  p - Wo - 1, p - Wo, p - 1, p
//         plane (0,1): taps (1,0) (1,2)              ->  p - 1, p
//         plane (1,0): taps (0,1) (2,1)              ->  p - Wo, p
//         plane (0,0): tap  (1,1)                    ->  p
//     so a tile of 256 output positions needs, per 128-byte channel chunk, four WINDOWS of 256 (+ Wo + 1) consecutive
//     plane pixels — each input pixel enters LDS once per (M tile, channel chunk, N tile) instead of 2.25 times.  The planes
//     exist only in LDS: LDS-DMA (buffer_load ... lds) takes a per-lane SOURCE address, the lane computes the NHWC address of
//     its plane pixel (one exact floor-division by Wo per 8-pixel piece) — no VGPR round trip, hardware zero fill outside
//     the tensor; halo taps (ho = 0 / wo = 0) read a zero row (one select per fragment row, 9-bit mask per position).
//   * k ordered (chunk, plane, tap): a chunk's nine k-tiles run plane by plane (4 + 2 + 2 + 1).  The four windows of a
//     chunk do not fit LDS twice, so windows live in a RING OF THREE buffers (37 KB each): while the taps of one plane are
//     read, the next plane's window is complete and the one after it is landing.
//   * the eight waves form two groups of four half a k-tile apart (waves w and w + 4 share a SIMD), two barriers per
//     k-tile: one group stages (fragment reads + DMA issue) while the other issues its 16 MFMAs — the ping-pong of
//     conv3x3_ring_pp_kernel.  New here, the two groups have different DMA duties:
//       group 0 issues ALL weight pieces (k-tile t + 1 at the start of its staging slot R(t), confirmed by its own counted
//         vmcnt at the end of its MFMA slot M(t): two stages suffice);
//       group 1 issues ALL window pieces on a static schedule and confirms a window one slot before its first reader.
//     A wave's vmcnt completes in order: with weights and windows in different waves' queues, an HBM-latency window piece
//     never stands in front of an L2-latency weight piece that is needed next.
//   * weights [N][chunk][tap in plane order][128 B] stream through the two stages by LDS-DMA; 128-byte rows XOR-swizzled on
//     the SOURCE address (chunk ^ ((row >> 1) & 7)), every ds_read_b128 conflict-free;
//   * persistent workgroups walk contiguous (M tile, N tile) items, N inner; the previous item's epilogue runs in each
//     group's first staging slot of the next item, without LDS: v_permlane32_swap turns the accumulator's (lane -> position,
//     register -> channel) into eight consecutive channels per lane = one 16-byte store.
// y = act(conv + shift[n]) with the folded-BN scale already in the weight rows (scale = NULL); bf16 in, fp32 accumulate, bf16 out.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "../../include/cadre_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

int cadre_fail(const char* msg);

#define S2_BM 256                  // output positions per item
#define S2_NTILE 128               // output channels per item
// a window buffer holds 4 * NPW pieces of 8 pixels (NPW = 9: 288 pixels >= 256 + Wo + 1 for Wo <= 31; NPW = 10: 320 pixels, Wo <= 39):
// every piece of a window is then requested unconditionally — no per-piece "inside the window?" select in the staging slots
#define S2_STG_B (S2_NTILE * 128)

// -DS2_ABL=<bits>: timing ablations (results wrong by construction, never in the product build): 1 no MFMAs, 2 no fragment
// reads, 4 no window DMA after the prologue, 8 no weight DMA after the prologue, 16 no epilogue
#ifndef S2_ABL
#define S2_ABL 0
#endif
// L2 warming of the plane windows — built, measured, REMOVED (git history: "L2-warming touches").  A window can be requested
// only once its ring buffer is free — one or two k-tiles before its first reader for the windows behind the short phases —
// and a first-touch line comes from HBM: the ablation of the first build put 17-33 % of a launch on the window DMA.  The idea:
// group 1 TOUCHES a window's lines three k-tiles before it requests them (4-byte LDS-DMA loads, one lane per 64-byte sector,
// into a junk area).  Measured (profiles/r05_s2_l2_touch_ab.txt, same box, interleaved): 614 / 499 / 456 us with the touches
// against 515 / 423 / 403 without — the three touch instructions per odd k-tile carry ~15 VALU each (a floor-division by Wo per
// lane), and vector-ALU work inside a staging slot is starved by the other group's MFMAs; even with NO window
// DMA the touches alone cost 158 us.  Whatever they save in HBM latency is far less.

struct s2_args {
  const void* x;          // [F][H][W][Cin] bf16, H = 2 Ho, W = 2 Wo
  const void* w;          // [N][NC][9][128 B]: chunk-major, tap IN PLANE ORDER (0,0) (0,2) (2,0) (2,2) (1,0) (1,2) (0,1) (2,1) (1,1)
  const float* shift;     // [N] or null (the folded-BN scale belongs in the weights)
  void* out;              // [M][N] bf16
  int M, Min;             // output positions F * Ho * Wo, input pixels F * H * W
  int Ho, Wo, W, Cin, N, NC;
  int act;                // 0 none, 1 ReLU
  int mtiles, ntiles, items, ipw;
};

template <int N>
__device__ __forceinline__ void s2_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// NPW: window pieces per wave of group 1 and window (4 waves: 4 * NPW >= PA): 9 (Wo <= 31) or 10
template <int NPW>
__global__ __launch_bounds__(512, 2) void conv3x3_s2_kernel(s2_args a) {
  constexpr int NH = (NPW + 1) / 2;                         // a window's pieces go out in two k-tiles: NH, then NPW - NH
  constexpr int S2_WIN_B = 4 * NPW * 1024;                  // bytes of one window buffer
  constexpr int NST = 8;                                   // epilogue stores per wave: two 32-position x 64-channel blocks x four 1 KB stores
  constexpr unsigned OOB = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;                 // 64-position block, 64-channel half
  const int grp = wave >> 2, pb = wave & 3;                // ping-pong group (waves w, w + 4 share a SIMD); DMA piece lane of the wave
  char* win0 = smem;                                       // three window buffers, two weight stages, the zero row / dump, folded BN
  char* bst = smem + 3 * (4 * NPW * 1024);
  char* dump = bst + 2 * S2_STG_B;
  float* sc_lds = reinterpret_cast<float*>(dump + 1024);

  const int i_begin = blockIdx.x * a.ipw, i_end = min(a.items, i_begin + a.ipw);
  const int nitems = i_end - i_begin;
  if (nitems <= 0) return;
  const int cin_b = a.Cin * 2;
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.Min * cin_b, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsX0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, 0, 0x00020000);      // zero records: every request reads zeros
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.N * a.NC * 9 * 128, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, a.M * a.N * 2, 0x00020000);
  for (int i = tid; i < 256; i += 512) reinterpret_cast<unsigned*>(dump)[i] = 0u;       // ZERO ROW (halo taps) and dummy DMA target
  for (int i = tid; i < a.ntiles * S2_NTILE; i += 512) sc_lds[i] = (a.shift && i < a.N) ? a.shift[i] : 0.f;      // folded-BN shift of every channel
  auto swz = [](int idx) constexpr -> int { return (idx >> 1) & 7; };
  const int Wo = a.Wo, Wi = a.W;
  const float inv_wo = 1.0f / (float)Wo, inv_ho = 1.0f / (float)a.Ho;

  // ---- window DMA (group 1).  Piece j = 4 n + pb of a window: LDS rows 8 j .. 8 j + 7; this lane brings row 8 j + (lane >> 3),
  // LDS chunk (lane & 7) <- source chunk (lane & 7) ^ swz(row); swz(row) = (lane >> 4) ^ 4 (j & 1), and j has the parity of pb.
  // Row r of plane ph's window of M tile mt holds plane position pp = mt * 256 + start(ph) + r, i.e. input pixel
  //   (2 q + pr) * W + 2 (pp - q Wo) + pc = 2 pp + q W + pr W + pc,   q = floor(pp / Wo)   (W = 2 Wo; q counts rows across frames)
  // Positions before the tensor give a negative offset (as unsigned: beyond num_records), positions past its end lie beyond
  // num_records: both arrive as zeros.
  const int sw_lane = (((lane & 7) ^ (lane >> 4) ^ (4 * (pb & 1)))) << 4;
  bool abl_pro = true;
  // (the compiler hoists the plane-position arithmetic of a chunk's 40 pieces out of the k-tile loop: ~40 live registers, two
  //  VALU per request left in the staging slots — computed IN the slots the same arithmetic, ~15 VALU per piece, made them the
  //  longest part of the kernel)
  // (An odd-row plane's pixel is the even-row plane's pixel of the same column parity ONE INPUT ROW UP: row r of window (1,1) /
  //  (1,0) = row r of window (0,1) / (0,0) minus W pixels.  The two families share the division: 25 hoisted offsets per chunk
  //  instead of 40 — the difference between fitting 256 registers and spilling — and one subtraction left in the staging slot.)
  // (a window of the chunk AFTER the workgroup's last one is requested through a descriptor with zero records — zeros land in a
  //  buffer nobody reads — instead of a per-piece "live?" select of offset and target)
  auto send_win = [&](const __amdgpu_buffer_rsrc_t rs, int mt_n, int c_n, auto ph_c, int bufsel, int n) {
    constexpr int ph = decltype(ph_c)::value;              // 0: plane (1,1), 1: (0,1), 2: (1,0), 3: (0,0)
    constexpr int pc = (ph == 0 || ph == 1) ? 1 : 0;       // column parity; the family's even-row window starts at -pc
    constexpr bool up = (ph == 0 || ph == 2);
    const int lrow = lane >> 3;
    const int j = 4 * n + pb;
    const int pp = mt_n * S2_BM - pc + 8 * j + lrow;       // plane position in the family's EVEN-row plane
    int q = (int)((float)pp * inv_wo);                     // exact after one correction step (|pp| < 2^24)
    const int r = pp - q * Wo;
    q += (r >= Wo) ? 1 : 0;
    q -= (r < 0) ? 1 : 0;
    const int px = 2 * pp + q * Wi + pc;
    unsigned voff = (unsigned)(px * cin_b + c_n * 128 + sw_lane);
    if constexpr (up) {
      asm volatile("" : "+v"(voff));                       // (the shared value is hoisted, the row-up subtraction stays here)
      voff -= (unsigned)(Wi * cin_b);
    }
    char* dst = win0 + bufsel * S2_WIN_B + j * 1024;
    if ((S2_ABL & 4) && !abl_pro) { voff = OOB; dst = dump; }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, (int)voff, 0, 0, 0);
  };
  // ---- weight DMA (group 0): stage piece pc4 = 4 pb + k (k = 0 .. 3): rows 8 pc4 .. + 7
  int b_lane[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = (4 * pb + k) * 8 + (lane >> 3);
    b_lane[k] = r * a.NC * 9 * 128 + (((lane & 7) ^ swz(r)) << 4);
  }
  auto send_wts = [&](const __amdgpu_buffer_rsrc_t rs, int nt_b, int c, int tap, int stg) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      unsigned voff = (unsigned)(nt_b * (S2_NTILE * a.NC * 9 * 128) + (c * 9 + tap) * 128 + b_lane[k]);
      char* dst = bst + stg * S2_STG_B + (4 * pb + k) * 1024;
      if ((S2_ABL & 8) && !abl_pro) { voff = OOB; dst = dump; }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, (int)voff, 0, 0, 0);
    }
  };
  // ---- lane constants of the fragment reads.  k-step s of this lane half reads the logical 16-byte chunk 2 s + lh = (s << 1) | lh;
  // with the row swizzle sw the physical chunk is ((s << 1) | lh) ^ sw = (lh ^ sw) ^ (s << 1): ONE base per fragment row with the
  // lane-half and swizzle bits folded in, XOR-ed with the literal s << 5 (rows start on 128-byte boundaries, so base + x == base ^ x)
  const unsigned lh4 = (unsigned)lh << 4;
  const unsigned bbase = (unsigned)((64 * wn + l31) * 128) ^ ((unsigned)swz(64 * wn + l31) << 4) ^ lh4;      // weight row n = 64 wn + 32 cb + l31: swz(n + 32) == swz(n)
  const unsigned zrow_off = (unsigned)(dump - smem);

  // (ho, wo) of this lane's two fragment rows (positions 64 wm + 32 rb + l31 of the current M tile)
  int ph_[2], pw_[2];
  int mt = i_begin / a.ntiles, nt = i_begin - mt * a.ntiles;
  {
    const int HW = a.Ho * Wo;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const int m = mt * S2_BM + 64 * wm + 32 * rb + l31;
      const int rem = m % HW;
      ph_[rb] = rem / Wo;
      pw_[rb] = rem - ph_[rb] * Wo;
    }
  }
  auto advance_mtile = [&]() {                             // + 256 positions, exact float-reciprocal floors (x < 2^16)
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const int x = pw_[rb] + S2_BM;
      const int q1 = (int)(((float)x + 0.5f) * inv_wo);
      pw_[rb] = x - q1 * Wo;
      const int y = ph_[rb] + q1;
      const int q2 = (int)(((float)y + 0.5f) * inv_ho);
      ph_[rb] = y - q2 * a.Ho;
    }
  };

  // ---- epilogue of one item, no BN arithmetic: the folded-BN scale is in the weights (the caller folds it: scale == NULL is
  // part of the contract), the shift is the accumulators' initial value.  Accumulator tile (rb, cb): lane (l31, lh), register r
  // holds position 32 rb + l31, channel 32 cb + 8 (r >> 2) + 4 lh + (r & 3).  ReLU and pairwise conversion to bf16 in registers.
  f32x16 acc[2][2];
  const float act_floor = (a.act & 15) == 1 ? 0.f : -__builtin_inff();
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  auto acc_init = [&](int nt_i) {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const float* sh = sc_lds + nt_i * S2_NTILE + 64 * wn + 32 * cb + 4 * lh;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(sh + 8 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[0][cb][4 * g + e] = t[e]; acc[1][cb][4 * g + e] = t[e]; }
      }
    }
  };
  // Stores.  The lane-half exchange leaves a lane with 16 bytes of a position: a store instruction then writes 32 rows x 32 bytes,
  // four instructions per 128-byte line — measured against (wrong) fully coalesced stores that pattern costs 6-8 % of a launch
  // (profiles/r05_s2_store_coalescing_ablation.txt).  So the bf16 pairs of a 32-position x 64-channel block go through a 4 KB
  // slab of LDS instead (rows of 128 bytes, 16-byte chunks XOR-swizzled by the row) and come back as (position lane >> 3,
  // chunk lane & 7): every store instruction writes EIGHT FULL 128-byte lines, four instructions per block instead of eight
  // pieces.  The slabs live in the window buffer that is free while the epilogue runs (`slab`: the (1,0) buffer of the chunk that
  // begins — read last two slots ago, requested again at its k-tile 1).
  auto epilogue = [&](int mt_e, int nt_e, unsigned slab) {
    if constexpr ((S2_ABL & 16) != 0) {
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int r = 0; r < 16; ++r) { const float t = acc[rb][cb][r]; asm volatile("" :: "v"(t)); }
    } else {
      typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
      char* sl = smem + slab + wave * 4096;
      const int prow = lane >> 3, pch = lane & 7;           // read-back: position 8 i + prow, 16-byte chunk pch (channels 8 pch .. + 7)
      const bool ch_ok = nt_e * S2_NTILE + 64 * wn + 32 * (pch >> 2) < a.N;     // (N % 32 == 0)
      const int eb = (mt_e * S2_BM + 64 * wm + prow) * a.N + nt_e * S2_NTILE + 64 * wn + 8 * pch;
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            // registers 4 g .. 4 g + 3 of tile (rb, cb): channels 32 cb + 8 g + 4 lh .. + 3 of position l31 -> 8 bytes of chunk 4 cb + g
            const bf16x2 p0 = {(__bf16)fmaxf(acc[rb][cb][4 * g], act_floor), (__bf16)fmaxf(acc[rb][cb][4 * g + 1], act_floor)};
            const bf16x2 p1 = {(__bf16)fmaxf(acc[rb][cb][4 * g + 2], act_floor), (__bf16)fmaxf(acc[rb][cb][4 * g + 3], act_floor)};
            *reinterpret_cast<u32x2_t*>(sl + l31 * 128 + (((4 * cb + g) ^ (l31 & 7)) << 4) + 8 * lh) =
                u32x2_t{__builtin_bit_cast(unsigned, p0), __builtin_bit_cast(unsigned, p1)};
          }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int pos = 8 * i + prow;
          const u32x4 v = *reinterpret_cast<const u32x4*>(sl + pos * 128 + ((pch ^ (pos & 7)) << 4));
          const int eo = eb + (32 * rb + 8 * i) * a.N;
          const int bo = (int)((unsigned)(eo * 2) | (ch_ok ? 0u : OOB));        // (a position past M lies past num_records)
          __builtin_amdgcn_raw_buffer_store_b128(v, rsC, bo, 0, 0);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // (slab reads done before this buffer's next window may land)
    }
  };

  // ---- prologue: group 1 brings window (1,1) and the first half of window (0,1) of the first chunk, group 0 the weights of
  // k-tile 0; everything landed and published
  if (grp == 1) {
#pragma unroll
    for (int n = 0; n < NPW; ++n) send_win(rsX, mt, 0, std::integral_constant<int, 0>{}, 0, n);
#pragma unroll
    for (int n = 0; n < NH; ++n) send_win(rsX, mt, 0, std::integral_constant<int, 1>{}, 1, n);
  } else {
    send_wts(rsW, nt, 0, 0, 0);
  }
  s2_wait_vm<0>();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (grp == 1) __builtin_amdgcn_s_barrier();              // group 1 runs one slot behind
  abl_pro = false;

  // ---- main loop.  The ping-pong GROUP is a compile-time parameter of the whole item loop (two instances, one per group):
  // with the group's DMA duty and the end-of-item cases as run-time conditions every staging slot carried two or three scalar
  // branches — on the stride-1 ping-pong kernel taking them out of the k-loop was worth 5-10 %.  What stays: the
  // loops and, per chunk, one branch at its first k-tile (first of the item?) and one at its last (last of the item?).
  using std::integral_constant;
  auto run = [&](auto grp_c) {
    constexpr int GRP = decltype(grp_c)::value;
    int b0 = 0, b1 = 1, b2 = 2;                            // window buffers of the current chunk's planes (1,1) / (0,1) / (1,0); (0,0) re-uses b0
    int kpar = 0;                                          // weight stage of the current chunk's k-tile 0 (9 k-tiles per chunk: flips)
    int mt_p = 0, nt_p = 0;
    bool have_prev = false;
    for (int li = 0; li < nitems; ++li) {
      int mt1 = mt, nt1 = nt + 1;
      if (nt1 == a.ntiles) { nt1 = 0; ++mt1; }
      const bool more = li + 1 < nitems;
      // halo masks in k-tile order: bit t set = the tap's pixel exists (kh = 0 needs ho > 0, kw = 0 needs wo > 0)
      unsigned mask[2];
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        const unsigned top = ph_[rb] > 0 ? 1u : 0u, left = pw_[rb] > 0 ? 1u : 0u;
        mask[rb] = (top & left) | (top << 1) | (left << 2) | (1u << 3) | (left << 4) | (1u << 5) | (top << 6) | (1u << 7) | (1u << 8);
      }
      // ---- head of the item's first staging slot: group 0 requests the weights of k-tile 1 FIRST (its stores then stand
      // behind them in the queue: the wait at the end of M(0) need not drain them), then the previous item's epilogue, both
      // groups side by side (group 1 is still in its M slot of the previous item's last k-tile: it takes that slot's closing
      // barrier only now)
      __builtin_amdgcn_s_setprio(2);
      if constexpr (GRP == 0) send_wts(rsW, nt, 0, 1, kpar ^ 1);
      if (have_prev) {
        epilogue(mt_p, nt_p, (unsigned)(b2 * S2_WIN_B));
        if constexpr (GRP == 1) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
      }
      acc_init(nt);                                        // the sums start at the folded-BN shift

      for (int c = 0; c < a.NC; ++c) {
        const bool last_c = c + 1 == a.NC;
        const int mt_n = last_c ? mt1 : mt, nt_n = last_c ? nt1 : nt, c_n = last_c ? 0 : c + 1;
        const bool live_n = !last_c || more;
        const __amdgpu_buffer_rsrc_t rsXn = live_n ? rsX : rsX0, rsWn = live_n ? rsW : rsW0;      // (past the workgroup's last chunk: zero records)
        const unsigned wb0 = (unsigned)(b0 * S2_WIN_B), wb1 = (unsigned)(b1 * S2_WIN_B), wb2 = (unsigned)(b2 * S2_WIN_B);
        const unsigned st_even = (unsigned)((bst - smem) + kpar * S2_STG_B), st_odd = (unsigned)((bst - smem) + (kpar ^ 1) * S2_STG_B);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          // ================= R slot: fragment reads of k-tile t, DMA issue (at raised priority: between the other group's MFMAs)
          __builtin_amdgcn_s_setprio(2);
          // window buffer and row offset of k-tile t (static)
          const unsigned wbase = t < 4 ? wb0 : (t < 6 ? wb1 : (t < 8 ? wb2 : wb0));
          const int ro = (t == 1 || t == 5) ? 1 : ((t == 2 || t == 7) ? Wo : (t == 3 ? Wo + 1 : 0));
          const unsigned sbase = (t & 1) ? st_odd : st_even;
          f32x4 afr[2][4], bfr[2][4];
          {
            unsigned arow_sw[2];
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
              const int idx = 64 * wm + 32 * rb + l31 + ro;
              const unsigned row = ((mask[rb] >> t) & 1u) ? wbase + (unsigned)(idx << 7) : zrow_off;
              arow_sw[rb] = row ^ (unsigned)(swz(idx) << 4) ^ lh4;
            }
            const unsigned bsw = sbase + bbase;              // (stages start on 16 KB boundaries)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              bfr[0][s] = *reinterpret_cast<const f32x4*>(smem + (bsw ^ (unsigned)(s << 5)));
#pragma unroll
              for (int rb = 0; rb < 2; ++rb) afr[rb][s] = *reinterpret_cast<const f32x4*>(smem + (arow_sw[rb] ^ (unsigned)(s << 5)));
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) bfr[1][s] = *reinterpret_cast<const f32x4*>(smem + (bsw ^ (unsigned)(s << 5)) + 32 * 128);
          }
          if constexpr ((S2_ABL & 2) != 0) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
              for (int rb = 0; rb < 2; ++rb) afr[rb][s] = f32x4{(float)(lane * 3 + s), 1.5f + rb, -0.75f * lane, 0.3f};
#pragma unroll
              for (int cb = 0; cb < 2; ++cb) bfr[cb][s] = f32x4{0.01f * lane, -2.5f + cb, 0.125f * s, 1.f};
            }
          }
          __builtin_amdgcn_sched_barrier(0);               // (the reads go out first: their latency runs under the issue below)
          if constexpr (GRP == 0) {
            // weights of k-tile t + 1 into the stage k-tile t - 1 was read from (its last readers, group 1, finished a slot ago)
            if (t == 0) { if (c > 0) send_wts(rsW, nt, c, 1, kpar ^ 1); }      // (the item's first chunk: sent in the head)
            else if (t < 8) send_wts(rsW, nt, c, t + 1, ((t + 1) & 1) ^ kpar);
            else send_wts(rsWn, nt_n, c_n, 0, kpar ^ 1);   // next chunk's k-tile 0: the stage parity flips with the chunk
          } else {
            // window pieces, static schedule (the buffer a window goes into was read last two phases ago):
            //   t0: (0,1) of this chunk, second half | t1, t2: (1,0) of this chunk | t4, t5: (0,0) of this chunk (re-uses b0)
            //   t6, t7: (1,1) of the NEXT chunk -> b1 | t8: (0,1) of the next chunk, first half -> b2
            if (t == 0) {
#pragma unroll
              for (int n = NH; n < NPW; ++n) send_win(rsX, mt, c, integral_constant<int, 1>{}, b1, n);
            } else if (t == 1) {
#pragma unroll
              for (int n = 0; n < NH; ++n) send_win(rsX, mt, c, integral_constant<int, 2>{}, b2, n);
            } else if (t == 2) {
#pragma unroll
              for (int n = NH; n < NPW; ++n) send_win(rsX, mt, c, integral_constant<int, 2>{}, b2, n);
            } else if (t == 4) {
#pragma unroll
              for (int n = 0; n < NH; ++n) send_win(rsX, mt, c, integral_constant<int, 3>{}, b0, n);
            } else if (t == 5) {
#pragma unroll
              for (int n = NH; n < NPW; ++n) send_win(rsX, mt, c, integral_constant<int, 3>{}, b0, n);
            } else if (t == 6) {
#pragma unroll
              for (int n = 0; n < NH; ++n) send_win(rsXn, mt_n, c_n, integral_constant<int, 0>{}, b1, n);
            } else if (t == 7) {
#pragma unroll
              for (int n = NH; n < NPW; ++n) send_win(rsXn, mt_n, c_n, integral_constant<int, 0>{}, b1, n);
            } else if (t == 8) {
#pragma unroll
              for (int n = 0; n < NH; ++n) send_win(rsXn, mt_n, c_n, integral_constant<int, 1>{}, b2, n);
            }
            // a window is confirmed (in-order completion: everything but the pieces issued after its last one) in the slot
            // before group 0 first reads it: (0,1) at t3, (1,0) at t5, (0,0) at t7, the next chunk's (1,1) at t8
            if (t == 3) s2_wait_vm<NPW>();                 // younger: t1 + t2
            if (t == 5) s2_wait_vm<NPW>();                 // younger: t4 + t5
            if (t == 7) s2_wait_vm<NPW>();                 // younger: t6 + t7
            if (t == 8) s2_wait_vm<NH>();                  // younger: t8
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
          __builtin_amdgcn_s_setprio(0);
          __builtin_amdgcn_sched_barrier(0);
          // ================= M slot
#pragma unroll
          for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              if constexpr ((S2_ABL & 1) != 0) {
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) asm volatile("" :: "v"(bfr[cb][s]), "v"(afr[rb][s]));
              } else {
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
                  acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bfr[cb][s]), __builtin_bit_cast(bf16x8, afr[rb][s]),
                                                                        acc[rb][cb], 0, 0, 0);
              }
            }
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (GRP == 0) {
            // the weights of k-tile t + 1 (issued in R(t)) have landed; after the item's first k-tile the previous item's
            // stores (issued behind them) may stay in flight
            if (t == 0) {
              if (c == 0 && have_prev && !(S2_ABL & 16)) s2_wait_vm<NST>();
              else s2_wait_vm<0>();
            } else {
              s2_wait_vm<0>();
            }
            __builtin_amdgcn_s_barrier();
          } else {
            if (t == 8) { if (!last_c) __builtin_amdgcn_s_barrier(); }      // (group 1, end of an item: the closing barrier comes after its epilogue)
            else __builtin_amdgcn_s_barrier();
          }
          asm volatile("" ::: "memory");
        }
        // next chunk: planes (1,1) / (0,1) / (1,0) in b1 / b2 / b0
        { const int t0 = b0; b0 = b1; b1 = b2; b2 = t0; }
        kpar ^= 1;
      }
      mt_p = mt; nt_p = nt; have_prev = true;
      if (mt1 != mt) advance_mtile();
      mt = mt1; nt = nt1;
    }
    // ---- tail: the last item's epilogue (group 0 one slot before group 1)
    __builtin_amdgcn_s_setprio(0);
    epilogue(mt_p, nt_p, (unsigned)(b2 * S2_WIN_B));
    __builtin_amdgcn_s_barrier();
  };
  if (grp == 0) run(integral_constant<int, 0>{});
  else run(integral_constant<int, 1>{});
}

// ---------------------------------------------------------------------------------------------------------------
static int s2_capable(int F, int H, int W, int Cin, int N) {
  if (F < 1 || H < 2 || W < 2 || (H & 1) || (W & 1)) return 0;
  const int Wo = W / 2;
  if (Wo < 1 || Wo > 39) return 0;                         // a window is 256 + Wo + 1 <= 296 pixels
  if (Cin % 64 != 0 || N % 32 != 0 || Cin < 64) return 0;
  const long long Min = (long long)F * H * W, M = Min / 4, lim = 1ll << 31;
  if (Min * Cin * 2 >= lim || (long long)N * Cin * 9 * 2 >= lim || M * N * 2 >= lim) return 0;
  if (M >= (1 << 23)) return 0;                            // the plane-position division runs in fp32
  const int ntiles = (N + S2_NTILE - 1) / S2_NTILE;
  const int npw = (S2_BM + Wo + 1 + 7) / 8 <= 36 ? 9 : 10;
  if ((size_t)3 * 4 * npw * 1024 + 2 * S2_STG_B + 1024 + (size_t)ntiles * S2_NTILE * 4 > 160 * 1024) return 0;
  return 1;
}

static const int g_s2_on = [] { const char* e = getenv("CADRE_S2_CONV"); return e ? atoi(e) : 1; }();

// 1: cadre_conv3x3_s2 takes this geometry (bf16 NHWC, even H and W, Cin % 64 == 0, N % 32 == 0); CADRE_S2_CONV=0: never
extern "C" int cadre_conv3x3_s2_supported(int32_t F, int32_t H, int32_t W, int32_t Cin, int32_t N) {
  return (g_s2_on && s2_capable(F, H, W, Cin, N)) ? 1 : 0;
}

extern "C" int cadre_conv3x3_s2(const void* x, const void* w, const float* scale, const float* shift, void* out, int32_t F,
                                int32_t H, int32_t W, int32_t Cin, int32_t N, int32_t act, void* stream) {
  if (!x || !w || !out) return cadre_fail("cadre_conv3x3_s2: null operand");
  if (scale) return cadre_fail("cadre_conv3x3_s2: fold the BN scale into the weight rows and pass scale = NULL (the kernel has no epilogue multiply)");
  if (!s2_capable(F, H, W, Cin, N))
    return cadre_fail("cadre_conv3x3_s2: unsupported geometry (even H, W; W <= 78; Cin % 64 == 0; N % 32 == 0; every tensor < 2 GiB: chunk the batch)");
  if ((act & 15) > 1 || (act & 16)) return cadre_fail("cadre_conv3x3_s2: act 0 (none) or 1 (ReLU)");
  if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)out) & 15) return cadre_fail("cadre_conv3x3_s2: operands must be 16-byte aligned");
  s2_args a;
  a.x = x; a.w = w; a.shift = shift; a.out = out;
  a.Ho = H / 2; a.Wo = W / 2; a.W = W; a.Cin = Cin; a.N = N; a.NC = Cin / 64; a.act = act;
  a.Min = F * H * W; a.M = F * a.Ho * a.Wo;
  a.mtiles = (a.M + S2_BM - 1) / S2_BM;
  a.ntiles = (N + S2_NTILE - 1) / S2_NTILE;
  a.items = a.mtiles * a.ntiles;
  const int wgs = a.items < 256 ? a.items : 256;           // persistent workgroups: one per CU
  a.ipw = (a.items + wgs - 1) / wgs;
  const int grid = (a.items + a.ipw - 1) / a.ipw;
  const int npw = (S2_BM + a.Wo + 1 + 7) / 8 <= 36 ? 9 : 10;
  const size_t lds = (size_t)3 * 4 * npw * 1024 + 2 * S2_STG_B + 1024 + (size_t)a.ntiles * S2_NTILE * 4;
  hipStream_t st = (hipStream_t)stream;
  if (npw == 9) {
    (void)hipFuncSetAttribute((const void*)conv3x3_s2_kernel<9>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((conv3x3_s2_kernel<9>), dim3(grid), dim3(512), lds, st, a);
  } else {
    (void)hipFuncSetAttribute((const void*)conv3x3_s2_kernel<10>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((conv3x3_s2_kernel<10>), dim3(grid), dim3(512), lds, st, a);
  }
  return (int)hipGetLastError();
}
