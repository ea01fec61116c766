// conv3x3_ring.hip — 3x3 / stride 1 / pad 1 convolution on dense NHWC for gfx950, fp32 (v_mfma_f32_32x32x2_f32,
// exact fp32 fma chain) and bf16 (v_mfma_f32_32x32x16_bf16), the workhorse of the DANet trunk and head
// (resnet.py:26-55 BasicBlock conv1/conv2 of layer1-4, danet.py:21-41 conv5a/5c/51/52): 16 of the encoder's 20 3x3
// convolutions are stride 1.
//
// What the implicit-GEMM tile kernels (gemm_f32.hip / gemm_bf16.hip, a_mode 2) pay for on these layers is operand
// staging: every input pixel is gathered 9 times (once per tap) from L2 into LDS through a VGPR round trip.  Here:
//   * k is ordered (channel chunk of 128 bytes, tap): for one chunk the workgroup brings the WINDOW of pixels its
//     256 output positions can touch — positions [P - W - 1, P + 256 + W + 1) of the flattened N*H*W axis, 128 B each
//     — into LDS ONCE, by LDS-DMA (buffer_load ... lds: no VGPR round trip, hardware zero fill outside the tensor);
//     the nine taps are nine row offsets (kh*W + kw) into the same resident window; taps that fall outside the frame
//     are zeroed on the fragment with a per-position 9-bit mask.  A traffic L2 -> LDS drops ~7x;
//   * the weights [N][chunk][tap][128 B] stream through a 3-stage LDS ring, two k-tiles ahead, also by LDS-DMA;
//   * the next chunk's window (or the next tile's first window) is loaded in slices during the current chunk's first
//     k-tiles — a whole phase of lead time; everything is ordered by each wave's own counted vmcnt plus raw s_barriers
//     (a __syncthreads() fence would drain vmcnt to 0);
//   * 128-byte rows are XOR-swizzled on the SOURCE address (chunk ^ ((row >> 1) & 7)), LDS stays lane-linear as
//     LDS-DMA requires, all ds_read_b128 fragment reads are bank-conflict free;
//   * persistent workgroups (one per CU) walk a contiguous run of (M-tile, N-tile) items, N inner: the epilogue of one
//     item runs while the loads of the next are already in flight.
// Two kernels share this scheme: conv3x3_ring_kernel (all eight waves in lockstep, one barrier per k-tile) and
// conv3x3_ring_pp_kernel (two groups of four waves half a k-tile apart, one staging while the other computes: every
// bf16 shape and the fp32 64-channel stage; see the comment above it).  -DRING_TRACE=1/2/3 compiles shader-clock
// stamps into them (tools/ring_trace.py) — where an item's time goes was measured, not guessed.
// Same epilogue contract as cadre_gemm_*: y = act(conv * scale[n] + shift[n] (+ resid)) (+ resid after act).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "../../include/cadre_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

int cadre_fail(const char* msg);

#define RG_BM_MAX 256               // output positions per workgroup tile: 64 * WVM
#define RG_SLAB 4608               // epilogue slab per wave: 32 rows x 36 floats

struct ring_args {
  const void* x;          // [M][Cin] elements (fp32 or bf16), M = F*H*W
  const void* w;          // [N][NC][9][128 B]: chunk-major, tap, then the chunk's channels (k contiguous)
  const float* scale;     // [N] or null
  const float* shift;     // [N] or null
  const void* resid;      // [M][N] (fp32 / bf16) or null
  void* out;              // [M][N] (fp32 / bf16)
  int M, H, W, Cin, N, NC;
  int act;                // 0 none, 1 ReLU; |16: resid added after the activation
  int out_bf16, resid_bf16;
  int mtiles, ntiles, items, ipw;      // M tiles of 256, N tiles, work items, items per workgroup
  int WPX;                // window pixels (multiple of 8, >= 288)
  int prio;               // ping-pong kernel: raise the wave priority in the staging slots
#ifdef RING_TRACE
  long long* trace;       // tools/ring_trace.py: per (workgroup, item) 6 shader-clock stamps of wave 0
#endif
};
#define RG_STAMP_(k) do { if (a.trace && (tid == 0 || tid == 256) && li < 64 && (tid == 0) == (blockIdx.x % 2 == 0)) a.trace[((size_t)blockIdx.x * 64 + li) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#if defined(RING_TRACE) && RING_TRACE == 1      // sections of an item
#define RG_STAMP(k) RG_STAMP_(k)
#else
#define RG_STAMP(k) do { } while (0)
#endif
#if defined(RING_TRACE) && RING_TRACE == 3      // ping-pong kernel: inside the staging slot of k-tile 4
#define RG_SLOT(k) RG_STAMP_(k)
#else
#define RG_SLOT(k) do { } while (0)
#endif
#if defined(RING_TRACE) && RING_TRACE == 4      // in-kernel clock: shader clock and 100 MHz wall clock at a workgroup's entry and exit
#define RG_CLK(k) do { if (a.trace && threadIdx.x == 0) { a.trace[(size_t)blockIdx.x * 4 + 2 * (k)] = __builtin_amdgcn_s_memtime(); \
    a.trace[(size_t)blockIdx.x * 4 + 2 * (k) + 1] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define RG_CLK(k) do { } while (0)
#endif
#if defined(RING_TRACE) && RING_TRACE == 2      // inside k-tiles 4 and 5 of the first chunk
#define RG_STEP(k) RG_STAMP_(k)
#else
#define RG_STEP(k) do { } while (0)
#endif

// -DRING_ABL=<bits>: timing ablations of the ping-pong kernel (tools/ring_ablate.py; results are WRONG by construction, never
// in the product build): 1 no MFMAs, 2 no fragment reads, 4 no window DMA after the prologue, 8 no weight DMA after the
// prologue, 16 no global stores, 32 no epilogue at all, 64 no residual loads.  Skipped values stay live through empty asm
// (a skipped producer must not let the compiler delete its consumers: cdna_hip_programming.md 5.4 rule 17).
#ifndef RING_ABL
#define RING_ABL 0
#endif

// The staging slots' raised wave priority (+1-3 % on every bf16 shape) is a compile-time choice: as the run-time flag
// a.prio it put two scalar branches into every slot of the ping-pong kernels (RING_PRIO_CONST=0 restores the flag for A/B).
#ifndef RING_PRIO_CONST
#define RING_PRIO_CONST 1
#endif
#if RING_PRIO_CONST
#define RING_PRIO_ON true
#else
#define RING_PRIO_ON (a.prio)
#endif

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_vm_n(int n) {        // wave-uniform n in [0, 40]
  switch (n) {
#define RG_CASE(k) case k: wait_vm<k>(); break;
    RG_CASE(0) RG_CASE(1) RG_CASE(2) RG_CASE(3) RG_CASE(4) RG_CASE(5) RG_CASE(6) RG_CASE(7) RG_CASE(8) RG_CASE(9)
    RG_CASE(10) RG_CASE(11) RG_CASE(12) RG_CASE(13) RG_CASE(14) RG_CASE(15) RG_CASE(16) RG_CASE(17) RG_CASE(18) RG_CASE(19)
    RG_CASE(20) RG_CASE(21) RG_CASE(22) RG_CASE(23) RG_CASE(24) RG_CASE(25) RG_CASE(26) RG_CASE(27) RG_CASE(28) RG_CASE(29)
    RG_CASE(30) RG_CASE(31) RG_CASE(32) RG_CASE(33) RG_CASE(34) RG_CASE(35) RG_CASE(36) RG_CASE(37) RG_CASE(38) RG_CASE(39)
#undef RG_CASE
    default: wait_vm<0>(); break;
  }
}

template <int V>
struct int_k { static constexpr int value = V; };

// NTILE: output channels per workgroup tile (64 or 128); 8 waves as 4 (positions) x 2 (channels): a wave owns
// 64 positions x NTILE/2 channels = 2 x WN MFMA tiles of 32x32.
// RES: 0 no residual, 1 fp32 residual, 2 bf16 residual; OUTB: output bf16 (else fp32)
// WVM: waves along the positions (4: 256-position tile, 8 waves, 3 weight stages two k-tiles ahead, ONE workgroup
// per CU; 2: 128-position tile, 4 waves, 2 weight stages one k-tile ahead, <= 80 KB of LDS so that TWO workgroups share
// a CU and each one's per-k-tile turnover runs under the other's MFMAs — for the layers with small maps).
template <bool BF16, int NTILE, int RES, bool OUTB, int WVM>
__global__ __launch_bounds__(128 * WVM, 2) void conv3x3_ring_kernel(ring_args a) {
  constexpr int WN = NTILE / 64;                           // 32-channel column blocks per wave
  constexpr int EB = BF16 ? 2 : 4;
  constexpr int NWAVES = 2 * WVM, NTHR = 64 * NWAVES;
  constexpr int RG_BM = 64 * WVM;
  constexpr int RG_NSTB = WVM == 4 ? 3 : 2, LEAD = RG_NSTB - 1;
  constexpr int STG_B = NTILE * 128;                       // bytes of one weight stage: [NTILE rows][128 B]
  constexpr int NBPW = (NTILE / 8) / NWAVES;               // weight pieces (8 rows x 128 B) per wave per stage
  // window slices of the NEXT phase, per wave and step: front-loaded, two per step on the first NFL steps (a slice
  // needs an HBM round trip, 2-4 us under load: issued late in the phase it is waited on at the next phase's first step)
  constexpr int NFL = WVM == 4 ? 4 : 5;
  auto sl_of = [](int tap) constexpr -> int { return tap < NFL ? 2 : 0; };
  constexpr unsigned OOB = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int win_bytes = a.WPX * 128;
  char* win0 = smem;                                       // two windows, then the weight stages, then a 1 KiB dump
  char* bst = smem + 2 * win_bytes;
  char* dump = bst + RG_NSTB * STG_B;                      // ... a 1 KiB dump, the folded-BN table

  const int i_begin = blockIdx.x * a.ipw, i_end = min(a.items, i_begin + a.ipw);
  const int nitems = i_end - i_begin;                      // uniform over the workgroup
  if (nitems <= 0) return;
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.M * a.Cin * EB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.N * a.NC * 9 * 128, 0x00020000);
  constexpr int ESZ = OUTB ? 2 : 4, RSZ = RES == 2 ? 2 : 4;
  const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, a.M * a.N * ESZ, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void*)(RES ? a.resid : a.x), 0, RES ? a.M * a.N * RSZ : 0, 0x00020000);
  const int cin_b = a.Cin * EB;
  const int PA = a.WPX >> 3;                               // window pieces
  for (int i = tid; i < 256; i += NTHR) reinterpret_cast<unsigned*>(dump)[i] = 0u;      // the dump doubles as the ZERO ROW
  float* sc_lds = reinterpret_cast<float*>(dump + 1024);   // folded BN of every output channel: [scale | shift], ntiles*NTILE each
  for (int i = tid; i < a.ntiles * NTILE; i += NTHR) {
    sc_lds[i] = (a.scale && i < a.N) ? a.scale[i] : 1.f;
    sc_lds[a.ntiles * NTILE + i] = (a.shift && i < a.N) ? a.shift[i] : 0.f;
  }

  // ---- lane constants of the staging path.  A window piece j (8 pixels): this lane brings pixel 8j + (lane>>3),
  // LDS chunk (lane&7), i.e. source chunk (lane&7) ^ ((idx>>1)&7) with idx = 8j + (lane>>3): (idx>>1)&7 =
  // (lane>>4) ^ 4*(j&1), and j = NWAVES*slot + wave has the parity of the wave — a per-lane constant.  Pixels outside the
  // tensor need no test: a negative position wraps to a huge unsigned offset, one past the end lies beyond
  // num_records — both arrive as zeros.
  const int a_lane = (lane >> 3) * cin_b + ((((lane & 7) ^ (lane >> 4) ^ (4 * (wave & 1)))) << 4);
  int b_lane[NBPW];                                        // weight piece k of this wave: stage row r = (wave*NBPW + k)*8 + lane>>3
#pragma unroll
  for (int k = 0; k < NBPW; ++k) {
    const int r = (wave * NBPW + k) * 8 + (lane >> 3);
    b_lane[k] = r * a.NC * 9 * 128 + (((lane & 7) ^ ((r >> 1) & 7)) << 4);
  }
  auto issue_a = [&](int mt_n, int c_n, int wsel, int j, bool live) {      // one window piece of the NEXT phase (or a dummy)
    const bool ok = live && j < PA;
    const unsigned voff = ok ? (unsigned)((mt_n * RG_BM - a.W - 1 + 8 * j) * cin_b + c_n * 128 + a_lane) : OOB;
    char* dst = ok ? win0 + wsel * win_bytes + j * 1024 : dump;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (__attribute__((address_space(3))) void*)dst, 16, (int)voff, 0, 0, 0);
  };
  auto issue_b = [&](int nt_b, int c, int tap, int stg, bool live) {       // this wave's weight pieces of one step
#pragma unroll
    for (int k = 0; k < NBPW; ++k) {
      const unsigned voff = live ? (unsigned)(nt_b * (NTILE * a.NC * 9 * 128) + (c * 9 + tap) * 128 + b_lane[k]) : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(bst + stg * STG_B + (wave * NBPW + k) * 1024),
                                               16, (int)voff, 0, 0, 0);
    }
  };
  // ---- lane constants of the fragment reads
  const int arow0 = (64 * wm + l31) * 128;                 // window byte offset of fragment row (rb = 0) at tap (0,0)
  int kc[4];                                               // logical 16-B chunk of k-step s for this lane half
#pragma unroll
  for (int s = 0; s < 4; ++s) kc[s] = 2 * s + lh;
  int boff[WN][4];                                         // weight fragment offsets inside a stage (swizzle is per row: constant)
#pragma unroll
  for (int cb = 0; cb < WN; ++cb) {
    const int n = (NTILE / 2) * wn + 32 * cb + l31;
#pragma unroll
    for (int s = 0; s < 4; ++s) boff[cb][s] = n * 128 + (((2 * s + lh) ^ ((n >> 1) & 7)) << 4);
  }

  // (h, w) of this lane's two fragment rows (positions 64*wm + 32*rb + l31 of the current M tile)
  const float inv_w = 1.0f / (float)a.W, inv_h = 1.0f / (float)a.H;
  int ph[2], pw[2];
  int mt = i_begin / a.ntiles, nt = i_begin - mt * a.ntiles;      // the only divisions of the kernel
  {
    const int HW = a.H * a.W;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const int m = mt * RG_BM + 64 * wm + 32 * rb + l31;
      const int rem = m % HW;
      ph[rb] = rem / a.W;
      pw[rb] = rem - ph[rb] * a.W;
    }
  }
  auto advance_mtile = [&]() {                             // + 256 positions, exact float-reciprocal floors (x < 2^16)
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const int x = pw[rb] + RG_BM;
      const int q1 = (int)(((float)x + 0.5f) * inv_w);
      pw[rb] = x - q1 * a.W;
      const int y = ph[rb] + q1;
      const int q2 = (int)(((float)y + 0.5f) * inv_h);
      ph[rb] = y - q2 * a.H;
    }
  };

  // ---- prologue: window of the first phase (all pieces), weights of steps 0 and 1
  const int total_ph = nitems * a.NC;
  for (int j = wave; j < PA; j += NWAVES) issue_a(mt, 0, 0, j, true);  // (j parity == wave parity: a_lane holds)
  issue_b(nt, 0, 0, 0, true);
  if constexpr (LEAD == 2) issue_b(nt, 0, 1, 1, true);
  int extra = 0, extra_steps = 0;                          // epilogue stores still behind the pieces a wait must cover (2 steps)
  int stg = 0;                                             // weight stage of the current step
  bool first_step = true;

  for (int li = 0; li < nitems; ++li) {
    // next item (N inner): its tiles feed the look-ahead issues
    int mt1 = mt, nt1 = nt + 1;
    if (nt1 == a.ntiles) { nt1 = 0; ++mt1; }
    const bool more = li + 1 < nitems;
    RG_STAMP(0);
    unsigned mask[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const int m = mt * RG_BM + 64 * wm + 32 * rb + l31;
      unsigned mk = 0;
      if (m < a.M) {
        unsigned colm = 0;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
          if ((unsigned)(pw[rb] - 1 + kw) < (unsigned)a.W) colm |= 1u << kw;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
          if ((unsigned)(ph[rb] - 1 + kh) < (unsigned)a.H) mk |= colm << (3 * kh);
      }
      mask[rb] = mk;
    }
    f32x16 acc[2][WN];
    // epilogue addressing (also used by the early residual request inside the k-loop)
    constexpr int NBLK = 2 * WN;                           // 32 x 32 blocks of this wave: b = cb*2 + rb
    const int row0 = lane >> 3, c4 = (lane & 7) * 4;       // lane -> (row 8*i + lane/8, channels 4*(lane&7) .. +3)
    auto eoff = [&](int b, int i) -> unsigned {            // element offset of this lane's 4 outputs, or OOB
      const int cb = b >> 1, rb = b & 1;
      const int pos = mt * RG_BM + 64 * wm + 32 * rb + 8 * i + row0;
      const int ch = nt * NTILE + (NTILE / 2) * wn + 32 * cb + c4;
      return (pos < a.M && ch < a.N) ? (unsigned)(pos * a.N + ch) : OOB;
    };
    u32x4 rq[2][4];
    auto req = [&](int b, u32x4* dst) {
      if constexpr (RES != 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const unsigned eo = eoff(b, i);
          if constexpr (RES == 2) {
            const u32x2 t = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rsR, eo == OOB ? (int)OOB : (int)(eo * 2), 0, 0));
            dst[i] = u32x4{t[0], t[1], 0u, 0u};
          } else {
            dst[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsR, eo == OOB ? (int)OOB : (int)(eo * 4), 0, 0));
          }
        }
      }
    };
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int cb = 0; cb < WN; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rb][cb][r] = 0.f;

    for (int c = 0; c < a.NC; ++c) {
      const int phg = li * a.NC + c;                       // phase index of this workgroup
      const char* win = win0 + (phg & 1) * win_bytes;
      const bool has_next = phg + 1 < total_ph;
      const bool last_c = c + 1 == a.NC;
      const int mt_n = last_c ? mt1 : mt, c_n = last_c ? 0 : c + 1;
      char* const zrow = dump;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        // ---- everything issued LEAD steps ago (weights of this step, window slices) has landed; publish
        // (in-order completion: all but the youngest (LEAD-1) steps' operations — plus, for LEAD steps after an
        //  epilogue, its stores — must have completed)
        if (c == 0 && tap == 4) RG_STEP(0);
        if (c == 0 && tap == 5) RG_STEP(4);
        if (first_step) { wait_vm<0>(); first_step = false; }
        else {
          int allow = (LEAD - 1) * (NBPW + sl_of((tap + 8) % 9));                  // (folds: tap is unrolled)
          if (RES != 0 && LEAD == 2 && tap == 8 && last_c) allow += 4;               // the early residual request (below)
          if (extra_steps > 0) { wait_vm_n(allow + extra); --extra_steps; }
          else wait_vm_n(allow);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (c == 0 && tap == 0) RG_STAMP(1);
        if (c == 0 && tap == 4) RG_STEP(1);
        if (c == 0 && tap == 5) RG_STEP(5);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (c == 0 && tap == 0) RG_STAMP(2);
        if (c == 0 && tap == 4) RG_STEP(2);
        if (c == 0 && tap == 5) RG_STEP(6);
        // ---- issue: weights LEAD steps ahead, window slices of the next phase
        {
          const int s2 = stg + LEAD >= RG_NSTB ? stg + LEAD - RG_NSTB : stg + LEAD;
          if (tap + LEAD < 9) issue_b(nt, c, tap + LEAD, s2, true);
          else if (!last_c) issue_b(nt, c + 1, tap + LEAD - 9, s2, true);
          else issue_b(nt1, 0, tap + LEAD - 9, s2, more);
#pragma unroll
          for (int i = 0; i < sl_of(tap); ++i)
            issue_a(mt_n, c_n, (phg + 1) & 1, (2 * tap + i) * NWAVES + wave, has_next);
          // the residual of the item's first output block: requested two steps before the epilogue needs it
          if constexpr (RES != 0) { if (tap == 7 && last_c) req(0, rq[0]); }
        }
        if (c == 0 && tap == 4) RG_STEP(3);
        // ---- compute this step.  The WEIGHTS are the MFMA's A operand, the pixels its B operand: the accumulator then
        // holds (lane -> position, register -> channel), four consecutive channels per register quad — the epilogue
        // stores a quad with one ds_write_b128.
        const char* bs = bst + stg * STG_B;
        const int toff = (tap / 3) * a.W + (tap % 3);
        f32x4 afr[2][4];
        const char* arow[2];
        unsigned sw[2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
          const int idx = 64 * wm + 32 * rb + l31 + toff;
          sw[rb] = (unsigned)((idx >> 1) & 7);
          // halo tap of this position: read the zero row instead (one select per row, not per fragment register)
          arow[rb] = ((mask[rb] >> tap) & 1u) ? win + arow0 + (32 * rb + toff) * 128 : zrow;
        }
        if constexpr (BF16) {
          // A bf16 k-tile is 8 x WN MFMAs of 32 cycles: left to itself the compiler sinks every fragment read to its
          // MFMA (read, s_waitcnt, MFMA, read ...), one LDS round trip per pair.  All reads of the step are issued
          // first, in the order the MFMAs consume them, and pinned there: the waits become counted lgkmcnt(n).
          f32x4 bfr[WN][4];
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            bfr[0][s] = *reinterpret_cast<const f32x4*>(bs + boff[0][s]);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) afr[rb][s] = *reinterpret_cast<const f32x4*>(arow[rb] + ((kc[s] ^ sw[rb]) << 4));
          }
#pragma unroll
          for (int cb = 1; cb < WN; ++cb)
#pragma unroll
            for (int s = 0; s < 4; ++s) bfr[cb][s] = *reinterpret_cast<const f32x4*>(bs + boff[cb][s]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int cb = 0; cb < WN; ++cb)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
              for (int rb = 0; rb < 2; ++rb)
                acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bfr[cb][s]), __builtin_bit_cast(bf16x8, afr[rb][s]),
                                                                      acc[rb][cb], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        } else {
#pragma unroll
          for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int s = 0; s < 4; ++s) afr[rb][s] = *reinterpret_cast<const f32x4*>(arow[rb] + ((kc[s] ^ sw[rb]) << 4));
#pragma unroll
          for (int cb = 0; cb < WN; ++cb) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              const f32x4 bv = *reinterpret_cast<const f32x4*>(bs + boff[cb][s]);
#pragma unroll
              for (int e = 0; e < 4; ++e)                  // alternate the two accumulators: consecutive MFMAs independent
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[e], afr[rb][s][e], acc[rb][cb], 0, 0, 0);
            }
          }
        }
        stg = stg + 1 == RG_NSTB ? 0 : stg + 1;
      }
    }
    // ---- epilogue of the item: slabs live in the window that was just read (all waves must be done with it).
    // Per 32 x 32 block: the raw accumulator goes to the slab as [position][channel] (4 ds_write_b128), comes back as
    // (row 8i + lane/8, channels 4*(lane&7)..+3) for folded BN (this lane's two float4 from LDS), residual, ReLU and a
    // coalesced store.  The residual of block b+1 is requested before block b is processed (block 0: at tap 7).
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    RG_STAMP(3);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    RG_STAMP(4);
    {
      const int phl = (li * a.NC + a.NC - 1) & 1;
      float* cs = reinterpret_cast<float*>(win0 + phl * win_bytes + wave * RG_SLAB);
      const bool relu = (a.act & 15) == 1, post = (a.act & 16) != 0;
      f32x4 sc4[WN], sh4[WN];
#pragma unroll
      for (int cb = 0; cb < WN; ++cb) {
        const int nl = nt * NTILE + (NTILE / 2) * wn + 32 * cb + c4;
        sc4[cb] = *reinterpret_cast<const f32x4*>(sc_lds + nl);
        sh4[cb] = *reinterpret_cast<const f32x4*>(sc_lds + a.ntiles * NTILE + nl);
      }
#pragma unroll
      for (int b = 0; b < NBLK; ++b) {
        const int cb = b >> 1, rb = b & 1;
        if (b + 1 < NBLK) req(b + 1, rq[(b + 1) & 1]);
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<f32x4*>(cs + l31 * 36 + 8 * g + 4 * lh) =
              f32x4{acc[rb][cb][4 * g], acc[rb][cb][4 * g + 1], acc[rb][cb][4 * g + 2], acc[rb][cb][4 * g + 3]};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          f32x4 v = *reinterpret_cast<const f32x4*>(cs + (8 * i + row0) * 36 + c4);
          v = v * sc4[cb] + sh4[cb];
          f32x4 rv = {0.f, 0.f, 0.f, 0.f};
          if constexpr (RES == 2) {
            const bf16x4 t = __builtin_bit_cast(bf16x4, u32x2{rq[b & 1][i][0], rq[b & 1][i][1]});
            rv = f32x4{(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
          } else if constexpr (RES == 1) {
            rv = __builtin_bit_cast(f32x4, rq[b & 1][i]);
          }
          if constexpr (RES != 0) { if (!post) v += rv; }
          if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          if constexpr (RES != 0) { if (post) v += rv; }
          const unsigned eo = eoff(b, i);
          if constexpr (OUTB) {
            bf16x4 o;
            o[0] = (__bf16)v[0]; o[1] = (__bf16)v[1]; o[2] = (__bf16)v[2]; o[3] = (__bf16)v[3];
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), rsC, eo == OOB ? (int)OOB : (int)(eo * 2), 0, 0);
          } else {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsC, eo == OOB ? (int)OOB : (int)(eo * 4), 0, 0);
          }
        }
      }
      RG_STAMP(5);
      extra = NBLK * 4;                                    // these stores sit behind the in-flight pieces in the queue:
      extra_steps = LEAD;                                  // the next LEAD waits reach back over them
    }
    if (mt1 != mt) advance_mtile();
    mt = mt1; nt = nt1;
  }
}


// ---------------------------------------------------------------------------------------------------------------
// PING-PONG variant (8 waves, 256-position tile).  In the kernel above all eight waves cross one barrier per k-tile
// together: right after it nobody has an MFMA to issue — every wave is issuing its loads and waiting for its fragment
// reads — and the matrix pipe idles for that turn, every k-tile (traced with -DRING_TRACE=2: 35-40 % of a bf16
// k-tile).  Here the waves form two groups of four (waves w and w+4 share a SIMD) that run HALF A STEP APART, two
// barriers per k-tile:
//     slot 2t   : group 0  R(t)  = issue the loads of k-tile t+LEAD, read the fragments of k-tile t into registers
//                 group 1  M(t-1) = its MFMAs of k-tile t-1
//     slot 2t+1 : group 0  M(t),   group 1  R(t)
// so that on every SIMD one wave computes while the other stages.  Per-item work (the epilogue of the previous item,
// masks, accumulator clear) sits in the group's R(0) slot, under the other group's MFMAs.
//   * weights: stage (t+LEAD) % (LEAD+1) is rewritten from slot 2t on; its last readers (group 1, k-tile t-1) finished
//     in slot 2t-1.  A wave confirms (counted vmcnt) at the end of R(t) the weights it issued in R(t-1) (k-tile t+1):
//     one slot before group 0 reads them.
//   * windows: the slices of the next phase go out on k-tiles 1..NFL (not 0: the epilogue slabs of the previous item
//     live in that buffer until both groups have passed their R(0)).
//   * M16 (bf16): the wave's 64 positions x NTILE/2 channels as 16 x 16 tiles of v_mfma_f32_16x16x32_bf16 instead of
//     32 x 32 tiles of v_mfma_f32_32x32x16_bf16 — the same LDS fragment reads (16 ds_read_b128 per k-tile), the same MFMA
//     cycles, the same accumulator registers; the chip can hold a higher clock on the 16x16x32 shape under load
//     (MI355X_MICROARCH.md, DVFS give-back item 7) — built, measured 5-10 % slower here, kept as an A/B option (g_ring_m16).  Lane (r = lane & 15, kq = lane >> 4) reads rows r + 16*pt, 16-byte
//     chunk 4*ks + kq: conflict-free under the swizzle chunk ^ (row & 7) (the 32 x 32 pattern needs chunk ^ ((row >> 1) & 7)).
template <bool BF16, int NTILE, int RES, bool OUTB, bool M16 = false>
__global__ __launch_bounds__(512, 2) void conv3x3_ring_pp_kernel(ring_args a) {
  static_assert(!M16 || BF16, "the 16x16x32 form is a bf16 MFMA");
  constexpr int WN = NTILE / 64;
  constexpr int PR = M16 ? 4 : 2;                           // position sub-tiles per lane (of RW rows)
  constexpr int RW = M16 ? 16 : 32;
  constexpr int KS = M16 ? 2 : 4;                           // fragment reads (k-steps) per 128-byte k-tile
  constexpr int CT = M16 ? 2 * WN : WN;                     // channel sub-tiles per wave (of RW channels)
  constexpr int EB = BF16 ? 2 : 4;
  constexpr int NWAVES = 8, NTHR = 512, RG_BM = 256;
  constexpr int LEAD = 2, RG_NSTB = 3;
  constexpr bool AHEAD = NTILE == 64 || !BF16;              // staging-slot address arithmetic done one slot ahead (see the k-loop)
  constexpr int STG_B = NTILE * 128;
  constexpr int NBPW = (NTILE / 8) / NWAVES;
  constexpr int NFL = 4;
  auto sl_of = [](int tap) constexpr -> int { return (tap >= 1 && tap <= NFL) ? 2 : 0; };
  constexpr unsigned OOB = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform: scalar registers, scalar branches
  const int l31 = M16 ? (lane & 15) : (lane & 31), lh = M16 ? (lane >> 4) : (lane >> 5);   // fragment row / k part of the lane
  const int wm = wave >> 1, wn = wave & 1;
  const int grp = wave >> 2;                               // waves w and w + 4 sit on the same SIMD
  auto swz = [](int idx) constexpr -> int { return M16 ? (idx & 7) : ((idx >> 1) & 7); };
  const int win_bytes = a.WPX * 128;
  char* win0 = smem;
  char* bst = smem + 2 * win_bytes;
  char* dump = bst + RG_NSTB * STG_B;

  const int i_begin = blockIdx.x * a.ipw, i_end = min(a.items, i_begin + a.ipw);
  const int nitems = i_end - i_begin;
  if (nitems <= 0) return;
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.M * a.Cin * EB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.N * a.NC * 9 * 128, 0x00020000);
  constexpr int ESZ = OUTB ? 2 : 4, RSZ = RES == 2 ? 2 : 4;
  const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, a.M * a.N * ESZ, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void*)(RES ? a.resid : a.x), 0, RES ? a.M * a.N * RSZ : 0, 0x00020000);
  const int cin_b = a.Cin * EB;
  const int PA = a.WPX >> 3;
  for (int i = tid; i < 256; i += NTHR) reinterpret_cast<unsigned*>(dump)[i] = 0u;
  float* sc_lds = reinterpret_cast<float*>(dump + 1024);
  for (int i = tid; i < a.ntiles * NTILE; i += NTHR) {
    sc_lds[i] = (a.scale && i < a.N) ? a.scale[i] : 1.f;
    sc_lds[a.ntiles * NTILE + i] = (a.shift && i < a.N) ? a.shift[i] : 0.f;
  }
  // (window piece j: pixel idx = 8j + (lane >> 3), j = NWAVES*slot + wave; swz(idx) is a per-lane constant either way)
  const int a_lane = (lane >> 3) * cin_b + ((M16 ? ((lane & 7) ^ (lane >> 3)) : ((lane & 7) ^ (lane >> 4) ^ (4 * (wave & 1)))) << 4);
  int b_lane[NBPW];
#pragma unroll
  for (int k = 0; k < NBPW; ++k) {
    const int r = (wave * NBPW + k) * 8 + (lane >> 3);
    b_lane[k] = r * a.NC * 9 * 128 + (((lane & 7) ^ swz(r)) << 4);
  }
  // The per-lane source offsets of a slot's loads (vector ALU work) are computed apart from their issue (scalar + VMEM
  // only): in the k-loop they are prepared one slot ahead, under the wave's own MFMAs.
  // (a dead request = bit 31 of the offset set: out of range for the descriptor.  OR-ing a scalar select into the offset keeps
  // the k-loop free of control flow — as "cond ? offset : OOB" hipcc branches around the offset arithmetic, and a scalar branch
  // in a staging slot costs more than the arithmetic it skips)
  auto voff_a = [&](int mt_n, int c_n, int j, bool live) -> unsigned {
    const unsigned kill = ((int)live & (int)(j < PA)) ? 0u : OOB;
    return (unsigned)((mt_n * RG_BM - a.W - 1 + 8 * j) * cin_b + c_n * 128 + a_lane) | kill;
  };
  bool abl_pro = true;                                     // (ablation builds: the prologue's loads always go out)
  auto send_a = [&](unsigned voff, int wsel, int j, bool live) {
    char* dst = ((int)live & (int)(j < PA)) ? win0 + wsel * win_bytes + j * 1024 : dump;
    if ((RING_ABL & 4) && !abl_pro) { voff = OOB; dst = dump; }      // the DMA still issues, nothing is fetched, the window keeps its data
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (__attribute__((address_space(3))) void*)dst, 16, (int)voff, 0, 0, 0);
  };
  auto issue_a = [&](int mt_n, int c_n, int wsel, int j, bool live) { send_a(voff_a(mt_n, c_n, j, live), wsel, j, live); };
  auto voff_b = [&](int nt_b, int c, int tap, bool live, int k) -> unsigned {
    return (unsigned)(nt_b * (NTILE * a.NC * 9 * 128) + (c * 9 + tap) * 128 + b_lane[k]) | (live ? 0u : OOB);
  };
  auto send_b = [&](unsigned voff, int stg, int k) {
    char* dst = bst + stg * STG_B + (wave * NBPW + k) * 1024;
    if ((RING_ABL & 8) && !abl_pro) { voff = OOB; dst = dump; }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)dst, 16, (int)voff, 0, 0, 0);
  };
  auto issue_b = [&](int nt_b, int c, int tap, int stg, bool live) {
#pragma unroll
    for (int k = 0; k < NBPW; ++k) send_b(voff_b(nt_b, c, tap, live, k), stg, k);
  };
  const int arow0 = (64 * wm + l31) * 128;
  int kc[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) kc[s] = ((M16 ? 4 : 2) * s + lh) << 4;      // byte offset of the logical 16-B chunk of k-step s
  int boff[CT][KS];
#pragma unroll
  for (int cb = 0; cb < CT; ++cb) {
    const int n = (NTILE / 2) * wn + RW * cb + l31;
#pragma unroll
    for (int s = 0; s < KS; ++s) boff[cb][s] = n * 128 + ((((M16 ? 4 : 2) * s + lh) ^ swz(n)) << 4);
  }
  const float inv_w = 1.0f / (float)a.W, inv_h = 1.0f / (float)a.H;
  int ph[PR], pw[PR];
  int mt = i_begin / a.ntiles, nt = i_begin - mt * a.ntiles;
  {
    const int HW = a.H * a.W;
#pragma unroll
    for (int rb = 0; rb < PR; ++rb) {
      const int m = mt * RG_BM + 64 * wm + RW * rb + l31;
      const int rem = m % HW;
      ph[rb] = rem / a.W;
      pw[rb] = rem - ph[rb] * a.W;
    }
  }
  auto advance_mtile = [&]() {
#pragma unroll
    for (int rb = 0; rb < PR; ++rb) {
      const int x = pw[rb] + RG_BM;
      const int q1 = (int)(((float)x + 0.5f) * inv_w);
      pw[rb] = x - q1 * a.W;
      const int y = ph[rb] + q1;
      const int q2 = (int)(((float)y + 0.5f) * inv_h);
      ph[rb] = y - q2 * a.H;
    }
  };

  // ---- epilogue of one item (coordinates passed in: it runs in the R(0) slot of the NEXT item)
  constexpr int NBLK = 2 * WN;
  // bf16 output (and bf16 / no residual): a lane stores EIGHT channels = 16 bytes per instruction (two passes of 16 rows per
  // 32 x 32 block instead of four passes of 8 rows with 8-byte stores): the store tail is issue-bound, half the store and
  // residual-load instructions at the same bytes (cdna_hip_programming.md T21)
  constexpr bool WIDE = OUTB && RES != 1;
  constexpr int NPASS = WIDE ? 2 : 4, RSTEP = WIDE ? 16 : 8;
  const int row0 = WIDE ? (lane >> 2) : (lane >> 3), c4 = WIDE ? (lane & 3) * 8 : (lane & 7) * 4;
  typedef typename std::conditional<M16, f32x4, f32x16>::type acc_t;
  acc_t acc[PR][CT];                                        // [position sub-tile][channel sub-tile]
  typedef typename std::conditional<RES == 2 && !WIDE, u32x2, u32x4>::type rq_t;
  rq_t rq[NBLK][NPASS];                                    // residual pieces of every block: requested during the last k-tiles
  // Output addressing: element offset = ebase(item) + (32*rb + 8*i) * N + 32*cb, one add per store; a position past
  // M lies past num_records (dropped by the buffer unit), a channel past N is sent there by hand.
  auto ebase_of = [&](int mt_e, int nt_e) -> int {
    return (mt_e * RG_BM + 64 * wm + row0) * a.N + nt_e * NTILE + (NTILE / 2) * wn + c4;
  };
  auto req = [&](int eb, unsigned chm, int b, rq_t* dst) {  // residual quads of block b
    if constexpr (RES != 0 && (RING_ABL & 64) != 0) {
#pragma unroll
      for (int i = 0; i < NPASS; ++i) dst[i] = rq_t{};
    } else if constexpr (RES != 0) {
      const int cb = b >> 1, rb = b & 1;
#pragma unroll
      for (int i = 0; i < NPASS; ++i) {
        const int eo = eb + (32 * rb + RSTEP * i) * a.N + 32 * cb;
        const int bo = (int)((unsigned)(eo * RSZ) | (((chm >> cb) & 1u) ? 0u : OOB));      // (select of constants + OR: no divergent branch around the offset)
        if constexpr (RES == 2 && !WIDE) dst[i] = __builtin_bit_cast(rq_t, __builtin_amdgcn_raw_buffer_load_b64(rsR, bo, 0, 0));
        else dst[i] = __builtin_bit_cast(rq_t, __builtin_amdgcn_raw_buffer_load_b128(rsR, bo, 0, 0));
      }
    }
  };
  auto chmask_of = [&](int nt_e) -> unsigned {             // bit cb: this lane's 4 channels of column block cb exist
    unsigned m = 0;
#pragma unroll
    for (int cb = 0; cb < WN; ++cb)
      if (nt_e * NTILE + (NTILE / 2) * wn + 32 * cb + c4 < a.N) m |= 1u << cb;
    return m;
  };
  const float act_floor = (a.act & 15) == 1 ? 0.f : -__builtin_inff();      // y = max(x, floor): ReLU or identity
  const bool post = (a.act & 16) != 0;
  auto epilogue = [&](int mt_e, int nt_e, int phl) {
    if constexpr ((RING_ABL & 32) != 0) {                   // no epilogue: the accumulators stay live
#pragma unroll
      for (int rb = 0; rb < PR; ++rb)
#pragma unroll
        for (int cb = 0; cb < CT; ++cb)
#pragma unroll
          for (int r = 0; r < (M16 ? 4 : 16); ++r) { const float t = acc[rb][cb][r]; asm volatile("" :: "v"(t)); }
    } else {
    float* cs = reinterpret_cast<float*>(win0 + phl * win_bytes + wave * RG_SLAB);
    const int eb = ebase_of(mt_e, nt_e);
    const unsigned chm = chmask_of(nt_e);
    constexpr int NV = WIDE ? 2 : 1;                        // f32x4 groups per lane and pass
    f32x4 sc4[WN][NV], sh4[WN][NV];
#pragma unroll
    for (int cb = 0; cb < WN; ++cb)
#pragma unroll
      for (int h = 0; h < NV; ++h) {
        const int nl = nt_e * NTILE + (NTILE / 2) * wn + 32 * cb + c4 + 4 * h;
        sc4[cb][h] = *reinterpret_cast<const f32x4*>(sc_lds + nl);
        sh4[cb][h] = *reinterpret_cast<const f32x4*>(sc_lds + a.ntiles * NTILE + nl);
      }
#pragma unroll
    for (int b = 2; b < NBLK; ++b) req(eb, chm, b, rq[b]);
#pragma unroll
    for (int b = 0; b < NBLK; ++b) {
      const int cb = b >> 1, rb = b & 1;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if constexpr (M16) {       // 16 x 16 tiles: (lane & 15) = position, registers = channels 4*(lane >> 4) .. +3 of the tile
          const int dp = g >> 1, dc = g & 1;
          *reinterpret_cast<f32x4*>(cs + (16 * dp + l31) * 36 + 16 * dc + 4 * lh) = acc[2 * rb + dp][2 * cb + dc];
        } else {
          *reinterpret_cast<f32x4*>(cs + l31 * 36 + 8 * g + 4 * lh) =
              f32x4{acc[rb][cb][4 * g], acc[rb][cb][4 * g + 1], acc[rb][cb][4 * g + 2], acc[rb][cb][4 * g + 3]};
        }
      }
#pragma unroll
      for (int i = 0; i < NPASS; ++i) {
        f32x4 v[NV];
#pragma unroll
        for (int h = 0; h < NV; ++h) {
          v[h] = *reinterpret_cast<const f32x4*>(cs + (RSTEP * i + row0) * 36 + c4 + 4 * h);
          v[h] = v[h] * sc4[cb][h] + sh4[cb][h];
        }
        if constexpr (RES != 0) {
          f32x4 rv[NV];
          if constexpr (WIDE) {
            const bf16x8 t = __builtin_bit_cast(bf16x8, rq[b][i]);
            rv[0] = f32x4{(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
            rv[1] = f32x4{(float)t[4], (float)t[5], (float)t[6], (float)t[7]};
          } else if constexpr (RES == 2) {
            const bf16x4 t = __builtin_bit_cast(bf16x4, rq[b][i]);
            rv[0] = f32x4{(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
          } else {
            rv[0] = __builtin_bit_cast(f32x4, rq[b][i]);
          }
          const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int h = 0; h < NV; ++h) {
            v[h] += post ? zero : rv[h];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[h][e] = fmaxf(v[h][e], act_floor);
            v[h] += post ? rv[h] : zero;
          }
        } else {
#pragma unroll
          for (int h = 0; h < NV; ++h)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[h][e] = fmaxf(v[h][e], act_floor);
        }
        const int eo = eb + (32 * rb + RSTEP * i) * a.N + 32 * cb;
        const int bo = (int)((unsigned)(eo * ESZ) | (((chm >> cb) & 1u) ? 0u : OOB));
        if constexpr ((RING_ABL & 16) != 0) {                 // no global stores: the values stay live
#pragma unroll
          for (int h = 0; h < NV; ++h) asm volatile("" :: "v"(v[h]), "v"(bo));
        } else if constexpr (WIDE) {
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) { o[e] = (__bf16)v[0][e]; o[4 + e] = (__bf16)v[1][e]; }
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rsC, bo, 0, 0);
        } else if constexpr (OUTB) {
          bf16x4 o;
          o[0] = (__bf16)v[0][0]; o[1] = (__bf16)v[0][1]; o[2] = (__bf16)v[0][2]; o[3] = (__bf16)v[0][3];
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), rsC, bo, 0, 0);
        } else {
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[0]), rsC, bo, 0, 0);
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // slab reads done before this wave's loads may land in that window
    }
  };

  // byte offsets (from the LDS base) of this lane's pixel fragments of k-tile `tap`.  Rows start on 128-byte boundaries,
  // so row + (chunk << 4) == row ^ (chunk << 4) and the swizzle (chunk ^ sw) << 4 folds into two XORs.
  typedef const __attribute__((address_space(3))) char* lds_cptr;
  lds_cptr a_addr[PR][KS];
  const lds_cptr lds0 = (lds_cptr)smem;
  const unsigned zrow_off = (unsigned)(dump - smem);
  auto frag_addr = [&](int tap, int win_off, const unsigned* mask, lds_cptr (*out)[KS]) {
    const int toff = (tap / 3) * a.W + (tap % 3);
#pragma unroll
    for (int rb = 0; rb < PR; ++rb) {
      const int idx = 64 * wm + RW * rb + l31 + toff;
      const unsigned row = ((mask[rb] >> tap) & 1u) ? (unsigned)(win_off + arow0 + (RW * rb + toff) * 128) : zrow_off;
      const unsigned rsw = row ^ (unsigned)(swz(idx) << 4);
#pragma unroll
      for (int s = 0; s < KS; ++s) out[rb][s] = lds0 + (rsw ^ (unsigned)kc[s]);
    }
  };

  unsigned dma_b[NBPW], dma_a[2];                          // source offsets of the next staging slot's loads

  // ---- prologue: window of the first phase, weights of k-tiles 0 .. LEAD-1; everything landed and published
  const int total_ph = nitems * a.NC;
  for (int j = wave; j < PA; j += NWAVES) issue_a(mt, 0, 0, j, true);
  if constexpr ((RING_ABL & 4) != 0) { for (int j = wave; j < PA; j += NWAVES) issue_a(mt, 0, 1, j, true); }
#pragma unroll
  for (int s = 0; s < LEAD; ++s) issue_b(nt, 0, s, s, true);                          // (LEAD <= 3 < 9 k-tiles of a chunk)
  if constexpr ((RING_ABL & 8) != 0) issue_b(nt, 0, LEAD, LEAD, true);
  wait_vm<0>();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (grp == 1) __builtin_amdgcn_s_barrier();              // group 1 runs one slot behind
  abl_pro = false;
  RG_CLK(0);
  int mt_p = 0, nt_p = 0, phl_p = 0;
  bool have_prev = false;

  for (int li = 0; li < nitems; ++li) {
    int mt1 = mt, nt1 = nt + 1;
    if (nt1 == a.ntiles) { nt1 = 0; ++mt1; }
    const bool more = li + 1 < nitems;
    unsigned mask[PR];
#pragma unroll
    for (int rb = 0; rb < PR; ++rb) {
      const int m = mt * RG_BM + 64 * wm + RW * rb + l31;
      unsigned mk = 0;
      if (m < a.M) {
        unsigned colm = 0;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
          if ((unsigned)(pw[rb] - 1 + kw) < (unsigned)a.W) colm |= 1u << kw;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
          if ((unsigned)(ph[rb] - 1 + kh) < (unsigned)a.H) mk |= colm << (3 * kh);
      }
      mask[rb] = mk;
    }
    // one channel chunk (9 k-tiles).  Whether it is the item's LAST chunk is a compile-time parameter (two instances of the body):
    // the counted waits, the early residual requests and the closing barrier depend on it, and as run-time conditions they put
    // half a dozen scalar branches into every staging slot (wait_vm_n alone is a branch tree over the count)
    auto chunk = [&](auto last_t, const int c) {
      constexpr int LT = decltype(last_t)::value;            // 0: not the last chunk, 1: the last chunk, 2: decided at run time
      const bool last_c = LT == 2 ? (c + 1 == a.NC) : (LT != 0);
      const int phg = li * a.NC + c;
      const int win_off = (phg & 1) * win_bytes;
      const bool has_next = phg + 1 < total_ph;
      const int mt_n = last_c ? mt1 : mt, c_n = last_c ? 0 : c + 1;
      auto slot_voff = [&](int tap) {                      // source offsets of the loads staging slot `tap` issues
#pragma unroll
        for (int k = 0; k < NBPW; ++k)
          dma_b[k] = tap + LEAD < 9 ? voff_b(nt, c, tap + LEAD, true, k)
                                    : (!last_c ? voff_b(nt, c + 1, tap + LEAD - 9, true, k) : voff_b(nt1, 0, tap + LEAD - 9, more, k));
#pragma unroll
        for (int i = 0; i < sl_of(tap); ++i) dma_a[i] = voff_a(mt_n, c_n, (2 * (tap - 1) + i) * NWAVES + wave, has_next);
      };
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        // ================= R slot (at raised priority: its few instructions go between the other group's MFMAs)
        if (RING_PRIO_ON) __builtin_amdgcn_s_setprio(2);
        if (tap == 4 && c == 0) RG_STAMP(2);
        if (tap == 4 && c == 0) RG_SLOT(0);
        // operations this wave issues in this R slot / issued in its previous one (folds: tap is unrolled)
        // (residual of output blocks 0 and 1: requested at k-tiles 7 and 8 of the item's last chunk — an HBM round trip
        //  ahead of the epilogue that adds them; the later blocks when the epilogue starts, two blocks ahead)
        auto n_res = [](int t) constexpr -> int { return (RES != 0 && !(RING_ABL & 64) && t >= 7) ? NPASS : 0; };
        const int n_now = NBPW + sl_of(tap) + (last_c ? n_res(tap) : 0);
        const char* bs = bst + (tap % RG_NSTB) * STG_B;     // 9 k-tiles per chunk, 3 stages: the stage of k-tile `tap` is tap % 3 — static
        f32x4 afr[PR][KS], bfr[CT][KS];
        {
          // AHEAD: the pixel-fragment addresses and load offsets of this k-tile were computed in the wave's previous MFMA
          // slot (below) — while the OTHER group's MFMAs run back to back the vector ALU port of the SIMD is theirs, and
          // a staging slot with VALU work in it lasts as long as their MFMA slot (traced: fp32 2430 vs 2100 cycles,
          // 280 without).  Either way the VALU work takes its time on the shared port: worth it where the staging slot
          // is the longer one (64-channel tile: 12 reads per 8 MFMAs, layer1 0.50 / 0.55 vs 0.54 / 0.61 ms; fp32), not
          // where the MFMA slot is (128-channel tile: 0.37 vs 0.34 ms on layer2).
          if constexpr (AHEAD) {
            if (tap == 0) frag_addr(0, win_off, mask, a_addr);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
              bfr[0][s] = *reinterpret_cast<const f32x4*>(bs + boff[0][s]);
#pragma unroll
              for (int rb = 0; rb < PR; ++rb) afr[rb][s] = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(a_addr[rb][s]);
            }
          } else {
            const int toff = (tap / 3) * a.W + (tap % 3);
            unsigned arow_sw[PR];
#pragma unroll
            for (int rb = 0; rb < PR; ++rb) {
              const int idx = 64 * wm + RW * rb + l31 + toff;
              const unsigned row = ((mask[rb] >> tap) & 1u) ? (unsigned)(win_off + arow0 + (RW * rb + toff) * 128) : zrow_off;
              arow_sw[rb] = row ^ (unsigned)(swz(idx) << 4);
            }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
              bfr[0][s] = *reinterpret_cast<const f32x4*>(bs + boff[0][s]);
#pragma unroll
              for (int rb = 0; rb < PR; ++rb) afr[rb][s] = *reinterpret_cast<const f32x4*>(smem + (arow_sw[rb] ^ (unsigned)kc[s]));
            }
          }
#pragma unroll
          for (int cb = 1; cb < CT; ++cb)
#pragma unroll
            for (int s = 0; s < KS; ++s) bfr[cb][s] = *reinterpret_cast<const f32x4*>(bs + boff[cb][s]);
        }
        if constexpr ((RING_ABL & 2) != 0) {                 // no fragment reads: lane-dependent junk operands instead
#pragma unroll
          for (int s = 0; s < KS; ++s) {
#pragma unroll
            for (int rb = 0; rb < PR; ++rb) afr[rb][s] = f32x4{(float)(lane * 3 + s), 1.5f + rb, -0.75f * lane, 0.3f};
#pragma unroll
            for (int cb = 0; cb < CT; ++cb) bfr[cb][s] = f32x4{0.01f * lane, -2.5f + cb, 0.125f * s, 1.f};
          }
        }
        __builtin_amdgcn_sched_barrier(0);                   // (the reads go out first: their latency runs under the issue below)
        {
          const int s2 = (tap + LEAD) % RG_NSTB;
          if constexpr (AHEAD) {
            if (tap == 0) slot_voff(0);                      // (first k-tile of a phase: no previous MFMA slot of this phase)
#pragma unroll
            for (int k = 0; k < NBPW; ++k) send_b(dma_b[k], s2, k);
#pragma unroll
            for (int i = 0; i < sl_of(tap); ++i)
              send_a(dma_a[i], (phg + 1) & 1, (2 * (tap - 1) + i) * NWAVES + wave, has_next);
          } else {                                           // offsets computed as the loads go out
            if (tap + LEAD < 9) issue_b(nt, c, tap + LEAD, s2, true);
            else if (!last_c) issue_b(nt, c + 1, tap + LEAD - 9, s2, true);
            else issue_b(nt1, 0, tap + LEAD - 9, s2, more);
#pragma unroll
            for (int i = 0; i < sl_of(tap); ++i)
              issue_a(mt_n, c_n, (phg + 1) & 1, (2 * (tap - 1) + i) * NWAVES + wave, has_next);
          }
          if constexpr (RES != 0) {
            if (tap >= 7 && last_c) req(ebase_of(mt, nt), chmask_of(nt), tap - 7, rq[tap - 7]);
          }
        }
        if (tap == 4 && c == 0) RG_SLOT(1);
        // confirm the weights this wave issued in its previous R slot (in-order completion), all fragment reads returned
        // (a slot issues weights first, then window slices / the early residual: those younger operations of the
        //  oldest slot still counted may stay in flight too — the slices are HBM reads, not needed before the next phase)
        // (tap 0 of an item's first chunk: between the previous slot's weights and this slot's loads lie the previous item's early
        //  residual requests of tap 8 and its epilogue — the later blocks' residual requests and every store; counting them too
        //  keeps this wait from draining the epilogue's stores.  In the launch's first item nothing older is in flight.  Every other
        //  tap: tap is unrolled and last_c a template parameter — the switch folds to one s_waitcnt)
        constexpr int N_EPI = (RING_ABL & 32) ? 0 : ((RING_ABL & 16) ? 0 : NBLK * NPASS) + ((RES != 0 && !(RING_ABL & 64)) ? (NBLK - 2) * NPASS : 0);
        const int young = sl_of((tap + 8) % 9) + (tap == 0 ? (c == 0 ? n_res(8) + N_EPI : 0) : (last_c ? n_res(tap - 1) : 0));
        wait_vm_n(n_now + young);
        if (tap == 4 && c == 0) RG_SLOT(2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (tap == 4 && c == 0) RG_STAMP(3);
        if (tap == 4 && c == 0) RG_SLOT(3);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (tap == 4 && c == 0) RG_STAMP(4);
        if (RING_PRIO_ON) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        // ================= M slot
#pragma unroll
        for (int cb = 0; cb < CT; ++cb)
#pragma unroll
          for (int s = 0; s < KS; ++s) {
            if constexpr ((RING_ABL & 1) != 0) {             // no MFMAs: the operands stay live
#pragma unroll
              for (int rb = 0; rb < PR; ++rb) asm volatile("" :: "v"(bfr[cb][s]), "v"(afr[rb][s]));
            } else if constexpr (M16) {
#pragma unroll
              for (int rb = 0; rb < PR; ++rb)
                acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bfr[cb][s]), __builtin_bit_cast(bf16x8, afr[rb][s]),
                                                                      acc[rb][cb], 0, 0, 0);
            } else if constexpr (BF16) {
#pragma unroll
              for (int rb = 0; rb < 2; ++rb)
                acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bfr[cb][s]), __builtin_bit_cast(bf16x8, afr[rb][s]),
                                                                      acc[rb][cb], 0, 0, 0);
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(bfr[cb][s][e], afr[rb][s][e], acc[rb][cb], 0, 0, 0);
            }
          }
        if (AHEAD && tap < 8) {                              // next k-tile's fragment addresses and load offsets: VALU work under OUR MFMAs
          frag_addr(tap + 1, win_off, mask, a_addr);
          slot_voff(tap + 1);
#pragma unroll
          for (int k = 0; k < NBPW; ++k) asm volatile("" : "+v"(dma_b[k]));
#pragma unroll
          for (int i = 0; i < 2; ++i) asm volatile("" : "+v"(dma_a[i]));
#pragma unroll
          for (int rb = 0; rb < PR; ++rb)
#pragma unroll
            for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(a_addr[rb][s]));      // (materialised HERE: not sunk to the reads)
        }
        __builtin_amdgcn_sched_barrier(0);
        if (tap == 4 && c == 0) RG_STAMP(5);
        if (!(tap == 8 && last_c && grp == 1)) __builtin_amdgcn_s_barrier();      // (group 1, end of an item: after its epilogue)
        asm volatile("" ::: "memory");
        if (tap == 4 && c == 0) RG_STAMP(6);
      }
    };
    // ---- head of the item's first staging slot (R(0) of chunk 0; ONE copy, in front of both instances of the chunk body)
    if (RING_PRIO_ON) __builtin_amdgcn_s_setprio(2);
    RG_STAMP(0);
    // the previous item's epilogue, both groups side by side: group 0 is in its R(0) slot, group 1 — one slot
    // behind — still in its M(8) slot, whose closing barrier it takes only now
    if (have_prev) {
      epilogue(mt_p, nt_p, phl_p);
      if (grp == 1) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
    }
#pragma unroll
    for (int rb = 0; rb < PR; ++rb)
#pragma unroll
      for (int cb = 0; cb < CT; ++cb)
#pragma unroll
        for (int r = 0; r < (M16 ? 4 : 16); ++r) acc[rb][cb][r] = 0.f;
    RG_STAMP(1);
    // (two chunks per item: every chunk would alternate between the two instances — measured 2-3 % slower than one body with the
    //  run-time conditions, 2 x 1152-channel layers; from four chunks on the instance without the end-of-item code repeats: -4..-7 %)
    if (a.NC == 2) {
      chunk(int_k<2>{}, 0);
      chunk(int_k<2>{}, 1);
    } else {
      for (int c = 0; c + 1 < a.NC; ++c) chunk(int_k<0>{}, c);
      chunk(int_k<1>{}, a.NC - 1);
    }
    mt_p = mt; nt_p = nt; phl_p = (li * a.NC + a.NC - 1) & 1; have_prev = true;
    if (mt1 != mt) advance_mtile();
    mt = mt1; nt = nt1;
  }
  // ---- tail: the last item's epilogue, group 0 one slot before group 1
  RG_CLK(1);
  epilogue(mt_p, nt_p, phl_p);
  __builtin_amdgcn_s_barrier();
}



#ifdef CADRE_AB_KERNELS
// ---------------------------------------------------------------------------------------------------------------
// PING-PONG, G K-TILES PER SLOT (bf16; round 4) — A/B BUILD ONLY (CADRE_BUILD_AB=1, then CADRE_RING_G=2): built, parity-green,
// and measured SLOWER than the kernel above on every layer (same box, 2048 frames at 288 x 288: 64-channel tile 1.01 / 1.46
// vs 0.96 / 1.04 ms, 128-channel tile 0.82 / 0.90 vs 0.65 / 0.75 ms; DESIGN.md 3.3 has the ablation and the slot trace).  The kernel above pays its fixed slot cost — two barriers, the turn in
// which the staging group's reads come back, the priority flips — once per k-tile: traced, 8 MFMAs = 256 pipe cycles in a
// 380-cycle slot + ~100 of barrier on the 64-channel tile, 16 MFMAs = 512 in 615 + 80-250 on the 128-channel tile.  Here a
// slot carries G consecutive k-tiles (G = 3 with the 64-channel tile: one row of taps; G = 2 with the 128-channel tile),
// the fixed cost once per 24 / 32 MFMAs.  What changes with it:
//   * the two groups split the CHANNELS of the tile, not its positions: wave w = (position block w & 3, group w >> 2),
//     group g owns channels [g * NTILE/2, (g+1) * NTILE/2).  A group then reads only ITS half of every weight stage, and
//     a half-stage is written by the group that reads it: no weight byte crosses the groups, so two sets of G half-stages
//     (read one, fill the other) replace the three stages — a group issues the weights of super-step u+1 at the start
//     of R(u), right after its own last read of that set (end of M(u-1)), and confirms them at the END of M(u) by its
//     counted vmcnt: almost two slots of lead for every piece, for both groups alike;
//   * R(u) reads the fragments of the super-step's FIRST k-tile only; inside M(u) the fragments of k-tile j + 1 are read
//     behind the MFMAs of k-tile j, one k-step at a time into the registers that k-step's MFMAs have just released (MFMA
//     order: k-step outer) — the register budget of ONE k-tile of fragments, as in the kernel above;
//   * window slices go out on a static per-group schedule (SL below) that follows from who reads which buffer when
//     (group 1 runs one slot behind: a buffer whose last k-tile is read inside an M slot is free for group 1's next R
//     slot but not yet for group 0's), confirmed by in-order completion: a slice issued in R(x) is complete when the
//     weights issued in R(x+1) are confirmed (end of M(x+1)) — 3.9 slots of G k-tiles instead of 4 slots of one k-tile
//     for an HBM round trip;
//   * the epilogue needs no LDS: the accumulator holds (lane -> position, register quad -> 4 channels, lane half -> which
//     4 of 8); one v_permlane32_swap per register pair leaves every lane with EIGHT consecutive channels = one 16-byte
//     store (and one 16-byte residual load) per 32 x 16 half-tile; folded BN comes from the LDS table in that layout.
//     No slab in a window, so the next phase's slices may start in the item's first slot, and no LDS round trip sits
//     between the last MFMA and the stores.  Both groups run their epilogue in their own R slot of the next item's first
//     super-step — under the other group's MFMAs.
// Needs (9 * NC) % G == 0 (G = 2: an even number of 128-byte channel chunks); other shapes stay on the kernel above.
template <int G_> struct pp2_geo {
  static constexpr int L = (G_ == 2) ? 18 : 9;             // k-tiles per period = lcm(9, G)
  static constexpr int P = L / G_;                         // super-steps per period (odd: 9 or 3)
  static constexpr int CPP = L / 9;                        // channel chunks per period
};

template <int NTILE, int RES, bool OUTB, int G>
__global__ __launch_bounds__(512, 2) void conv3x3_ring_pp2_kernel(ring_args a) {
  static_assert(G == 2 || G == 3, "G k-tiles per slot");
  static_assert(RES == 0 || RES == 2, "bf16 operands take a bf16 residual");
  using geo = pp2_geo<G>;
  constexpr int P = geo::P, CPP = geo::CPP;
  constexpr int WN = NTILE / 64;                           // 32-channel blocks per wave
  constexpr int RG_BM = 256, NWAVES = 8;
  constexpr int PPK = NTILE / 16;                          // 1-KiB pieces of a half-stage (NTILE/2 rows x 128 B): 4 or 8
  constexpr int NQ = PPK / 4;                              // pieces of one half-stage per wave (4 waves per group)
  constexpr int NWP = G * NQ;                              // weight pieces per wave and super-step: piece i = (k-tile i / NQ, rows 8 (pb + 4 (i % NQ)) ..)
  constexpr int HS_B = (NTILE / 2) * 128, SET_B = G * HS_B;
  constexpr bool AHEAD = true;                             // R(ss + 1)'s addresses / offsets computed under the wave's own MFMAs of M(ss)
  constexpr unsigned OOB = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int pb = wave & 3, grp = wave >> 2;                // position block, channel half (waves w and w + 4 share a SIMD)
  const int win_bytes = a.WPX * 128;
  char* win0 = smem;
  char* wts = smem + 2 * win_bytes;                        // [group][set][k-tile of the super-step][NTILE/2 rows][128 B]
  char* dump = wts + 4 * SET_B;
  const int i_begin = blockIdx.x * a.ipw, i_end = min(a.items, i_begin + a.ipw);
  const int nitems = i_end - i_begin;
  if (nitems <= 0) return;
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.M * a.Cin * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.N * a.NC * 9 * 128, 0x00020000);
  constexpr int ESZ = OUTB ? 2 : 4;
  const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, a.M * a.N * ESZ, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void*)(RES ? a.resid : a.x), 0, RES ? a.M * a.N * 2 : 0, 0x00020000);
  const int cin_b = a.Cin * 2;
  const int PA = a.WPX >> 3;
  const int NPER = a.NC / CPP;
  for (int i = tid; i < 256; i += 512) reinterpret_cast<unsigned*>(dump)[i] = 0u;      // the ZERO ROW (halo taps) and the dummy DMA target
  float* sc_lds = reinterpret_cast<float*>(dump + 1024);
  for (int i = tid; i < a.ntiles * NTILE; i += 512) {
    sc_lds[i] = (a.scale && i < a.N) ? a.scale[i] : 1.f;
    sc_lds[a.ntiles * NTILE + i] = (a.shift && i < a.N) ? a.shift[i] : 0.f;
  }
  auto swz = [](int idx) constexpr -> int { return (idx >> 1) & 7; };
  // window piece j = 8 n + wave (n = 0 .. 7): pixel 8 j + (lane >> 3), LDS chunk (lane & 7) <- source chunk ^ swz(pixel);
  // swz(pixel) = (lane >> 4) ^ 4 (j & 1) and j has the parity of the wave: a per-lane constant
  const int a_lane = (lane >> 3) * cin_b + ((((lane & 7) ^ (lane >> 4) ^ (4 * (wave & 1)))) << 4);
  int b_lane[NQ];
#pragma unroll
  for (int k = 0; k < NQ; ++k) {
    const int r = 8 * (pb + 4 * k) + (lane >> 3);          // row inside the group's half-stage
    b_lane[k] = (grp * (NTILE / 2) + r) * a.NC * 9 * 128 + (((lane & 7) ^ swz(r)) << 4);
  }
  bool abl_pro = true;                                     // (ablation builds: the prologue's loads always go out)
  auto send_a = [&](unsigned voff, int wsel, int j, bool live) {
    char* dst = (live && j < PA) ? win0 + wsel * win_bytes + j * 1024 : dump;
    if ((RING_ABL & 4) && !abl_pro) { voff = OOB; dst = dump; }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (__attribute__((address_space(3))) void*)dst, 16, (int)voff, 0, 0, 0);
  };
  auto voff_a = [&](int mt_n, int c_n, int j, bool live) -> unsigned {
    return (live && j < PA) ? (unsigned)((mt_n * RG_BM - a.W - 1 + 8 * j) * cin_b + c_n * 128 + a_lane) : OOB;
  };
  auto voff_b = [&](int nt_b, int c, int tap, bool live, int k) -> unsigned {
    return live ? (unsigned)(nt_b * (NTILE * a.NC * 9 * 128) + (c * 9 + tap) * 128 + b_lane[k]) : OOB;
  };
  auto send_b = [&](unsigned voff, int set, int i) {
    char* dst = wts + (grp * 2 + set) * SET_B + (i / NQ) * HS_B + (pb + 4 * (i % NQ)) * 1024;
    if ((RING_ABL & 8) && !abl_pro) { voff = OOB; dst = dump; }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)dst, 16, (int)voff, 0, 0, 0);
  };
  // ---- fragment addressing
  const int base_idx = 64 * pb + l31;
  int kc[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) kc[s] = (2 * s + lh) << 4;
  int boff[4];                                             // weight fragment of column block 0 (block cb: + 32 * 128 * cb)
#pragma unroll
  for (int s = 0; s < 4; ++s) boff[s] = l31 * 128 + (((2 * s + lh) ^ swz(l31)) << 4);
  const float inv_w = 1.0f / (float)a.W, inv_h = 1.0f / (float)a.H;
  int ph[2], pw[2];
  int mt = i_begin / a.ntiles, nt = i_begin - mt * a.ntiles;
  {
    const int HW = a.H * a.W;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const int m = mt * RG_BM + 64 * pb + 32 * rb + l31;
      const int rem = m % HW;
      ph[rb] = rem / a.W;
      pw[rb] = rem - ph[rb] * a.W;
    }
  }
  auto advance_mtile = [&]() {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const int x = pw[rb] + RG_BM;
      const int q1 = (int)(((float)x + 0.5f) * inv_w);
      pw[rb] = x - q1 * a.W;
      const int y = ph[rb] + q1;
      const int q2 = (int)(((float)y + 0.5f) * inv_h);
      ph[rb] = y - q2 * a.H;
    }
  };
  typedef const __attribute__((address_space(3))) char* lds_cptr;
  typedef const __attribute__((address_space(3))) f32x4* lds_f4;
  const lds_cptr lds0 = (lds_cptr)smem;
  const unsigned zrow_off = (unsigned)(dump - smem);
  const unsigned wts_off = (unsigned)(wts - smem);
  lds_cptr a_addr[2][4];                                    // pixel fragment addresses of the next R slot's k-tile
  // (bi = 64 pb + l31 and the two halo masks come in as per-super-step copies behind an empty asm: every address is then
  //  computed where it is used — left to itself the compiler hoists the tap-dependent parts of all nine taps out of the
  //  period loop and spills them)
  auto frag_addr = [&](int tap, int win_off, int bi, unsigned mk0, unsigned mk1, int Wx, lds_cptr (*out)[4]) {
    const int toff = (tap / 3) * Wx + (tap % 3);
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const int idx = bi + 32 * rb + toff;                 // window row of this lane's pixel; its bytes start at idx * 128
      const unsigned row = (((rb ? mk1 : mk0) >> tap) & 1u) ? (unsigned)(win_off + (idx << 7)) : zrow_off;
      const unsigned rsw = row ^ (unsigned)(swz(idx) << 4);
#pragma unroll
      for (int s = 0; s < 4; ++s) out[rb][s] = lds0 + (rsw ^ (unsigned)kc[s]);
    }
  };

  // ---- epilogue (no LDS slab): tile b = (cb, rb); after the lane-half exchange lane (l31, lh) holds, for position
  // 32 rb + l31, channels 32 cb + 16 h + 8 lh .. + 7 (h = 0, 1): one 16-byte store / residual load per (tile, h)
  constexpr int NT2 = 2 * WN;
  f32x16 acc[2][WN];
  u32x4 rq[NT2][2];
  auto ebase_of = [&](int mt_e, int nt_e) -> int {
    return (mt_e * RG_BM + 64 * pb + l31) * a.N + nt_e * NTILE + grp * (NTILE / 2) + 8 * lh;
  };
  auto chmask_of = [&](int nt_e) -> unsigned {             // bit cb: the 32-channel block exists (N % 32 == 0)
    unsigned m = 0;
#pragma unroll
    for (int cb = 0; cb < WN; ++cb)
      if (nt_e * NTILE + grp * (NTILE / 2) + 32 * cb < a.N) m |= 1u << cb;
    return m;
  };
  auto req = [&](int eb, unsigned chm, int b) {
    if constexpr (RES != 0 && (RING_ABL & 64) == 0) {
      const int cb = b >> 1, rb = b & 1;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int eo = eb + 32 * rb * a.N + 32 * cb + 16 * h;
        rq[b][h] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsR, ((chm >> cb) & 1u) ? eo * 2 : (int)OOB, 0, 0));
      }
    }
  };
  const float act_floor = (a.act & 15) == 1 ? 0.f : -__builtin_inff();      // y = max(x, floor): ReLU or identity
  const bool post = (a.act & 16) != 0;
  auto epilogue = [&](int mt_e, int nt_e) {
    if constexpr ((RING_ABL & 32) != 0) {
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < WN; ++cb)
#pragma unroll
          for (int r = 0; r < 16; ++r) { const float t = acc[rb][cb][r]; asm volatile("" :: "v"(t)); }
    } else {
    const int eb = ebase_of(mt_e, nt_e);
    const unsigned chm = chmask_of(nt_e);
#pragma unroll
    for (int b = 0; b < NT2; ++b) {
      const int cb = b >> 1, rb = b & 1;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        // register quads g = 2h (channels 16h + 4 lh ..) and g = 2h + 1 (16h + 8 + 4 lh ..): exchanging the upper lane
        // half of the first with the lower lane half of the second leaves channels 16h + 8 lh .. + 7 in this lane
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // v_permlane32_swap x, y: x <- [x.lo, y.lo], y <- [x.hi, y.hi].  As inline asm: ROCm 7.2's
          // __builtin_amdgcn_permlane32_swap hands back its first result for BOTH elements of the pair (measured: channels
          // 0-3 twice), and __builtin_bit_cast of a vector-element lvalue reads element 0.  The accumulator was written by
          // MFMAs at least a barrier ago; the s_nop covers the VALU -> permlane operand wait states hipcc does not see here.
          float fx = acc[rb][cb][8 * h + e], fy = acc[rb][cb][8 * h + 4 + e];
          asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(fx), "+v"(fy));
          v[e] = fx;
          v[4 + e] = fy;
        }
        const int nl = nt_e * NTILE + grp * (NTILE / 2) + 32 * cb + 16 * h + 8 * lh;
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(sc_lds + nl), s1 = *reinterpret_cast<const f32x4*>(sc_lds + nl + 4);
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(sc_lds + a.ntiles * NTILE + nl);
        const f32x4 t1 = *reinterpret_cast<const f32x4*>(sc_lds + a.ntiles * NTILE + nl + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = v[e] * s0[e] + t0[e]; v[4 + e] = v[4 + e] * s1[e] + t1[e]; }
        if constexpr (RES != 0) {
          const bf16x8 t = __builtin_bit_cast(bf16x8, rq[b][h]);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float rv = (float)t[e];
            v[e] += post ? 0.f : rv;
            v[e] = fmaxf(v[e], act_floor);
            v[e] += post ? rv : 0.f;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], act_floor);
        }
        const int eo = eb + 32 * rb * a.N + 32 * cb + 16 * h;
        const int bo = ((chm >> cb) & 1u) ? eo * ESZ : (int)OOB;
        if constexpr ((RING_ABL & 16) != 0) {
#pragma unroll
          for (int e = 0; e < 8; ++e) asm volatile("" :: "v"(v[e]), "v"(bo));
        } else if constexpr (OUTB) {
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rsC, bo, 0, 0);
        } else {
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{v[0], v[1], v[2], v[3]}), rsC, bo, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{v[4], v[5], v[6], v[7]}), rsC, bo == (int)OOB ? bo : bo + 16, 0, 0);
        }
      }
    }
    }
  };

  // ---- static slice schedule: SL(g, ss) slices of a phase's window go out in R(ss) of a wave of group g (slot of G0's
  // R(ss): 2 ss, M(ss): 2 ss + 1; group 1 one later).  A slice issued in R(x) is complete at the end of M(x + 1).
  // G = 3, one chunk per period: the next phase's buffer was read last in the previous period's M(2) — by group 1 in this
  //   period's slot 0 — and is read first in the next period's R(0) (slot 6): group 0 may issue in R(1) only, group 1 in
  //   R(0) and R(1) (its R(1) slices forced complete at the end of its R(2), below).  Group 1's waves therefore carry two
  //   thirds of the pieces: j = pb + 4 n (n = 0 .. 9), group 0's j = 40 + pb + 4 n (n = 0 .. 4); 60 pieces >= WPX / 8.
  // G = 2, two chunks per period: chunk 1's buffer was read last in the previous period's M(8) (group 1: slot 0) and is
  //   read first in M(4) (k-tile 9: slot 9): group 0 issues in R(1), R(2), group 1 in R(0) .. R(2); chunk 0's buffer is
  //   read last in R(4) (k-tile 8) and first in the next period's R(0): R(5) .. R(7) for group 0, R(5), R(6) for group 1
  //   (its slices complete one slot later).  Pieces j = 8 n + wave, n = 0 .. 7.
  auto SL = [](int g, int ss) constexpr -> int {
    if (G == 3) return g == 0 ? (ss == 1 ? 5 : 0) : (ss < 2 ? 5 : 0);
    if (g == 0) return (ss == 1 || ss == 2) ? 4 : ((ss == 5 || ss == 6) ? 3 : (ss == 7 ? 2 : 0));
    return (ss == 0 || ss == 1) ? 3 : (ss == 2 ? 2 : ((ss == 5 || ss == 6) ? 4 : 0));
  };
  auto SL0 = [&](int g, int ss) constexpr -> int {          // slices of the same phase this wave issued in earlier slots
    int n = 0;
    for (int t = (G == 3 || ss < 5) ? 0 : 5; t < ss; ++t) n += SL(g, t);
    return n;
  };
  auto piece_of = [&](int n) -> int { return G == 3 ? (grp == 0 ? 40 + pb + 4 * n : pb + 4 * n) : 8 * n + wave; };
  constexpr int MAXSL = 5;
  constexpr int NRQ = (RES != 0 && (RING_ABL & 64) == 0) ? NT2 * 2 : 0;      // early residual loads per lane

  unsigned dma_b[NWP], dma_a[MAXSL] = {OOB, OOB, OOB, OOB, OOB};

  // ---- prologue: first window (all pieces), the group's weights of super-step 0 (set 0)
  for (int j = wave; j < PA; j += NWAVES) send_a(voff_a(mt, 0, j, true), 0, j, true);
  if constexpr ((RING_ABL & 4) != 0) { for (int j = wave; j < PA; j += NWAVES) send_a(voff_a(mt, 0, j, true), 1, j, true); }
#pragma unroll
  for (int i = 0; i < NWP; ++i) send_b(voff_b(nt, 0, i / NQ, true, i % NQ), 0, i);      // (G <= 3 < 9: chunk 0)
  if constexpr ((RING_ABL & 8) != 0) {
#pragma unroll
    for (int i = 0; i < NWP; ++i) send_b(voff_b(nt, 0, i / NQ, true, i % NQ), 1, i);
  }
  wait_vm<0>();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (grp == 1) __builtin_amdgcn_s_barrier();              // group 1 runs one slot behind
  abl_pro = false;
  RG_CLK(0);
  int mt_p = 0, nt_p = 0;
  bool have_prev = false;
  int par = 0;                                             // weight set of the current period's super-step 0 (P is odd)

  for (int li = 0; li < nitems; ++li) {
    int mt1 = mt, nt1 = nt + 1;
    if (nt1 == a.ntiles) { nt1 = 0; ++mt1; }
    const bool more = li + 1 < nitems;
    unsigned mask[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const int m = mt * RG_BM + 64 * pb + 32 * rb + l31;
      unsigned mk = 0;
      if (m < a.M) {
        unsigned colm = 0;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
          if ((unsigned)(pw[rb] - 1 + kw) < (unsigned)a.W) colm |= 1u << kw;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
          if ((unsigned)(ph[rb] - 1 + kh) < (unsigned)a.H) mk |= colm << (3 * kh);
      }
      mask[rb] = mk;
    }
    for (int per = 0; per < NPER; ++per) {
      const int c0 = per * CPP;                            // first chunk of the period
      const int phg0 = li * a.NC + c0;                     // its phase: window buffer phg0 & 1 (chunk c0 + 1: the other one)
      const bool last_per = per + 1 == NPER;
      const bool has_next = !last_per || more;             // another period follows in this workgroup
      // what R(ss) needs (static ss): source offsets of the weights of super-step ss + 1 and of its slices ...
      auto slot_prep = [&](auto ss_c, int a_ln, const int* b_ln) {
        constexpr int ss = decltype(ss_c)::value;
        auto voff_b2 = [&](int nt_b, int c, int tap, bool live, int k) -> unsigned {
          return live ? (unsigned)(nt_b * (NTILE * a.NC * 9 * 128) + (c * 9 + tap) * 128 + b_ln[k]) : OOB;
        };
        auto voff_a2 = [&](int mt_n, int c_n, int j, bool live) -> unsigned {
          return (live && j < PA) ? (unsigned)((mt_n * RG_BM - a.W - 1 + 8 * j) * cin_b + c_n * 128 + a_ln) : OOB;
        };
#pragma unroll
        for (int i = 0; i < NWP; ++i) {
          if constexpr (ss + 1 < P) {
            constexpr int ktg = (ss + 1) * G;              // first k-tile of super-step ss + 1 inside the period
            dma_b[i] = voff_b2(nt, c0 + (ktg + i / NQ) / 9, (ktg + i / NQ) % 9, true, i % NQ);
          } else {                                         // first super-step of the next period: taps 0 .. G-1 of its first chunk
            dma_b[i] = !last_per ? voff_b2(nt, c0 + CPP, i / NQ, true, i % NQ) : voff_b2(nt1, 0, i / NQ, more, i % NQ);
          }
        }
        constexpr bool tgt_next = (G == 3) || (ss >= 5);   // slices for the next period's first chunk (else: this period's chunk 1)
        const int n0 = grp == 0 ? SL0(0, ss) : SL0(1, ss);
#pragma unroll
        for (int i = 0; i < MAXSL; ++i) {
          if (i < SL(0, ss) || i < SL(1, ss)) {
            const int j = piece_of(n0 + i);
            if (tgt_next) dma_a[i] = !last_per ? voff_a2(mt, c0 + CPP, j, true) : voff_a2(mt1, 0, j, more);
            else dma_a[i] = voff_a2(mt, c0 + 1, j, true);
          }
        }
      };
      // ... and the LDS addresses of the pixel fragments of its first k-tile
      auto addr_prep = [&](auto ss_c, int bi, unsigned mk0, unsigned mk1, int Wx) {
        constexpr int ss = decltype(ss_c)::value;
        frag_addr((ss * G) % 9, ((phg0 + (ss * G) / 9) & 1) * win_bytes, bi, mk0, mk1, Wx, a_addr);
      };
      auto super_step = [&](auto ss_c) {
        constexpr int ss = decltype(ss_c)::value;
        const int set = (ss & 1) ^ par;
        // per-super-step copies of the lane constants the address arithmetic starts from (see frag_addr)
        int bi = base_idx, a_ln = a_lane, b_ln[NQ], Wx = a.W;
        unsigned mk0 = mask[0], mk1 = mask[1];
        asm volatile("" : "+v"(bi), "+v"(a_ln), "+v"(mk0), "+v"(mk1), "+s"(Wx));
#pragma unroll
        for (int q = 0; q < NQ; ++q) { b_ln[q] = b_lane[q]; asm volatile("" : "+v"(b_ln[q])); }
        // ================= R slot (at raised priority: its instructions go between the other group's MFMAs)
        if (RING_PRIO_ON) __builtin_amdgcn_s_setprio(2);
        if (ss == 0 && per == 0) RG_STAMP(0);
        if (ss == 1 && per == 0) RG_STAMP(2);
        if (ss == 0 && per == 0) {
          // the previous item's epilogue — each group in its own R slot, under the other group's MFMAs — BEFORE this slot's
          // loads: nothing is in flight behind the residual it waits for
          if (have_prev) epilogue(mt_p, nt_p);
#pragma unroll
          for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int cb = 0; cb < WN; ++cb)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[rb][cb][r] = 0.f;
          RG_STAMP(1);
        }
        if (!AHEAD || ss == 0) { slot_prep(ss_c, a_ln, b_ln); addr_prep(ss_c, bi, mk0, mk1, Wx); }      // (R(0): new period, possibly a new item)
        f32x4 afr[2][4], bfr[WN][4];
        const unsigned wbase = wts_off + (grp * 2 + set) * SET_B;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
          for (int cb = 0; cb < WN; ++cb) bfr[cb][s] = *reinterpret_cast<lds_f4>(lds0 + wbase + boff[s] + 32 * 128 * cb);
#pragma unroll
          for (int rb = 0; rb < 2; ++rb) afr[rb][s] = *reinterpret_cast<lds_f4>(a_addr[rb][s]);
        }
        auto junk = [&](int j) {                             // (ablation: no fragment reads — lane-dependent junk operands instead)
#pragma unroll
          for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) afr[rb][s] = f32x4{(float)(lane * 3 + s), 1.5f + rb, -0.75f * lane, 0.3f + j};
#pragma unroll
            for (int cb = 0; cb < WN; ++cb) bfr[cb][s] = f32x4{0.01f * lane, -2.5f + cb, 0.125f * s, 1.f + j};
          }
        };
        if constexpr ((RING_ABL & 2) != 0) junk(0);
        __builtin_amdgcn_sched_barrier(0);                   // (the reads go out first: their latency runs under the issue below)
        // this slot's loads: the weights of super-step ss + 1 FIRST (the end of M(ss) confirms them; what is issued after
        // them may stay in flight), then window slices, then the early residual request
        const int n_sl = grp == 0 ? SL(0, ss) : SL(1, ss);
#pragma unroll
        for (int i = 0; i < NWP; ++i) send_b(dma_b[i], set ^ 1, i);
        {
          constexpr bool tgt_next = (G == 3) || (ss >= 5);
          const int wsel = tgt_next ? ((phg0 + CPP) & 1) : ((phg0 + 1) & 1);
          const bool live = tgt_next ? has_next : true;
          const int n0 = grp == 0 ? SL0(0, ss) : SL0(1, ss);
#pragma unroll
          for (int i = 0; i < MAXSL; ++i)
            if ((i < SL(0, ss) || i < SL(1, ss)) && i < n_sl) send_a(dma_a[i], wsel, piece_of(n0 + i), live);
        }
        int n_after = n_sl;                                  // operations of this wave younger than the weight pieces
        if constexpr (NRQ != 0 && ss == P - 2) {
          if (last_per) {                                    // residual of the item's tiles: four slots ahead of the epilogue
            const int eb = ebase_of(mt, nt);
            const unsigned chm = chmask_of(nt);
#pragma unroll
            for (int b = 0; b < NT2; ++b) req(eb, chm, b);
            n_after += NRQ;
          }
        }
        // G = 3, group 1: its slices of R(1) must be complete — and published by this slot's barrier — before group 0 reads
        // that window in R(0) of the next period, two slots from here: wait for everything older than this slot's weights
        if constexpr (G == 3 && ss == 2) { if (grp == 1) wait_vm<NWP>(); }
        if (ss == 1 && per == 0) RG_STAMP(3);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (ss == 1 && per == 0) RG_STAMP(4);
        if (RING_PRIO_ON) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        // ================= M slot: G k-tiles, k-step outer; behind the MFMAs of k-step s of k-tile j go the reads of k-step s
        // of k-tile j + 1 — into the registers those MFMAs have just released
#pragma unroll
        for (int j = 0; j < G; ++j) {
          lds_cptr n_addr[2][4];
          if (j + 1 < G) frag_addr((ss * G + j + 1) % 9, ((phg0 + (ss * G + j + 1) / 9) & 1) * win_bytes, bi, mk0, mk1, Wx, n_addr);
          f32x4 nafr[2][4], nbfr[WN][4];
#pragma unroll
          for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int cb = 0; cb < WN; ++cb)
#pragma unroll
              for (int rb = 0; rb < 2; ++rb) {
                if constexpr ((RING_ABL & 1) != 0) asm volatile("" :: "v"(bfr[cb][s]), "v"(afr[rb][s]));
                else acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bfr[cb][s]), __builtin_bit_cast(bf16x8, afr[rb][s]),
                                                                           acc[rb][cb], 0, 0, 0);
              }
            if (j + 1 < G) {
              if constexpr ((RING_ABL & 2) == 0) {
#pragma unroll
                for (int cb = 0; cb < WN; ++cb) nbfr[cb][s] = *reinterpret_cast<lds_f4>(lds0 + wbase + (j + 1) * HS_B + boff[s] + 32 * 128 * cb);
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) nafr[rb][s] = *reinterpret_cast<lds_f4>(n_addr[rb][s]);
              }
            }
          }
          if (j + 1 < G) {
            // issue order per k-step: its 2 WN MFMAs, then its WN + 2 reads
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              __builtin_amdgcn_sched_group_barrier(0x008, 2 * WN, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, WN + 2, 0);
            }
            if constexpr ((RING_ABL & 2) == 0) {
#pragma unroll
              for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int cb = 0; cb < WN; ++cb) bfr[cb][s] = nbfr[cb][s];
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) afr[rb][s] = nafr[rb][s];
              }
            } else junk(j + 1);
          }
        }
        if constexpr (AHEAD && ss + 1 < P) {                 // R(ss + 1)'s offsets and addresses: VALU work under OUR MFMAs
          slot_prep(int_k<ss + 1>{}, a_ln, b_ln);
          addr_prep(int_k<ss + 1>{}, bi, mk0, mk1, Wx);
#pragma unroll
          for (int i = 0; i < NWP; ++i) asm volatile("" : "+v"(dma_b[i]));
#pragma unroll
          for (int i = 0; i < MAXSL; ++i) asm volatile("" : "+v"(dma_a[i]));
#pragma unroll
          for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(a_addr[rb][s]));      // (materialised HERE: not sunk to the reads)
        }
        __builtin_amdgcn_sched_barrier(0);
        if (ss == 1 && per == 0) RG_STAMP(5);
        wait_vm_n(n_after);                                  // this slot's weight pieces have landed (in-order completion)
        if (ss == 1 && per == 0) RG_STAMP(6);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (ss == 1 && per == 0) RG_STAMP(7);
      };
      if constexpr (P == 3) { super_step(int_k<0>{}); super_step(int_k<1>{}); super_step(int_k<2>{}); }
      else {
        super_step(int_k<0>{}); super_step(int_k<1>{}); super_step(int_k<2>{}); super_step(int_k<3>{}); super_step(int_k<4>{});
        super_step(int_k<5>{}); super_step(int_k<6>{}); super_step(int_k<7>{}); super_step(int_k<8>{});
      }
      par ^= 1;
    }
    mt_p = mt; nt_p = nt; have_prev = true;
    if (mt1 != mt) advance_mtile();
    mt = mt1; nt = nt1;
  }
  // ---- tail: the last item's epilogue; group 0 then meets the barrier that closes group 1's last M slot
  RG_CLK(1);
  epilogue(mt_p, nt_p);
  if (grp == 0) __builtin_amdgcn_s_barrier();
}


#endif      // CADRE_AB_KERNELS

// ---------------------------------------------------------------------------------------------------------------
// WEIGHT-STATIONARY 64 -> 64 CHANNEL STAGE (bf16; round 4): conv3x3_c64s_kernel.  resnet.py:26-55, layer1: Cin = N = 64, K = 576.
// Measured on the ping-pong kernels (tools/ring_ablate.py, tools/ring_trace.py): on this stage the matrix pipe is busy 36 %
// of the time at 1.96 GHz; an item's 73 KB of weights are re-streamed into LDS per 256 positions (more DMA bytes than the
// pixels), 12 ds_read_b128 feed 8 MFMAs (LDS 75 % busy), and eighteen barriers per item order a weight ring that never
// changes.  Here the weights never move:
//   * 4 waves per workgroup, ONE per SIMD, up to 512 registers each: wave (ph, ch) owns 128 positions x 32 channels of a
//     256-position item and keeps ITS 32 x 576 weights as MFMA A fragments in registers (144 VGPRs) for the whole launch —
//     no weight DMA, no weight stage in LDS, no weight fragment reads: one ds_read_b128 (pixels) per MFMA;
//   * LDS holds only pixels: THREE windows (item i being read, i + 1 landed or landing, i + 2 being requested): an HBM
//     round trip has one to two items of lead;
//   * ONE barrier per item, between its taps 7 and 8: behind it every wave's slices of the next window have landed and
//     every wave has issued (and waited for) its last read of the current one, so the tap-0 fragments of the next item
//     are prefetched under tap 8's MFMAs and the freed buffer takes window i + 2.  No other synchronisation: inside an item
//     the four waves run free;
//   * the fragments of tap t + 1 are read behind the MFMAs of tap t, one k-step at a time into the registers that
//     k-step's four MFMAs have just released;
//   * TWO accumulator sets: the epilogue of item i - 1 (no LDS: lane-half exchange -> eight consecutive channels per lane,
//     folded BN from registers, residual, ReLU, one 16-byte store per 32 x 16 piece) is cut into eight pieces, one per tap
//     of item i, and runs under item i's MFMAs — with one wave per SIMD nothing else could hide it;
//   * loads are ordered by in-order completion and ONE counted vmcnt per item: between the last slice of window i + 1 and
//     the wait in front of barrier i + 1 a wave issues exactly 13 slices, 8 residual loads and 8 stores.
// Same k order (tap, k-step) and the same epilogue arithmetic as the ping-pong kernels: bit-identical outputs.
template <int RES, bool OUTB>
__global__ __launch_bounds__(256, 1) void conv3x3_c64s_kernel(ring_args a) {
  static_assert(RES == 0 || RES == 2, "bf16 operands take a bf16 residual");
  constexpr int RG_BM = 256, NWAVES = 4, NSLW = 13;        // 13 window slices of 8 pixels per wave and item: 416 pixels >= WPX
  constexpr unsigned OOB = 0x80000000u;
  constexpr int ESZ = OUTB ? 2 : 4;
  constexpr int NST = OUTB ? 8 : 16, NRQ = RES ? 8 : 0;    // stores / residual loads per lane and item
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int phf = wave >> 1, chh = wave & 1;               // position half (128 positions), channel half (32 channels)
  const int win_bytes = a.WPX * 128;
  char* win0 = smem;
  char* dump = smem + 3 * win_bytes;
  const int i_begin = blockIdx.x * a.ipw, i_end = min(a.items, i_begin + a.ipw);
  const int nitems = i_end - i_begin;
  if (nitems <= 0) return;
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.M * 128, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, 64 * 9 * 128, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, a.M * 64 * ESZ, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void*)(RES ? a.resid : a.x), 0, RES ? a.M * 128 : 0, 0x00020000);
  const int PA = a.WPX >> 3;
  for (int i = tid; i < 256; i += 256) reinterpret_cast<unsigned*>(dump)[i] = 0u;      // the ZERO ROW (halo taps) and the dummy DMA target
  // ---- this wave's weights: A fragments of its 32 output channels, all nine taps (w: [64][1][9][128 B]).  The folded-BN
  // SCALE of an output channel multiplies its weight row once, here (fp32 product, rounded to bf16 again: a caller that
  // wants single rounding folds the scale into the weights itself and passes scale = NULL, as cadre_amd/encoder.py does);
  // the SHIFT is the accumulator's initial value (the C operand of an item's first MFMAs): the epilogue has no BN arithmetic.
  f32x4 wfr[9][4];
  {
    const float wsc = a.scale ? a.scale[32 * chh + l31] : 1.f;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rsW, (32 * chh + l31) * 1152 + t * 128 + (2 * s + lh) * 16, 0, 0);
        bf16x8 wv = __builtin_bit_cast(bf16x8, raw);
        if (a.scale) {
#pragma unroll
          for (int e = 0; e < 8; ++e) wv[e] = (__bf16)((float)wv[e] * wsc);
        }
        wfr[t][s] = __builtin_bit_cast(f32x4, wv);
      }
  }
  f32x16 cinit;                                            // register r of a 32 x 32 tile: channel 32 chh + 8 (r >> 2) + 4 lh + (r & 3)
#pragma unroll
  for (int r = 0; r < 16; ++r) cinit[r] = a.shift ? a.shift[32 * chh + 8 * (r >> 2) + 4 * lh + (r & 3)] : 0.f;
  auto swz = [](int idx) constexpr -> int { return (idx >> 1) & 7; };
  // window slice n of this wave: piece j = 4 n + wave: pixel 8 j + (lane >> 3); LDS chunk (lane & 7) <- source chunk
  // (lane & 7) ^ swz(pixel), swz(pixel) = (lane >> 4) ^ 4 (j & 1) and j has the parity of the wave
  const int a_lane = (lane >> 3) * 128 + ((((lane & 7) ^ (lane >> 4) ^ (4 * (wave & 1)))) << 4);
  bool abl_pro = true;
  auto send_a = [&](int mt_n, int buf, int n, bool live) {
    const int j = 4 * n + wave;
    const bool ok = live && j < PA;
    unsigned voff = ok ? (unsigned)((mt_n * RG_BM - a.W - 1 + 8 * j) * 128 + a_lane) : OOB;
    char* dst = ok ? win0 + buf * win_bytes + j * 1024 : dump;
    if ((RING_ABL & 4) && !abl_pro) { voff = OOB; dst = dump; }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (__attribute__((address_space(3))) void*)dst, 16, (int)voff, 0, 0, 0);
  };
  typedef const __attribute__((address_space(3))) char* lds_cptr;
  typedef const __attribute__((address_space(3))) f32x4* lds_f4;
  const lds_cptr lds0 = (lds_cptr)smem;
  const unsigned zrow_off = (unsigned)(dump - smem);
  const int base_idx = 128 * phf + l31;                    // window row of this lane's position (rb = 0) at tap (0, 0)
  const unsigned lhb = (unsigned)lh << 4;                  // k-step s reads 16-byte chunk 2 s + lh: (lh << 4) ^ (s << 5), the latter an immediate
  const float inv_w = 1.0f / (float)a.W, inv_h = 1.0f / (float)a.H;
  // (h, w) of this lane's four fragment rows (positions 128 phf + 32 rb + l31 of an item), advanced by 256 per item
  int ph[4], pw[4];
  {
    const int HW = a.H * a.W;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      const int m = i_begin * RG_BM + 128 * phf + 32 * rb + l31;
      const int rem = m % HW;
      ph[rb] = rem / a.W;
      pw[rb] = rem - ph[rb] * a.W;
    }
  }
  auto advance_item = [&]() {
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      const int x = pw[rb] + RG_BM;
      const int q1 = (int)(((float)x + 0.5f) * inv_w);
      pw[rb] = x - q1 * a.W;
      const int y = ph[rb] + q1;
      const int q2 = (int)(((float)y + 0.5f) * inv_h);
      ph[rb] = y - q2 * a.H;
    }
  };
  auto masks_of = [&](int mt_i, unsigned* mk) {             // 9-bit tap validity of this lane's four positions of item mt_i (branch-free)
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      const int m = mt_i * RG_BM + 128 * phf + 32 * rb + l31;
      unsigned colm = 0, v = 0;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) colm |= ((unsigned)(pw[rb] - 1 + kw) < (unsigned)a.W) ? (1u << kw) : 0u;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) v |= ((unsigned)(ph[rb] - 1 + kh) < (unsigned)a.H) ? (colm << (3 * kh)) : 0u;
      mk[rb] = m < a.M ? v : 0u;
    }
  };
  // pixel fragment addresses of one tap: row idx = base_idx + 32 rb + toff of window buffer `buf` (or the zero row)
  auto frag_addr = [&](int tap, int buf, int bi, const unsigned* mk, int Wx, lds_cptr (*out)[4]) {
    const int toff = (tap / 3) * Wx + (tap % 3);
    const int win_off = buf * win_bytes;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      const int idx = bi + 32 * rb + toff;
      unsigned row = ((mk[rb] >> tap) & 1u) ? (unsigned)(win_off + (idx << 7)) : zrow_off;
      if constexpr ((RING_ABL & 128) != 0) row = (unsigned)(win_off + (idx << 7));      // (padded-layout emulation: no halo masks)
      const unsigned rsw = row ^ (unsigned)(swz(idx) << 4) ^ lhb;
#pragma unroll
      for (int s = 0; s < 4; ++s) out[rb][s] = lds0 + (rsw ^ (unsigned)(s << 5));
    }
  };
  // ---- epilogue pieces: piece p = (rb = p >> 1, h = p & 1) of a finished accumulator set: position 128 phf + 32 rb + l31,
  // channels 32 chh + 16 h + 8 lh .. + 7 after the lane-half exchange
  f32x16 acc[2][4];
  u32x4 rq[8];
  const float act_floor = (a.act & 15) == 1 ? 0.f : -__builtin_inff();
  const bool post = (a.act & 16) != 0;
  auto ebyte = [&](int mt_e, int p, bool live, int esz) -> int {      // byte offset of piece p's 8 channels, or out of range
    const int pos = mt_e * RG_BM + 128 * phf + 32 * (p >> 1) + l31;
    return live ? (pos * 64 + 32 * chh + 16 * (p & 1) + 8 * lh) * esz : (int)OOB;      // (pos >= M lies past num_records)
  };
  auto rq_load = [&](int mt_e, int p, bool live) {
    if constexpr (RES != 0 && (RING_ABL & 64) == 0)
      rq[p] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsR, ebyte(mt_e, p, live, 2), 0, 0));
  };
  // The piece's arithmetic as single VALU instructions: left to the SLP vectoriser these become v_pk_* pairs fed by register
  // shuffles (measured: 200 v_mov per item) — beside MFMAs a packed op costs more than its two halves
  // (MI355X_MICROARCH.md, per-instruction constants).
  auto vadd = [](float x, float y) -> float { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; };
  auto vmax = [](float x, float y) -> float { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; };
  auto epi_piece = [&](auto q_c, int mt_e, int p, bool live) {
    constexpr int Q = decltype(q_c)::value;
    if constexpr ((RING_ABL & 32) != 0) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float t = acc[Q][p >> 1][8 * (p & 1) + e]; asm volatile("" :: "v"(t)); }
    } else {
      const int rb = p >> 1, h = p & 1;
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = acc[Q][rb][8 * h + e]; v[4 + e] = acc[Q][rb][8 * h + 4 + e]; }
      // v_permlane32_swap x, y: x <- [x.lo, y.lo], y <- [x.hi, y.hi] (inline asm: see conv3x3_ring_pp2_kernel); the four
      // exchanges of a piece in ONE statement: one leading s_nop covers the VALU -> permlane wait states of all of them
      asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %4\n\tv_permlane32_swap_b32 %1, %5\n\tv_permlane32_swap_b32 %2, %6\n\t"
          "v_permlane32_swap_b32 %3, %7\n\ts_nop 1"
          : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
      if constexpr (RES != 0) {
        const u32x4 t = rq[p];                               // (residual BEFORE the activation: act | 16 stays on the ping-pong kernels)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const unsigned rbits = (e & 1) ? (t[e >> 1] & 0xffff0000u) : (t[e >> 1] << 16);      // bf16 -> f32
          v[e] = vmax(vadd(v[e], __builtin_bit_cast(float, rbits)), act_floor);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = vmax(v[e], act_floor);
      }
      const int bo = ebyte(mt_e, p, live, ESZ);
      if constexpr ((RING_ABL & 16) != 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) asm volatile("" :: "v"(v[e]), "v"(bo));
      } else if constexpr (OUTB) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rsC, bo, 0, 0);
      } else {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{v[0], v[1], v[2], v[3]}), rsC, bo, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{v[4], v[5], v[6], v[7]}), rsC, bo == (int)OOB ? bo : bo + 16, 0, 0);
      }
    }
  };

  // ---- prologue: windows of the first two items (buffers 0, 1), everything landed and published; then the two slices of
  // window 2 that the steady state issues in the tap-8 slot of the previous item
  const int mt0 = i_begin;
#pragma unroll 1
  for (int n = 0; n < NSLW; ++n) { send_a(mt0, 0, n, true); send_a(mt0 + 1, 1, n, nitems > 1); }
  wait_vm<0>();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  abl_pro = false;
  RG_CLK(0);
  // slices of window i + 2 per tap slot (slot 8 belongs to the PREVIOUS item's tap 8): 2 2 2 2 1 1 1 1 | 1
  auto sl_n = [](int slot) constexpr -> int { return slot == 8 ? 1 : (slot < 4 ? 2 : 1); };
  auto sl_0 = [](int slot) constexpr -> int { return slot == 8 ? 0 : (slot < 4 ? 1 + 2 * slot : 5 + slot); };      // first slice index of the slot
  send_a(mt0 + 2, 2, 0, nitems > 2);                       // (slot 8 of "item -1")
  unsigned mk[4];
  masks_of(mt0, mk);
  lds_cptr p_addr[4][4];
  f32x4 pfr[4][4];
  {
    int bi = base_idx, Wx = a.W;
    frag_addr(0, 0, bi, mk, Wx, p_addr);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) pfr[rb][s] = *reinterpret_cast<lds_f4>(p_addr[rb][s]);
  }
  bool have_prev = false;

  // one item on accumulator set Q; the epilogue pieces of the previous item (set Q ^ 1) ride on its taps 0 .. 7
  auto item_body = [&](auto q_c, int li) {
    constexpr int Q = decltype(q_c)::value;
    const int mt = mt0 + li;
    const int buf = li % 3, buf_n = (li + 1) % 3, buf_nn = (li + 2) % 3;
    const bool more = li + 1 < nitems, more2 = li + 2 < nitems;
    unsigned mkn[4];                                         // masks of the NEXT item (its tap-0 fragments are read in this item's tap 8)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      // per-tap copies of the lane constants behind an empty asm: the address arithmetic stays where it is used
      int bi = base_idx, Wx = a.W;
      asm volatile("" : "+v"(bi), "+s"(Wx));
      if (tap == 8) {
        // ---- the item's barrier.  This wave's slices of window li + 1 have landed: since the last of them (tap 7 of the
        // previous item) it issued 13 slices, NRQ residual loads, NST stores — in-order completion; its reads of window li
        // are done (tap 8's fragments are in registers).
        wait_vm<NSLW + NRQ + NST>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      if (tap == 4 && (RING_ABL & 128) == 0) {               // the next item's positions and halo masks: vector work under this tap's MFMAs
        advance_item();
        masks_of(mt + 1, mkn);
      }
      // addresses of the next tap's fragments (tap 8: tap 0 of the next item, window li + 1)
      lds_cptr n_addr[4][4];
      if (tap < 8) frag_addr(tap + 1, buf, bi, mk, Wx, n_addr);
      else frag_addr(0, buf_n, bi, mkn, Wx, n_addr);
      // this tap slot's slices of window li + 2 (tap 8: the first slice of window li + 3, into the buffer the barrier freed)
#pragma unroll
      for (int i = 0; i < sl_n(tap); ++i) {
        if (tap < 8) send_a(mt + 2, buf_nn, sl_0(tap) + i, more2);
        else send_a(mt + 3, buf, 0, li + 3 < nitems);
      }
      // residual of THIS item's piece `tap`, for the epilogue piece that runs in the next item's tap `tap`: a whole item
      // (~2.5 us) ahead — an HBM round trip under load is 2-4 us, the loads complete in order behind the window slices, and
      // with one wave per SIMD a wait on them idles the matrix pipe (measured: 230 us of a 1.1 ms launch with two taps
      // of lead).  Issued after the previous item's piece `tap` has consumed the same registers (below).
      (void)0;
      f32x4 nfr[4][4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
          if constexpr ((RING_ABL & 1) != 0) asm volatile("" :: "v"(wfr[tap][s]), "v"(pfr[rb][s]));
          else if (tap == 0 && s == 0)
            acc[Q][rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wfr[tap][s]), __builtin_bit_cast(bf16x8, pfr[rb][s]),
                                                                 cinit, 0, 0, 0);
          else
            acc[Q][rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wfr[tap][s]), __builtin_bit_cast(bf16x8, pfr[rb][s]),
                                                                 acc[Q][rb], 0, 0, 0);
        }
        if (tap < 8 || more) {
          if constexpr ((RING_ABL & 2) == 0) {
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) nfr[rb][s] = *reinterpret_cast<lds_f4>(n_addr[rb][s]);
          } else {
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) nfr[rb][s] = f32x4{(float)(lane * 3 + s), 1.5f + rb, -0.75f * lane, 0.3f + tap};
          }
        }
      }
      // the previous item's epilogue piece `tap` (item 0: the same loads and stores, out of range — the vmcnt count is static)
      if (tap < 8) {
        epi_piece(int_k<Q ^ 1>{}, mt - 1, tap, have_prev);
        if constexpr (RES != 0) rq_load(mt, tap, true);
      }
      // issue order per k-step: one MFMA, then a share of the tap's vector work (address arithmetic, the epilogue piece), four
      // times; then the four reads into the registers the k-step released
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) pfr[rb][s] = nfr[rb][s];
      if (tap == 8) {
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) mk[rb] = mkn[rb];
      }
    }
    have_prev = true;
  };
  for (int li = 0; li < nitems; li += 2) {
    item_body(int_k<0>{}, li);
    if (li + 1 < nitems) item_body(int_k<1>{}, li + 1);
  }
  RG_CLK(1);
  // ---- tail: the last item's epilogue (its residual pieces were requested during the item)
  {
    const int mt_l = mt0 + nitems - 1;
    if ((nitems & 1) != 0) {
#pragma unroll
      for (int p = 0; p < 8; ++p) epi_piece(int_k<0>{}, mt_l, p, true);
    } else {
#pragma unroll
      for (int p = 0; p < 8; ++p) epi_piece(int_k<1>{}, mt_l, p, true);
    }
  }
}


#ifdef CADRE_AB_KERNELS
// ---------------------------------------------------------------------------------------------------------------
// ONE WAVE PER SIMD, STREAMED WEIGHTS (bf16, 128-channel tile; round 4) — A/B BUILD ONLY (CADRE_BUILD_AB=1, CADRE_RING_1W=1):
// parity-green; without a residual it TIES the ping-pong kernel (same box, 2048 frames: 0.663 / 0.621 / 0.614 vs 0.658 /
// 0.616 / 0.624 ms on layer2 / 3 / 4 — both at 64-65 % matrix-pipe duty and the same 1.65-1.69 GHz sustained clock: the
// chip is power-limited on these layers), with one it spills (16 residual pieces held in registers).
// conv3x3_ring1w_kernel — the layers whose weights
// do not fit registers (resnet.py:26-55 layer2-4, danet.py:21-41 conv5a/5c/51) on the scheme of conv3x3_c64s_kernel:
//   * 4 waves per workgroup, one per SIMD, up to 512 registers: wave w owns 64 positions x ALL 128 channels of a
//     256-position x 128-channel item (accumulator 128 registers): per k-tile 16 weight + 8 pixel fragment reads feed 32
//     MFMAs (0.75 ds_read_b128 per MFMA; the 8-wave ping-pong kernel: 1.0) and nothing is read twice by a SIMD;
//   * weights [128 rows][128 B] per k-tile stream through FOUR LDS stages by LDS-DMA, three k-tiles ahead; pixels: two
//     windows (the current chunk's and the next one's), as in the ping-pong kernels;
//   * ONE barrier per k-tile: behind it the stage of k-tile t + 1 is complete (every wave waited for its own pieces by
//     counted vmcnt: in-order completion) and nobody reads the stage the DMA of k-tile t + 3 is about to overwrite.  The
//     fragments of k-tile t + 1 are read behind the MFMAs of k-tile t, k-step by k-step into the registers just released:
//     a wave comes out of the barrier with 32 MFMAs to issue;
//   * the epilogue needs no LDS (lane-half exchange -> 16-byte stores, conv3x3_ring_pp2_kernel); the folded-BN SHIFT
//     is the accumulator's initial value, read from an LDS table in accumulator layout; the SCALE must be folded into the
//     weights by the caller (scale = NULL; cadre_amd/encoder.py does) — otherwise the ping-pong kernel runs;
//   * the residual of an item is requested during its last chunk (two pieces per k-tile), a chunk ahead of the epilogue.
// Same k order and epilogue arithmetic as the ping-pong kernel on folded weights: bit-identical outputs.
template <int RES, bool OUTB>
__global__ __launch_bounds__(256, 1) void conv3x3_ring1w_kernel(ring_args a) {
  static_assert(RES == 0 || RES == 2, "bf16 operands take a bf16 residual");
  constexpr int NTILE = 128, RG_BM = 256, NWAVES = 4, NSTG = 4, LEAD = 3;
  constexpr int STG_B = NTILE * 128, NBPW = 4;              // weight pieces (8 rows x 128 B) per wave and stage
  constexpr unsigned OOB = 0x80000000u;
  constexpr int ESZ = OUTB ? 2 : 4;
  constexpr int NPC = 16;                                   // epilogue pieces per item and lane: (rb, cb, h)
  constexpr int NST = OUTB ? NPC : 2 * NPC;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int win_bytes = a.WPX * 128;
  char* win0 = smem;
  char* wst = smem + 2 * win_bytes;
  char* dump = wst + NSTG * STG_B;
  float* sh_lds = reinterpret_cast<float*>(dump + 1024);   // shift in accumulator layout: [nt][cb][lh][16]
  const int i_begin = blockIdx.x * a.ipw, i_end = min(a.items, i_begin + a.ipw);
  const int nitems = i_end - i_begin;
  if (nitems <= 0) return;
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.M * a.Cin * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.N * a.NC * 9 * 128, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, a.M * a.N * ESZ, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void*)(RES ? a.resid : a.x), 0, RES ? a.M * a.N * 2 : 0, 0x00020000);
  const int cin_b = a.Cin * 2;
  const int PA = a.WPX >> 3;
  for (int i = tid; i < 256; i += 256) reinterpret_cast<unsigned*>(dump)[i] = 0u;      // the ZERO ROW and the dummy DMA target
  for (int i = tid; i < a.ntiles * 128; i += 256) {         // i = ((nt*4 + cb)*2 + lh)*16 + r -> channel nt*128 + 32 cb + 8 (r >> 2) + 4 lh + (r & 3)
    const int r = i & 15, lh_ = (i >> 4) & 1, cb = (i >> 5) & 3, nt_ = i >> 7;
    const int n = nt_ * NTILE + 32 * cb + 8 * (r >> 2) + 4 * lh_ + (r & 3);
    sh_lds[i] = (a.shift && n < a.N) ? a.shift[n] : 0.f;
  }
  auto swz = [](int idx) constexpr -> int { return (idx >> 1) & 7; };
  const int a_lane = (lane >> 3) * cin_b + ((((lane & 7) ^ (lane >> 4) ^ (4 * (wave & 1)))) << 4);      // window piece j = 4 n + wave
  int b_lane[NBPW];
#pragma unroll
  for (int k = 0; k < NBPW; ++k) {
    const int r = (wave * NBPW + k) * 8 + (lane >> 3);
    b_lane[k] = r * a.NC * 9 * 128 + (((lane & 7) ^ swz(r)) << 4);
  }
  bool abl_pro = true;
  auto send_a = [&](int mt_n, int c_n, int buf, int n, bool live) {
    const int j = 4 * n + wave;
    const bool ok = live && j < PA;
    unsigned voff = ok ? (unsigned)((mt_n * RG_BM - a.W - 1 + 8 * j) * cin_b + c_n * 128 + a_lane) : OOB;
    char* dst = ok ? win0 + buf * win_bytes + j * 1024 : dump;
    if ((RING_ABL & 4) && !abl_pro) { voff = OOB; dst = dump; }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (__attribute__((address_space(3))) void*)dst, 16, (int)voff, 0, 0, 0);
  };
  auto send_b = [&](int nt_b, int c, int tap, int stg, bool live) {      // this wave's four pieces of one weight stage
#pragma unroll
    for (int k = 0; k < NBPW; ++k) {
      unsigned voff = live ? (unsigned)(nt_b * (NTILE * a.NC * 9 * 128) + (c * 9 + tap) * 128 + b_lane[k]) : OOB;
      char* dst = wst + stg * STG_B + (wave * NBPW + k) * 1024;
      if ((RING_ABL & 8) && !abl_pro) { voff = OOB; dst = dump; }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)dst, 16, (int)voff, 0, 0, 0);
    }
  };
  typedef const __attribute__((address_space(3))) char* lds_cptr;
  typedef const __attribute__((address_space(3))) f32x4* lds_f4;
  const lds_cptr lds0 = (lds_cptr)smem;
  const unsigned zrow_off = (unsigned)(dump - smem);
  const unsigned wst_off = (unsigned)(wst - smem);
  const int base_idx = 64 * wave + l31;
  const unsigned lhb = (unsigned)lh << 4;
  int boff[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) boff[s] = l31 * 128 + (((2 * s + lh) ^ swz(l31)) << 4);
  const float inv_w = 1.0f / (float)a.W, inv_h = 1.0f / (float)a.H;
  int ph[2], pw[2];
  int mt = i_begin / a.ntiles, nt = i_begin - mt * a.ntiles;
  {
    const int HW = a.H * a.W;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const int m = mt * RG_BM + 64 * wave + 32 * rb + l31;
      const int rem = m % HW;
      ph[rb] = rem / a.W;
      pw[rb] = rem - ph[rb] * a.W;
    }
  }
  auto advance_mtile = [&]() {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const int x = pw[rb] + RG_BM;
      const int q1 = (int)(((float)x + 0.5f) * inv_w);
      pw[rb] = x - q1 * a.W;
      const int y = ph[rb] + q1;
      const int q2 = (int)(((float)y + 0.5f) * inv_h);
      ph[rb] = y - q2 * a.H;
    }
  };
  auto masks_of = [&](int mt_i, unsigned* mk) {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const int m = mt_i * RG_BM + 64 * wave + 32 * rb + l31;
      unsigned colm = 0, v = 0;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) colm |= ((unsigned)(pw[rb] - 1 + kw) < (unsigned)a.W) ? (1u << kw) : 0u;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) v |= ((unsigned)(ph[rb] - 1 + kh) < (unsigned)a.H) ? (colm << (3 * kh)) : 0u;
      mk[rb] = m < a.M ? v : 0u;
    }
  };
  auto frag_addr = [&](int tap, int win_off, int bi, const unsigned* mk, int Wx, lds_cptr (*out)[4]) {
    const int toff = (tap / 3) * Wx + (tap % 3);
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const int idx = bi + 32 * rb + toff;
      const unsigned row = ((mk[rb] >> tap) & 1u) ? (unsigned)(win_off + (idx << 7)) : zrow_off;
      const unsigned rsw = row ^ (unsigned)(swz(idx) << 4) ^ lhb;
#pragma unroll
      for (int s = 0; s < 4; ++s) out[rb][s] = lds0 + (rsw ^ (unsigned)(s << 5));
    }
  };
  // ---- epilogue piece p = (rb = p >> 3, cb = (p >> 1) & 3, h = p & 1): position 64 wave + 32 rb + l31, channels
  // nt * 128 + 32 cb + 16 h + 8 lh .. + 7
  f32x16 acc[2][4];
  u32x4 rq[NPC];
  const float act_floor = (a.act & 15) == 1 ? 0.f : -__builtin_inff();
  auto vadd = [](float x, float y) -> float { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; };
  auto vmax = [](float x, float y) -> float { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; };
  auto ebyte = [&](int mt_e, int nt_e, int p, int esz) -> int {
    const int rb = p >> 3, cb = (p >> 1) & 3, h = p & 1;
    const int pos = mt_e * RG_BM + 64 * wave + 32 * rb + l31;
    const int ch = nt_e * NTILE + 32 * cb + 16 * h + 8 * lh;
    return ch < a.N ? (pos * a.N + ch) * esz : (int)OOB;      // (pos >= M lies past num_records; N % 32 == 0)
  };
  auto rq_load = [&](int mt_e, int nt_e, int p) {
    if constexpr (RES != 0 && (RING_ABL & 64) == 0)
      rq[p] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsR, ebyte(mt_e, nt_e, p, 2), 0, 0));
  };
  auto epi_piece = [&](int mt_e, int nt_e, int p) {
    const int rb = p >> 3, cb = (p >> 1) & 3, h = p & 1;
    if constexpr ((RING_ABL & 32) != 0) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float t = acc[rb][cb][8 * h + e]; asm volatile("" :: "v"(t)); }
    } else {
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = acc[rb][cb][8 * h + e]; v[4 + e] = acc[rb][cb][8 * h + 4 + e]; }
      asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %4\n\tv_permlane32_swap_b32 %1, %5\n\tv_permlane32_swap_b32 %2, %6\n\t"
          "v_permlane32_swap_b32 %3, %7\n\ts_nop 1"
          : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
      if constexpr (RES != 0) {
        const u32x4 t = rq[p];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const unsigned rbits = (e & 1) ? (t[e >> 1] & 0xffff0000u) : (t[e >> 1] << 16);
          v[e] = vmax(vadd(v[e], __builtin_bit_cast(float, rbits)), act_floor);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = vmax(v[e], act_floor);
      }
      const int bo = ebyte(mt_e, nt_e, p, ESZ);
      if constexpr ((RING_ABL & 16) != 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) asm volatile("" :: "v"(v[e]), "v"(bo));
      } else if constexpr (OUTB) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rsC, bo, 0, 0);
      } else {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{v[0], v[1], v[2], v[3]}), rsC, bo, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{v[4], v[5], v[6], v[7]}), rsC, bo == (int)OOB ? bo : bo + 16, 0, 0);
      }
    }
  };

  // window slices of the next phase per tap slot (11 per wave: 44 pieces of 8 pixels >= WPX / 8 for W <= 47)
  auto sl_n = [](int tap) constexpr -> int { return tap < 5 ? 2 : (tap == 5 ? 1 : 0); };
  auto sl_0 = [](int tap) constexpr -> int { return tap < 5 ? 2 * tap : 10; };
  // residual loads per tap slot of an item's LAST chunk: pieces 2 (tap - 1), 2 (tap - 1) + 1 for taps 1 .. 8
  auto rq_n = [](int tap) constexpr -> int { return (RES != 0 && (RING_ABL & 64) == 0 && tap >= 1) ? 2 : 0; };

  // ---- prologue: first window (all 11 slices), weights of k-tiles 0 .. LEAD - 1 into stages 0 .. LEAD - 1
#pragma unroll 1
  for (int n = 0; n < 11; ++n) send_a(mt, 0, 0, n, true);
#pragma unroll
  for (int t = 0; t < LEAD; ++t) send_b(nt, 0, t, t, true);
  wait_vm<0>();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  abl_pro = false;
  RG_CLK(0);
  unsigned mk[2];
  masks_of(mt, mk);
  f32x4 wfr[4][4], pfr[2][4];                               // fragments of the current k-tile: weights [cb][s], pixels [rb][s]
  {
    lds_cptr p_addr[2][4];
    frag_addr(0, 0, base_idx, mk, a.W, p_addr);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) wfr[cb][s] = *reinterpret_cast<lds_f4>(lds0 + wst_off + boff[s] + 4096 * cb);
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) pfr[rb][s] = *reinterpret_cast<lds_f4>(p_addr[rb][s]);
    }
  }
  int g = 0;                                                // k-tiles this workgroup has started (stage of k-tile g: g & 3)
  int prev_tail = 0;                                        // VMEM operations the previous slot issued after its weight pieces
  int prev2_tail = 0;

  for (int li = 0; li < nitems; ++li) {
    int mt1 = mt, nt1 = nt + 1;
    if (nt1 == a.ntiles) { nt1 = 0; ++mt1; }
    const bool more = li + 1 < nitems;
    unsigned mkn[2] = {mk[0], mk[1]};                       // masks of the next item's M tile
    if (mt1 != mt) { advance_mtile(); masks_of(mt1, mkn); }
    // the accumulator starts from the folded-BN shift (LDS table in accumulator layout)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      const float* tb = sh_lds + ((nt * 4 + cb) * 2 + lh) * 16;
      f32x16 c0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(tb + 4 * q);
        c0[4 * q] = t[0]; c0[4 * q + 1] = t[1]; c0[4 * q + 2] = t[2]; c0[4 * q + 3] = t[3];
      }
      acc[0][cb] = c0;
      acc[1][cb] = c0;
    }
    for (int c = 0; c < a.NC; ++c) {
      const int phg = li * a.NC + c;
      const int wb = phg & 1;
      const bool last_c = c + 1 == a.NC;
      const bool has_next = !last_c || more;
      const int mt_n = last_c ? mt1 : mt, c_n = last_c ? 0 : c + 1;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        int bi = base_idx, Wx = a.W;
        asm volatile("" : "+v"(bi), "+s"(Wx));
        // ---- the k-tile's barrier: this wave's pieces of stage g + 1 (issued two slots ago) have landed — everything it
        // issued since stays in flight: the two slots' tails and the last slot's four weight pieces
        wait_vm_n(prev2_tail + NBPW + prev_tail);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // weights of k-tile g + LEAD into the stage k-tile g - 1 occupied (its fragments were read two slots ago)
        {
          const int stg = (g + LEAD) & (NSTG - 1);
          if (tap + LEAD < 9) send_b(nt, c, tap + LEAD, stg, true);
          else if (!last_c) send_b(nt, c + 1, tap + LEAD - 9, stg, true);
          else send_b(nt1, 0, tap + LEAD - 9, stg, more);
        }
        int tail = 0;
#pragma unroll
        for (int i = 0; i < sl_n(tap); ++i) send_a(mt_n, c_n, wb ^ 1, sl_0(tap) + i, has_next);
        tail += sl_n(tap);
        if constexpr (RES != 0) {
          if (last_c && tap >= 1) {
            rq_load(mt, nt, 2 * (tap - 1));
            rq_load(mt, nt, 2 * (tap - 1) + 1);
            tail += rq_n(tap);
          }
        }
        // fragments of the next k-tile: addresses (tap 8: tap 0 of the next chunk / item)
        lds_cptr n_addr[2][4];
        if (tap < 8) frag_addr(tap + 1, wb * win_bytes, bi, mk, Wx, n_addr);
        else frag_addr(0, (wb ^ 1) * win_bytes, bi, last_c ? mkn : mk, Wx, n_addr);
        const unsigned wnext = wst_off + (unsigned)(((g + 1) & (NSTG - 1)) * STG_B);
        f32x4 nwf[4][4], npf[2][4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
          for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
              if constexpr ((RING_ABL & 1) != 0) asm volatile("" :: "v"(wfr[cb][s]), "v"(pfr[rb][s]));
              else acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wfr[cb][s]), __builtin_bit_cast(bf16x8, pfr[rb][s]),
                                                                         acc[rb][cb], 0, 0, 0);
            }
          if constexpr ((RING_ABL & 2) == 0) {
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) nwf[cb][s] = *reinterpret_cast<lds_f4>(lds0 + wnext + boff[s] + 4096 * cb);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) npf[rb][s] = *reinterpret_cast<lds_f4>(n_addr[rb][s]);
          } else {
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) nwf[cb][s] = f32x4{0.01f * lane, -2.5f + cb, 0.125f * s, 1.f + tap};
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) npf[rb][s] = f32x4{(float)(lane * 3 + s), 1.5f + rb, -0.75f * lane, 0.3f + tap};
          }
        }
        // issue order per k-step: its eight MFMAs, then the six reads into the registers they released
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
          for (int cb = 0; cb < 4; ++cb) wfr[cb][s] = nwf[cb][s];
#pragma unroll
          for (int rb = 0; rb < 2; ++rb) pfr[rb][s] = npf[rb][s];
        }
        prev2_tail = prev_tail;
        prev_tail = tail;
        ++g;
      }
    }
    // ---- the item's epilogue (16 pieces; its stores count in the next two slots' waits)
#pragma unroll
    for (int p = 0; p < NPC; ++p) epi_piece(mt, nt, p);
    if ((RING_ABL & 48) == 0) prev_tail += NST;
    if (mt1 != mt) { mk[0] = mkn[0]; mk[1] = mkn[1]; }
    mt = mt1; nt = nt1;
  }
  RG_CLK(1);
}

#endif      // CADRE_AB_KERNELS

// Tile configuration (host logic): 256 positions x 64 / 128 channels on 8 waves, one persistent workgroup per CU.
// Channel tile 128 unless N < 128 (CADRE_RING_NTILE forces one for A/B runs).  (A 128-position / 4-wave shape with two
// workgroups per CU existed until the ping-pong kernel beat it on every shape it was picked for: layer3 bf16 1134 vs
// 990 TFLOP/s; the kernel template still takes WVM = 2.)
struct ring_cfg { int wvm, ntile, bm, wpx, wgs, pp, g; size_t lds; long long items; };
// pflags: 1 = residual after the activation (act | 16), 2 = a folded-BN scale vector is present — either keeps the launch
// on the ping-pong kernels (the weight-stationary / one-wave kernels add the residual before the activation; the one-wave
// kernel wants the scale folded into the weights)
static void ring_pick(long long M, int W, int N, int bf16, int NC, ring_cfg* c, int pflags = 0) {
  const bool no_c64s = (pflags & 1) != 0;
  static const int force_nt = [] { const char* e = getenv("CADRE_RING_NTILE"); return e ? atoi(e) : 0; }();
  static const int force_pp = [] { const char* e = getenv("CADRE_RING_PP"); return e ? atoi(e) : 1; }();
  int ntile = N >= 128 ? 128 : 64;
  if (force_nt == 64 || (force_nt == 128 && N >= 128)) ntile = force_nt;
  const int wvm = 4, bm = 64 * wvm;
  int wpx = (bm + 2 * W + 2 + 7) & ~7;
  const int min_wpx = (2 * wvm * RG_SLAB + 127) / 128;             // the epilogue slabs live in a window
  if (wpx < min_wpx) wpx = (min_wpx + 7) & ~7;
  const size_t bn_table = (size_t)((N + ntile - 1) / ntile) * ntile * 8;      // folded BN of every channel: scale | shift
  c->wvm = wvm; c->ntile = ntile; c->bm = bm; c->wpx = wpx;
  c->lds = (size_t)2 * wpx * 128 + (size_t)3 * ntile * 128 + 1024 + bn_table;
  // The ping-pong kernel: every bf16 shape, and the fp32 64-channel tile (layer1: 2.99 vs 3.03 ms once its staging
  // slots carry no vector-ALU work; fp32 with 128-channel tiles is not instantiated — those layers run on the tile
  // kernels, and a forced window conv takes the lockstep kernel).  CADRE_RING_PP=0: lockstep everywhere, for A/B runs.
  c->pp = (force_pp > 0 && (bf16 || ntile == 64)) ? 1 : 0;
  // G k-tiles per slot (conv3x3_ring_pp2_kernel, bf16; A/B build + CADRE_RING_G=2 only — measured slower): G = 3 with the
  // 64-channel tile, G = 2 with the 128-channel tile when the item has an even number of k-tiles (NC even) and the two
  // sets of G weight half-stages per group fit beside the windows.
  static const int force_g = [] { const char* e = getenv("CADRE_RING_G"); return e ? atoi(e) : 1; }();
  c->g = 1;
#ifdef CADRE_AB_KERNELS
  const bool pp2_built = true;
#else
  const bool pp2_built = false;
#endif
  if (pp2_built && c->pp && bf16 && force_g == 2 && NC > 0) {
    const int g = ntile == 64 ? 3 : 2;
    const size_t lds2 = (size_t)2 * wpx * 128 + (size_t)4 * g * (ntile / 2) * 128 + 1024 + bn_table;
    if ((9 * NC) % g == 0 && lds2 <= 160 * 1024) { c->g = g; c->lds = lds2; }
  }
  // the 64 -> 64 channel stage in bf16 (one 128-byte chunk, one channel tile): conv3x3_c64s_kernel — weights resident in
  // registers, three pixel windows in LDS (CADRE_RING_C64S=0: off, for A/B runs); reported as g = 9
  static const int c64s_on = [] { const char* e = getenv("CADRE_RING_C64S"); return e ? atoi(e) : 1; }();
  if (bf16 && N == 64 && NC == 1 && c64s_on && force_pp > 0 && !no_c64s) {
    const int wpx3 = (bm + 2 * W + 2 + 7) & ~7;
    if (wpx3 <= 416) { c->g = 9; c->pp = 1; c->wpx = wpx3; c->lds = (size_t)3 * wpx3 * 128 + 1024; }
  }
  // the 128-channel tile in bf16 with the scale folded into the weights: conv3x3_ring1w_kernel — one wave per SIMD, four
  // weight stages, two windows (A/B build + CADRE_RING_1W=1 only: ties the ping-pong kernel); reported as g = 8
  static const int r1w_on = [] { const char* e = getenv("CADRE_RING_1W"); return e ? atoi(e) : 0; }();
  if (pp2_built && bf16 && ntile == 128 && NC >= 1 && r1w_on && force_pp > 0 && pflags == 0 && c->g == 1) {
    const int wpx2 = (bm + 2 * W + 2 + 7) & ~7;
    const size_t lds1 = (size_t)2 * wpx2 * 128 + (size_t)4 * 128 * 128 + 1024 + (size_t)((N + 127) / 128) * 512;
    if (wpx2 <= 352 && lds1 <= 160 * 1024) { c->g = 8; c->pp = 1; c->wpx = wpx2; c->lds = lds1; }
  }
  c->items = ((M + bm - 1) / bm) * ((N + ntile - 1) / ntile);
  c->wgs = (int)(c->items < 256 ? c->items : 256);                  // persistent workgroups: one per CU
}

static int g_ring_mode = [] { const char* e = getenv("CADRE_RING_CONV"); return e ? atoi(e) : 1; }();   // 0 off, 1 auto, 2 wherever supported

static int ring_capable(int F, int H, int W, int Cin, int N, int bf16, int out_bytes, int resid_bytes) {   // geometry the kernel can run at all
  const int eb = bf16 ? 2 : 4;
  if (F < 1 || H < 1 || W < 2 || W > 95) return 0;
  if ((Cin * eb) % 128 != 0 || N % 32 != 0) return 0;
  const long long M = (long long)F * H * W, lim = 1ll << 31;
  // 32-bit buffer offsets: input, weights, output and residual each at their own element size (resid_bytes 0 = none)
  if (M * Cin * eb >= lim || (long long)N * Cin * 9 * eb >= lim || M * N * out_bytes >= lim || M * N * resid_bytes >= lim) return 0;
  ring_cfg c;
  ring_pick(M, W, N, bf16, Cin * eb / 128, &c);
  return c.lds <= 160 * 1024;
}
// flags: the launch's word (1 bf16 operands, 2 bf16 output, 4 bf16 residual) + 8 = a residual is present; a caller that
// commits to this kernel on a 1 here gets no geometry failure from cadre_conv3x3_ring with the same flags
extern "C" int cadre_conv3x3_ring_supported(int32_t F, int32_t H, int32_t W, int32_t Cin, int32_t N, int32_t flags) {
  const int bf16 = flags & 1;
  if (g_ring_mode == 0 || !ring_capable(F, H, W, Cin, N, bf16, (flags & 2) ? 2 : 4, (flags & 8) ? ((flags & 4) ? 2 : 4) : 0)) return 0;
  if (g_ring_mode == 1 && !bf16 && N > 64) return 0;      // fp32: the tile kernels win from N = 128 on
  return 1;
}

// tile configuration cadre_conv3x3_ring would use, as ntile (64 / 128) + 1000 * WVM + 100000 * ping-pong + 1000000 * G
// (host logic; names the kernel instantiation for profiles: conv3x3_ring_kernel<bf16, ntile, res, out_bf16, WVM>,
// conv3x3_ring_pp_kernel<bf16, ntile, res, out_bf16, false> or, G > 0, conv3x3_ring_pp2_kernel<ntile, res, out_bf16, G>)
extern "C" int cadre_conv3x3_ring_ntile(int32_t F, int32_t H, int32_t W, int32_t Cin, int32_t N, int32_t bf16) {
  ring_cfg c;
  ring_pick((long long)F * H * W, W, N, bf16, Cin * (bf16 ? 2 : 4) / 128, &c);
  return c.ntile + 1000 * c.wvm + 100000 * c.pp + 1000000 * (c.g > 1 ? c.g : 0);
}

#ifdef RING_TRACE
static long long* g_ring_trace = nullptr;
extern "C" void cadre_ring_set_trace(void* p) { g_ring_trace = (long long*)p; }
#endif

// bf16 on 16 x 16 x 32 MFMAs: opt-in (CADRE_RING_M16=1).  Measured on every layer of the bf16 encoder (2048 frames at 288 x 288,
// same box, alternating runs): 5-10 % SLOWER than the 32 x 32 x 16 form (layer2 0.835 vs 0.795 ms, layer4 0.731 vs 0.695 ms,
// whole forward 18.0 vs 17.3 ms) — the kernel is paced by its staging slots and LDS reads, not by the matrix pipe's clock.
static const int g_ring_m16 = [] { const char* e = getenv("CADRE_RING_M16"); return e ? atoi(e) : 0; }();

template <bool BF, int NT, int RS, bool OB>
static void ring_launch_pp(const ring_args& a, int grid, size_t lds, hipStream_t st) {
#ifdef CADRE_AB_KERNELS      // (A/B build only: measured 5-10 % slower, see above)
  if constexpr (BF) {
    if (g_ring_m16) {
      (void)hipFuncSetAttribute((const void*)conv3x3_ring_pp_kernel<BF, NT, RS, OB, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipLaunchKernelGGL((conv3x3_ring_pp_kernel<BF, NT, RS, OB, true>), dim3(grid), dim3(512), lds, st, a);
      return;
    }
  }
#endif
  if constexpr (BF || NT == 64) {                          // (fp32: the 64-channel tile only — see ring_pick)
    (void)hipFuncSetAttribute((const void*)conv3x3_ring_pp_kernel<BF, NT, RS, OB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((conv3x3_ring_pp_kernel<BF, NT, RS, OB>), dim3(grid), dim3(512), lds, st, a);
  }
}

extern "C" int cadre_conv3x3_ring(const void* x, const void* w, const float* scale, const float* shift, const void* resid,
                                  void* out, int32_t F, int32_t H, int32_t W, int32_t Cin, int32_t N, int32_t act,
                                  int32_t flags, void* stream) {
  const int bf16 = flags & 1, out_bf16 = (flags >> 1) & 1, resid_bf16 = (flags >> 2) & 1;
  if (!x || !w || !out) return cadre_fail("cadre_conv3x3_ring: null operand");
  if (!ring_capable(F, H, W, Cin, N, bf16, out_bf16 ? 2 : 4, resid ? (resid_bf16 ? 2 : 4) : 0))
    return cadre_fail("cadre_conv3x3_ring: unsupported geometry (W in 2..95, Cin a multiple of 128 bytes, N % 32 == 0, every tensor < 2 GiB: chunk the batch)");
  if ((act & 15) > 1) return cadre_fail("cadre_conv3x3_ring: the window kernels implement act 0 (none) and 1 (ReLU) only");
  if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)out | (uintptr_t)resid) & 15) return cadre_fail("cadre_conv3x3_ring: operands must be 16-byte aligned");
  ring_args a;
  a.x = x; a.w = w; a.scale = scale; a.shift = shift; a.resid = resid; a.out = out;
  a.M = F * H * W; a.H = H; a.W = W; a.Cin = Cin; a.N = N; a.NC = Cin * (bf16 ? 2 : 4) / 128;
  a.act = act; a.out_bf16 = out_bf16; a.resid_bf16 = resid_bf16;
  ring_cfg cfg;
  ring_pick(a.M, W, N, bf16, a.NC, &cfg, ((act & 16) ? 1 : 0) | (scale ? 2 : 0));
  const int ntile = cfg.ntile;
  a.mtiles = (a.M + cfg.bm - 1) / cfg.bm;
  a.ntiles = (N + ntile - 1) / ntile;
  a.items = (int)cfg.items;
  a.ipw = (a.items + cfg.wgs - 1) / cfg.wgs;
  const int grid = (a.items + a.ipw - 1) / a.ipw;
  a.WPX = cfg.wpx;
  static const int prio = [] { const char* e = getenv("CADRE_RING_PRIO"); return e ? atoi(e) : 1; }();   // (+1-3 % on every bf16 shape)
  a.prio = prio;
#ifdef RING_TRACE
  a.trace = g_ring_trace;
#endif
  const size_t lds = cfg.lds;
  if (lds > 160 * 1024) return cadre_fail("cadre_conv3x3_ring: window does not fit LDS");
  hipStream_t st = (hipStream_t)stream;
#define RG_LAUNCH(BF, NT_, RS_, OB_, WV_)                                                                        \
  do {                                                                                                           \
    (void)hipFuncSetAttribute((const void*)conv3x3_ring_kernel<BF, NT_, RS_, OB_, WV_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    hipLaunchKernelGGL((conv3x3_ring_kernel<BF, NT_, RS_, OB_, WV_>), dim3(grid), dim3(128 * WV_), lds, st, a); \
  } while (0)
#define RG_PP(BF, NT_, RS_, OB_) ring_launch_pp<BF, NT_, RS_, OB_>(a, grid, lds, st)
#define RG_NT(BF, RS_, OB_)                                                                                      \
  do {                                                                                                           \
    if (cfg.pp) {                                                                                \
      if (ntile == 128) RG_PP(BF, 128, RS_, OB_); else RG_PP(BF, 64, RS_, OB_);                                  \
    } else {                                                                                                     \
      if (ntile == 128) RG_LAUNCH(BF, 128, RS_, OB_, 4);                                                      \
      else RG_LAUNCH(BF, 64, RS_, OB_, 4);                                                                    \
    }                                                                                                            \
  } while (0)
#ifdef CADRE_AB_KERNELS
#define RG_PP2(NT_, RS_, OB_, G_)                                                                                  \
  do {                                                                                                           \
    (void)hipFuncSetAttribute((const void*)conv3x3_ring_pp2_kernel<NT_, RS_, OB_, G_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    hipLaunchKernelGGL((conv3x3_ring_pp2_kernel<NT_, RS_, OB_, G_>), dim3(grid), dim3(512), lds, st, a);        \
  } while (0)
#endif
#define RG_C64S(RS_, OB_)                                                                                        \
  do {                                                                                                           \
    (void)hipFuncSetAttribute((const void*)conv3x3_c64s_kernel<RS_, OB_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    hipLaunchKernelGGL((conv3x3_c64s_kernel<RS_, OB_>), dim3(grid), dim3(256), lds, st, a);                     \
  } while (0)
#ifdef CADRE_AB_KERNELS
#define RG_R1W(RS_, OB_)                                                                                         \
  do {                                                                                                           \
    (void)hipFuncSetAttribute((const void*)conv3x3_ring1w_kernel<RS_, OB_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    hipLaunchKernelGGL((conv3x3_ring1w_kernel<RS_, OB_>), dim3(grid), dim3(256), lds, st, a);                   \
  } while (0)
#endif
  if (false) {
#ifdef CADRE_AB_KERNELS
  } else if (bf16 && cfg.g == 8) {
    if (resid && !resid_bf16) return cadre_fail("cadre_conv3x3_ring: bf16 operands take a bf16 residual");
    if (resid) { if (out_bf16) RG_R1W(2, true); else RG_R1W(2, false); }
    else { if (out_bf16) RG_R1W(0, true); else RG_R1W(0, false); }
#endif
  } else if (bf16 && cfg.g == 9) {
    if (resid && !resid_bf16) return cadre_fail("cadre_conv3x3_ring: bf16 operands take a bf16 residual");
    if (resid) { if (out_bf16) RG_C64S(2, true); else RG_C64S(2, false); }
    else { if (out_bf16) RG_C64S(0, true); else RG_C64S(0, false); }
#ifdef CADRE_AB_KERNELS
  } else if (bf16 && cfg.g > 1) {
    if (resid && !resid_bf16) return cadre_fail("cadre_conv3x3_ring: bf16 operands take a bf16 residual");
    if (ntile == 64) {
      if (resid) { if (out_bf16) RG_PP2(64, 2, true, 3); else RG_PP2(64, 2, false, 3); }
      else { if (out_bf16) RG_PP2(64, 0, true, 3); else RG_PP2(64, 0, false, 3); }
    } else {
      if (resid) { if (out_bf16) RG_PP2(128, 2, true, 2); else RG_PP2(128, 2, false, 2); }
      else { if (out_bf16) RG_PP2(128, 0, true, 2); else RG_PP2(128, 0, false, 2); }
    }
#endif
  } else if (bf16) {
    if (resid && !resid_bf16) return cadre_fail("cadre_conv3x3_ring: bf16 operands take a bf16 residual");
    if (resid) { if (out_bf16) RG_NT(true, 2, true); else RG_NT(true, 2, false); }
    else { if (out_bf16) RG_NT(true, 0, true); else RG_NT(true, 0, false); }
  } else {
    if (out_bf16 || (resid && resid_bf16)) return cadre_fail("cadre_conv3x3_ring: fp32 operands take fp32 residual / output");
    if (resid) RG_NT(false, 1, false); else RG_NT(false, 0, false);
  }
#ifdef CADRE_AB_KERNELS
#undef RG_R1W
#endif
#undef RG_C64S
#ifdef CADRE_AB_KERNELS
#undef RG_PP2
#endif
#undef RG_NT
#undef RG_PP
#undef RG_LAUNCH
  return (int)hipGetLastError();
}
