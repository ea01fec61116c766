// conv3x3_s1x.hip — the SECOND conv of a down-sampling BasicBlock with the block's shortcut folded into it, bf16 model, gfx950:
//     out = relu( bn2(conv3x3_s1(t)) + bn_d(conv1x1_s2(x)) )           (resnet.py:40-55 with downsample = resnet.py:152-158)
// as ONE accumulation:  out = relu( conv3x3_s1(t; W2 * s2) + conv1x1_s2(x; Wd * sd) + (b2 + bd) ).
// Until round 5 the shortcut was its own launch on the tile kernel (gemm_bf16.hip: K = 64 / 128 / 256 — 150 / 276 / 441
// TFLOP/s, the kernels furthest below either roof), its output written to HBM and read back as conv2's residual.  Here it is
// K-EXTENSION: 1 / 2 / 4 extra k-tiles on top of conv2's 18 / 36 / 72, no shortcut tensor at all.  The bf16 model then also
// skips one rounding (the shortcut was rounded to bf16 before it was added; now the sum is formed in fp32).
//
// Structure = the ping-pong window kernel of conv3x3_ring.hip (two groups of four waves half a k-tile apart, a 256-position
// x 128-channel item, windows of the item's positions resident in LDS, weights streamed by LDS-DMA, XOR swizzle on the
// source address) with the DMA duties split as in conv3x3_s2.hip: group 0 issues all weight pieces (two stages, k-tile t + 1
// requested in R(t), confirmed at the end of M(t)), group 1 all window pieces.  Per 128-byte chunk c of conv2's input:
//     S_c  nine k-tiles (taps) from the stride-1 window [P - W - 1, P + 256 + W + 1) of t, double-buffered across chunks;
//     E_c  (c < NCd) one k-tile from the 256-row window of x's even-row / even-column pixels (2h, 2w), chunk c of x —
//          the lane computes the NHWC address of its pixel (one floor-division by W per 8-pixel piece, once per M tile).
// Weights arrive in exactly that k-tile order: [N][9 NC + NCd][64].  The epilogue is conv3x3_s2.hip's: sums start at the
// shift, ReLU, pairwise conversion to bf16, a 4 KB LDS slab per wave, stores of eight full 128-byte lines per instruction.
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

#define SX_BM 256
#define SX_NTILE 128
#define SX_STG_B (SX_NTILE * 128)
#define SX_XWIN_B (SX_BM * 128)

struct s1x_args {
  const void* x;          // [F][H][W][C1] bf16: conv2's input t
  const void* x2;         // [F][2H][2W][Cd] bf16: the block input (shortcut conv 1x1 / s2 reads pixel (2h, 2w))
  const void* w;          // [N][KT][128 B], KT = 9 NC + NCd k-tiles in execution order: per chunk c nine taps (kh*3 + kw) of W2, then (c < NCd) chunk c of Wd
  const float* shift;     // [N] = b2 + bd, or null
  void* out;              // [M][N] bf16
  int M, M2;              // positions F*H*W; pixels of x2 (4 M)
  int H, W, C1, Cd, N, NC, NCd, KT;
  int act;
  int mtiles, ntiles, items, ipw;
  int WPX, PA;            // stride-1 window: pixels (multiple of 8) and 8-pixel pieces
};

template <int N>
__device__ __forceinline__ void sx_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// NPS: stride-1 window pieces per wave of group 1 (4 * NPS >= PA): 9, 10 or 11
// NSTG: weight stages.  2: k-tile t + 1 requested in R(t), confirmed at the end of M(t) (one slot and a half of lead).  3 (where
//       LDS allows): k-tile t + 2 requested in R(t), k-tile t + 1 confirmed at the end of R(t) — three slots of lead, and group
//       0's MFMA slot ends without a wait, as in conv3x3_ring_pp_kernel.
template <int NPS, int NSTG>
__global__ __launch_bounds__(512, 2) void conv3x3_s1x_kernel(s1x_args a) {
  constexpr int NST = 8;                                   // epilogue stores per wave
  constexpr unsigned OOB = 0x80000000u;
  // SYM (three weight stages, opt-in): every wave issues weights AND window pieces, as in conv3x3_ring_pp_kernel.  With two stages
  // the duties are split by group (group 1's weight pieces would land a slot too late).  At equal work (a plain stride-1 conv, no
  // shortcut) the split form runs 6-8 % behind conv3x3_ring_pp_kernel and the symmetric form 10-15 %: that kernel's staging slot is
  // leaner — 9 k-tiles per chunk on 3 stages make every stage offset an immediate and every DMA target a scalar constant, here the
  // 9- or 10-k-tile chunks on 2 stages leave both to run-time arithmetic (4 + 2 VALU per slot).
  constexpr bool SYM = NSTG == 3;
  constexpr int NBW = SYM ? 2 : 4;                         // weight pieces per issuing wave and k-tile
  constexpr int NS8 = (4 * NPS + 7) / 8;                   // SYM: stride-1 window pieces per wave (8 waves)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int grp = wave >> 2, pb = wave & 3;
  const int win_bytes = a.WPX * 128;
  char* win0 = smem;                                       // two stride-1 windows, the shortcut window, two weight stages, zero row / dump, shifts
  char* xwin = smem + 2 * win_bytes;
  char* bst = xwin + SX_XWIN_B;
  char* dump = bst + NSTG * SX_STG_B;
  float* sh_lds = reinterpret_cast<float*>(dump + 1024);

  const int i_begin = blockIdx.x * a.ipw, i_end = min(a.items, i_begin + a.ipw);
  const int nitems = i_end - i_begin;
  if (nitems <= 0) return;
  const int cin_b = a.C1 * 2, cd_b = a.Cd * 2;
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.M * cin_b, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsX2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.x2, 0, a.M2 * cd_b, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.N * a.KT * 128, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, a.M * a.N * 2, 0x00020000);
  for (int i = tid; i < 256; i += 512) reinterpret_cast<unsigned*>(dump)[i] = 0u;       // ZERO ROW (halo taps) and dummy DMA target
  for (int i = tid; i < a.ntiles * SX_NTILE; i += 512) sh_lds[i] = (a.shift && i < a.N) ? a.shift[i] : 0.f;
  auto swz = [](int idx) constexpr -> int { return (idx >> 1) & 7; };
  const int W = a.W;
  const float inv_w = 1.0f / (float)W, inv_h = 1.0f / (float)a.H;

  // ---- window DMA (group 1).  Piece j = 4 n + pb: LDS rows 8 j .. 8 j + 7, this lane row 8 j + (lane >> 3), LDS chunk (lane & 7) <-
  // source chunk (lane & 7) ^ swz(row), swz(row) = (lane >> 4) ^ 4 (j & 1), j has the parity of pb.
  const int sw_lane = (((lane & 7) ^ (lane >> 4) ^ (4 * (pb & 1)))) << 4;
  const int s_lane = (lane >> 3) * cin_b + sw_lane;        // stride-1 window: linear in the position
  auto send_s = [&](int mt_n, int c_n, int bufsel, int n, bool live) {
    const int j = SYM ? 8 * n + wave : 4 * n + pb;
    const bool ok = live && j < a.PA;
    const unsigned voff = (unsigned)((mt_n * SX_BM - W - 1 + 8 * j) * cin_b + c_n * 128 + s_lane) | (ok ? 0u : OOB);
    char* dst = ok ? win0 + bufsel * win_bytes + j * 1024 : dump;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (__attribute__((address_space(3))) void*)dst, 16, (int)voff, 0, 0, 0);
  };
  // shortcut window: row r of M tile mt holds position p = mt * 256 + r = (q, w) with q = floor(p / W) (rows counted across
  // frames), i.e. pixel (2 q) * (2 W) + 2 w = 2 p + 2 q W of x2.  The per-lane pixel offsets of the wave's 8 pieces are computed
  // once per M tile (xoff), not in the staging slots (vector-ALU work there is starved by the other group's MFMAs).
  constexpr int NXP = SYM ? 4 : 8;                         // shortcut-window pieces per issuing wave
  int xoff[NXP];
  auto plan_x = [&](int mt_x) {
#pragma unroll
    for (int n = 0; n < NXP; ++n) {
      const int p = mt_x * SX_BM + 8 * (SYM ? 8 * n + wave : 4 * n + pb) + (lane >> 3);
      int q = (int)((float)p * inv_w);                     // exact after one correction step (p < 2^23)
      const int r = p - q * W;
      q += (r >= W) ? 1 : 0;
      q -= (r < 0) ? 1 : 0;
      xoff[n] = (2 * p + 2 * q * W) * cd_b + sw_lane;      // (past the tensor: beyond num_records, zero fill)
    }
  };
  auto send_x = [&](int c_x, int n, bool live) {
    const unsigned voff = (unsigned)(xoff[n] + c_x * 128) | (live ? 0u : OOB);
    char* dst = live ? xwin + (SYM ? 8 * n + wave : 4 * n + pb) * 1024 : dump;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX2, (__attribute__((address_space(3))) void*)dst, 16, (int)voff, 0, 0, 0);
  };
  // ---- weight DMA (group 0): stage piece 4 pb + k (k = 0 .. 3)
  int b_lane[NBW];
#pragma unroll
  for (int k = 0; k < NBW; ++k) {
    const int r = ((SYM ? 2 * wave : 4 * pb) + k) * 8 + (lane >> 3);
    b_lane[k] = r * a.KT * 128 + (((lane & 7) ^ swz(r)) << 4);
  }
  auto send_wts = [&](int nt_b, int kt, int stg, bool live) {
#pragma unroll
    for (int k = 0; k < NBW; ++k) {
      const unsigned voff = (unsigned)(nt_b * (SX_NTILE * a.KT * 128) + kt * 128 + b_lane[k]) | (live ? 0u : OOB);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(bst + stg * SX_STG_B + ((SYM ? 2 * wave : 4 * pb) + k) * 1024),
                                               16, (int)voff, 0, 0, 0);
    }
  };
  // ---- lane constants of the fragment reads (conv3x3_s2.hip: one base per row, XOR-ed with the literal s << 5)
  const unsigned lh4 = (unsigned)lh << 4;
  const unsigned bbase = (unsigned)((64 * wn + l31) * 128) ^ ((unsigned)swz(64 * wn + l31) << 4) ^ lh4;
  const unsigned zrow_off = (unsigned)(dump - smem);
  const unsigned xw_off = (unsigned)(xwin - smem), bst_off = (unsigned)(bst - smem);

  int ph_[2], pw_[2];
  int mt = i_begin / a.ntiles, nt = i_begin - mt * a.ntiles;
  {
    const int HW = a.H * W;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const int m = mt * SX_BM + 64 * wm + 32 * rb + l31;
      const int rem = m % HW;
      ph_[rb] = rem / W;
      pw_[rb] = rem - ph_[rb] * W;
    }
  }
  auto advance_mtile = [&]() {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const int x = pw_[rb] + SX_BM;
      const int q1 = (int)(((float)x + 0.5f) * inv_w);
      pw_[rb] = x - q1 * W;
      const int y = ph_[rb] + q1;
      const int q2 = (int)(((float)y + 0.5f) * inv_h);
      ph_[rb] = y - q2 * a.H;
    }
  };

  // ---- epilogue (conv3x3_s2.hip): sums start at the shift; ReLU, bf16 pairs, an LDS slab turns them into full-line stores
  f32x16 acc[2][2];
  const float act_floor = (a.act & 15) == 1 ? 0.f : -__builtin_inff();
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  auto acc_init = [&](int nt_i) {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const float* sh = sh_lds + nt_i * SX_NTILE + 64 * wn + 32 * cb + 4 * lh;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(sh + 8 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[0][cb][4 * g + e] = t[e]; acc[1][cb][4 * g + e] = t[e]; }
      }
    }
  };
  // (stores through a 4 KB slab per wave, as in conv3x3_s2.hip: every store instruction writes eight full 128-byte lines.  `slab`:
  //  the stride-1 window buffer that is free while the epilogue runs — the last chunk's, read last two slots ago, requested again
  //  at the new item's first k-tile, after the epilogue)
  auto epilogue = [&](int mt_e, int nt_e, unsigned slab) {
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    char* sl = smem + slab + wave * 4096;
    const int prow = lane >> 3, pch = lane & 7;
    const bool ch_ok = nt_e * SX_NTILE + 64 * wn + 32 * (pch >> 2) < a.N;
    const int eb = (mt_e * SX_BM + 64 * wm + prow) * a.N + nt_e * SX_NTILE + 64 * wn + 8 * pch;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
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
        const int bo = (int)((unsigned)(eo * 2) | (ch_ok ? 0u : OOB));
        __builtin_amdgcn_raw_buffer_store_b128(v, rsC, bo, 0, 0);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };

  // ---- prologue: group 1 brings the first stride-1 window, group 0 the weights of k-tile 0
  plan_x(mt);
  if constexpr (SYM) {
#pragma unroll
    for (int n = 0; n < NS8; ++n) send_s(mt, 0, 0, n, true);
    send_wts(nt, 0, 0, true);
    send_wts(nt, 1, 1, true);
  } else if (grp == 1) {
#pragma unroll
    for (int n = 0; n < NPS; ++n) send_s(mt, 0, 0, n, true);
  } else {
    send_wts(nt, 0, 0, true);
  }
  sx_wait_vm<0>();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (grp == 1) __builtin_amdgcn_s_barrier();              // group 1 runs one slot behind

  // ---- main loop.  The ping-pong GROUP is a compile-time parameter of the whole item loop (two instances, one per group) and so
  // is "this chunk has a shortcut k-tile": the staging slots then carry no control flow — as run-time conditions the group's DMA
  // duty, the shortcut requests and the end-of-item cases put 25 scalar branches into every chunk, and on the stride-1 ping-pong
  // kernel taking the branches out of the k-loop was worth 5-10 %.  What stays: the loops, one branch per chunk
  // at its first k-tile (first k-tile of the item?) and one at its last (last k-tile of the item?).
  using std::integral_constant;
  auto run = [&](auto grp_c) {
    constexpr int GRP = decltype(grp_c)::value;
    int sbuf = 0;                                          // window buffer of the current stride-1 phase
    int kpar = 0;                                          // weight stage of the current k-tile (advances every k-tile, across items)
    auto stg_next = [&](int st) -> int { return NSTG == 2 ? (st ^ 1) : (st == 2 ? 0 : st + 1); };
    int mt_p = 0, nt_p = 0;
    bool have_prev = false;
    for (int li = 0; li < nitems; ++li) {
      int mt1 = mt, nt1 = nt + 1;
      if (nt1 == a.ntiles) { nt1 = 0; ++mt1; }
      const bool more = li + 1 < nitems;
      unsigned mask[2];
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        unsigned colm = 0, mk = 0;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
          if ((unsigned)(pw_[rb] - 1 + kw) < (unsigned)W) colm |= 1u << kw;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
          if ((unsigned)(ph_[rb] - 1 + kh) < (unsigned)a.H) mk |= colm << (3 * kh);
        mask[rb] = mk;
      }
      // head of the item's first staging slot: group 0's weights of k-tile 1 first (its stores then stand behind them in the
      // queue), the previous item's epilogue — group 1, still in the M slot of that item's last k-tile, takes its closing barrier now
      __builtin_amdgcn_s_setprio(2);
      if constexpr (GRP == 0 && NSTG == 2) send_wts(nt, 1, kpar ^ 1, true);
      if (have_prev) {
        epilogue(mt_p, nt_p, (unsigned)((sbuf ^ 1) * win_bytes));
        if constexpr (GRP == 1) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
      }
      acc_init(nt);
      int kt = 0;                                          // k-tile index inside the item's weight stream

      auto chunk = [&](auto has_e_c, const int c) {
        constexpr bool HAS_E = decltype(has_e_c)::value;
        const bool last_c = c + 1 == a.NC;
        const int mt_n = last_c ? mt1 : mt, c_n = last_c ? 0 : c + 1;
        const bool live_n = !last_c || more;
        const unsigned wbase = (unsigned)(sbuf * win_bytes);
        // one k-tile: KIND 0 = tap T of the stride-1 window, KIND 1 = the shortcut k-tile of this chunk.  FIRST: the chunk's first
        // k-tile (may be the item's first: item_first), LAST: the chunk's last k-tile (may be the item's last: item_last)
        auto step = [&](auto kind_c, auto t_c, const bool item_first, const bool item_last) {
          constexpr int KIND = decltype(kind_c)::value, T = decltype(t_c)::value;
          constexpr bool FIRST = KIND == 0 && T == 0, LAST = HAS_E ? KIND == 1 : (KIND == 0 && T == 8);
          // ================= R slot
          __builtin_amdgcn_s_setprio(2);
          const unsigned sbase = bst_off + (unsigned)(kpar * SX_STG_B);
          f32x4 afr[2][4], bfr[2][4];
          {
            unsigned arow_sw[2];
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
              if constexpr (KIND == 0) {
                const int toff = (T / 3) * W + (T % 3);
                const int idx = 64 * wm + 32 * rb + l31 + toff;
                const unsigned row = ((mask[rb] >> T) & 1u) ? wbase + (unsigned)(idx << 7) : zrow_off;
                arow_sw[rb] = row ^ (unsigned)(swz(idx) << 4) ^ lh4;
              } else {
                const int idx = 64 * wm + 32 * rb + l31;
                arow_sw[rb] = (xw_off + (unsigned)(idx << 7)) ^ (unsigned)(swz(idx) << 4) ^ lh4;
              }
            }
            const unsigned bsw = sbase + bbase;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              bfr[0][s] = *reinterpret_cast<const f32x4*>(smem + (bsw ^ (unsigned)(s << 5)));
#pragma unroll
              for (int rb = 0; rb < 2; ++rb) afr[rb][s] = *reinterpret_cast<const f32x4*>(smem + (arow_sw[rb] ^ (unsigned)(s << 5)));
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) bfr[1][s] = *reinterpret_cast<const f32x4*>(smem + (bsw ^ (unsigned)(s << 5)) + 32 * 128);
          }
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (SYM) {
            // every wave: its two pieces of k-tile kt + 2 of the weight stream (past the item's end: k-tile 0 / 1 of the next
            // item) into the stage k-tile kt - 1 was read from, then its window pieces — the next stride-1 window two pieces per
            // k-tile at taps 1 .. 3, this chunk's shortcut window two per k-tile at taps 4, 5 (conv3x3_ring_pp_kernel's scheme).
            // Then k-tile kt + 1 — requested a slot pair ago, FIRST in its slot — is confirmed: all but the window pieces of the
            // previous slot and this slot's operations (and, in the item's first slot, the previous item's stores) has completed.
            const bool wrap = kt + 2 >= a.KT;
            send_wts(wrap ? nt1 : nt, wrap ? kt + 2 - a.KT : kt + 2, stg_next(stg_next(kpar)), wrap ? more : true);
            auto w_of = [](int kind, int t) constexpr -> int {       // window pieces a wave issues in the slot of (kind, t)
              if (kind != 0) return 0;
              if (t == 1 || t == 2) return 2;
              if (t == 3) return NS8 - 4;
              if (t == 4 || t == 5) return HAS_E ? 2 : 0;
              return 0;
            };
            if constexpr (KIND == 0 && T >= 1 && T <= 3) {
              if constexpr (2 * (T - 1) < NS8) send_s(mt_n, c_n, sbuf ^ 1, 2 * (T - 1), live_n);
              if constexpr (2 * (T - 1) + 1 < NS8) send_s(mt_n, c_n, sbuf ^ 1, 2 * (T - 1) + 1, live_n);
            }
            if constexpr (KIND == 0 && HAS_E && (T == 4 || T == 5)) {
              send_x(c, 2 * (T - 4), true);
              send_x(c, 2 * (T - 4) + 1, true);
            }
            constexpr int W_PREV = KIND == 1 ? w_of(0, 8) : (T == 0 ? 0 : w_of(0, T - 1));
            constexpr int ALLOW = W_PREV + NBW + w_of(KIND, T);
            if constexpr (FIRST) {
              if (item_first && have_prev) sx_wait_vm<ALLOW + NST>();
              else sx_wait_vm<ALLOW>();
            } else {
              sx_wait_vm<ALLOW>();
            }
          } else if constexpr (GRP == 0) {
            // weights of the next k-tile of the stream into the stage the previous k-tile was read from; after the item's last
            // k-tile: k-tile 0 of the next item.  (The item's first k-tile: k-tile 1 went out in the head.)
            if constexpr (FIRST) {
              if (!item_first) send_wts(nt, kt + 1, kpar ^ 1, true);
            } else if constexpr (LAST) {
              // (selects, no branch: the next k-tile of this item, or k-tile 0 of the next item)
              send_wts(item_last ? nt1 : nt, item_last ? 0 : kt + 1, kpar ^ 1, item_last ? more : true);
            } else {
              send_wts(nt, kt + 1, kpar ^ 1, true);
            }
          } else {
            if constexpr (KIND == 0) {
              // static schedule: the next stride-1 window (other buffer) two pieces per k-tile at taps 0 .. 5, this chunk's
              // shortcut window two pieces per k-tile at taps 0 .. 3 (both buffers were read last before this chunk began)
              if constexpr (T <= 5) {
                if constexpr (2 * T < NPS) send_s(mt_n, c_n, sbuf ^ 1, 2 * T, live_n);
                if constexpr (2 * T + 1 < NPS) send_s(mt_n, c_n, sbuf ^ 1, 2 * T + 1, live_n);
              }
              if constexpr (HAS_E && T <= 3) {
                send_x(c, 2 * T, true);
                send_x(c, 2 * T + 1, true);
              }
              // everything this wave requested has landed before the chunk's last stride-1 k-tile ends: the shortcut window (read
              // next, if the chunk has one) and the next stride-1 window
              if constexpr (T == 8) sx_wait_vm<0>();
            }
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
            for (int s = 0; s < 4; ++s)
#pragma unroll
              for (int rb = 0; rb < 2; ++rb)
                acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bfr[cb][s]), __builtin_bit_cast(bf16x8, afr[rb][s]),
                                                                      acc[rb][cb], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (GRP == 0) {
            if constexpr (NSTG == 2) {
              // the weights requested in this k-tile's R slot have landed (after the item's first k-tile the previous item's
              // stores, issued behind k-tile 1's weights, may stay in flight)
              if constexpr (FIRST) {
                if (item_first && have_prev) sx_wait_vm<NST>();
                else sx_wait_vm<0>();
              } else {
                sx_wait_vm<0>();
              }
            }
            __builtin_amdgcn_s_barrier();
          } else {
            if constexpr (LAST) {
              if (!item_last) __builtin_amdgcn_s_barrier();  // (group 1, end of an item: the closing barrier comes after its epilogue)
            } else {
              __builtin_amdgcn_s_barrier();
            }
          }
          asm volatile("" ::: "memory");
          kpar = stg_next(kpar);
          ++kt;
        };
        step(integral_constant<int, 0>{}, integral_constant<int, 0>{}, c == 0, false);
        step(integral_constant<int, 0>{}, integral_constant<int, 1>{}, false, false);
        step(integral_constant<int, 0>{}, integral_constant<int, 2>{}, false, false);
        step(integral_constant<int, 0>{}, integral_constant<int, 3>{}, false, false);
        step(integral_constant<int, 0>{}, integral_constant<int, 4>{}, false, false);
        step(integral_constant<int, 0>{}, integral_constant<int, 5>{}, false, false);
        step(integral_constant<int, 0>{}, integral_constant<int, 6>{}, false, false);
        step(integral_constant<int, 0>{}, integral_constant<int, 7>{}, false, false);
        step(integral_constant<int, 0>{}, integral_constant<int, 8>{}, false, HAS_E ? false : last_c);
        if constexpr (HAS_E) step(integral_constant<int, 1>{}, integral_constant<int, 0>{}, false, last_c);
        sbuf ^= 1;
      };
      for (int c = 0; c < a.NCd; ++c) chunk(integral_constant<bool, true>{}, c);
      for (int c = a.NCd; c < a.NC; ++c) chunk(integral_constant<bool, false>{}, c);
      mt_p = mt; nt_p = nt; have_prev = true;
      if (mt1 != mt) { advance_mtile(); plan_x(mt1); }
      mt = mt1; nt = nt1;
    }
    // tail: the last item's epilogue (group 0 one slot before group 1)
    __builtin_amdgcn_s_setprio(0);
    epilogue(mt_p, nt_p, (unsigned)((sbuf ^ 1) * win_bytes));
    __builtin_amdgcn_s_barrier();
  };
  if (grp == 0) run(integral_constant<int, 0>{});
  else run(integral_constant<int, 1>{});
}

// ---------------------------------------------------------------------------------------------------------------
static int s1x_wpx(int W) { return (SX_BM + 2 * W + 2 + 7) & ~7; }

// Weight stages.  Default 2 (DMA duties split by group).  CADRE_S1X_STAGES=3: three stages with every wave issuing weights and
// window pieces (conv3x3_ring_pp_kernel's scheme) where they fit beside the windows — built for A/B runs and measured SLOWER
// here (same box, 2048 frames, conv2 + shortcut: 0.735 / 0.745 vs 0.717 / 0.710 ms on layer3.0 / layer4.0; as a plain conv 0.693 /
// 0.712 vs 0.658 / 0.642: profiles/r05_s1x_three_stages_symmetric_issue.txt, r05_s1x_two_stages_split_issue.txt).
static int s1x_stages(int W, int N) {
  static const int want = [] { const char* e = getenv("CADRE_S1X_STAGES"); return e ? atoi(e) : 2; }();
  const int ntiles = (N + SX_NTILE - 1) / SX_NTILE;
  const size_t fixed = (size_t)2 * ((SX_BM + 2 * W + 2 + 7) & ~7) * 128 + SX_XWIN_B + 1024 + (size_t)ntiles * SX_NTILE * 4;
  return (want == 3 && fixed + 3 * SX_STG_B <= 160 * 1024) ? 3 : 2;
}

// weight stages cadre_conv3x3_s1x runs with at this map width and channel count (the profiling key / kernel name of the launch:
// conv3x3_s1x_kernel<nps, stages>; a request for 3 falls back to 2 where three stages do not fit beside the windows)
extern "C" int cadre_conv3x3_s1x_stages(int32_t W, int32_t N) { return s1x_stages(W, N); }

static int s1x_capable(int F, int H, int W, int C1, int Cd, int N) {
  if (F < 1 || H < 1 || W < 2 || W > 46) return 0;          // 4 * 11 pieces of 8 pixels >= 256 + 2 W + 2
  if (C1 % 64 != 0 || Cd % 64 != 0 || C1 < 64 || Cd < 0 || N % 32 != 0) return 0;      // (Cd == 0: no shortcut, a plain 3x3 / s1 conv)
  if (Cd / 64 > C1 / 64) return 0;                          // a shortcut k-tile rides behind each of the first NCd chunks
  const long long M = (long long)F * H * W, lim = 1ll << 31;
  if (M * C1 * 2 >= lim || 4 * M * Cd * 2 >= lim || (long long)N * (9 * C1 + Cd) * 2 >= lim || M * N * 2 >= lim) return 0;
  if (M >= (1 << 23)) return 0;
  const int ntiles = (N + SX_NTILE - 1) / SX_NTILE;
  if ((size_t)2 * s1x_wpx(W) * 128 + SX_XWIN_B + 2 * SX_STG_B + 1024 + (size_t)ntiles * SX_NTILE * 4 > 160 * 1024) return 0;
  return 1;
}

static const int g_s1x_on = [] { const char* e = getenv("CADRE_S1X_CONV"); return e ? atoi(e) : 1; }();

extern "C" int cadre_conv3x3_s1x_supported(int32_t F, int32_t H, int32_t W, int32_t C1, int32_t Cd, int32_t N) {
  return (g_s1x_on && s1x_capable(F, H, W, C1, Cd, N)) ? 1 : 0;
}

extern "C" int cadre_conv3x3_s1x(const void* x, const void* x2, const void* w, const float* shift, void* out, int32_t F, int32_t H,
                                 int32_t W, int32_t C1, int32_t Cd, int32_t N, int32_t act, void* stream) {
  if (!x || (!x2 && Cd > 0) || !w || !out) return cadre_fail("cadre_conv3x3_s1x: null operand");
  if (!x2) x2 = x;
  if (!s1x_capable(F, H, W, C1, Cd, N))
    return cadre_fail("cadre_conv3x3_s1x: unsupported geometry (W in 2..46; C1, Cd multiples of 64 with Cd <= C1; N % 32 == 0; every tensor < 2 GiB: chunk the batch)");
  if ((act & 15) > 1 || (act & 16)) return cadre_fail("cadre_conv3x3_s1x: act 0 (none) or 1 (ReLU)");
  if (((uintptr_t)x | (uintptr_t)x2 | (uintptr_t)w | (uintptr_t)out) & 15) return cadre_fail("cadre_conv3x3_s1x: operands must be 16-byte aligned");
  s1x_args a;
  a.x = x; a.x2 = x2; a.w = w; a.shift = shift; a.out = out;
  a.H = H; a.W = W; a.C1 = C1; a.Cd = Cd; a.N = N; a.NC = C1 / 64; a.NCd = Cd / 64; a.KT = 9 * a.NC + a.NCd; a.act = act;
  a.M = F * H * W; a.M2 = 4 * a.M;
  a.mtiles = (a.M + SX_BM - 1) / SX_BM;
  a.ntiles = (N + SX_NTILE - 1) / SX_NTILE;
  a.items = a.mtiles * a.ntiles;
  const int wgs = a.items < 256 ? a.items : 256;
  a.ipw = (a.items + wgs - 1) / wgs;
  const int grid = (a.items + a.ipw - 1) / a.ipw;
  a.WPX = s1x_wpx(W); a.PA = a.WPX / 8;
  const int nstg = s1x_stages(W, N);
  const size_t lds = (size_t)2 * a.WPX * 128 + SX_XWIN_B + nstg * SX_STG_B + 1024 + (size_t)a.ntiles * SX_NTILE * 4;
  hipStream_t st = (hipStream_t)stream;
#define SX_LAUNCH2(NPS_, NSTG_)                                                                                                       \
  do {                                                                                                                                \
    (void)hipFuncSetAttribute((const void*)conv3x3_s1x_kernel<NPS_, NSTG_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    hipLaunchKernelGGL((conv3x3_s1x_kernel<NPS_, NSTG_>), dim3(grid), dim3(512), lds, st, a);                                        \
  } while (0)
#define SX_LAUNCH(NPS_) do { if (nstg == 3) SX_LAUNCH2(NPS_, 3); else SX_LAUNCH2(NPS_, 2); } while (0)
  if (a.PA <= 36) SX_LAUNCH(9);
  else if (a.PA <= 40) SX_LAUNCH(10);
  else SX_LAUNCH(11);
#undef SX_LAUNCH2
#undef SX_LAUNCH
  return (int)hipGetLastError();
}
