// gemm_f32.hip — LDS-tiled fp32 GEMM / implicit-GEMM convolution for gfx950 (MI355X).
//
// One kernel template covers every matmul-shaped op on the Cadre PPO hot path:
//   * a_mode 0/1 x b_mode 0/1: NT / NN / TN products for nn.Linear and LSTMCell forward,
//     dX and dW (reference ppo_agent/models.py:130-177, distributions.py:34-40,
//     carla_perception/Networks/danet_blocks/intertask_att.py:39-80);
//   * a_mode 2: NHWC implicit-GEMM conv, Cin % 32 == 0 (resnet.py:26-55,152-166,
//     danet.py:21-41,96,108); a_mode 3: the Cin==4 7x7 stem (resnet.py:111-112).
//
// MI355X mapping: 256-thread workgroup = 4 wave64, 2x2 waves, each wave WM x WN tiles of
// v_mfma_f32_32x32x2_f32 (exact f32 fma chain, 64 FLOP/clk/SIMD = the fp32 roof).  BK = 32.
// Tiles are register-staged (16-B global loads, zero fill for halo / tails) into a
// double-buffered LDS image; k-contiguous operands use a 36-float row pitch so that
// ds_read_b128 fragment reads are bank-conflict free (MI355X_MICROARCH.md §LDS), k-major
// operands are stored as they arrive and read with conflict-free ds_read_b32.
// The k order inside a 8-deep step is permuted (lane half h owns k = 8*kb + 4*h + i) — the
// same permutation on A and B, so the products pair correctly.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "../../include/cadre_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static_assert(sizeof(cadre_gemm_t) == 264, "cadre_gemm_t layout is part of the C ABI (ctypes mirror in cadre_amd/hip.py)");

#define BK 32
#define LDS_PITCH 36
#ifndef GEMM_NSETS
#define GEMM_NSETS 2      // register sets = k-tiles of global loads in flight per wave
#endif

// WVM x WVN = waves along M x N (64 threads each); each wave owns WM x WN MFMA tiles of 32x32.
// NS = register sets = k-tiles of global loads in flight per wave (3 and 4 measured: no gain, also not on the
// update's skinny weight-streaming GEMMs — those are bound by one MFMA chain per SIMD, not by load latency).
template <int WM, int WN, int AMODE, int BMODE, int WVN = 2, int NS = GEMM_NSETS, int WVM = 2>
__global__ __launch_bounds__(64 * WVM * WVN, ((WVM * WM + WVN * WN) * 2 * 32 * 36 * 4 > 80 * 1024 ? 1 : 2)) void gemm_f32_kernel(cadre_gemm_t p) {
  constexpr int NT = 64 * WVM * WVN;   // threads
  constexpr int RP = NT / 8;      // rows staged per pass (8 x 16-B chunks per 128-B row)
  constexpr int BM = WVM * WM * 32;
  constexpr int BN = WVN * WN * 32;
  constexpr int RA = BM / RP;     // 16-B chunks per thread per A tile
  constexpr int RB = BN / RP;
  __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * LDS_PITCH];
  float* As = lds;
  float* Bs = lds + 2 * BM * LDS_PITCH;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WVN, wn = wave % WVN;
  const int l31 = lane & 31, lh = lane >> 5;

  // XCD-aware tile order (cdna_hip_programming.md T1, bijective form): workgroups are dealt
  // round-robin over the 8 XCDs, so give each XCD a CONTIGUOUS run of tiles — neighbouring M-tiles
  // share conv halo rows and the N-tiles of one M-tile share the A panel, which then hit that
  // XCD's private L2 instead of being re-fetched by up to 8 L2s.  Placement only affects speed.
  const int tilesN = (p.N + BN - 1) / BN;
  int bid = blockIdx.x;
#ifndef GEMM_NO_XCD_REMAP
  if (p.seg_mode != 3) {      // (compact rows: the live tiles are the grid's first ones — dealt round-robin they cover all XCDs)
    const int nwg = gridDim.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
#endif
  const int tile_m = bid / tilesN, tile_n = bid % tilesN;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int z = blockIdx.z;
  const float* A = p.A;
  const float* B = p.B;
  float* C = p.C;
  // batch entry -> operand slot: (z / div) % mod.  A scalar division is ~60 dependent SALU instructions, and the
  // update's skinny GEMMs are short enough to notice: take the common cases (div 1, mod not reached) by branch.
  auto slot = [](int zz, int dv, int md) {
    const int q = dv == 1 ? zz : zz / dv;
    return q < md ? q : q % md;
  };
  if (p.batch > 1) {
    A += (int64_t)slot(z, p.a_div, p.a_mod) * p.a_str;
    B += (int64_t)slot(z, p.b_div, p.b_mod) * p.b_str;
    C += (int64_t)slot(z, p.c_div, p.c_mod) * p.c_str;
  }

  // row segments (rows sorted by command): skip what this batch entry does not own
  int seg_beg = 0, seg_cnt = 0;
  if (p.seg_mode) {
    const int32_t* sg = p.row_seg + 2 * (p.seg_div == 1 ? z : z / p.seg_div);
    seg_beg = sg[0];
    seg_cnt = sg[1];
    if (p.seg_mode == 1) {
      const int b_lo = m0 % p.seg_period;
      if (seg_cnt <= 0 || b_lo + BM <= seg_beg || b_lo >= seg_beg + seg_cnt) return;
    }
  }
  // seg_mode 3: the M index runs over the batch entry's OWN rows only, period after period — compact row mc is row
  // (mc / cnt) * seg_period + beg + mc % cnt of A and C — so a tile holds BM useful rows wherever the run starts
  int c_cnt = 1, c_rows = 0;
  if constexpr (AMODE == 0 && BMODE == 0) {
    if (p.seg_mode == 3) {
      c_cnt = max(seg_cnt, 1);
      c_rows = seg_cnt > 0 ? (p.M / p.seg_period) * seg_cnt : 0;
      if (m0 >= c_rows) return;
    }
  }
  auto compact_row = [&](int mc) {          // mc < c_rows
    const int t = mc / c_cnt;
    return t * p.seg_period + seg_beg + (mc - t * c_cnt);
  };
  const int nk_total = (p.K + BK - 1) / BK;
  int kt_begin = 0, kt_end = nk_total;
  int k_per = 1, k_first = 0, k_run = 1;       // seg_mode 2: k-tiles [k_first, k_first+k_run) of each period
  if (p.seg_mode == 2) {
    k_per = p.seg_period / BK;
    k_first = seg_beg / BK;
    k_run = seg_cnt > 0 ? (seg_beg + seg_cnt + BK - 1) / BK - k_first : 0;
    kt_end = (nk_total / k_per) * k_run;
  }
  if (p.split_k > 1) {
    const int per = (nk_total + p.split_k - 1) / p.split_k;
    kt_begin = blockIdx.y * per;
    kt_end = min(nk_total, kt_begin + per);
    // slab s of a batched problem holds every z-slot: [split][slots][M][ldc]
    const int64_t slots = p.batch > 1 ? min((p.batch + p.c_div - 1) / p.c_div, p.c_mod) : 1;
    C += (int64_t)blockIdx.y * (p.batch > 1 ? slots * p.c_str : (int64_t)p.M * p.ldc);
  }

  // ---------------------------------------------------------------- staging setup
  // All global->register staging goes through raw buffer loads (SRSRC form): the byte offset is a
  // 32-bit VGPR, and any offset >= num_records (2 GiB window) returns zeros in hardware.  Halo taps,
  // M/N tails and the K tail are therefore "loads from OOB" — no branches, no exec masking, no
  // 64-bit address arithmetic in the k-loop, so the compiler can interleave the loads with MFMAs.
  // k-contiguous operand: thread owns chunk column cc (4 floats of the 32-deep tile) of rows
  // rr + 32*i.  k-major operand: tile is [32 k][BM] floats, thread owns chunks id = tid+256*i.
  constexpr unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)OOB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (int)OOB, 0x00020000);
  const int cc = tid & 7, rr = tid >> 3;
  unsigned aoff[RA];            // byte offset of the row's first staged chunk (or OOB)
  unsigned amask[RA];           // conv: bit t set = tap t of the receptive field is inside the image
  if constexpr (AMODE == 0) {
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const int m = m0 + rr + RP * i;
      aoff[i] = m < p.M ? (unsigned)(((int64_t)m * p.lda + cc * 4) * 4) : OOB;
      if constexpr (BMODE == 0) {
        if (p.seg_mode == 3) aoff[i] = m < c_rows ? (unsigned)(((int64_t)compact_row(m) * p.lda + cc * 4) * 4) : OOB;
      }
      amask[i] = 0;
    }
  } else if constexpr (AMODE >= 2) {
    // implicit-GEMM gather, decoded once per staged row: byte offset of the receptive field's
    // top-left tap (negative in the top halo -> wraps to OOB, masked anyway) + in-bounds tap mask.
    // The tile's first row m0 is decoded with scalar divisions (wave-uniform -> SALU); a staged row is
    // m0 + r with r < BM, so its (img, ho, wo) follow from two small carries.  floor(x / d) for
    // 0 <= x < 2^16 is exact as (int)((x + 0.5f) * (1.0f / d)): the argument stays 0.5/d away from
    // every integer, far more than the rounding error — no per-lane integer division.
    const int hw = p.Ho * p.Wo;
    const int img0 = m0 / hw, rem0 = m0 % hw;
    const int ho0 = rem0 / p.Wo, wo0 = rem0 % p.Wo;
    const float inv_wo = 1.0f / (float)p.Wo, inv_ho = 1.0f / (float)p.Ho;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const int r = rr + RP * i;
      const int m = m0 + r;
#ifdef ABL_NOPRO   // ablation: no im2col decode
      aoff[i] = (unsigned)(((m > 2 * p.W + 4 ? m - 2 * p.W - 4 : 0) * p.Cin + cc * 4) * 4); amask[i] = m < p.M ? 0x1FF : 0; continue;
#endif
      const int x = wo0 + r;
      const int q1 = (int)(((float)x + 0.5f) * inv_wo);
      const int wo = x - __mul24(q1, p.Wo);
      const int y = ho0 + q1;
      const int q2 = (int)(((float)y + 0.5f) * inv_ho);
      const int ho = y - __mul24(q2, p.Ho);
      const int img = img0 + q2;
      const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
      unsigned mask = 0;
      if constexpr (AMODE == 2) {
        aoff[i] = (unsigned)((((img * p.H + hi0) * p.W + wi0) * p.Cin + cc * 4) * 4);
        if (m < p.M) {     // separable: (rows inside) x (columns inside)
          unsigned colm = 0;
          for (int kw = 0; kw < p.KW; ++kw)
            if ((unsigned)(wi0 + kw) < (unsigned)p.W) colm |= 1u << kw;
          for (int kh = 0; kh < p.KH; ++kh)
            if ((unsigned)(hi0 + kh) < (unsigned)p.H) mask |= colm << (kh * p.KW);
        }
      } else {
        // Cin == 4 stem, "row" formulation: k-tile kt = kernel row kh; its 32 k-values are the 8
        // consecutive NHWC4 pixels wi0 .. wi0+7 of input row hi0+kh (128 contiguous bytes), of which the
        // first KW carry weights (B is [N][KH][32], zero beyond KW*4).  One scalar delta per k-tile, no
        // per-thread tap decode.  This thread's chunk cc is pixel wi0+cc: masked when outside the row.
        aoff[i] = (unsigned)((((img * p.H + hi0) * p.W + wi0) * 4 + cc * 4) * 4);
        if (m < p.M && cc < p.KW && (unsigned)(wi0 + cc) < (unsigned)p.W) {
          for (int kh = 0; kh < p.KH; ++kh)
            if ((unsigned)(hi0 + kh) < (unsigned)p.H) mask |= 1u << kh;
        }
      }
      amask[i] = mask;
    }
  }
  unsigned boff[RB];
  if constexpr (BMODE == 0) {
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int n = n0 + rr + RP * i;
      boff[i] = n < p.N ? (unsigned)(((int64_t)n * p.ldb + cc * 4) * 4) : OOB;
    }
  }

  // NS register sets hold tiles in flight; set s is written to LDS and re-requested NS tiles ahead.
  constexpr int U = (NS % 2 == 0) ? NS : 2 * NS;     // steps per unrolled group: set and LDS-buffer parity both static
  f32x4 areg[NS][RA], breg[NS][RB];
  auto ldg = [](const __amdgpu_buffer_rsrc_t& rs, unsigned off) -> f32x4 {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
  };

  // Request k-tile kt_ into register set rs.  Tiles past the end resolve to OOB offsets (zero fill),
  // so the k-loop can issue its loads unconditionally.
  auto load_tiles = [&](int kt_, int rs) {
#ifdef ABL_NOGLOAD   // ablation: keep the LDS writes, skip the global loads after the first tile
    if (kt_ != kt_begin) return;
#endif
    const int kr = max(k_run, 1);
    const int kt = p.seg_mode == 2 ? (kt_ / kr) * k_per + k_first + kt_ % kr : kt_;
    const int k0 = kt * BK;
    // ---- A
    if constexpr (AMODE == 0) {
      const unsigned kb_ = (k0 + cc * 4 < p.K) ? (unsigned)k0 * 4u : OOB;   // K % 4 == 0
#pragma unroll
      for (int i = 0; i < RA; ++i) areg[rs][i] = ldg(rsA, aoff[i] + kb_);
    } else if constexpr (AMODE == 1) {
#pragma unroll
      for (int i = 0; i < RA; ++i) {
        const int id = tid + NT * i;
        const int kk = id / (BM / 4), mc = id % (BM / 4);
        const int k = k0 + kk, m = m0 + mc * 4;
        areg[rs][i] = ldg(rsA, (k < p.K && m < p.M) ? (unsigned)(((int64_t)k * p.lda + m) * 4) : OOB);
      }
    } else if constexpr (AMODE == 2) {
      const int pos = k0 / p.Cin, ci = k0 % p.Cin;             // uniform per tile (Cin % 32 == 0)
      const unsigned delta = (unsigned)((((pos / p.KW) * p.W + (pos % p.KW)) * p.Cin + ci) * 4);
      const unsigned bit = pos < 32 ? 1u << pos : 0u;
#pragma unroll
      for (int i = 0; i < RA; ++i) areg[rs][i] = ldg(rsA, (amask[i] & bit) ? aoff[i] + delta : OOB);
    } else {  // stem rows: k-tile kt is kernel row kh
      const unsigned delta = (unsigned)(kt * p.W * 16);
      const unsigned bit = kt < 32 ? 1u << kt : 0u;
#pragma unroll
      for (int i = 0; i < RA; ++i) areg[rs][i] = ldg(rsA, (amask[i] & bit) ? aoff[i] + delta : OOB);
    }
    // ---- B
    if constexpr (BMODE == 0) {
      const unsigned kb_ = (k0 + cc * 4 < p.K) ? (unsigned)k0 * 4u : OOB;
#pragma unroll
      for (int i = 0; i < RB; ++i) breg[rs][i] = ldg(rsB, boff[i] + kb_);
    } else {
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        const int id = tid + NT * i;
        const int kk = id / (BN / 4), nc = id % (BN / 4);
        const int k = k0 + kk, n = n0 + nc * 4;
        breg[rs][i] = ldg(rsB, (k < p.K && n < p.N) ? (unsigned)(((int64_t)k * p.ldb + n) * 4) : OOB);
      }
    }
  };

  auto store_tiles = [&](int buf, int rs) {
    float* as = As + buf * BM * LDS_PITCH;
    float* bs = Bs + buf * BN * LDS_PITCH;
#ifdef ABL_NOSTORE   // ablation: keep the loads alive, skip the LDS writes
#pragma unroll
    for (int i = 0; i < RA; ++i) asm volatile("" ::"v"(areg[rs][i]));
#pragma unroll
    for (int i = 0; i < RB; ++i) asm volatile("" ::"v"(breg[rs][i]));
    return;
#endif
    if constexpr (AMODE == 1) {
#pragma unroll
      for (int i = 0; i < RA; ++i) *reinterpret_cast<f32x4*>(as + (tid + NT * i) * 4) = areg[rs][i];
    } else {
#pragma unroll
      for (int i = 0; i < RA; ++i)
        *reinterpret_cast<f32x4*>(as + (rr + RP * i) * LDS_PITCH + cc * 4) = areg[rs][i];
    }
    if constexpr (BMODE == 1) {
#pragma unroll
      for (int i = 0; i < RB; ++i) *reinterpret_cast<f32x4*>(bs + (tid + NT * i) * 4) = breg[rs][i];
    } else {
#pragma unroll
      for (int i = 0; i < RB; ++i)
        *reinterpret_cast<f32x4*>(bs + (rr + RP * i) * LDS_PITCH + cc * 4) = breg[rs][i];
    }
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // One k-tile of MFMAs from LDS buffer `buf`; `staging()` runs behind the first fragment reads.
  auto compute = [&](int buf, auto&& staging) {
    const float* as = As + buf * BM * LDS_PITCH;
    const float* bs = Bs + buf * BN * LDS_PITCH;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      f32x4 af[WM], bf[WN];
      const int kq = kb * 8 + lh * 4;
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        const int row = (wm * WM + i) * 32 + l31;
        if constexpr (AMODE == 1) {
          af[i][0] = as[(kq + 0) * BM + row];
          af[i][1] = as[(kq + 1) * BM + row];
          af[i][2] = as[(kq + 2) * BM + row];
          af[i][3] = as[(kq + 3) * BM + row];
        } else {
#ifdef ABL_NOLDSREAD
          af[i] = f32x4{(float)row, (float)kq, 1.f, 2.f};
#else
          af[i] = *reinterpret_cast<const f32x4*>(as + row * LDS_PITCH + kq);
#endif
        }
      }
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int col = (wn * WN + j) * 32 + l31;
        if constexpr (BMODE == 1) {
          bf[j][0] = bs[(kq + 0) * BN + col];
          bf[j][1] = bs[(kq + 1) * BN + col];
          bf[j][2] = bs[(kq + 2) * BN + col];
          bf[j][3] = bs[(kq + 3) * BN + col];
        } else {
#ifdef ABL_NOLDSREAD
          bf[j] = f32x4{(float)col, (float)kq, 1.f, 2.f};
#else
          bf[j] = *reinterpret_cast<const f32x4*>(bs + col * LDS_PITCH + kq);
#endif
        }
      }
      if (kb == 0) staging();
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int j = 0; j < WN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
    }
  };

  // k-loop.  Iteration t: barrier; compute tile t from LDS buffer t&1; behind its first fragment reads,
  // write tile t+1 (register set (t+1)%NS, requested NS iterations ago) into the buffer compute(t-1)
  // just released and re-request that set for tile t+1+NS.  No wave then waits on its own loads or LDS
  // writes in front of a barrier (the old order — request, compute, write, barrier — cost 6-8 %: every
  // tile ended in s_waitcnt vmcnt(0) -> 4 ds_write -> s_waitcnt lgkmcnt(0) -> s_barrier), and a global
  // load has NS iterations to land.  The loop runs in groups of U steps with nothing conditional
  // inside, so hipcc's counted s_waitcnt vmcnt(N) stay exact.
  auto step = [&](auto uc, int kt) {
    constexpr int u = decltype(uc)::value;
#ifndef ABL_NOBAR
    __syncthreads();
#endif
    compute(u & 1, [&] {
#ifndef ABL_NOLOAD
      store_tiles((u + 1) & 1, (u + 1) % NS);
      load_tiles(kt + 1 + NS, (u + 1) % NS);
#endif
    });
  };
  load_tiles(kt_begin, 0);
  store_tiles(0, 0);
#pragma unroll
  for (int j = 1; j <= NS; ++j) load_tiles(kt_begin + j, j % NS);
  int kt = kt_begin;
  for (; kt + U <= kt_end; kt += U) {
    step(std::integral_constant<int, 0>{}, kt);
    step(std::integral_constant<int, 1>{}, kt + 1);
    if constexpr (U > 2) {
      step(std::integral_constant<int, 2>{}, kt + 2);
      step(std::integral_constant<int, 3>{}, kt + 3);
    }
    if constexpr (U > 4) {
      step(std::integral_constant<int, 4>{}, kt + 4);
      step(std::integral_constant<int, 5>{}, kt + 5);
    }
  }
  if (kt < kt_end) step(std::integral_constant<int, 0>{}, kt);
  if (kt + 1 < kt_end) step(std::integral_constant<int, 1>{}, kt + 1);
  if constexpr (U > 2) {
    if (kt + 2 < kt_end) step(std::integral_constant<int, 2>{}, kt + 2);
    if (kt + 3 < kt_end) step(std::integral_constant<int, 3>{}, kt + 3);
  }
  if constexpr (U > 4) {
    if (kt + 4 < kt_end) step(std::integral_constant<int, 4>{}, kt + 4);
  }
  __syncthreads();      // every wave is done reading: the epilogue re-uses the staging buffers


  // ---------------------------------------------------------------- epilogue
#ifdef ABL_NOEPI   // ablation: no epilogue (one never-taken store keeps the accumulators alive)
  if (acc[0][0][0] == 12345.678f) p.C[0] = acc[0][0][1] + acc[WM - 1][WN - 1][15];
  return;
#endif
  const bool raw = p.split_k > 1;
  const int actk = p.act & 15;
  const bool post = (p.act & 16) != 0;   // residual added after the activation
  const float* scale = raw ? nullptr : p.scale;
  const float* shift = raw ? nullptr : p.shift;
  const float* resid = raw ? nullptr : p.resid;
  if (p.batch > 1) {
    const int64_t so = (int64_t)slot(z, p.s_div, p.s_mod) * p.s_str;
    if (scale) scale += so;
    if (shift) shift += so;
    if (resid) resid += (int64_t)slot(z, p.r_div, p.r_mod) * p.r_str;
  }
  const bool vec_ok = ((p.N | p.ldc | (resid ? p.ldr : 0)) & 3) == 0 && (((uintptr_t)C | (uintptr_t)resid) & 15) == 0;
  if (vec_ok) {
    // Stage each wave's accumulator tile through its own LDS slice so that global traffic is whole
    // 16-B-per-lane row segments: residual read and output write are each R/rows_per_instr wide
    // instructions per lane instead of 16*WM*WN scalar ones.  (The k-loop's last barrier has passed,
    // so the staging buffers are free.)  The body is instantiated per (activation, residual) so that
    // it is straight-line code — with the modes as run-time uniforms hipcc branched per element — and
    // C / resid go through buffer descriptors rebased at the tile's first row: 32-bit offsets, rows
    // past M dropped by the hardware bounds check, no exec masking.  (Short-K convs spend up to 6 %
    // of their time here: tools/ablate_proepi.sh.)
    constexpr int CW = WN * 32, P = CW + 4;
    constexpr int LPR = CW / 4;          // lanes per row
    constexpr int RPI = 64 / LPR;        // rows per wave-instruction
    constexpr int NIT = 32 / RPI;        // instructions per 32-row slab
    float* cs = lds + wave * (32 * P);
    const int c4 = (lane % LPR) * 4;
    const int col = n0 + wn * CW + c4;
    const bool cvalid = col < p.N;
    const bool c16 = (p.flags & 2) != 0;      // C is bf16 (config C3: fp32 stem feeding the bf16 trunk)
    const int esz = c16 ? 2 : 4;
    bool seg3 = false;
    if constexpr (AMODE == 0 && BMODE == 0) seg3 = p.seg_mode == 3;
    const int64_t rows_left = seg3 ? (int64_t)p.M : (int64_t)p.M - m0;      // (compact rows: offsets from the operand's first row)
    auto window = [](int64_t bytes) { return (int)(bytes < 0x7fffffff ? bytes : 0x7fffffff); };
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<char*>(C) + (seg3 ? 0 : (int64_t)m0 * p.ldc * esz)), 0, window(rows_left * p.ldc * esz), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(resid ? resid + (int64_t)m0 * p.ldr : p.C), 0, resid ? window(rows_left * p.ldr * 4) : 0, 0x00020000);
    const int lrow = lane / LPR;
    const unsigned coff = cvalid ? (unsigned)((lrow * p.ldc + col) * esz) : OOB;     // row `lrow` of the tile
    const unsigned roff = cvalid ? (unsigned)((lrow * p.ldr + col) * 4) : OOB;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (cvalid && scale) sc = *reinterpret_cast<const f32x4*>(scale + col);
    if (cvalid && shift) sh = *reinterpret_cast<const f32x4*>(shift + col);
    const float slope = p.slope;

    auto body = [&](auto actc, auto resc) {
      constexpr int ACT = decltype(actc)::value;      // -1 raw split-K slab, 0 none, 1 ReLU, 2 LeakyReLU
      constexpr bool RES = decltype(resc)::value;
#pragma unroll
      for (int i = 0; i < WM; ++i) {          // one 32-row slab of the wave tile at a time
        const int r0 = (wm * WM + i) * 32;    // slab's first row inside the tile
        f32x4 rv[NIT];
        if constexpr (RES) {
#pragma unroll
          for (int it = 0; it < NIT; ++it)
            rv[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsR, (int)(roff + (unsigned)((r0 + it * RPI) * p.ldr * 4)), 0, 0));
        }
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            cs[((r & 3) + 8 * (r >> 2) + 4 * lh) * P + j * 32 + l31] = acc[i][j][r];
        // same-wave producer/consumer: LDS ops of one wave execute in order, no barrier needed
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          f32x4 v = *reinterpret_cast<const f32x4*>(cs + (it * RPI + lrow) * P + c4);
          if constexpr (ACT >= 0) {
            v = v * sc + sh;
            if constexpr (RES) { if (!post) v += rv[it]; }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if constexpr (ACT == 1) v[e] = fmaxf(v[e], 0.f);
              if constexpr (ACT == 2) v[e] = v[e] > 0.f ? v[e] : v[e] * slope;
            }
            if constexpr (RES) { if (post) v += rv[it]; }
          }
          unsigned off = coff + (unsigned)((r0 + it * RPI) * p.ldc * esz);
          if constexpr (AMODE == 0 && BMODE == 0) {
            if (seg3) {
              const int mc = m0 + r0 + it * RPI + lrow;
              off = (cvalid && mc < c_rows) ? (unsigned)((compact_row(mc) * p.ldc + col) * esz) : OOB;
            }
          }
          if (c16) {
            typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            bf16x4 o;
            o[0] = (__bf16)v[0]; o[1] = (__bf16)v[1]; o[2] = (__bf16)v[2]; o[3] = (__bf16)v[3];
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), rsC, (int)off, 0, 0);
          } else {
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsC, (int)off, 0, 0);
          }
        }
      }
    };
    using std::integral_constant;
    if (raw) body(integral_constant<int, -1>{}, integral_constant<bool, false>{});
    else if (resid) {
      if (actk == 1) body(integral_constant<int, 1>{}, integral_constant<bool, true>{});
      else if (actk == 2) body(integral_constant<int, 2>{}, integral_constant<bool, true>{});
      else body(integral_constant<int, 0>{}, integral_constant<bool, true>{});
    } else {
      if (actk == 1) body(integral_constant<int, 1>{}, integral_constant<bool, false>{});
      else if (actk == 2) body(integral_constant<int, 2>{}, integral_constant<bool, false>{});
      else body(integral_constant<int, 0>{}, integral_constant<bool, false>{});
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    const int col = n0 + (wn * WN + j) * 32 + l31;
    if (col >= p.N) continue;
    const float sc = scale ? scale[col] : 1.f;
    const float sh = shift ? shift[col] : 0.f;
#pragma unroll
    for (int i = 0; i < WM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + (wm * WM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row >= p.M) continue;
        float v = acc[i][j][r];
        if (!raw) {
          v = v * sc + sh;
          const float rv = resid ? resid[(int64_t)row * p.ldr + col] : 0.f;
          if (!post) v += rv;
          if (actk == 1) v = fmaxf(v, 0.f);
          else if (actk == 2) v = v > 0.f ? v : v * p.slope;
          if (post) v += rv;
        }
        C[(int64_t)row * p.ldc + col] = v;
      }
    }
  }
}

__global__ void splitk_reduce_kernel(const float* slabs, int split_k, int64_t slab_stride, int64_t lds_,
                                     float* C, int64_t ldc, int M, int N, const float* scale,
                                     const float* shift, int act, float slope, const int32_t* row_seg, int period) {
  const int64_t total = (int64_t)M * N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int m = (int)(i / N), n = (int)(i % N);
    if (row_seg) {       // rows [z][period] sorted by command: only the 32-row tiles of net z's run were written by the GEMM
      const int z = m / period, b = m - z * period;
      const int beg = row_seg[2 * z], cnt = row_seg[2 * z + 1];
      if (cnt <= 0 || b < (beg & ~31) || b >= ((beg + cnt + 31) & ~31)) continue;
    }
    float v = 0.f;
    for (int s = 0; s < split_k; ++s) v += slabs[s * slab_stride + (int64_t)m * lds_ + n];
    v = v * (scale ? scale[n] : 1.f) + (shift ? shift[n] : 0.f);
    if (act == 1) v = fmaxf(v, 0.f);
    else if (act == 2) v = v > 0.f ? v : v * slope;
    C[(int64_t)m * ldc + n] = v;
  }
}

extern thread_local char g_cadre_err[256];
int cadre_fail(const char* msg);

#ifdef CADRE_AB_KERNELS      // A/B build only (csrc/ab/, include/cadre_hip_ab.h): kernels the dispatch superseded
int cadre_conv_stream_f32_launch(const cadre_gemm_t& p, void* stream);     // ab/conv_stream_f32.hip (tile 12)
int cadre_gemm_f32_skinny_ok(const cadre_gemm_t& p);                       // ab/gemm_f32_skinny.hip (tile 11)
int cadre_gemm_f32_skinny_launch(const cadre_gemm_t& p, void* stream);
#endif

#define GEMM_CHECK(cond, msg) \
  if (!(cond)) return cadre_fail("cadre_gemm_f32: " msg)

// Auto tile: estimated efficiency = (measured steady-state factor of the tile shape) x (wave
// quantisation over 256 CUs x resident workgroups per CU).  Factors from tools/gemm_bench.py on
// MI355X (profiles/r01_gemm_tile_sweep.txt).
#ifdef CADRE_AB_KERNELS
int cadre_gemm_stream_f32_launch(const cadre_gemm_t& p, void* stream);   // ab/gemm_stream_f32.hip (tile 13)
#endif

static int pick_tile(const cadre_gemm_t& p) {
  const int batch = p.batch < 1 ? 1 : p.batch, sk = p.split_k < 1 ? 1 : p.split_k;
  struct Cand { int id, bm, bn, per_cu; double base; };
  // measured with the staged k-loop (profiles/r01_gemm_tile_sweep_staged.txt), F=512:
  //   dense 4096^3: 128x128 on 8 waves 146, 128x128 on 4 waves 142, 64x64 131 TFLOP/s
  //   conv N=128 (layer2): 8-wave 128x128 130 vs 64x64 126;  N=256: 129 vs 131;  N=512: 121 vs 131
  // (the 64x64 tile keeps four workgroups per CU, which hides the gather better once N tiles multiply)
  static const Cand big[2] = {{8, 128, 128, 2, 1.00}, {3, 64, 64, 4, 0.92}};
  static const Cand conv128[2] = {{8, 128, 128, 2, 1.03}, {3, 64, 64, 4, 1.00}};
  static const Cand conv_wide[2] = {{8, 128, 128, 2, 0.93}, {3, 64, 64, 4, 1.00}};
  static const Cand narrow[2] = {{2, 128, 64, 2, 0.88}, {3, 64, 64, 4, 1.00}};
  const Cand* big_conv = p.N <= 128 ? conv128 : conv_wide;
#ifdef CADRE_AB_KERNELS
  // Cin = 4 stem (7 k-tiles per 64x64 tile): several M-tiles per workgroup with the prefetch running across
  // tile boundaries, ab/conv_stream_f32.hip — 101 vs 94 TFLOP/s; on K >= 576 the one-tile kernel's second
  // register set is worth more than the hidden start-up (121 vs 125, 128 vs 134).  (The product's stem is the fused
  // front, stem_pool.hip: this only serves geometries that one does not cover.)
  if (p.a_mode == 3 && batch == 1 && sk == 1 && !p.seg_mode && p.M >= 64 * 2048 &&
      ((p.N | p.ldc | (p.resid ? p.ldr : 0)) & 3) == 0 && (((uintptr_t)p.C | (uintptr_t)p.resid) & 15) == 0)
    return 12;
  // row-sorted minibatch (each batch entry owns one run of rows per period): 32-row tiles skip the most.  The
  // register-direct kernel (ab/gemm_f32_skinny.hip, tile 11) is opt-in: measured on the update's launches it only wins
  // the backward at B = 64 (31 vs 39 us with the split-K pass) and loses the forward at B = 256 (39.5 vs 23.8 us)
  static const int skinny = [] { const char* e = getenv("CADRE_SKINNY_GEMM"); return e ? atoi(e) : 0; }();
  if (p.seg_mode == 1 && skinny && cadre_gemm_f32_skinny_ok(p)) return 11;
#endif
  if (p.seg_mode == 1 && (p.N >= 96 || p.seg_period % 64 != 0)) return 9;
  if (p.seg_mode == 3) return p.N >= 96 ? 9 : 3;
  const Cand* c = p.N <= 64 ? narrow : (p.a_mode >= 2 ? big_conv : big);
  int best = c[0].id;
  double best_e = -1.0;
  // matrix-pipe saturation with r workgroups resident on a CU (4-wave / 8-wave workgroups)
  static const double sat4[5] = {0.55, 0.55, 0.80, 0.93, 1.00}, sat8[3] = {0.85, 0.85, 1.00};
  for (int i = 0; i < 2; ++i) {
    const double tiles = (double)((p.M + c[i].bm - 1) / c[i].bm) * ((p.N + c[i].bn - 1) / c[i].bn) * batch * sk;
    // useful fraction of the padded tile grid (edge tiles)
    const double useful = ((double)p.M * p.N * batch * sk) / (tiles * c[i].bm * c[i].bn);
    // a CU's throughput does not depend on how many of its tiles are resident at once, so the tail is
    // per CU: ceil(tiles/256) tile-times for tiles/256 tiles of work
    const double per_cu = tiles / 256.0;
    const double quant = per_cu / (double)(int64_t)(per_cu + 0.999999);
    const double r = per_cu < c[i].per_cu ? (per_cu < 1.0 ? 1.0 : per_cu) : (double)c[i].per_cu;
    const double* st = c[i].id == 8 ? sat8 : sat4;
    const int r0 = (int)r;
    const double s_r = st[r0] + (r - r0) * ((r0 + 1 <= c[i].per_cu ? st[r0 + 1] : st[r0]) - st[r0]);
    const double e = c[i].base * useful * quant * s_r / st[c[i].per_cu];
    if (e > best_e) { best_e = e; best = c[i].id; }
  }
  return best;
}

extern "C" int cadre_gemm_pick_tile(const cadre_gemm_t* pp) {
  cadre_gemm_t p = *pp;
  return p.tile ? p.tile : pick_tile(p);
}

extern "C" int cadre_gemm_f32(const cadre_gemm_t* pp, void* stream) {
  cadre_gemm_t p = *pp;
  GEMM_CHECK(p.A && p.B && p.C, "null operand");
  GEMM_CHECK(p.M > 0 && p.N > 0 && p.K > 0, "empty problem");
  GEMM_CHECK(p.a_mode >= 0 && p.a_mode <= 3 && p.b_mode >= 0 && p.b_mode <= 1, "bad operand mode");
  GEMM_CHECK(((uintptr_t)p.A & 15) == 0 && ((uintptr_t)p.B & 15) == 0, "operands must be 16-byte aligned");
  if (p.a_mode == 0) GEMM_CHECK(p.K % 4 == 0 && p.lda % 4 == 0, "a_mode 0 needs K%4==0, lda%4==0");
  if (p.a_mode == 1) GEMM_CHECK(p.M % 4 == 0 && p.lda % 4 == 0, "a_mode 1 needs M%4==0, lda%4==0");
  if (p.b_mode == 0) GEMM_CHECK(p.K % 4 == 0 && p.ldb % 4 == 0, "b_mode 0 needs K%4==0, ldb%4==0");
  if (p.b_mode == 1) GEMM_CHECK(p.N % 4 == 0 && p.ldb % 4 == 0, "b_mode 1 needs N%4==0, ldb%4==0");
  if (p.a_mode == 2) GEMM_CHECK(p.Cin % 32 == 0 && p.K == p.KH * p.KW * p.Cin, "conv needs Cin%32==0, K==KH*KW*Cin");
  if (p.a_mode == 2) GEMM_CHECK(p.KH * p.KW <= 32, "conv window larger than 32 taps");
  {  // every staged byte offset must fit the 2 GiB buffer window of the raw-buffer loads
    const int64_t lim = 1ll << 31;
    int64_t a_bytes, b_bytes;
    if (p.a_mode >= 2) a_bytes = (int64_t)(p.M / (p.Ho * p.Wo)) * p.H * p.W * p.Cin * 4;
    else a_bytes = (int64_t)(p.a_mode == 0 ? p.M : p.K) * p.lda * 4;
    b_bytes = (int64_t)(p.b_mode == 0 ? p.N : p.K) * p.ldb * 4;
    GEMM_CHECK(a_bytes < lim, "A operand spans >= 2 GiB: chunk the batch");
    GEMM_CHECK(b_bytes < lim, "B operand spans >= 2 GiB");
  }
  if (p.a_mode == 3) GEMM_CHECK(p.Cin == 4 && p.KW <= 8 && p.KH <= 32 && p.K == p.KH * 32, "stem conv needs Cin==4, KW<=8, K==KH*32 (B rows [KH][32], zero past KW*4)");
  if (p.a_mode >= 2) GEMM_CHECK(p.M % (p.Ho * p.Wo) == 0 && p.stride > 0, "conv M must be Nimg*Ho*Wo");
  if (p.batch < 1) p.batch = 1;
  if (p.split_k < 1) p.split_k = 1;
  if (p.a_div < 1) p.a_div = 1;
  if (p.b_div < 1) p.b_div = 1;
  if (p.c_div < 1) p.c_div = 1;
  if (p.s_div < 1) p.s_div = 1;
  if (p.r_div < 1) p.r_div = 1;
  if (p.a_mod < 1) p.a_mod = 1 << 30;
  if (p.b_mod < 1) p.b_mod = 1 << 30;
  if (p.c_mod < 1) p.c_mod = 1 << 30;
  if (p.s_mod < 1) p.s_mod = 1 << 30;
  if (p.r_mod < 1) p.r_mod = 1 << 30;
  GEMM_CHECK(p.split_k == 1 || p.batch == 1 || p.c_str >= (int64_t)p.M * p.ldc, "batched split_k needs c_str >= M*ldc");
  if (p.seg_mode) {
    GEMM_CHECK(p.row_seg && p.seg_period > 0 && p.seg_div > 0 && p.seg_mode >= 1 && p.seg_mode <= 3, "bad row segment fields");
    if (p.seg_mode == 3)
      GEMM_CHECK(p.a_mode == 0 && p.b_mode == 0 && p.split_k == 1 && !p.resid && !(p.flags & 2) && p.M % p.seg_period == 0 &&
                     ((p.N | p.ldc) & 3) == 0 && ((uintptr_t)p.C & 15) == 0 && (int64_t)p.M * p.ldc * 4 < (1ll << 31),
                 "seg_mode 3 (compact rows) needs an NT product, no split / residual, M%seg_period==0, N%4==0, ldc%4==0, C below 2 GiB");
    if (p.seg_mode == 1) GEMM_CHECK(p.seg_period % 32 == 0 && p.M % p.seg_period == 0, "seg_mode 1 needs seg_period%32==0, M%seg_period==0");
    if (p.seg_mode == 2) GEMM_CHECK(p.seg_period % 32 == 0 && p.K % p.seg_period == 0 && p.split_k == 1 && p.a_mode == 1 && p.b_mode == 1,
                                    "seg_mode 2 needs k-major operands, seg_period%32==0, K%seg_period==0");
  }
  if (p.flags & 2)
    GEMM_CHECK(p.batch == 1 && p.split_k == 1 && ((p.N | p.ldc) & 3) == 0 && !p.resid, "bf16 output needs the vector epilogue, no batch/split/resid");
  int tile = p.tile ? p.tile : pick_tile(p);
#ifdef CADRE_AB_KERNELS
  if (tile == 13) return cadre_gemm_stream_f32_launch(p, stream);      // short-K dense NT product, several M-tiles per workgroup
  if (tile == 12) return cadre_conv_stream_f32_launch(p, stream);      // 64x64 conv, several M-tiles per workgroup
  if (tile == 11) {                                                    // skinny products of the PPO update
    if (!cadre_gemm_f32_skinny_ok(p)) return cadre_fail("cadre_gemm_f32: tile 11 (skinny) does not take this descriptor");
    return cadre_gemm_f32_skinny_launch(p, stream);
  }
#else
  if (tile >= 11 && tile <= 13) return cadre_fail("cadre_gemm_f32: tiles 11 / 12 / 13 exist only in the A/B build (CADRE_BUILD_AB=1)");
#endif
  hipStream_t st = (hipStream_t)stream;
  if (tile < 1 || tile > 10 || tile == 7) return cadre_fail("cadre_gemm_f32: bad tile");
  // 9: 32x128 on 4 waves (1x4) for row-sorted skinny GEMMs; 10: 128x64 on 8 waves (4x2) for N <= 64 convs
  static const int BMS[11] = {0, 128, 128, 64, 256, 128, 256, 0, 128, 32, 128}, BNS[11] = {0, 128, 64, 64, 128, 256, 64, 0, 128, 128, 64};
  if (p.seg_mode == 1 && p.seg_period % BMS[tile] != 0) tile = p.seg_period % 64 == 0 ? 3 : 9;     // the M tile must divide the period
  const int bm = BMS[tile], bn = BNS[tile];
  dim3 block(tile == 8 || tile == 10 ? 512 : 256);
  if (p.a_mode >= 2 && p.b_mode != 0) return cadre_fail("cadre_gemm_f32: conv needs b_mode 0");
  dim3 grid(((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn), p.split_k, p.batch);
#define LAUNCH(WM_, WN_, AM_, BM_) hipLaunchKernelGGL((gemm_f32_kernel<WM_, WN_, AM_, BM_>), grid, block, 0, st, p)
#define LAUNCH_TILE(AM_, BM_)                 \
  do {                                        \
    if (tile == 1) LAUNCH(2, 2, AM_, BM_);    \
    else if (tile == 2) LAUNCH(2, 1, AM_, BM_); \
    else if (tile == 3) LAUNCH(1, 1, AM_, BM_); \
    else if (tile == 4) LAUNCH(4, 2, AM_, BM_); \
    else if (tile == 5) LAUNCH(2, 4, AM_, BM_); \
    else if (tile == 6) LAUNCH(4, 1, AM_, BM_); \
    else if (tile == 9) hipLaunchKernelGGL((gemm_f32_kernel<1, 1, AM_, BM_, 4, GEMM_NSETS, 1>), grid, block, 0, st, p); \
    else if (tile == 10) hipLaunchKernelGGL((gemm_f32_kernel<1, 1, AM_, BM_, 2, GEMM_NSETS, 4>), grid, block, 0, st, p); \
    else hipLaunchKernelGGL((gemm_f32_kernel<2, 1, AM_, BM_, 4>), grid, block, 0, st, p); \
  } while (0)
  switch (p.a_mode * 2 + p.b_mode) {
    case 0: LAUNCH_TILE(0, 0); break;
    case 1: LAUNCH_TILE(0, 1); break;
    case 2: LAUNCH_TILE(1, 0); break;
    case 3: LAUNCH_TILE(1, 1); break;
    case 4: LAUNCH_TILE(2, 0); break;
    case 6: LAUNCH_TILE(3, 0); break;
    default: return cadre_fail("cadre_gemm_f32: bad operand mode");
  }
#undef LAUNCH_TILE
#undef LAUNCH
  return (int)hipGetLastError();
}

extern "C" int cadre_splitk_reduce(const float* slabs, int32_t split_k, int64_t slab_stride, int64_t lds_,
                                   float* C, int64_t ldc, int32_t M, int32_t N, const float* scale,
                                   const float* shift, int32_t act, float slope, const int32_t* row_seg, int32_t period,
                                   void* stream) {
  if (!slabs || !C || split_k < 1 || M < 1 || N < 1) return cadre_fail("cadre_splitk_reduce: bad argument");
  if (row_seg && (period < 32 || period % 32 != 0 || M % period != 0)) return cadre_fail("cadre_splitk_reduce: row segments need period % 32 == 0, M % period == 0");
  const int64_t total = (int64_t)M * N;
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, slabs, split_k,
                     slab_stride, lds_, C, ldc, M, N, scale, shift, act, slope, row_seg, period);
  return (int)hipGetLastError();
}
