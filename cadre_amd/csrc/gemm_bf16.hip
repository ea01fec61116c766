// gemm_bf16.hip — bf16-input / fp32-accumulate GEMM and implicit-GEMM convolution for gfx950.
//
// Same tile/loader/epilogue architecture as gemm_f32.hip (raw buffer loads with hardware
// zero-fill -> double-buffered LDS, XCD-contiguous tile order, LDS-staged vector epilogue) with
// 2-byte operands: a 128-B LDS row holds BK = 64 bf16, and the matrix core is
// v_mfma_f32_32x32x16_bf16 (lane half h owns k = 16*s + 8*h .. +7 of k-step s, i.e. the SAME
// 16-B chunk (2s+h) the fp32 kernel reads — the staging code is byte-for-byte the same geometry).
// Used for BASELINE config C3 ("bf16 encoder / fp32 losses"): DANet conv stack
// (carla_perception/Networks/danet_blocks/resnet.py:26-55,152-166, danet.py:21-41,96,108) and the
// inter-task first-layer GEMM (intertask_att.py:39-80).  At 16x the fp32 MFMA rate these layers are
// bound by operand staging (L2 -> LDS), so the large tiles are preferred.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "../../include/cadre_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define BKE 64          // k elements per tile (128 bytes)
#ifndef BF16_NS_SMALL
#define BF16_NS_SMALL 2  // register sets of the 64-wide tiles
#endif
#define PITCH_F 36      // LDS row pitch in 4-byte words (128 B data + 16 B pad)

int cadre_fail(const char* msg);

// flags of cadre_gemm_t used here: bit1 = C is bf16 (else f32), bit2 = resid is bf16 (else f32)
// WVM x WVN = waves along M x N (64 threads each).
// NS = register sets of staged tiles (see gemm_f32.hip: the k-loop stages the next tile behind the first
// fragment reads of the current one); 1 for the 256-wide tiles, whose accumulators leave no room for two.
template <int WM, int WN, int AMODE, int WVN, int NS, int WVM = 2>
__global__ __launch_bounds__(64 * WVM * WVN, ((WVM * WM + WVN * WN) * 2 * 32 * 36 * 4 > 80 * 1024 ? 1 : 2)) void gemm_bf16_kernel(cadre_gemm_t p) {
  constexpr int NT = 64 * WVM * WVN;   // threads
  constexpr int RP = NT / 8;      // rows staged per pass (8 x 16-B chunks per 128-B row)
  constexpr int BM = WVM * WM * 32;
  constexpr int BN = WVN * WN * 32;
  constexpr int RA = BM / RP;
  constexpr int RB = BN / RP;
  __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * PITCH_F];
  float* As = lds;
  float* Bs = lds + 2 * BM * PITCH_F;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WVN, wn = wave % WVN;
  const int l31 = lane & 31, lh = lane >> 5;

  const int tilesN = (p.N + BN - 1) / BN;
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tile_m = bid / tilesN, tile_n = bid % tilesN;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int z = blockIdx.z;
  const char* A = reinterpret_cast<const char*>(p.A);
  const char* B = reinterpret_cast<const char*>(p.B);
  const bool c_bf16 = (p.flags & 2) != 0, r_bf16 = (p.flags & 4) != 0;
  char* C = reinterpret_cast<char*>(p.C);
  if (p.batch > 1) {
    A += (int64_t)((z / p.a_div) % p.a_mod) * p.a_str * 2;
    B += (int64_t)((z / p.b_div) % p.b_mod) * p.b_str * 2;
    C += (int64_t)((z / p.c_div) % p.c_mod) * p.c_str * (c_bf16 ? 2 : 4);
  }

  const int nk_total = (p.K + BKE - 1) / BKE;
  int kt_begin = 0, kt_end = nk_total;
  if (p.split_k > 1) {      // raw f32 slabs [split][M][ldc]
    const int per = (nk_total + p.split_k - 1) / p.split_k;
    kt_begin = blockIdx.y * per;
    kt_end = min(nk_total, kt_begin + per);
    C += (int64_t)blockIdx.y * p.M * p.ldc * 4;
  }

  constexpr unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)OOB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (int)OOB, 0x00020000);
  const int cc = tid & 7, rr = tid >> 3;      // 16-B chunk column (8 bf16), staged row
  unsigned aoff[RA], amask[RA];
  if constexpr (AMODE == 0) {
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const int m = m0 + rr + RP * i;
      aoff[i] = m < p.M ? (unsigned)((int64_t)m * p.lda * 2 + cc * 16) : OOB;
      amask[i] = 0;
    }
  } else {
    // conv rows: scalar decode of the tile's first row + float-reciprocal carries (gemm_f32.hip)
    const int hw = p.Ho * p.Wo;
    const int img0 = m0 / hw, rem0 = m0 % hw;
    const int ho0 = rem0 / p.Wo, wo0 = rem0 % p.Wo;
    const float inv_wo = 1.0f / (float)p.Wo, inv_ho = 1.0f / (float)p.Ho;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const int r = rr + RP * i;
      const int m = m0 + r;
      const int x = wo0 + r;
      const int q1 = (int)(((float)x + 0.5f) * inv_wo);
      const int wo = x - __mul24(q1, p.Wo);
      const int y = ho0 + q1;
      const int q2 = (int)(((float)y + 0.5f) * inv_ho);
      const int ho = y - __mul24(q2, p.Ho);
      const int img = img0 + q2;
      if constexpr (AMODE == 4) {
        // Cin == 4 stem on a ZERO-PADDED bf16 NHWC4 image [Nimg][H][W][4] (H, W = padded sizes, the halo is
        // real zeros in memory, so no tap masks): k-tile kt holds kernel rows 2kt and 2kt+1; chunk cc is
        // the pixel pair 2(cc&3), 2(cc&3)+1 of row 2kt + (cc>>2), counted from padded pixel (stride*ho,
        // stride*wo).  B is [N][KH/2 rounded up][64] with zeros where kh >= KH or kw >= KW.
        aoff[i] = m < p.M ? (unsigned)((((img * p.H + ho * p.stride + (cc >> 2)) * p.W + wo * p.stride + 2 * (cc & 3)) * 4) * 2) : OOB;
        amask[i] = 0;
      } else {
        const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
        aoff[i] = (unsigned)((((img * p.H + hi0) * p.W + wi0) * p.Cin) * 2 + cc * 16);
        unsigned mask = 0;
        if (m < p.M) {     // separable: (rows inside) x (columns inside)
          unsigned colm = 0;
          for (int kw = 0; kw < p.KW; ++kw)
            if ((unsigned)(wi0 + kw) < (unsigned)p.W) colm |= 1u << kw;
          for (int kh = 0; kh < p.KH; ++kh)
            if ((unsigned)(hi0 + kh) < (unsigned)p.H) mask |= colm << (kh * p.KW);
        }
        amask[i] = mask;
      }
    }
  }
  unsigned boff[RB];
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    const int n = n0 + rr + RP * i;
    boff[i] = n < p.N ? (unsigned)((int64_t)n * p.ldb * 2 + cc * 16) : OOB;
  }

  f32x4 areg[NS][RA], breg[NS][RB];
  auto ldg = [](const __amdgpu_buffer_rsrc_t& rs, unsigned off) -> f32x4 {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
  };
  // request k-tile kt into register set rs; tiles past the end resolve to OOB offsets (zero fill)
  auto load_tiles = [&](int kt, int rs) {
    const int k0 = kt * BKE;
    const unsigned kb_ = (k0 + cc * 8 < p.K) ? (unsigned)k0 * 2u : OOB;     // K % 8 == 0
    if constexpr (AMODE == 0) {
#pragma unroll
      for (int i = 0; i < RA; ++i) areg[rs][i] = ldg(rsA, aoff[i] + kb_);
    } else if constexpr (AMODE == 4) {
      const unsigned delta = (unsigned)(2 * kt * p.W * 8);                    // two padded rows per k-tile
      const bool in = kt < nk_total;                                          // (past-the-end tiles must not touch memory)
#pragma unroll
      for (int i = 0; i < RA; ++i) areg[rs][i] = ldg(rsA, in ? aoff[i] + delta : OOB);   // OOB rows stay >= 2 GiB
    } else {
      const int pos = k0 / p.Cin, ci = k0 % p.Cin;                           // uniform (Cin % 64 == 0)
      const unsigned delta = (unsigned)((((pos / p.KW) * p.W + (pos % p.KW)) * p.Cin + ci) * 2);
      const unsigned bit = pos < 32 ? 1u << pos : 0u;
#pragma unroll
      for (int i = 0; i < RA; ++i) areg[rs][i] = ldg(rsA, (amask[i] & bit) ? aoff[i] + delta : OOB);
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) breg[rs][i] = ldg(rsB, boff[i] + kb_);
  };
  auto store_tiles = [&](int buf, int rs) {
    float* as = As + buf * BM * PITCH_F;
    float* bs = Bs + buf * BN * PITCH_F;
#pragma unroll
    for (int i = 0; i < RA; ++i) *reinterpret_cast<f32x4*>(as + (rr + RP * i) * PITCH_F + cc * 4) = areg[rs][i];
#pragma unroll
    for (int i = 0; i < RB; ++i) *reinterpret_cast<f32x4*>(bs + (rr + RP * i) * PITCH_F + cc * 4) = breg[rs][i];
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // one k-tile of MFMAs from LDS buffer `buf`; `staging()` runs behind the first fragment reads
  auto compute = [&](int buf, auto&& staging) {
    const float* as = As + buf * BM * PITCH_F;
    const float* bs = Bs + buf * BN * PITCH_F;
#pragma unroll
    for (int s = 0; s < 4; ++s) {             // k-step s: 16 bf16; this lane half reads chunk 2s+lh
      bf16x8 af[WM], bf[WN];
#pragma unroll
      for (int i = 0; i < WM; ++i)
        af[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(as + ((wm * WM + i) * 32 + l31) * PITCH_F + (2 * s + lh) * 4));
#pragma unroll
      for (int j = 0; j < WN; ++j)
        bf[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(bs + ((wn * WN + j) * 32 + l31) * PITCH_F + (2 * s + lh) * 4));
      if (s == 0) staging();
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  };
  // k-loop (same order as gemm_f32.hip): barrier; first fragment reads of tile t; write tile t+1 (register
  // set (t+1)%NS) into the buffer compute(t-1) released and re-request the set for tile t+1+NS; MFMAs.
  constexpr int U = (NS % 2 == 0) ? NS : 2 * NS;
  auto step = [&](auto uc, int kt) {
    constexpr int u = decltype(uc)::value;
    __syncthreads();
    compute(u & 1, [&] {
      store_tiles((u + 1) & 1, (u + 1) % NS);
      load_tiles(kt + 1 + NS, (u + 1) % NS);
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
  }
  if (kt < kt_end) step(std::integral_constant<int, 0>{}, kt);
  if (kt + 1 < kt_end) step(std::integral_constant<int, 1>{}, kt + 1);
  if constexpr (U > 2) {
    if (kt + 2 < kt_end) step(std::integral_constant<int, 2>{}, kt + 2);
  }
  __syncthreads();      // every wave is done reading: the epilogue re-uses the staging buffers

  // ---------------------------------------------------------------- epilogue (fp32 math)
  // Straight-line per (activation, residual kind) with buffer loads/stores rebased at the tile's first
  // row (see gemm_f32.hip): 32-bit offsets, rows >= M dropped by the hardware bounds check.
  const bool raw = p.split_k > 1;
  const int actk = p.act & 15;
  const bool post = (p.act & 16) != 0;
  const float* scale = raw ? nullptr : p.scale;
  const float* shift = raw ? nullptr : p.shift;
  const char* resid = raw ? nullptr : reinterpret_cast<const char*>(p.resid);
  if (p.batch > 1) {
    const int64_t so = (int64_t)((z / p.s_div) % p.s_mod) * p.s_str;
    if (scale) scale += so;
    if (shift) shift += so;
    if (resid) resid += (int64_t)((z / p.r_div) % p.r_mod) * p.r_str * (r_bf16 ? 2 : 4);
  }
  constexpr int CW = WN * 32, P = CW + 4;
  constexpr int LPR = CW / 4, RPI = 64 / LPR, NIT = 32 / RPI;
  float* cs = lds + wave * (32 * P);
  const int c4 = (lane % LPR) * 4;
  const int col = n0 + wn * CW + c4;
  const bool cvalid = col < p.N;              // N % 4 == 0
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
  if (cvalid && scale) sc = *reinterpret_cast<const f32x4*>(scale + col);
  if (cvalid && shift) sh = *reinterpret_cast<const f32x4*>(shift + col);
  const bool out_bf16 = c_bf16 && !raw;
  const int esz = out_bf16 ? 2 : 4, rsz = r_bf16 ? 2 : 4;
  const int64_t rows_left = (int64_t)p.M - m0;
  auto window = [](int64_t bytes) { return (int)(bytes < 0x7fffffff ? bytes : 0x7fffffff); };
  const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc((void*)(C + (int64_t)m0 * p.ldc * esz), 0,
                                                                       window(rows_left * p.ldc * esz), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(resid ? resid + (int64_t)m0 * p.ldr * rsz : (const char*)p.C), 0, resid ? window(rows_left * p.ldr * rsz) : 0, 0x00020000);
  const int lrow = lane / LPR;
  const unsigned coff = cvalid ? (unsigned)((lrow * p.ldc + col) * esz) : OOB;
  const unsigned roff = cvalid ? (unsigned)((lrow * p.ldr + col) * rsz) : OOB;
  const float slope = p.slope;
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

  auto body = [&](auto actc, auto resc) {
    constexpr int ACT = decltype(actc)::value;      // -1 raw split-K slab, 0 none, 1 ReLU, 2 LeakyReLU
    constexpr int RES = decltype(resc)::value;      // 0 none, 1 f32 residual, 2 bf16 residual
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      const int r0 = (wm * WM + i) * 32;
      f32x4 rv[NIT];
      if constexpr (RES != 0) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const unsigned off = roff + (unsigned)((r0 + it * RPI) * p.ldr * rsz);
          if constexpr (RES == 2) {
            const bf16x4 t = __builtin_bit_cast(bf16x4, __builtin_amdgcn_raw_buffer_load_b64(rsR, (int)off, 0, 0));
            rv[it] = f32x4{(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
          } else {
            rv[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsR, (int)off, 0, 0));
          }
        }
      }
#pragma unroll
      for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) cs[((r & 3) + 8 * (r >> 2) + 4 * lh) * P + j * 32 + l31] = acc[i][j][r];
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        f32x4 v = *reinterpret_cast<const f32x4*>(cs + (it * RPI + lrow) * P + c4);
        if constexpr (ACT >= 0) {
          v = v * sc + sh;
          if constexpr (RES != 0) { if (!post) v += rv[it]; }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if constexpr (ACT == 1) v[e] = fmaxf(v[e], 0.f);
            if constexpr (ACT == 2) v[e] = v[e] > 0.f ? v[e] : v[e] * slope;
          }
          if constexpr (RES != 0) { if (post) v += rv[it]; }
        }
        const unsigned off = coff + (unsigned)((r0 + it * RPI) * p.ldc * esz);
        if (out_bf16) {
          bf16x4 o;
          o[0] = (__bf16)v[0]; o[1] = (__bf16)v[1]; o[2] = (__bf16)v[2]; o[3] = (__bf16)v[3];
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), rsC, (int)off, 0, 0);
        } else {
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsC, (int)off, 0, 0);
        }
      }
    }
  };
  using std::integral_constant;
#define BODY_ACT(RES_)                                                                   \
  do {                                                                                   \
    if (actk == 1) body(integral_constant<int, 1>{}, integral_constant<int, RES_>{});    \
    else if (actk == 2) body(integral_constant<int, 2>{}, integral_constant<int, RES_>{}); \
    else body(integral_constant<int, 0>{}, integral_constant<int, RES_>{});              \
  } while (0)
  if (raw) body(integral_constant<int, -1>{}, integral_constant<int, 0>{});
  else if (!resid) BODY_ACT(0);
  else if (r_bf16) BODY_ACT(2);
  else BODY_ACT(1);
#undef BODY_ACT
}

#ifdef CADRE_AB_KERNELS      // A/B build only (csrc/ab/, include/cadre_hip_ab.h)
int cadre_conv_stream_bf16_launch(const cadre_gemm_t& p, void* stream);     // ab/conv_stream_bf16.hip (tile 12)
#endif

#define BCHECK(cond, msg) \
  if (!(cond)) return cadre_fail("cadre_gemm_bf16: " msg)

// Tile choice (host logic).  12 = ab/conv_stream_bf16.hip (64x64, several M-tiles per workgroup; A/B build only).
static int pick_tile_bf16(const cadre_gemm_t& p) {
  int tile = p.tile;
  const int batch = p.batch < 1 ? 1 : p.batch, sk = p.split_k < 1 ? 1 : p.split_k;
#ifdef CADRE_AB_KERNELS
  // N <= 64 convs (stage-1 convs, padded stem): several M-tiles per workgroup, ab/conv_stream_bf16.hip (444 vs 431);
  // the product runs these layers on the window kernel (conv3x3_ring.hip) and the fused front (stem_pool.hip)
  if (tile == 0 && p.N <= 64 && p.a_mode >= 2 && batch == 1 && sk == 1 && p.K >= 128 && p.M >= 64 * 2048) tile = 12;
#endif
  if (tile == 0) {
    // staging-bound regime: shape factor (dense 8192^3: 256x256 on 8 waves 1054, 128x128 826 TFLOP/s;
    // tools/gemm_bf16_bench.py) x wave quantisation over 256 CUs x resident workgroups per CU
    struct Cand { int id, bm, bn, per_cu; double base; };
    static const Cand wide[3] = {{7, 256, 256, 1, 1.00}, {1, 128, 128, 2, 0.80}, {3, 64, 64, 4, 0.45}};
    static const Cand narrow[2] = {{10, 128, 64, 2, 1.05}, {3, 64, 64, 4, 1.00}};     // layer 1 @72x72: 437 vs 416 TFLOP/s
    const Cand* c = p.N <= 64 ? narrow : wide;
    const int nc = p.N <= 64 ? 2 : 3;
    double best_e = -1.0;
    for (int i = 0; i < nc; ++i) {
      if (c[i].id == 7 && p.N < 256) continue;
      const double tiles = (double)((p.M + c[i].bm - 1) / c[i].bm) * ((p.N + c[i].bn - 1) / c[i].bn) * batch * sk;
      const double slots = 256.0 * c[i].per_cu;
      const double waves = tiles <= slots ? 1.0 : (double)(int64_t)((tiles + slots - 1) / slots);
      const double useful = ((double)p.M * p.N * batch * sk) / (tiles * c[i].bm * c[i].bn);
      const double e = c[i].base * useful * (tiles / (waves * slots));
      if (e > best_e) { best_e = e; tile = c[i].id; }
    }
  }
  return tile;
}

extern "C" int cadre_gemm_bf16_pick_tile(const cadre_gemm_t* pp) { return pick_tile_bf16(*pp); }

extern "C" int cadre_gemm_bf16(const cadre_gemm_t* pp, void* stream) {
  cadre_gemm_t p = *pp;
  BCHECK(p.A && p.B && p.C, "null operand");
  BCHECK(p.M > 0 && p.N > 0 && p.K > 0, "empty problem");
  BCHECK((p.a_mode == 0 || p.a_mode == 2 || p.a_mode == 4) && p.b_mode == 0, "a_mode must be 0, 2 or 4, b_mode 0");
  if (p.a_mode == 4)
    BCHECK(p.Cin == 4 && p.KW <= 8 && p.K == ((p.KH + 1) / 2) * 64 && p.M % (p.Ho * p.Wo) == 0 &&
               (p.Ho - 1) * p.stride + 2 * ((p.KH + 1) / 2) <= p.H && (p.Wo - 1) * p.stride + 8 <= p.W && (p.W % 2) == 0 &&
               (p.stride % 2) == 0,
           "padded stem: Cin==4, KW<=8, K==ceil(KH/2)*64, even stride/W, padded image must cover every tap row/pixel");
  BCHECK(((uintptr_t)p.A & 15) == 0 && ((uintptr_t)p.B & 15) == 0 && ((uintptr_t)p.C & 15) == 0, "operands must be 16-byte aligned");
  BCHECK(p.K % 8 == 0 && p.ldb % 8 == 0 && p.N % 4 == 0 && p.ldc % 4 == 0, "needs K%8==0, ldb%8==0, N%4==0, ldc%4==0");
  if (p.a_mode == 0) BCHECK(p.lda % 8 == 0, "lda%8==0");
  if (p.a_mode == 2) BCHECK(p.Cin % 64 == 0 && p.K == p.KH * p.KW * p.Cin && p.KH * p.KW <= 32 && p.M % (p.Ho * p.Wo) == 0, "conv needs Cin%64==0");
  if (p.resid) BCHECK(p.ldr % 4 == 0 && ((uintptr_t)p.resid & 15) == 0, "resid alignment");
  if (p.batch < 1) p.batch = 1;
  if (p.split_k < 1) p.split_k = 1;
  BCHECK(p.split_k == 1 || p.batch == 1, "split_k with batch unsupported");
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
  {
    const int64_t lim = 1ll << 31;
    const int64_t a_bytes = p.a_mode >= 2 ? (int64_t)(p.M / (p.Ho * p.Wo)) * p.H * p.W * p.Cin * 2 : (int64_t)p.M * p.lda * 2;
    BCHECK(a_bytes < lim && (int64_t)p.N * p.ldb * 2 < lim, "operand spans >= 2 GiB: chunk the batch");
  }
  const int tile = pick_tile_bf16(p);
  // 10: 128x64 on 8 waves (4x2), 11: 256x64 on 8 waves (4x2, each wave 64x32) — the N <= 64 layers are bound by
  // L2 -> LDS staging bytes per FLOP, which only a taller tile lowers
#ifdef CADRE_AB_KERNELS
  if (tile == 12) return cadre_conv_stream_bf16_launch(p, stream);       // 64x64 conv, several M-tiles per workgroup
#endif
  // 13: 256x128 on 8 waves (4x2, each wave 64x64, two register sets) — N = 128 layers: a quarter less operand traffic per FLOP than 128x128
  static const int BMS[14] = {0, 128, 128, 64, 256, 128, 256, 256, 0, 0, 128, 256, 0, 256}, BNS[14] = {0, 128, 64, 64, 128, 256, 64, 256, 0, 0, 64, 64, 0, 128};
  BCHECK(tile == 1 || tile == 2 || tile == 3 || tile == 4 || tile == 7 || tile == 10 || tile == 11 || tile == 13, "bad tile");
  const int bm = BMS[tile], bn = BNS[tile];
  dim3 grid(((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn), p.split_k, p.batch), block(tile == 7 || tile >= 10 ? 512 : 256);
  hipStream_t st = (hipStream_t)stream;
#define LB(WM_, WN_, WV_, NS_, WVM_)                                                                  \
  do {                                                                                                \
    if (p.a_mode == 0) hipLaunchKernelGGL((gemm_bf16_kernel<WM_, WN_, 0, WV_, NS_, WVM_>), grid, block, 0, st, p); \
    else if (p.a_mode == 2) hipLaunchKernelGGL((gemm_bf16_kernel<WM_, WN_, 2, WV_, NS_, WVM_>), grid, block, 0, st, p); \
    else hipLaunchKernelGGL((gemm_bf16_kernel<WM_, WN_, 4, WV_, NS_, WVM_>), grid, block, 0, st, p);  \
  } while (0)
  if (tile == 1) LB(2, 2, 2, 2, 2);
  else if (tile == 2) LB(2, 1, 2, 2, 2);
  else if (tile == 3) LB(1, 1, 2, BF16_NS_SMALL, 2);
  else if (tile == 4) LB(4, 2, 2, 1, 2);
  else if (tile == 10) LB(1, 1, 2, BF16_NS_SMALL, 4);
  else if (tile == 11) LB(2, 1, 2, BF16_NS_SMALL, 4);
  else if (tile == 13) LB(2, 2, 2, 2, 4);
  else LB(4, 2, 4, 1, 2);          // 256 x 256, 8 waves (2 x 4), each wave 128 x 64
  return (int)hipGetLastError();
}
