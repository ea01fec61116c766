// gemm_f32_skinny.hip — fp32 GEMM for the PPO update's skinny products on gfx950 (tile id 11 of cadre_gemm_f32).
//
// The LSTM steps of CadreAgent.update_policy (models.py:139-152 through agent.py:166-237) are chains of small
// products: per step and net a [B, 544] x [2120, 544]^T forward and a [B, 2120] x [2120, 544] backward with B = 64..256
// rows of which a net owns a quarter (rows sorted by command).  On the general tile kernel (gemm_f32.hip, 32 x 128 on
// four waves) such a launch puts about one workgroup on a CU — ONE wave per SIMD walking 17..66 k-tiles through a
// load -> LDS -> barrier -> MFMA turn, each turn exposed: 25-58 us per launch for 0.3-2.4 GFLOP, and the backward needs
// a split-K pass with its own reduction kernel to find parallelism.  Here:
//   * operand fragments go from global memory STRAIGHT to the MFMA operand registers: the k order inside a group of
//     32 is permuted (lane half h owns k = 32*g + 16*h + 0..15, on A and B alike), so a lane's share of a group is
//     64 contiguous bytes of its row — four 16-B loads per operand and group, no LDS staging, no barrier in the
//     k-loop, two groups of loads in flight per wave;
//   * a workgroup is 8 waves on a 32 x 64 tile: 2 column blocks x 4 K-SLICES; the slices meet in LDS once, are summed
//     in a fixed order (bit-reproducible) and leave through the usual epilogue (scale/shift, residual, activation) —
//     no split-K slabs in HBM, no reduction launch; 2..4 waves per SIMD hide each other's load latency.
// The k order and the K slices change the summation tree: results agree with the tile kernel to fp32 rounding
// (tests/test_kernels_gpu.py).
// MEASURED (tools/skinny_gemm_bench.py, MI355X, per launch inside a back-to-back run): forward B = 64 22.7 us (tile
// kernel 22.0), forward B = 256 39.5 (23.8), backward B = 64 31.1 (50.4, or 29 + a 9.6 us reduction with split-K 4),
// backward B = 256 54.0 (50.8).  Without LDS staging every wave fetches its own copy of the shared operand through
// L1/L2 — about twice the cache traffic per flop of the 32 x 128 tile — which costs more than the barriers it saves.
// The kernel stays available as tile 11 (CADRE_SKINNY_GEMM=1 makes the row-sorted update launches use it).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../../include/cadre_hip_ab.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

int cadre_fail(const char* msg);

#define SK_BM 32
#define SK_BN 64
#define SK_KS 4                    // K slices (waves along K)
#define SK_D 2                     // groups of 32 k in flight per wave
#define SK_PITCH 36

// BMODE 0: B[n][k] (k contiguous, nn.Linear weight);  1: B[k][n]
template <int BMODE>
__global__ __launch_bounds__(512) void gemm_f32_skinny_kernel(cadre_gemm_t p) {
  __shared__ __attribute__((aligned(16))) float red[SK_KS * 2 * 32 * SK_PITCH];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int wn = wave & 1, ks = wave >> 1;
  const int tilesN = (p.N + SK_BN - 1) / SK_BN;
  const int tile_m = blockIdx.x / tilesN, tile_n = blockIdx.x % tilesN;
  const int m0 = tile_m * SK_BM, n0 = tile_n * SK_BN;
  const int z = blockIdx.z;
  const float* A = p.A;
  const float* B = p.B;
  float* C = p.C;
  auto slot = [](int zz, int dv, int md) {
    const int q = dv == 1 ? zz : zz / dv;
    return q < md ? q : q % md;
  };
  if (p.batch > 1) {
    A += (int64_t)slot(z, p.a_div, p.a_mod) * p.a_str;
    B += (int64_t)slot(z, p.b_div, p.b_mod) * p.b_str;
    C += (int64_t)slot(z, p.c_div, p.c_mod) * p.c_str;
  }
  if (p.seg_mode == 1) {           // rows sorted by command: tiles outside this batch entry's run are skipped
    const int32_t* sg = p.row_seg + 2 * (p.seg_div == 1 ? z : z / p.seg_div);
    const int seg_beg = sg[0], seg_cnt = sg[1];
    const int b_lo = m0 % p.seg_period;
    if (seg_cnt <= 0 || b_lo + SK_BM <= seg_beg || b_lo >= seg_beg + seg_cnt) return;
  }
  constexpr unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)OOB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (int)OOB, 0x00020000);
  // this wave's K slice, in groups of 32 k.  Any permutation of k is a valid product order as long as A and B share
  // it: lane half h takes k = 32*g + 16*h .. +15 of a group — 64 CONTIGUOUS bytes of its row, so that a 128-byte line
  // is consumed by one wave within one group (four 16-byte loads per operand and group; with 8 k per step a line
  // was touched by four steps of eight waves and did not survive in the vector L1, and every load instruction
  // addressed 64 lines instead of 32).
  const int ng = (p.K + 31) >> 5;
  const int per = (ng + SK_KS - 1) / SK_KS;
  const int g0 = ks * per, g1 = min(ng, g0 + per);
  const int am = m0 + l31, bn = n0 + wn * 32 + l31;
  const bool a_ok = am < p.M, b_ok = bn < p.N;
  const unsigned a_base = (unsigned)(((int64_t)am * p.lda + 16 * lh) * 4);
  const unsigned b_base = BMODE == 0 ? (unsigned)(((int64_t)bn * p.ldb + 16 * lh) * 4) : (unsigned)(((int64_t)(16 * lh) * p.ldb + bn) * 4);
  const unsigned b_krow = (unsigned)(p.ldb * 4);             // BMODE 1: bytes between consecutive k
  struct frag { f32x4 q[4]; };
  auto load_a = [&](int g) -> frag {
    frag f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool ok = a_ok && g < g1 && 32 * g + 16 * lh + 4 * q < p.K;      // (K % 4 == 0: a chunk is all in or all out)
      f.q[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, ok ? (int)(a_base + (unsigned)g * 128u + 16u * q) : (int)OOB, 0, 0));
    }
    return f;
  };
  auto load_b = [&](int g) -> frag {
    frag f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool ok = b_ok && g < g1 && 32 * g + 16 * lh + 4 * q < p.K;
      if constexpr (BMODE == 0) {
        f.q[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, ok ? (int)(b_base + (unsigned)g * 128u + 16u * q) : (int)OOB, 0, 0));
      } else {
        const unsigned o0 = b_base + (unsigned)(32 * g + 4 * q) * b_krow;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          f.q[q][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsB, ok ? (int)(o0 + (unsigned)i * b_krow) : (int)OOB, 0, 0));
      }
    }
    return f;
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  frag ar[SK_D], br[SK_D];
#pragma unroll
  for (int d = 0; d < SK_D; ++d) { ar[d] = load_a(g0 + d); br[d] = load_b(g0 + d); }
  for (int g = g0; g < g1; g += SK_D) {
#pragma unroll
    for (int d = 0; d < SK_D; ++d) {
      if (g + d < g1) {                                      // (wave-uniform)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[d].q[q][i], br[d].q[q][i], acc, 0, 0, 0);
      }
      ar[d] = load_a(g + d + SK_D);
      br[d] = load_b(g + d + SK_D);
    }
  }
  // ---- the four K slices meet in LDS: slab [ks][wn] of 32 rows x 32 columns (pitch 36)
  {
    float* slab = red + (ks * 2 + wn) * (32 * SK_PITCH);
#pragma unroll
    for (int r = 0; r < 16; ++r) slab[((r & 3) + 8 * (r >> 2) + 4 * lh) * SK_PITCH + l31] = acc[r];
  }
  __syncthreads();
  // wave (wn, ks) finishes rows 8*ks .. 8*ks+7 of column block wn: lane -> (row 8*ks + lane/8, columns 4*(lane%8)..+3)
  const int row = 8 * ks + (lane >> 3), c4 = (lane & 7) * 4;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < SK_KS; ++s) v += *reinterpret_cast<const f32x4*>(red + (s * 2 + wn) * (32 * SK_PITCH) + row * SK_PITCH + c4);
  const int gm = m0 + row, col = n0 + wn * 32 + c4;
  if (gm >= p.M || col >= p.N) return;
  const float* scale = p.scale;
  const float* shift = p.shift;
  const float* resid = p.resid;
  if (p.batch > 1) {
    const int64_t so = (int64_t)slot(z, p.s_div, p.s_mod) * p.s_str;
    if (scale) scale += so;
    if (shift) shift += so;
    if (resid) resid += (int64_t)slot(z, p.r_div, p.r_mod) * p.r_str;
  }
  const int actk = p.act & 15;
  const bool post = (p.act & 16) != 0;
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f}, rv = {0.f, 0.f, 0.f, 0.f};
  if (scale) sc = *reinterpret_cast<const f32x4*>(scale + col);
  if (shift) sh = *reinterpret_cast<const f32x4*>(shift + col);
  if (resid) rv = *reinterpret_cast<const f32x4*>(resid + (int64_t)gm * p.ldr + col);
  v = v * sc + sh;
  if (!post) v += rv;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (actk == 1) v[e] = fmaxf(v[e], 0.f);
    else if (actk == 2) v[e] = v[e] > 0.f ? v[e] : v[e] * p.slope;
  }
  if (post) v += rv;
  *reinterpret_cast<f32x4*>(C + (int64_t)gm * p.ldc + col) = v;
}

// geometry the kernel takes (host logic; cadre_gemm_f32 falls back to the tile kernels otherwise)
int cadre_gemm_f32_skinny_ok(const cadre_gemm_t& p) {
  if (p.a_mode != 0 || p.b_mode < 0 || p.b_mode > 1) return 0;
  if (p.seg_mode != 0 && p.seg_mode != 1) return 0;
  if (p.seg_mode == 1 && p.seg_period % SK_BM != 0) return 0;
  if (p.split_k > 1 || (p.flags & 2)) return 0;
  if (p.K % 4 != 0 || p.lda % 4 != 0 || (p.b_mode == 0 && p.ldb % 4 != 0)) return 0;
  if (((p.N | p.ldc | (p.resid ? p.ldr : 0)) & 3) != 0) return 0;
  if ((((uintptr_t)p.A | (uintptr_t)p.B | (uintptr_t)p.C | (uintptr_t)p.resid | (uintptr_t)p.scale | (uintptr_t)p.shift) & 15) != 0) return 0;
  const int64_t lim = 1ll << 31;
  if ((int64_t)p.M * p.lda * 4 >= lim || (int64_t)(p.b_mode == 0 ? p.N : p.K) * p.ldb * 4 >= lim) return 0;
  if ((p.a_str | p.b_str | p.c_str | p.s_str | p.r_str) & 3) return 0;      // batch strides keep the 16-byte alignment
  return 1;
}

int cadre_gemm_f32_skinny_launch(const cadre_gemm_t& p, void* stream) {
  const int tiles = ((p.M + SK_BM - 1) / SK_BM) * ((p.N + SK_BN - 1) / SK_BN);
  dim3 grid(tiles, 1, p.batch < 1 ? 1 : p.batch);
  if (p.b_mode == 0) hipLaunchKernelGGL((gemm_f32_skinny_kernel<0>), grid, dim3(512), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL((gemm_f32_skinny_kernel<1>), grid, dim3(512), 0, (hipStream_t)stream, p);
  return (int)hipGetLastError();
}
