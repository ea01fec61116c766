// gemm_stream_f32.hip — fp32 NT product C = act((A . B^T) * scale + shift + resid) for SHORT K, several M-tiles per
// workgroup (gfx950): tile 13 of cadre_gemm_f32.
//
// Same math, operand layout (A [M][lda], B [N][ldb], k contiguous), k order and epilogue as gemm_f32_kernel<1,1,0,0>
// (64 x 64 tile, 2 x 2 wave64, v_mfma_f32_32x32x2_f32, BK = 32, raw buffer loads with hardware zero fill, 36-float
// LDS pitch) — results are bit-identical to it.  The difference is the schedule: a workgroup walks MT consecutive
// M-tiles of one N-tile (and one batch entry) as ONE stream of k-tiles; the register-staged prefetch runs two k-tiles
// ahead straight across tile boundaries, so only the first tile of a workgroup pays the global-load round trip in
// front of its first MFMA.  With one tile per workgroup that start-up (~2 us) is a quarter of the lifetime of a
// K = 128 tile (four k-tiles) and the resident workgroups of a CU do not cover it: the batched GEMMs of the Winograd
// convs (csrc/winograd.hip: 25 planes, K = Cin = 128 / 256) ran at 100 / 122 TFLOP/s on the one-tile kernels.
// MEASURED AND NOT ADOPTED (A/B build only): 98 / 113 / 121 TFLOP/s at K = 128 / 256 / 512 against 99 / 121 / 133 of the
// 128 x 128 8-wave tile (profiles/r04_winograd_gemm_streamed_tile.txt) — the start-up is not what binds those products: at
// K = 128 they move 3.8 GB (half of it written) in the time the matrix pipe needs, 64 % of either roof.
// At a tile boundary the finished tile's accumulators go out through the LDS buffer that was just consumed (the other
// buffer already holds the next tile's first k-tile), one extra barrier per tile.  The schedule is the one measured in
// the A/B build on short-K convolutions (ab/conv_stream_f32.hip), here on dense A with batch addressing.
// Reference ops served: the stride-1 3x3 convs of resnet.py:26-55 / danet.py:21-41 in their Winograd form.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "../../../include/cadre_hip_ab.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define GS_BK 32
#define GS_PITCH 36

int cadre_fail(const char* msg);

__global__ __launch_bounds__(256, 2) void gemm_stream_f32_kernel(cadre_gemm_t p, int MT) {
  constexpr int BM = 64, BN = 64, RA = 2, RB = 2, RP = 32;
  __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * GS_PITCH];
  float* As = lds;
  float* Bs = lds + 2 * BM * GS_PITCH;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;

  const int tilesN = (p.N + BN - 1) / BN, tilesM = (p.M + BM - 1) / BM;
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int grp = bid / tilesN, tile_n = bid % tilesN;
  const int n0 = tile_n * BN;
  const int tile_first = grp * MT;
  const int ntile = min(MT, tilesM - tile_first);
  const int nk = (p.K + GS_BK - 1) / GS_BK;

  // batch entry z: operands at (z / div) % mod strides (cadre_gemm_t)
  const int z = blockIdx.z;
  auto slot = [](int zz, int div, int mod) { return (zz / div) % mod; };
  const float* A = p.A;
  const float* B = p.B;
  float* C = p.C;
  const float* scale = p.scale;
  const float* shift = p.shift;
  const float* resid = p.resid;
  if (p.batch > 1) {
    A += (int64_t)slot(z, p.a_div, p.a_mod) * p.a_str;
    B += (int64_t)slot(z, p.b_div, p.b_mod) * p.b_str;
    C += (int64_t)slot(z, p.c_div, p.c_mod) * p.c_str;
    const int64_t so = (int64_t)slot(z, p.s_div, p.s_mod) * p.s_str;
    if (scale) scale += so;
    if (shift) shift += so;
    if (resid) resid += (int64_t)slot(z, p.r_div, p.r_mod) * p.r_str;
  }

  constexpr unsigned OOB = 0x80000000u;
  // rows >= M / >= N lie past num_records: the buffer unit returns zeros, no per-row mask
  auto window = [](int64_t bytes) { return (int)(bytes < 0x7fffffff ? bytes : 0x7fffffff); };
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, window(((int64_t)(p.M - 1) * p.lda + p.K) * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, window(((int64_t)(p.N - 1) * p.ldb + p.K) * 4), 0x00020000);
  const int cc = tid & 7, rr = tid >> 3;
  unsigned aoff[RA], boff[RB];
#pragma unroll
  for (int i = 0; i < RA; ++i) aoff[i] = (unsigned)(((int64_t)(tile_first * BM + rr + RP * i) * p.lda + cc * 4) * 4);
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    const int n = n0 + rr + RP * i;
    boff[i] = n < p.N ? (unsigned)(((int64_t)n * p.ldb + cc * 4) * 4) : OOB;
  }
  const unsigned tile_step = (unsigned)((int64_t)BM * p.lda * 4);

  f32x4 areg[RA], breg[RB];            // one register set (see ab/conv_stream_f32.hip)
  auto ldg = [](const __amdgpu_buffer_rsrc_t& rs, unsigned off) -> f32x4 {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
  };
  // load stream: (lt, lk) = local tile / k-tile of the next request; past the end -> OOB (zeros).  The wave-uniform
  // part of a request is computed one request ahead, so that the loads themselves issue at the top of the staging block.
  int lt = 0, lk = 0;
  unsigned ld_a = 0, ld_kb = 0;
  auto plan_next = [&]() {
    const bool live = lt < ntile;
    const int k0 = lk * GS_BK;
    ld_kb = (live && k0 + cc * 4 < p.K) ? (unsigned)k0 * 4u : OOB;          // K % 4 == 0
    ld_a = (unsigned)lt * tile_step;
    if (++lk == nk) { lk = 0; ++lt; }
  };
  auto load_next = [&]() {
#pragma unroll
    for (int i = 0; i < RA; ++i) areg[i] = ldg(rsA, ld_kb != OOB ? aoff[i] + ld_a + ld_kb : OOB);
#pragma unroll
    for (int i = 0; i < RB; ++i) breg[i] = ldg(rsB, (ld_kb != OOB && boff[i] != OOB) ? boff[i] + ld_kb : OOB);
    __builtin_amdgcn_sched_barrier(0);      // keep the requests here, ahead of the MFMAs
    plan_next();
  };
  auto store_tiles = [&](int buf) {
    float* as = As + buf * BM * GS_PITCH;
    float* bs = Bs + buf * BN * GS_PITCH;
#pragma unroll
    for (int i = 0; i < RA; ++i) *reinterpret_cast<f32x4*>(as + (rr + RP * i) * GS_PITCH + cc * 4) = areg[i];
#pragma unroll
    for (int i = 0; i < RB; ++i) *reinterpret_cast<f32x4*>(bs + (rr + RP * i) * GS_PITCH + cc * 4) = breg[i];
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  auto compute = [&](int buf, auto&& staging) {
    const float* as = As + buf * BM * GS_PITCH + (wm * 32 + l31) * GS_PITCH;
    const float* bs = Bs + buf * BN * GS_PITCH + (wn * 32 + l31) * GS_PITCH;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const int kq = kb * 8 + lh * 4;
      const f32x4 af = *reinterpret_cast<const f32x4*>(as + kq);
      const f32x4 bf = *reinterpret_cast<const f32x4*>(bs + kq);
      if (kb == 0) staging();
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], bf[s], acc, 0, 0, 0);
    }
  };

  // ---- epilogue pieces that do not depend on the M-tile
  const int actk = p.act & 15;
  const bool post = (p.act & 16) != 0;
  constexpr int P = 36, LPR = 8, RPI = 8, NIT = 4;
  const int c4 = (lane % LPR) * 4;
  const int col = n0 + wn * 32 + c4;
  const bool cvalid = col < p.N;
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
  if (cvalid && scale) sc = *reinterpret_cast<const f32x4*>(scale + col);
  if (cvalid && shift) sh = *reinterpret_cast<const f32x4*>(shift + col);
  const int lrow = lane / LPR;
  const unsigned coff = cvalid ? (unsigned)(((wm * 32 + lrow) * p.ldc + col) * 4) : OOB;
  const unsigned roff = cvalid ? (unsigned)(((wm * 32 + lrow) * p.ldr + col) * 4) : OOB;
  const float slope = p.slope;

  auto epilogue = [&](int t, int buf, auto actc, auto resc) {
    constexpr int ACT = decltype(actc)::value;
    constexpr bool RES = decltype(resc)::value;
    const int m0 = (tile_first + t) * BM;
    const int64_t rows_left = (int64_t)p.M - m0;
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc((void*)(C + (int64_t)m0 * p.ldc), 0, window(rows_left * p.ldc * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void*)(RES ? resid + (int64_t)m0 * p.ldr : C), 0,
                                                                         RES ? window(rows_left * p.ldr * 4) : 0, 0x00020000);
    // this wave's staging slice inside the LDS buffer the k-loop just released
    float* cs = (wave < 2 ? As + buf * BM * GS_PITCH : Bs + buf * BN * GS_PITCH) + (wave & 1) * (32 * P);
    f32x4 rv[NIT];
    if constexpr (RES) {
#pragma unroll
      for (int it = 0; it < NIT; ++it)
        rv[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsR, (int)(roff + (unsigned)(it * RPI * p.ldr * 4)), 0, 0));
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) cs[((r & 3) + 8 * (r >> 2) + 4 * lh) * P + l31] = acc[r];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      f32x4 v = *reinterpret_cast<const f32x4*>(cs + (it * RPI + lrow) * P + c4);
      v = v * sc + sh;
      if constexpr (RES) { if (!post) v += rv[it]; }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if constexpr (ACT == 1) v[e] = fmaxf(v[e], 0.f);
        if constexpr (ACT == 2) v[e] = v[e] > 0.f ? v[e] : v[e] * slope;
      }
      if constexpr (RES) { if (post) v += rv[it]; }
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsC, (int)(coff + (unsigned)(it * RPI * p.ldc * 4)), 0, 0);
    }
  };
  auto epilogue_dispatch = [&](int t, int buf) {
    using std::integral_constant;
    if (resid) {
      if (actk == 1) epilogue(t, buf, integral_constant<int, 1>{}, integral_constant<bool, true>{});
      else if (actk == 2) epilogue(t, buf, integral_constant<int, 2>{}, integral_constant<bool, true>{});
      else epilogue(t, buf, integral_constant<int, 0>{}, integral_constant<bool, true>{});
    } else {
      if (actk == 1) epilogue(t, buf, integral_constant<int, 1>{}, integral_constant<bool, false>{});
      else if (actk == 2) epilogue(t, buf, integral_constant<int, 2>{}, integral_constant<bool, false>{});
      else epilogue(t, buf, integral_constant<int, 0>{}, integral_constant<bool, false>{});
    }
  };

  // ---- the k-tile stream: the outer loop walks the workgroup's M-tiles, the inner one their k-tiles; the load stream
  // (lt, lk) runs two k-tiles ahead of `v` and does not care about the loop nest.  Step v: barrier; first fragment
  // reads of k-tile v; write k-tile v+1 into the other buffer and request v+2; MFMAs.
  plan_next();
  load_next();
  store_tiles(0);
  load_next();
  int v = 0;
  for (int ct = 0; ct < ntile; ++ct) {
    for (int ck = 0; ck < nk; ++ck, ++v) {
      const int buf = v & 1;
      __syncthreads();
      compute(buf, [&] {
        store_tiles(buf ^ 1);
        load_next();
      });
    }
    __syncthreads();                      // every wave is done reading the last buffer: it becomes the epilogue's staging area
    epilogue_dispatch(ct, (v - 1) & 1);
    // retire the epilogue's residual loads / stores here, once per tile (otherwise hipcc's waitcnt pass merges them into
    // the k-loop header and every k-tile waits for its prefetch in front of the barrier)
    __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  }
}

// Host logic: can tile 13 take this descriptor?
int cadre_gemm_stream_f32_ok(const cadre_gemm_t& p) {
  return p.a_mode == 0 && p.b_mode == 0 && p.split_k <= 1 && !p.seg_mode && !(p.flags & 2) && p.K >= 2 * GS_BK && (p.K & 3) == 0 &&
         ((p.N | p.ldc | p.lda | p.ldb | (p.resid ? p.ldr : 0)) & 3) == 0 &&
         (((uintptr_t)p.A | (uintptr_t)p.B | (uintptr_t)p.C | (uintptr_t)p.resid | (uintptr_t)p.scale | (uintptr_t)p.shift) & 15) == 0;
}

// Called by cadre_gemm_f32 for tile id 13 (arguments already validated and normalised there).
int cadre_gemm_stream_f32_launch(const cadre_gemm_t& p, void* stream) {
  if (!cadre_gemm_stream_f32_ok(p))
    return cadre_fail("cadre_gemm_f32: tile 13 (streamed short-K product) needs a dense NT product, K >= 64, K/N/ld* % 4 == 0, 16-byte aligned operands, no split-K / row segments / bf16 output");
  const int64_t tilesM = (p.M + 63) / 64, tilesN = (p.N + 63) / 64;
  const int batch = p.batch < 1 ? 1 : p.batch;
  // enough workgroups to fill 256 CUs x 4 slots several times over, otherwise as many tiles per workgroup as possible
  int64_t mt = tilesM * tilesN * batch / 8192;
  const int MT = (int)(mt < 1 ? 1 : (mt > 8 ? 8 : mt));
  const int64_t groups = (tilesM + MT - 1) / MT;
  dim3 grid((unsigned)(groups * tilesN), 1, (unsigned)batch), block(256);
  hipLaunchKernelGGL(gemm_stream_f32_kernel, grid, block, 0, (hipStream_t)stream, p, MT);
  return (int)hipGetLastError();
}
