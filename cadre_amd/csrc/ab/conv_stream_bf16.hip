// conv_stream_bf16.hip — bf16-input / fp32-accumulate implicit-GEMM convolution, several M-tiles per
// workgroup (gfx950).  The bf16 twin of conv_stream_f32.hip: same math, operand layout, k order and
// epilogue as gemm_bf16_kernel<1,1,AMODE,2,*> (64x64 tile, 2x2 wave64, v_mfma_f32_32x32x16_bf16,
// 64 bf16 per k-tile), bit-identical results; the k-tile stream runs across the MT M-tiles of a
// workgroup so that only the first one pays the global-load round trip in front of its first MFMA.
// At 16x the fp32 MFMA rate the short-K layers (N = 64 stage, 9 k-tiles; padded stem, 4) are bound by
// exactly that start-up and by workgroup turnover: 430 TFLOP/s whatever the tile shape.
// Reference ops: resnet.py:26-55,111-112,152-166.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "../../../include/cadre_hip_ab.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define SBK 64          // bf16 elements per k-tile (128 bytes)
#define SPITCH 36

template <int AMODE>
__global__ __launch_bounds__(256, 2) void conv_stream_bf16_kernel(cadre_gemm_t p, int MT) {
  constexpr int BM = 64, BN = 64, RA = 2, RB = 2, RP = 32;
  __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * SPITCH];
  float* As = lds;
  float* Bs = lds + 2 * BM * SPITCH;
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
  const int nk = (p.K + SBK - 1) / SBK;

  constexpr unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)OOB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)OOB, 0x00020000);
  const int cc = tid & 7, rr = tid >> 3;

  // im2col decode of the rows this thread stages for local tile `t` (see gemm_f32.hip)
  const int hw = p.Ho * p.Wo;
  const float inv_wo = 1.0f / (float)p.Wo, inv_ho = 1.0f / (float)p.Ho;
  auto decode = [&](int t, unsigned (&aoff)[RA], unsigned (&amask)[RA]) {
    const int m0 = (tile_first + t) * BM;
    const int img0 = m0 / hw, rem0 = m0 % hw;
    const int ho0 = rem0 / p.Wo, wo0 = rem0 % p.Wo;
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
      const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
      unsigned mask = 0;
      if constexpr (AMODE == 2) {
        aoff[i] = (unsigned)((((img * p.H + hi0) * p.W + wi0) * p.Cin) * 2 + cc * 16);
        if (m < p.M) {
          unsigned colm = 0;
          for (int kw = 0; kw < p.KW; ++kw)
            if ((unsigned)(wi0 + kw) < (unsigned)p.W) colm |= 1u << kw;
          for (int kh = 0; kh < p.KH; ++kh)
            if ((unsigned)(hi0 + kh) < (unsigned)p.H) mask |= colm << (kh * p.KW);
        }
      } else {      // Cin == 4 stem on the zero-padded NHWC4 image (gemm_bf16.hip a_mode 4): no tap masks
        aoff[i] = m < p.M ? (unsigned)((((img * p.H + ho * p.stride + (cc >> 2)) * p.W + wo * p.stride + 2 * (cc & 3)) * 4) * 2) : OOB;
        mask = m < p.M ? 0xffffffffu : 0u;
      }
      amask[i] = mask;
    }
  };
  unsigned aoff0[RA], amask0[RA], aoff1[RA], amask1[RA];     // even / odd local tiles
  decode(0, aoff0, amask0);
#pragma unroll
  for (int i = 0; i < RA; ++i) { aoff1[i] = OOB; amask1[i] = 0; }
  if (ntile > 1) decode(1, aoff1, amask1);
  unsigned boff[RB];
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    const int n = n0 + rr + RP * i;
    boff[i] = n < p.N ? (unsigned)((int64_t)n * p.ldb * 2 + cc * 16) : OOB;
  }

  f32x4 areg[RA], breg[RB];            // one register set: two would need static set/buffer parity across M-tile
                                       // boundaries (odd k-tile counts), and hipcc then drains vmcnt in the k-loop
  auto ldg = [](const __amdgpu_buffer_rsrc_t& rs, unsigned off) -> f32x4 {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
  };
  // load stream: (lt, lk) = local tile / k-tile of the next request; past the end -> OOB (zeros).  The
  // wave-uniform part of a request (tap offset, tap bit, k offset, decode-set parity) is computed one
  // request ahead, so that the loads themselves can issue at the top of the staging block.
  int lt = 0, lk = 0;
  unsigned ld_delta = 0, ld_bit = 0, ld_kb = 0;
  bool ld_odd = false;
  auto plan_next = [&]() {
    const bool live = lt < ntile;
    const int k0 = lk * SBK;
    if constexpr (AMODE == 2) {
      const int pos = k0 / p.Cin, ci = k0 % p.Cin;
      ld_delta = (unsigned)((((pos / p.KW) * p.W + (pos % p.KW)) * p.Cin + ci) * 2);
      ld_bit = (live && pos < 32) ? 1u << pos : 0u;
    } else {
      ld_delta = (unsigned)(2 * lk * p.W * 8);              // two padded rows per k-tile
      ld_bit = live ? 1u : 0u;
    }
    ld_kb = live ? (unsigned)k0 * 2u : OOB;
    ld_odd = (lt & 1) != 0;
    if (++lk == nk) { lk = 0; ++lt; }
  };
  auto load_next = [&]() {
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const unsigned off = ld_odd ? aoff1[i] : aoff0[i];
      const unsigned msk = ld_odd ? amask1[i] : amask0[i];
      areg[i] = ldg(rsA, (msk & ld_bit) ? off + ld_delta : OOB);
    }
    const unsigned kb_ = (ld_kb != OOB && (ld_kb >> 1) + cc * 8 < (unsigned)p.K) ? ld_kb : OOB;
#pragma unroll
    for (int i = 0; i < RB; ++i) breg[i] = ldg(rsB, boff[i] + kb_);
    __builtin_amdgcn_sched_barrier(0);      // keep the requests here, ahead of the MFMAs
    plan_next();
  };
  auto store_tiles = [&](int buf) {
    float* as = As + buf * BM * SPITCH;
    float* bs = Bs + buf * BN * SPITCH;
#pragma unroll
    for (int i = 0; i < RA; ++i) *reinterpret_cast<f32x4*>(as + (rr + RP * i) * SPITCH + cc * 4) = areg[i];
#pragma unroll
    for (int i = 0; i < RB; ++i) *reinterpret_cast<f32x4*>(bs + (rr + RP * i) * SPITCH + cc * 4) = breg[i];
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  auto compute = [&](int buf, auto&& staging) {
    const float* as = As + buf * BM * SPITCH + (wm * 32 + l31) * SPITCH;
    const float* bs = Bs + buf * BN * SPITCH + (wn * 32 + l31) * SPITCH;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {          // k-step: 16 bf16; this lane half reads chunk 2*ks + lh
      const bf16x8 af = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(as + (2 * ks + lh) * 4));
      const bf16x8 bf = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(bs + (2 * ks + lh) * 4));
      if (ks == 0) staging();
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc, 0, 0, 0);
    }
  };

  // ---- epilogue pieces that do not depend on the M-tile
  const int actk = p.act & 15;
  const bool post = (p.act & 16) != 0;
  constexpr int P = 36, LPR = 8, RPI = 8, NIT = 4;
  const int c4 = (lane % LPR) * 4;
  const int col = n0 + wn * 32 + c4;
  const bool cvalid = col < p.N;
  const bool c16 = (p.flags & 2) != 0;
  const int esz = c16 ? 2 : 4;
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
  if (cvalid && p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + col);
  if (cvalid && p.shift) sh = *reinterpret_cast<const f32x4*>(p.shift + col);
  const int lrow = lane / LPR;
  const unsigned coff = cvalid ? (unsigned)(((wm * 32 + lrow) * p.ldc + col) * esz) : OOB;
  const bool r16 = (p.flags & 4) != 0;      // residual is bf16
  const int rsz = r16 ? 2 : 4;
  const unsigned roff = cvalid ? (unsigned)(((wm * 32 + lrow) * p.ldr + col) * rsz) : OOB;
  const float slope = p.slope;
  auto window = [](int64_t bytes) { return (int)(bytes < 0x7fffffff ? bytes : 0x7fffffff); };

  auto epilogue = [&](int t, int buf, auto actc, auto resc) {
    constexpr int ACT = decltype(actc)::value;
    constexpr int RES = decltype(resc)::value;        // 0 none, 1 f32 residual, 2 bf16 residual
    const int m0 = (tile_first + t) * BM;
    const int64_t rows_left = (int64_t)p.M - m0;
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<char*>(p.C) + (int64_t)m0 * p.ldc * esz), 0, window(rows_left * p.ldc * esz), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(RES ? reinterpret_cast<const char*>(p.resid) + (int64_t)m0 * p.ldr * rsz : reinterpret_cast<const char*>(p.C)), 0,
        RES ? window(rows_left * p.ldr * rsz) : 0, 0x00020000);
    // this wave's staging slice inside the LDS buffer the k-loop just released
    float* cs = (wave < 2 ? As + buf * BM * SPITCH : Bs + buf * BN * SPITCH) + (wave & 1) * (32 * P);
    f32x4 rv[NIT];
    if constexpr (RES != 0) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const unsigned off = roff + (unsigned)(it * RPI * p.ldr * rsz);
        if constexpr (RES == 2) {
          const bf16x4 t = __builtin_bit_cast(bf16x4, __builtin_amdgcn_raw_buffer_load_b64(rsR, (int)off, 0, 0));
          rv[it] = f32x4{(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
        } else {
          rv[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsR, (int)off, 0, 0));
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) cs[((r & 3) + 8 * (r >> 2) + 4 * lh) * P + l31] = acc[r];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      f32x4 v = *reinterpret_cast<const f32x4*>(cs + (it * RPI + lrow) * P + c4);
      v = v * sc + sh;
      if constexpr (RES != 0) { if (!post) v += rv[it]; }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if constexpr (ACT == 1) v[e] = fmaxf(v[e], 0.f);
        if constexpr (ACT == 2) v[e] = v[e] > 0.f ? v[e] : v[e] * slope;
      }
      if constexpr (RES != 0) { if (post) v += rv[it]; }
      const unsigned off = coff + (unsigned)(it * RPI * p.ldc * esz);
      if (c16) {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        bf16x4 o;
        o[0] = (__bf16)v[0]; o[1] = (__bf16)v[1]; o[2] = (__bf16)v[2]; o[3] = (__bf16)v[3];
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), rsC, (int)off, 0, 0);
      } else {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsC, (int)off, 0, 0);
      }
    }
  };
  auto epilogue_dispatch = [&](int t, int buf) {
    using std::integral_constant;
#define SB_ACT(RES_)                                                                                       \
  do {                                                                                                     \
    if (actk == 1) epilogue(t, buf, integral_constant<int, 1>{}, integral_constant<int, RES_>{});          \
    else if (actk == 2) epilogue(t, buf, integral_constant<int, 2>{}, integral_constant<int, RES_>{});     \
    else epilogue(t, buf, integral_constant<int, 0>{}, integral_constant<int, RES_>{});                    \
  } while (0)
    if (!p.resid) SB_ACT(0);
    else if (r16) SB_ACT(2);
    else SB_ACT(1);
#undef SB_ACT
  };

  // ---- the k-tile stream: the outer loop walks the workgroup's M-tiles, the inner one their k-tiles; the
  // load stream (lt, lk) runs two k-tiles ahead of `v` and does not care about the loop nest.  Step v:
  // barrier; first fragment reads of k-tile v; write k-tile v+1 into the other buffer and request v+2; MFMAs.
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
    // retire the epilogue's residual loads / stores here, once per tile: otherwise hipcc's waitcnt pass merges
    // them into the k-loop header and makes every k-tile wait for its prefetch in front of the barrier
    __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (ct + 2 < ntile) {                 // the set of tile ct is free: the load stream is inside tile ct+1
      if (ct & 1) decode(ct + 2, aoff1, amask1);
      else decode(ct + 2, aoff0, amask0);
    }
  }
}

int cadre_fail(const char* msg);

// Called by cadre_gemm_bf16 for tile id 12 (arguments already validated there).
int cadre_conv_stream_bf16_launch(const cadre_gemm_t& p, void* stream) {
  if (!(p.a_mode == 2 || p.a_mode == 4) || p.b_mode != 0 || p.batch > 1 || p.split_k > 1)
    return cadre_fail("cadre_gemm_bf16: tile 12 (streamed conv) needs a plain conv launch");
  const int nk = (p.K + SBK - 1) / SBK;
  if (nk < 2) return cadre_fail("cadre_gemm_bf16: tile 12 needs K >= 128");
  const int64_t tilesM = (p.M + 63) / 64, tilesN = (p.N + 63) / 64;
  int64_t mt = tilesM * tilesN / 8192;
  const int MT = (int)(mt < 1 ? 1 : (mt > 8 ? 8 : mt));
  const int64_t groups = (tilesM + MT - 1) / MT;
  dim3 grid((unsigned)(groups * tilesN)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (p.a_mode == 2) hipLaunchKernelGGL((conv_stream_bf16_kernel<2>), grid, block, 0, st, p, MT);
  else hipLaunchKernelGGL((conv_stream_bf16_kernel<4>), grid, block, 0, st, p, MT);
  return (int)hipGetLastError();
}
