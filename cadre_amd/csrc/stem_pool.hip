// stem_pool.hip — the front of the DANet encoder as ONE kernel for gfx950:
//   u8 observation (packed RGB + route, cadre_pack_obs) -> /255 LUT (agent.py:46) -> 7x7/s2 conv +
//   folded eval-BatchNorm + ReLU -> 3x3/s2 max-pool (resnet.py:111-115, 168-172) -> NHWC pooled map.
// Replaces preprocess (1.36 GB f32 write at 1024 x 288x288), the stem conv's 5.4 GB output and the
// max-pool's re-read of it: HBM traffic of the front drops from ~14 GB to ~1.7 GB per 1024 frames, and the
// stem runs its MFMAs with K = 200 (49 taps x 4 channels, padded by ONE tap) instead of 224.
//
// Work decomposition: a PAIR of waves (one per 32-channel half) owns (frame, band of pooled rows) and walks down
//   the band one pooled row per iteration; four pairs per workgroup (2 waves per SIMD), each pair with its own ring.
//   Iteration p computes stem rows 2p, 2p+1 as NT MFMA tiles of "2 stem rows x 16 stem columns" x 64
//   channels (v_mfma_f32_32x32x2_f32 / v_mfma_f32_32x32x16_bf16).  Tile row i = (dy, dx) with
//   dy = (i>>2)&1, dx = (i&3) + 4*(i>>3): in the accumulator layout (row = (reg&3) + 8*(reg>>2) + 4*(lane>>5),
//   col = lane&31) lane half h then holds stem row 2p+h, columns x0 .. x0+15 in its 16 registers — the
//   horizontal 3-max is register-local, the vertical one a lane-half exchange (v_permlane32_swap) plus the
//   previous iteration's row kept in registers.  The A operand is formed from the pair's LDS ring of 12 input
//   rows (even / odd pixel planes: stride-2 taps become unit stride, bank-conflict free).  fp32: the ring holds the
//   PACKED u8 pixels and each fragment element is converted by div255() — bit-identical to the reference's
//   float32(rgb / 255.) for all 256 inputs (cadre_div255_selfcheck); bf16: converted once per pixel at staging.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/cadre_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

int cadre_fail(const char* msg);

#define SP_RING 12        // input rows in a ring (3 groups of 4)
#define SP_TAPS32 50      // fp32: 49 taps + 1 zero tap  -> K = 200
#define SP_WP32 204       // fp32 weight row pitch in floats ((pitch/4) odd: conflict-free ds_read_b128 over rows)
#define SP_TAPS16 52      // bf16: 49 taps + 3 zero taps -> K = 208 (13 k-steps of 16)
#define SP_WP16 216       // bf16 weight row pitch in bf16 elements (432 B: (pitch_bytes/16) odd)

struct stem_args {
  const uint32_t* img;    // [F][H][W] packed pixels: R | G<<8 | B<<16 | route<<24 (route byte 0 or 255)
  const void* wt;         // fp32 [64][50][4] or bf16 [64][52][4]: tap-major (tap = ky*7 + kx), zero padded
  const float* scale;     // folded BN, [64]
  const float* shift;
  void* out;              // pooled map
  int F, H, W, Hs, Ws, Hp, Wp;
  int PB, NB;             // pooled rows per band, bands per frame
  int PP;                 // ring plane pitch in pixels
  int total;              // F * NB (frame, band) units; one PAIR of waves per unit
  long long out_frame, out_row;      // output strides in elements: frame, pooled row
  int out_px;                        // pooled pixel stride in elements (>= 64)
  long long out_off;                 // element offset of pooled (0, 0, 0, ch 0)
};

__device__ __forceinline__ float lo_to_hi(float v) {     // lanes 32..63 receive the value of lane - 32
  const unsigned u = __builtin_bit_cast(unsigned, v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __builtin_bit_cast(float, r[0]);
}

// float32(i / 255.) for an integer-valued float i in [0, 255], EXACTLY the reference's value (agent.py:46 divides in
// double and stores float32): one multiply by fl(1/255) and one Newton correction — checked against the host table for
// all 256 inputs by cadre_div255_selfcheck (tests/test_kernels_gpu.py).  4 VALU per value, no table lookup, no LDS.
__device__ __forceinline__ float div255(float f) {
  const float r = 0.00392156862745098f;
  const float q0 = f * r;
  const float e = __builtin_fmaf(-q0, 255.0f, f);
  return __builtin_fmaf(e, r, q0);
}

__global__ void div255_check_kernel(const float* lut, int* bad) {
  const int i = threadIdx.x;
  if (div255((float)i) != lut[i]) atomicAdd(bad, 1);
}
extern "C" int cadre_div255_selfcheck(const float* lut255, int32_t* mismatches, void* stream) {
  if (!lut255 || !mismatches) return cadre_fail("cadre_div255_selfcheck: bad argument");
  hipError_t e = hipMemsetAsync(mismatches, 0, sizeof(int32_t), (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(div255_check_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, lut255, mismatches);
  return (int)hipGetLastError();
}

// 8 waves per workgroup = 4 (frame, band) units x 2 channel halves; the two waves of a unit share one input ring.
// Two workgroup barriers per iteration fence the ring (all reads of iteration p | overwrite group p by group p+3).
template <int NT, int CH, bool RAGGED, bool BF16>
__global__ __launch_bounds__(512, 2) void stem_pool_kernel(stem_args a) {
  static_assert(NT % CH == 0, "tiles per row pair must split into whole chunks");
  constexpr int NLD = (NT + 3) / 4;                 // 16-B chunk loads per lane per 4-row group (two waves, W <= 32*NT)
  constexpr int WROW = BF16 ? SP_WP16 / 2 : SP_WP32;   // weight row pitch in dwords
  constexpr int NK = BF16 ? SP_TAPS16 / 4 : SP_TAPS32 / 2;   // k-steps: bf16 4 taps (16 k), fp32 2 taps (8 k = 4 MFMAs)
  constexpr int PXD = BF16 ? 2 : 1;                 // ring dwords per pixel: bf16 x 4 channels | packed u8 x 4
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* wts = smem;                                // [64][WROW] dwords
  uint32_t* ring0 = reinterpret_cast<uint32_t*>(smem + 64 * WROW);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int pair = wave >> 1, hN = wave & 1;
  {
    constexpr int CPR = (BF16 ? SP_TAPS16 * 8 : SP_TAPS32 * 16) / 16;    // 16-B chunks per weight row
    const char* src = reinterpret_cast<const char*>(a.wt);
    for (int i = tid; i < 64 * CPR; i += 512) {
      const int n = i / CPR, c = i - n * CPR;
      *reinterpret_cast<f32x4*>(wts + n * WROW + c * 4) = *reinterpret_cast<const f32x4*>(src + ((size_t)n * CPR + c) * 16);
    }
  }
  const int u = blockIdx.x * 4 + pair;
  const bool valid = u < a.total;
  const int f = valid ? u / a.NB : 0, band = valid ? u - f * a.NB : 0;
  const int p0 = band * a.PB, p1 = valid ? min(a.Hp, p0 + a.PB) : 0;
  const int PP = a.PP, RP = 2 * PP * PXD;           // ring row pitch in dwords (two pixel planes)
  uint32_t* ring = ring0 + pair * (SP_RING * RP);
  for (int i = hN * 64 + lane; i < SP_RING * RP; i += 128) ring[i] = 0u;      // pads (3 px left, >= 5 right) stay zero
  const uint32_t* frame = a.img + (size_t)f * a.H * a.W;
  const int W4 = a.W >> 2;

  u32x4 pre[NLD];
  // group g = input rows 4g-3 .. 4g (contiguous in memory); rows outside the image are zeros (conv padding).
  // The two waves of the pair split the 16-B chunks (4 pixels each) between them.
  auto load_group = [&](int g) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int c = i * 128 + hN * 64 + lane;
      const int j = c / W4, r = 4 * g - 3 + j;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (c < a.W && r >= 0 && r < a.H) v = *reinterpret_cast<const u32x4*>(frame + (size_t)r * a.W + (c - j * W4) * 4);
      pre[i] = v;
    }
  };
  // ring pixel index q = x + 3: plane q & 1, index q >> 1.  A 4-pixel chunk at x (x % 4 == 0) lands as
  // plane 1 [x/2+1, x/2+2] <- px x, x+2 and plane 0 [x/2+2, x/2+3] <- px x+1, x+3.
  auto put = [&](uint32_t* dst, uint32_t px) {
    if constexpr (BF16) {                            // converted once per input pixel: bf16(float32(byte / 255.))
      typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
      bf16x4 o;
#pragma unroll
      for (int s = 0; s < 4; ++s) o[s] = (__bf16)div255((float)((px >> (8 * s)) & 255u));
      *reinterpret_cast<u32x2*>(dst) = __builtin_bit_cast(u32x2, o);
    } else {
      *dst = px;
    }
  };
  auto store_group = [&](int g) {
    const int gs = (g % 3) * 4;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int c = i * 128 + hN * 64 + lane;
      if (c < a.W) {
        const int j = c / W4, hx = (c - j * W4) * 2;               // hx = x / 2
        uint32_t* row = ring + (gs + j) * RP;
        put(row + (PP + hx + 1) * PXD, pre[i][0]);
        put(row + (hx + 2) * PXD, pre[i][1]);
        put(row + (PP + hx + 2) * PXD, pre[i][2]);
        put(row + (hx + 3) * PXD, pre[i][3]);
      }
    }
  };

  const int pstart = p0 > 0 ? p0 - 1 : 0;           // a band that does not start at the top first rebuilds stem row 2*p0-1
  __syncthreads();                                   // weights + zeroed rings
  for (int g = pstart; g < pstart + 3; ++g) {
    if (valid) { load_group(g); store_group(g); }
  }
  __syncthreads();

  const float sc = a.scale[32 * hN + l31], sh = a.shift[32 * hN + l31];
  // previous stem row's horizontal maxima.  bf16 build: kept as packed bf16 pairs (half the registers) — rounding is
  // monotonic, so max(bf16(a), b, c) rounded to bf16 is the same value as bf16(max(a, b, c)).
  constexpr int PHN = BF16 ? 4 : 8;
  float prevH[NT][PHN];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int k = 0; k < PHN; ++k) prevH[t][k] = 0.f;               // post-ReLU values are >= 0: 0 is the pool's -inf

  const int dy = (l31 >> 2) & 1, dx = (l31 & 3) + 4 * (l31 >> 3);
  char* outp = reinterpret_cast<char*>(a.out);
  const float* wrow = wts + (32 * hN + l31) * WROW;

  const int iters = a.PB + 1;                        // uniform over the workgroup (barriers inside)
  for (int it = 0; it < iters; ++it) {
    const int p = pstart + it;
    const bool act = p < p1;                         // wave-uniform
    if (act) {
      load_group(p + 3);                             // prefetch the next iteration's 4 new rows into registers
      const int sb = (p % 3) * 4 + 2 * dy;           // ring slot of input row 2*(2p+dy)-3 (+ ky, wrapped at 12)
      const bool emit = p >= p0;
      float carry = 0.f;
#pragma unroll
      for (int j = 0; j < NT / CH; ++j) {
        asm volatile("" ::: "memory");               // keep the B fragments of one chunk from being CSE'd (and kept live) across all chunks
        f32x16 acc[CH];
#pragma unroll
        for (int t = 0; t < CH; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        const int xb = j * CH * 16 + dx;
        if constexpr (!BF16) {
          // k-quad q: taps 2q (lane half 0) and 2q+1 (lane half 1), 4 channels each = 4 MFMAs of K = 2
          auto fetch = [&](int q, uint32_t* px) {
            const int tap = 2 * q + lh;
            const int ky = tap / 7, kx = tap - ky * 7;             // tap 49: ky 7, kx 0, zero weights
            int sl = sb + ky;
            sl = sl >= SP_RING ? sl - SP_RING : sl;
            const uint32_t* rp = ring + sl * RP + (kx & 1) * PP + xb + (kx >> 1);
#pragma unroll
            for (int t = 0; t < CH; ++t) px[t] = rp[16 * t];
          };
          // two-deep software pipeline, pinned with sched_barrier (hipcc otherwise sinks the ring read to its first
          // use and every k-quad starts with an exposed LDS round trip — 73 % MFMA-busy in profiles/r02 PMC):
          //   pixels of quad q+2 are requested, then the 4 MFMAs of quad q issue, then quad q+1 is converted.
          auto conv = [&](const uint32_t* px, float (*af)[4]) {
#pragma unroll
            for (int t = 0; t < CH; ++t)
#pragma unroll
              for (int s = 0; s < 4; ++s) af[t][s] = div255((float)((px[t] >> (8 * s)) & 255u));
          };
          uint32_t pxn[CH];
          float af[CH][4];
          fetch(0, pxn);
          conv(pxn, af);
          fetch(1, pxn);
          f32x4 b = *reinterpret_cast<const f32x4*>(wrow + lh * 4);
#pragma unroll
          for (int q = 0; q < NK; ++q) {
            f32x4 bn = b;
            if (q + 1 < NK) bn = *reinterpret_cast<const f32x4*>(wrow + (2 * (q + 1) + lh) * 4);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
              for (int t = 0; t < CH; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[t][s], b[s], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (q + 1 < NK) conv(pxn, af);
            if (q + 2 < NK) fetch(q + 2, pxn);
            b = bn;
          }
        } else {
          // k-step q: taps 4q + 2*half + {0,1}: two ring pixels (bf16 x 4 channels each) = the lane's 8 k-values
          typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
          auto fetch = [&](int q, u32x2 (*px)[2], f32x4& b) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const int tap = 4 * q + 2 * lh + e;
              const int ky = tap / 7, kx = tap - ky * 7;           // taps 49..51: ky 7, zero weights
              int sl = sb + ky;
              sl = sl >= SP_RING ? sl - SP_RING : sl;
              const uint32_t* rp = ring + sl * RP + ((kx & 1) * PP + xb + (kx >> 1)) * 2;
#pragma unroll
              for (int t = 0; t < CH; ++t) px[t][e] = *reinterpret_cast<const u32x2*>(rp + 32 * t);
            }
            b = *reinterpret_cast<const f32x4*>(wrow + (4 * q + 2 * lh) * 2);
          };
          u32x2 pxn[CH][2];
          f32x4 bn;
          fetch(0, pxn, bn);
#pragma unroll
          for (int q = 0; q < NK; ++q) {
            u32x4 av[CH];
#pragma unroll
            for (int t = 0; t < CH; ++t) av[t] = u32x4{pxn[t][0][0], pxn[t][0][1], pxn[t][1][0], pxn[t][1][1]};
            const bf16x8 b = __builtin_bit_cast(bf16x8, bn);
            if (q + 1 < NK) fetch(q + 1, pxn, bn);                 // one k-step ahead, no further (register budget)
#pragma unroll
            for (int t = 0; t < CH; ++t)
              acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[t]), b, acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        // ---- chunk epilogue: BN + ReLU, horizontal 3-max in registers, vertical 3-max across lane halves + previous row
#pragma unroll
        for (int t = 0; t < CH; ++t) {
          const int T = j * CH + t;
          float y[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = fmaxf(acc[t][r] * sc + sh, 0.f);
            if constexpr (RAGGED) {
              if (16 * T + r >= a.Ws || 2 * p + lh >= a.Hs) v = 0.f;       // outside the stem map: pool padding
            }
            y[r] = v;
          }
          float h[8];
          h[0] = fmaxf(fmaxf(carry, y[0]), y[1]);
#pragma unroll
          for (int k = 1; k < 8; ++k) h[k] = fmaxf(fmaxf(y[2 * k - 1], y[2 * k]), y[2 * k + 1]);
          carry = y[15];
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float lo = lo_to_hi(h[k]);                               // stem row 2p (held by lane half 0)
            float pv;
            if constexpr (BF16) {
              const unsigned w = __builtin_bit_cast(unsigned, prevH[T][k >> 1]);
              pv = __builtin_bit_cast(float, (k & 1) ? (w & 0xffff0000u) : (w << 16));
            } else {
              pv = prevH[T][k];
            }
            const float o = fmaxf(fmaxf(pv, lo), h[k]);                    // rows 2p-1, 2p, 2p+1 (lane half 1)
            if constexpr (BF16) {
              if (k & 1) {
                typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                bf16x2 pk;
                pk[0] = (__bf16)h[k - 1]; pk[1] = (__bf16)h[k];
                prevH[T][k >> 1] = __builtin_bit_cast(float, pk);
              }
            } else {
              prevH[T][k] = h[k];
            }
            const int c = 8 * T + k;
            if (emit && lh == 1 && (!RAGGED || c < a.Wp)) {
              const long long e = a.out_off + (long long)f * a.out_frame + (long long)p * a.out_row + (long long)c * a.out_px + 32 * hN + l31;
              if constexpr (BF16) reinterpret_cast<__bf16*>(outp)[e] = (__bf16)o;
              else reinterpret_cast<float*>(outp)[e] = o;
            }
          }
        }
      }
    }
    __syncthreads();                                 // every wave is done reading groups p .. p+2
    if (act) store_group(p + 3);                     // overwrites group p
    __syncthreads();
  }
}

template <int NT, int CH32, int CH16, bool RAGGED>
static int launch_stem(const stem_args& a, bool bf16, size_t lds, hipStream_t st) {
  const dim3 grid((a.total + 3) / 4), block(512);
  if (bf16) {
    auto k = stem_pool_kernel<NT, CH16, RAGGED, true>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(k, grid, block, lds, st, a);
  } else {
    auto k = stem_pool_kernel<NT, CH32, RAGGED, false>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(k, grid, block, lds, st, a);
  }
  return (int)hipGetLastError();
}

// Geometry supported by the fused front: NT = ceil(Ws / 16) in {3, 8, 9} (84x84, 144x256, 288x288 and anything
// with the same tile count), W % 4 == 0, W <= 32 * NT.  Returns 1 if supported (host logic, no launch).
extern "C" int cadre_stem_pool_supported(int32_t H, int32_t W) {
  if (H < 7 || W < 7 || (W & 3)) return 0;
  const int Ws = (W + 6 - 7) / 2 + 1;
  const int NT = (Ws + 15) / 16;
  if (!(NT == 3 || NT == 8 || NT == 9)) return 0;
  if (W > 32 * NT) return 0;
  return 1;
}

extern "C" int cadre_stem_pool(const uint32_t* img, const void* wt, const float* scale, const float* shift,
                               void* out, int32_t F, int32_t H, int32_t W, int32_t bf16,
                               int64_t out_frame, int64_t out_row, int32_t out_px, int64_t out_off, void* stream) {
  if (!img || !wt || !scale || !shift || !out || F < 1) return cadre_fail("cadre_stem_pool: bad argument");
  if (!cadre_stem_pool_supported(H, W)) return cadre_fail("cadre_stem_pool: unsupported geometry (see cadre_stem_pool_supported)");
  if (out_px < 64 || ((uintptr_t)img & 15) || ((uintptr_t)wt & 15)) return cadre_fail("cadre_stem_pool: bad output stride / alignment");
  stem_args a;
  a.img = img; a.wt = wt; a.scale = scale; a.shift = shift; a.out = out;
  a.F = F; a.H = H; a.W = W;
  a.Hs = (H + 6 - 7) / 2 + 1; a.Ws = (W + 6 - 7) / 2 + 1;
  a.Hp = (a.Hs + 2 - 3) / 2 + 1; a.Wp = (a.Ws + 2 - 3) / 2 + 1;
  const int NT = (a.Ws + 15) / 16;
  // bands: enough wave units to fill 256 CUs x 4 waves; a band start costs one extra iteration
  int NB = (1024 + F - 1) / F;
  if (NB < 1) NB = 1;
  if (NB > a.Hp) NB = a.Hp;
  a.PB = (a.Hp + NB - 1) / NB;
  a.NB = (a.Hp + a.PB - 1) / a.PB;
  if ((long long)F * a.NB > 0x7fffffffLL) return cadre_fail("cadre_stem_pool: too many frames");
  a.total = F * a.NB;
  int PP = (W + 8 + 1) / 2;
  if (PP < 16 * NT + 4) PP = 16 * NT + 4;
  PP = ((PP + 3) / 8) * 8 + 4;                        // == 4 (mod 8): the two stem rows of a tile read disjoint banks
  a.PP = PP;
  a.out_frame = out_frame; a.out_row = out_row; a.out_px = out_px; a.out_off = out_off;
  const int wrow = bf16 ? SP_WP16 / 2 : SP_WP32;
  const size_t lds = (size_t)64 * wrow * 4 + (size_t)4 * SP_RING * 2 * PP * (bf16 ? 8 : 4);
  if (lds > 160 * 1024) return cadre_fail("cadre_stem_pool: frame too wide for the LDS ring");
  hipStream_t st = (hipStream_t)stream;
  const bool ragged = (a.Ws % 16) != 0 || (a.Hs & 1) || a.Wp * 2 != a.Ws;
  // tiles per MFMA chunk (CH): fp32 1 (230 VGPRs, no spill at 2 waves per SIMD), bf16 3 / 2 (B fragment shared by the chunk)
  static const int ch9 = [] { const char* e = getenv("CADRE_STEM_CH"); return e ? atoi(e) : 3; }();     // A/B knob
  if (NT == 9 && ch9 == 1) return ragged ? launch_stem<9, 1, 3, true>(a, bf16 != 0, lds, st) : launch_stem<9, 1, 3, false>(a, bf16 != 0, lds, st);
  if (NT == 9) return ragged ? launch_stem<9, 3, 3, true>(a, bf16 != 0, lds, st) : launch_stem<9, 3, 3, false>(a, bf16 != 0, lds, st);
  if (NT == 8) return ragged ? launch_stem<8, 1, 2, true>(a, bf16 != 0, lds, st) : launch_stem<8, 1, 2, false>(a, bf16 != 0, lds, st);
  return launch_stem<3, 1, 3, true>(a, bf16 != 0, lds, st);
}
