// stem_pool.hip — the front of the DANet encoder as ONE kernel for gfx950:
//   u8 observation (packed RGB + route, cadre_pack_obs) -> /255 LUT (agent.py:46) -> 7x7/s2 conv +
//   folded eval-BatchNorm + ReLU -> 3x3/s2 max-pool (resnet.py:111-115, 168-172) -> NHWC pooled map.
// Replaces preprocess (1.36 GB f32 write at 1024 x 288x288), the stem conv's 5.4 GB output and the
// max-pool's re-read of it: HBM traffic of the front drops from ~14 GB to ~1.7 GB per 1024 frames, and the
// stem runs its MFMAs with K = 200 (49 taps x 4 channels, padded by ONE tap) instead of 224.
//
// Work decomposition (no workgroup barrier after start-up, no cross-wave traffic):
//   one WAVE owns (frame, band of pooled rows) and walks down the band one pooled row per iteration.
//   Iteration p computes stem rows 2p, 2p+1 as NT MFMA tiles of "2 stem rows x 16 stem columns" x 64
//   channels (v_mfma_f32_32x32x2_f32 / v_mfma_f32_32x32x16_bf16).  Tile row i = (dy, dx) with
//   dy = (i>>2)&1, dx = (i&3) + 4*(i>>3): in the accumulator layout (row = (reg&3) + 8*(reg>>2) + 4*(lane>>5),
//   col = lane&31) lane half h then holds stem row 2p+h, columns x0 .. x0+15 in its 16 registers — the
//   horizontal 3-max is register-local, the vertical one a lane-half exchange (v_permlane32_swap) plus the
//   previous iteration's row kept in registers.  The A operand is formed from a per-wave LDS ring of 16 input
//   rows holding the PACKED u8 pixels (even / odd pixel planes: stride-2 taps become unit stride, bank-conflict
//   free); each fragment element goes through the 256-entry /255 LUT in LDS, so the values entering the MFMA
//   are bit-identical to the reference's float32(rgb / 255.).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/cadre_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

int cadre_fail(const char* msg);

#define SP_RING 16        // input rows in the ring (4 groups of 4)
#define SP_TAPS32 50      // fp32: 49 taps + 1 zero tap  -> K = 200
#define SP_WP32 204       // fp32 weight row pitch in floats ((pitch/4) odd: conflict-free ds_read_b128 over rows)
#define SP_TAPS16 52      // bf16: 49 taps + 3 zero taps -> K = 208 (13 k-steps of 16)
#define SP_WP16 216       // bf16 weight row pitch in bf16 elements (432 B: (pitch_bytes/16) odd)

struct stem_args {
  const uint32_t* img;    // [F][H][W] packed pixels: R | G<<8 | B<<16 | route<<24 (route byte 0 or 255)
  const void* wt;         // fp32 [64][50][4] or bf16 [64][52][4]: tap-major (tap = ky*7 + kx), zero padded
  const float* scale;     // folded BN, [64]
  const float* shift;
  const float* lut;       // float32(i / 255.), [256]
  void* out;              // pooled map
  int F, H, W, Hs, Ws, Hp, Wp;
  int PB, NB;             // pooled rows per band, bands per frame
  int PP;                 // ring plane pitch in dwords
  int total;              // F * NB wave units
  long long out_frame, out_row;      // output strides in elements: frame, pooled row
  int out_px;                        // pooled pixel stride in elements (>= 64)
  long long out_off;                 // element offset of pooled (0, 0, 0, ch 0)
};

__device__ __forceinline__ float lo_to_hi(float v) {     // lanes 32..63 receive the value of lane - 32
  const unsigned u = __builtin_bit_cast(unsigned, v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __builtin_bit_cast(float, r[0]);
}

template <int NT, int CH, bool RAGGED, bool BF16>
__global__ __launch_bounds__(256, 1) void stem_pool_kernel(stem_args a) {
  static_assert(NT % CH == 0, "tiles per row pair must split into whole chunks");
  constexpr int NLD = (NT + 1) / 2;                 // 16-B chunk loads per lane per 4-row group (W <= 32*NT)
  constexpr int WROW = BF16 ? SP_WP16 / 2 : SP_WP32;   // weight row pitch in dwords
  constexpr int NK = BF16 ? SP_TAPS16 / 4 : SP_TAPS32 / 2;   // k-steps: bf16 4 taps (16 k), fp32 2 taps (8 k = 4 MFMAs)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* wts = smem;                                // [64][WROW] dwords
  float* lut = smem + 64 * WROW;                    // [256]
  uint32_t* ring0 = reinterpret_cast<uint32_t*>(lut + 256);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  {
    constexpr int CPR = (BF16 ? SP_TAPS16 * 8 : SP_TAPS32 * 16) / 16;    // 16-B chunks per weight row
    const char* src = reinterpret_cast<const char*>(a.wt);
    for (int i = tid; i < 64 * CPR; i += 256) {
      const int n = i / CPR, c = i - n * CPR;
      *reinterpret_cast<f32x4*>(wts + n * WROW + c * 4) = *reinterpret_cast<const f32x4*>(src + ((size_t)n * CPR + c) * 16);
    }
    lut[tid] = a.lut[tid];
  }
  __syncthreads();
  const int u = blockIdx.x * 4 + wave;
  if (u >= a.total) return;                          // (no barrier below)
  const int f = u / a.NB, band = u - f * a.NB;
  const int p0 = band * a.PB, p1 = min(a.Hp, p0 + a.PB);
  const int PP = a.PP, RP = 2 * PP;
  uint32_t* ring = ring0 + wave * (SP_RING * RP);
  for (int i = lane; i < SP_RING * RP; i += 64) ring[i] = 0u;      // pads (3 px left, >= 5 right) stay zero
  const uint32_t* frame = a.img + (size_t)f * a.H * a.W;
  const int W4 = a.W >> 2;

  u32x4 pre[NLD];
  // group g = input rows 4g-3 .. 4g (contiguous in memory); rows outside the image are zeros (conv padding)
  auto load_group = [&](int g) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int c = i * 64 + lane;
      const int j = c / W4, r = 4 * g - 3 + j;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (c < a.W && r >= 0 && r < a.H) v = *reinterpret_cast<const u32x4*>(frame + (size_t)r * a.W + (c - j * W4) * 4);
      pre[i] = v;
    }
  };
  // ring pixel index q = x + 3: plane q & 1, index q >> 1.  A 4-pixel chunk at x (x % 4 == 0) lands as
  // plane 1 [x/2+1, x/2+2] <- px x, x+2 and plane 0 [x/2+2, x/2+3] <- px x+1, x+3.
  auto store_group = [&](int g) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int c = i * 64 + lane;
      if (c < a.W) {
        const int j = c / W4, hx = (c - j * W4) * 2;               // hx = x / 2
        uint32_t* row = ring + ((4 * g + j) & (SP_RING - 1)) * RP;
        row[PP + hx + 1] = pre[i][0];
        row[hx + 2] = pre[i][1];
        row[PP + hx + 2] = pre[i][2];
        row[hx + 3] = pre[i][3];
      }
    }
  };

  const int pstart = p0 > 0 ? p0 - 1 : 0;           // a band that does not start at the top first rebuilds stem row 2*p0-1
  for (int g = pstart; g < pstart + 3; ++g) { load_group(g); store_group(g); }

  float sc[2], sh[2];
#pragma unroll
  for (int hN = 0; hN < 2; ++hN) { sc[hN] = a.scale[32 * hN + l31]; sh[hN] = a.shift[32 * hN + l31]; }
  float prevH[NT][2][8];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int hN = 0; hN < 2; ++hN)
#pragma unroll
      for (int k = 0; k < 8; ++k) prevH[t][hN][k] = 0.f;           // post-ReLU values are >= 0: 0 is the pool's -inf

  const int dy = (l31 >> 2) & 1, dx = (l31 & 3) + 4 * (l31 >> 3);
  char* outp = reinterpret_cast<char*>(a.out);

  for (int p = pstart; p < p1; ++p) {
    load_group(p + 3);                               // prefetch the next iteration's 4 new rows
    const int s4p = (4 * p) & (SP_RING - 1);
    const bool emit = p >= p0;
    float carry[2] = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NT / CH; ++j) {
      f32x16 acc[CH][2];
#pragma unroll
      for (int t = 0; t < CH; ++t)
#pragma unroll
        for (int hN = 0; hN < 2; ++hN)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[t][hN][r] = 0.f;
      if constexpr (!BF16) {
#pragma unroll
        for (int q = 0; q < NK; ++q) {               // k-quad: taps 2q (lane half 0) and 2q+1 (lane half 1)
          const int tap = 2 * q + lh;
          const int ky = tap / 7, kx = tap - ky * 7;                     // tap 49: ky 7, kx 0, zero weights
          const uint32_t* rp = ring + ((s4p + 2 * dy + ky) & (SP_RING - 1)) * RP + (kx & 1) * PP + (j * CH * 16 + dx + (kx >> 1));
          uint32_t px[CH];
#pragma unroll
          for (int t = 0; t < CH; ++t) px[t] = rp[16 * t];
          f32x4 b[2];
#pragma unroll
          for (int hN = 0; hN < 2; ++hN) b[hN] = *reinterpret_cast<const f32x4*>(wts + (32 * hN + l31) * WROW + tap * 4);
#pragma unroll
          for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int t = 0; t < CH; ++t) {
              const float af = lut[(px[t] >> (8 * s)) & 255u];
#pragma unroll
              for (int hN = 0; hN < 2; ++hN) acc[t][hN] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, b[hN][s], acc[t][hN], 0, 0, 0);
            }
          }
        }
      } else {
#pragma unroll
        for (int q = 0; q < NK; ++q) {               // k-step: taps 4q + 2*half, 4q + 2*half + 1 (8 bf16 per lane)
          uint32_t px[CH][2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int tap = 4 * q + 2 * lh + e;
            const int ky = tap / 7, kx = tap - ky * 7;                   // taps 49..51: ky 7, zero weights
            const uint32_t* rp = ring + ((s4p + 2 * dy + ky) & (SP_RING - 1)) * RP + (kx & 1) * PP + (j * CH * 16 + dx + (kx >> 1));
#pragma unroll
            for (int t = 0; t < CH; ++t) px[t][e] = rp[16 * t];
          }
          bf16x8 b[2];
#pragma unroll
          for (int hN = 0; hN < 2; ++hN)
            b[hN] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(wts + (32 * hN + l31) * WROW + (4 * q + 2 * lh) * 2));
#pragma unroll
          for (int t = 0; t < CH; ++t) {
            bf16x8 af;
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
              for (int s = 0; s < 4; ++s) af[4 * e + s] = (__bf16)lut[(px[t][e] >> (8 * s)) & 255u];
#pragma unroll
            for (int hN = 0; hN < 2; ++hN) acc[t][hN] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b[hN], acc[t][hN], 0, 0, 0);
          }
        }
      }
      // ---- chunk epilogue: BN + ReLU, horizontal 3-max in registers, vertical 3-max across lane halves + previous row
#pragma unroll
      for (int t = 0; t < CH; ++t) {
        const int T = j * CH + t;
#pragma unroll
        for (int hN = 0; hN < 2; ++hN) {
          float y[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = fmaxf(acc[t][hN][r] * sc[hN] + sh[hN], 0.f);
            if constexpr (RAGGED) {
              if (16 * T + r >= a.Ws || 2 * p + lh >= a.Hs) v = 0.f;       // outside the stem map: pool padding
            }
            y[r] = v;
          }
          float h[8];
          h[0] = fmaxf(fmaxf(carry[hN], y[0]), y[1]);
#pragma unroll
          for (int k = 1; k < 8; ++k) h[k] = fmaxf(fmaxf(y[2 * k - 1], y[2 * k]), y[2 * k + 1]);
          carry[hN] = y[15];
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float lo = lo_to_hi(h[k]);                               // stem row 2p (held by lane half 0)
            const float o = fmaxf(fmaxf(prevH[T][hN][k], lo), h[k]);       // rows 2p-1, 2p, 2p+1 (lane half 1)
            prevH[T][hN][k] = h[k];
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
    store_group(p + 3);
  }
}

template <int NT, int CH, bool RAGGED>
static int launch_stem(const stem_args& a, bool bf16, size_t lds, hipStream_t st) {
  const dim3 grid((a.total + 3) / 4), block(256);
  if (bf16) {
    auto k = stem_pool_kernel<NT, CH, RAGGED, true>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(k, grid, block, lds, st, a);
  } else {
    auto k = stem_pool_kernel<NT, CH, RAGGED, false>;
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
                               const float* lut255, void* out, int32_t F, int32_t H, int32_t W, int32_t bf16,
                               int64_t out_frame, int64_t out_row, int32_t out_px, int64_t out_off, void* stream) {
  if (!img || !wt || !scale || !shift || !lut255 || !out || F < 1) return cadre_fail("cadre_stem_pool: bad argument");
  if (!cadre_stem_pool_supported(H, W)) return cadre_fail("cadre_stem_pool: unsupported geometry (see cadre_stem_pool_supported)");
  if (out_px < 64 || ((uintptr_t)img & 15) || ((uintptr_t)wt & 15)) return cadre_fail("cadre_stem_pool: bad output stride / alignment");
  stem_args a;
  a.img = img; a.wt = wt; a.scale = scale; a.shift = shift; a.lut = lut255; a.out = out;
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
  const size_t lds = (size_t)(64 * wrow + 256) * 4 + (size_t)4 * SP_RING * 2 * PP * 4;
  if (lds > 160 * 1024) return cadre_fail("cadre_stem_pool: frame too wide for the LDS ring");
  hipStream_t st = (hipStream_t)stream;
  const bool ragged = (a.Ws % 16) != 0 || (a.Hs & 1) || a.Wp * 2 != a.Ws;
  if (NT == 9) return ragged ? launch_stem<9, 3, true>(a, bf16 != 0, lds, st) : launch_stem<9, 3, false>(a, bf16 != 0, lds, st);
  if (NT == 8) return ragged ? launch_stem<8, 4, true>(a, bf16 != 0, lds, st) : launch_stem<8, 4, false>(a, bf16 != 0, lds, st);
  return launch_stem<3, 3, true>(a, bf16 != 0, lds, st);
}
