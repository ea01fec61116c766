// cadre_kernels.hip — the non-GEMM kernels of the Cadre PPO learner hot path for gfx950.
// HBM-bound pointwise / small-reduction work: coalesced NHWC access, LDS staging per frame,
// wave64 shuffle reductions.  Reference citations (paths under /root/reference) are on each
// entry point in include/cadre_hip.h.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include "../../include/cadre_hip.h"
#ifdef CADRE_AB_KERNELS
#include "../../include/cadre_hip_ab.h"
#endif

thread_local char g_cadre_err[256] = {0};
int cadre_fail(const char* msg) {
  strncpy(g_cadre_err, msg, sizeof(g_cadre_err) - 1);
  return -1;
}
extern "C" int cadre_abi_version(void) { return CADRE_ABI_VERSION; }
extern "C" const char* cadre_last_error(void) { return g_cadre_err; }

#define ST(s) ((hipStream_t)(s))
#define FAIL_IF(cond, msg) \
  if (cond) return cadre_fail(msg)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// Device-side fill used instead of hipMemsetAsync: every launch function here may be captured into a hipGraph, and a
// memset NODE was observed to replay a 0xD3 byte pattern instead of zeros on ROCm 7.2 (act() replays encoded the new
// frame with garbage route maxima; the update graph's loss sums likewise).
__global__ void zero_u32_kernel(uint32_t* p, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = 0u;
}
static inline void zero_words(void* p, int64_t n_words, hipStream_t st) {
  const int blocks = (int)std::min<int64_t>((n_words + 255) / 256, 64);
  hipLaunchKernelGGL(zero_u32_kernel, dim3(blocks), dim3(256), 0, st, (uint32_t*)p, n_words);
}

// ============================================================================ pre_process
__global__ void route_max_kernel(const uint8_t* route, uint32_t* frame_max, int per_frame, const int64_t* frame_idx = nullptr) {
  const int f = blockIdx.y;
  const uint8_t* r = route + (frame_idx ? frame_idx[f] : (int64_t)f) * per_frame;
  uint32_t m = 0;
  const int stride = gridDim.x * blockDim.x, t0 = blockIdx.x * blockDim.x + threadIdx.x;
  if ((((uintptr_t)r) & 15) == 0) {                       // 16 bytes per lane
    const uint4* r4 = reinterpret_cast<const uint4*>(r);
    const int n4 = per_frame >> 4;
    for (int i = t0; i < n4; i += stride) {
      const uint4 v = r4[i];
      uint32_t w = v.x | v.y | v.z | v.w;                 // cheap pre-test: all-zero words skip the byte max
      if (w) {
        const uint32_t ws[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int b = 0; b < 4; ++b) m = max(m, (ws[k] >> (8 * b)) & 0xffu);
      }
    }
    for (int i = (n4 << 4) + t0; i < per_frame; i += stride) m = max(m, (uint32_t)r[i]);
  } else {
    for (int i = t0; i < per_frame; i += stride) m = max(m, (uint32_t)r[i]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
  if ((threadIdx.x & 63) == 0 && m) atomicMax(frame_max + f, m);
}

// One workgroup = a 32(h) x 32(w) pixel tile of one frame.  The route plane is stored [W][H] (h
// fastest), the output NHWC (w fastest): the tile is read coalesced along h, transposed through LDS,
// and written coalesced along w together with the LUT-converted RGB.
__global__ __launch_bounds__(256) void preprocess_kernel(const uint8_t* rgb, const uint8_t* route, const float* lut,
                                                         const uint32_t* frame_max, float* out, uint8_t* route_norm,
                                                         int F, int H, int W) {
  __shared__ float s_lut[256];
  __shared__ uint8_t s_r[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
  s_lut[threadIdx.x] = lut[threadIdx.x];
  const int f = blockIdx.z, h0 = blockIdx.y * 32, w0 = blockIdx.x * 32;
  const uint32_t mx = frame_max[f];
#pragma unroll
  for (int j = 0; j < 4; ++j) {                                // read route[f][w0+wl][h0+tx], wl = ty + 8j
    const int wl = ty + 8 * j, w = w0 + wl, h = h0 + tx;
    uint8_t rn = 0;
    if (w < W && h < H) {
      const int64_t ridx = ((int64_t)f * W + w) * H + h;
      const uint8_t rv = route[ridx];
      // agent.py:51-54: route[i] = 1.0*route[i]/max stored into uint8 -> truncates to {0,1}
      rn = mx > 0 ? (uint8_t)(rv == mx ? 1 : 0) : rv;
      if (route_norm) route_norm[ridx] = rn;
    }
    s_r[wl][tx] = rn;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {                                // write out[f][h0+hl][w0+tx][0..3], hl = ty + 8j
    const int hl = ty + 8 * j, h = h0 + hl, w = w0 + tx;
    if (h < H && w < W) {
      const int64_t i = ((int64_t)f * H + h) * W + w;
      const uint8_t* px = rgb + i * 3;
      float4 o;
      o.x = s_lut[px[0]];
      o.y = s_lut[px[1]];
      o.z = s_lut[px[2]];
      o.w = (float)s_r[tx][hl];
      reinterpret_cast<float4*>(out)[i] = o;
    }
  }
}

// bf16 variant for the C3 stem: writes the interior of a zero-padded NHWC4 bf16 image
// [F][Hp][Wp][4] at (pad_t, pad_l); the border is never written (allocated zeroed by the caller).
__global__ __launch_bounds__(256) void preprocess_bf16pad_kernel(const uint8_t* rgb, const uint8_t* route, const float* lut,
                                                                 const uint32_t* frame_max, __bf16* out, uint8_t* route_norm,
                                                                 int F, int H, int W, int Hp, int Wp, int pad_t, int pad_l) {
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  __shared__ float s_lut[256];
  __shared__ uint8_t s_r[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  s_lut[threadIdx.x] = lut[threadIdx.x];
  const int f = blockIdx.z, h0 = blockIdx.y * 32, w0 = blockIdx.x * 32;
  const uint32_t mx = frame_max[f];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int wl = ty + 8 * j, w = w0 + wl, h = h0 + tx;
    uint8_t rn = 0;
    if (w < W && h < H) {
      const int64_t ridx = ((int64_t)f * W + w) * H + h;
      const uint8_t rv = route[ridx];
      rn = mx > 0 ? (uint8_t)(rv == mx ? 1 : 0) : rv;
      if (route_norm) route_norm[ridx] = rn;
    }
    s_r[wl][tx] = rn;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int hl = ty + 8 * j, h = h0 + hl, w = w0 + tx;
    if (h < H && w < W) {
      const uint8_t* px = rgb + (((int64_t)f * H + h) * W + w) * 3;
      bf16x4 o;
      o[0] = (__bf16)s_lut[px[0]]; o[1] = (__bf16)s_lut[px[1]]; o[2] = (__bf16)s_lut[px[2]]; o[3] = (__bf16)(float)s_r[tx][hl];
      reinterpret_cast<bf16x4*>(out)[((int64_t)f * Hp + h + pad_t) * Wp + w + pad_l] = o;
    }
  }
}
extern "C" int cadre_preprocess_bf16pad(const uint8_t* rgb, const uint8_t* route, const float* lut255, void* out,
                                        uint8_t* route_norm, uint32_t* frame_max, int32_t F, int32_t H, int32_t W,
                                        int32_t Hp, int32_t Wp, int32_t pad_t, int32_t pad_l, void* stream) {
  FAIL_IF(!rgb || !route || !lut255 || !out || !frame_max || F < 1 || H < 1 || W < 1 || pad_t < 0 || pad_l < 0 ||
              Hp < H + pad_t || Wp < W + pad_l,
          "cadre_preprocess_bf16pad: bad argument");
  zero_words(frame_max, F, ST(stream));
  const int per = H * W;
  dim3 g1(min(64, (per + 255) / 256), F);
  hipLaunchKernelGGL(route_max_kernel, g1, dim3(256), 0, ST(stream), route, frame_max, per);
  hipLaunchKernelGGL(preprocess_bf16pad_kernel, dim3((W + 31) / 32, (H + 31) / 32, F), dim3(256), 0, ST(stream), rgb,
                     route, lut255, frame_max, (__bf16*)out, route_norm, F, H, W, Hp, Wp, pad_t, pad_l);
  return (int)hipGetLastError();
}

// Packed variant for the fused front (stem_pool.hip): one dword per pixel, R | G<<8 | B<<16 | route<<24 with the
// normalised route {0,1} stored as byte 0 / 255, so that EVERY byte goes through the same /255 LUT
// (LUT[255] = float32(255/255.) = 1.0 exactly).  4x less HBM than the f32 NHWC4 image.
__global__ __launch_bounds__(256) void pack_obs_kernel(const uint8_t* rgb, const uint8_t* route, const uint32_t* frame_max,
                                                       uint32_t* out, uint8_t* route_norm, int F, int H, int W,
                                                       const int64_t* frame_idx) {
  __shared__ uint8_t s_r[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int fo = blockIdx.z, h0 = blockIdx.y * 32, w0 = blockIdx.x * 32;
  const int64_t f = frame_idx ? frame_idx[fo] : fo;       // source frame (sliding windows repeat frames)
  const uint32_t mx = frame_max[f];                      // (per source frame: a window repeats each frame 8 times)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int wl = ty + 8 * j, w = w0 + wl, h = h0 + tx;
    uint8_t rn = 0;
    if (w < W && h < H) {
      const int64_t ridx = ((int64_t)f * W + w) * H + h;
      const uint8_t rv = route[ridx];
      rn = mx > 0 ? (uint8_t)(rv == mx ? 1 : 0) : rv;              // agent.py:51-54 (uint8 truncation quirk)
      if (route_norm) route_norm[ridx] = rn;
    }
    s_r[wl][tx] = rn;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int hl = ty + 8 * j, h = h0 + hl, w = w0 + tx;
    if (h < H && w < W) {
      const uint8_t* px = rgb + (((int64_t)f * H + h) * W + w) * 3;
      out[((int64_t)fo * H + h) * W + w] = (uint32_t)px[0] | ((uint32_t)px[1] << 8) | ((uint32_t)px[2] << 16) | (s_r[tx][hl] ? 0xff000000u : 0u);
    }
  }
}
// Four pixels per thread (W % 4 == 0): 12 RGB bytes as three aligned dwords, one 16-byte store; a workgroup covers a
// 32 x 32 tile in one pass (the byte-per-lane loads of the kernel above cost 259 us per 1024 frames at 288^2).
__global__ __launch_bounds__(256) void pack_obs4_kernel(const uint8_t* rgb, const uint8_t* route, const uint32_t* frame_max,
                                                        uint32_t* out, uint8_t* route_norm, int F, int H, int W,
                                                        const int64_t* frame_idx) {
  __shared__ uint8_t s_r[32][36];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int fo = blockIdx.z, h0 = blockIdx.y * 32, w0 = blockIdx.x * 32;
  const int64_t f = frame_idx ? frame_idx[fo] : fo;
  const uint32_t mx = frame_max[f];                      // (per source frame: a window repeats each frame 8 times)
  // (round 6) the pixel group's three dwords are requested BEFORE the route tile goes through LDS: one memory round trip of the
  // workgroup's dependent chain (frame index -> route bytes -> barrier -> pixels -> store) less — the kernel is 83 k small
  // workgroups per 1024 frames and paced by that chain, not by bytes
  const int hl = threadIdx.x >> 3, wq = (threadIdx.x & 7) * 4;      // 32 rows x 8 groups of 4 pixels
  const int h = h0 + hl, w = w0 + wq;
  const bool live = h < H && w < W;                      // (W % 4 == 0: a group is all in or all out)
  uint32_t a = 0, b = 0, c = 0;                          // R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3
  if (live) {
    const uint32_t* px = reinterpret_cast<const uint32_t*>(rgb + (((int64_t)f * H + h) * W + w) * 3);
    a = px[0]; b = px[1]; c = px[2];
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {                          // route tile [w][h] -> LDS (stored transposed in memory)
    const int wl = ty + 8 * j, w = w0 + wl, h = h0 + tx;
    uint8_t rn = 0;
    if (w < W && h < H) {
      const int64_t ridx = ((int64_t)f * W + w) * H + h;
      const uint8_t rv = route[ridx];
      rn = mx > 0 ? (uint8_t)(rv == mx ? 1 : 0) : rv;              // agent.py:51-54 (uint8 truncation quirk)
      if (route_norm) route_norm[ridx] = rn;
    }
    s_r[wl][tx] = rn;
  }
  __syncthreads();
  if (live) {
    uint4 o;
    o.x = (a & 0x00ffffffu) | (s_r[wq][hl] ? 0xff000000u : 0u);
    o.y = (a >> 24) | ((b & 0x0000ffffu) << 8) | (s_r[wq + 1][hl] ? 0xff000000u : 0u);
    o.z = (b >> 16) | ((c & 0x000000ffu) << 16) | (s_r[wq + 2][hl] ? 0xff000000u : 0u);
    o.w = (c >> 8) | (s_r[wq + 3][hl] ? 0xff000000u : 0u);
    *reinterpret_cast<uint4*>(out + ((int64_t)fo * H + h) * W + w) = o;
  }
}
extern "C" int cadre_pack_obs(const uint8_t* rgb, const uint8_t* route, uint32_t* out, uint8_t* route_norm,
                              uint32_t* frame_max, int32_t F, int32_t H, int32_t W, const int64_t* frame_idx, int32_t n_src,
                              void* stream) {
  FAIL_IF(!rgb || !route || !out || !frame_max || F < 1 || H < 1 || W < 1 || (frame_idx && n_src < 1), "cadre_pack_obs: bad argument");
  const int nmax = frame_idx ? n_src : F;                 // route maxima per SOURCE frame (frame_max holds that many)
  zero_words(frame_max, nmax, ST(stream));
  const int per = H * W;
  dim3 g1(min(64, (per + 255) / 256), nmax);
  hipLaunchKernelGGL(route_max_kernel, g1, dim3(256), 0, ST(stream), route, frame_max, per, (const int64_t*)nullptr);
  if (W % 4 == 0 && ((uintptr_t)rgb & 3) == 0 && ((uintptr_t)out & 15) == 0)
    hipLaunchKernelGGL(pack_obs4_kernel, dim3((W + 31) / 32, (H + 31) / 32, F), dim3(256), 0, ST(stream), rgb, route,
                       frame_max, out, route_norm, F, H, W, frame_idx);
  else
    hipLaunchKernelGGL(pack_obs_kernel, dim3((W + 31) / 32, (H + 31) / 32, F), dim3(256), 0, ST(stream), rgb, route,
                       frame_max, out, route_norm, F, H, W, frame_idx);
  return (int)hipGetLastError();
}

extern "C" int cadre_preprocess(const uint8_t* rgb, const uint8_t* route, const float* lut255,
                                float* out, uint8_t* route_norm, uint32_t* frame_max,
                                int32_t F, int32_t H, int32_t W, void* stream) {
  FAIL_IF(!rgb || !route || !lut255 || !out || !frame_max || F < 1 || H < 1 || W < 1,
          "cadre_preprocess: bad argument");
  zero_words(frame_max, F, ST(stream));
  const int per = H * W;
  dim3 g1(min(64, (per + 255) / 256), F);
  hipLaunchKernelGGL(route_max_kernel, g1, dim3(256), 0, ST(stream), route, frame_max, per);
  hipLaunchKernelGGL(preprocess_kernel, dim3((W + 31) / 32, (H + 31) / 32, F), dim3(256), 0, ST(stream), rgb, route,
                     lut255, frame_max, out, route_norm, F, H, W);
  return (int)hipGetLastError();
}

// ============================================================================ maxpool 3x3 s2 p1
// One workgroup per MP_ROWS consecutive output rows of a frame: the frame/row decode is scalar, a lane owns
// one 16-byte channel group of one output pixel.  HBM-bound (stem output read once + a quarter written:
// 6.8 GB in 1.4 ms at 288x288 x 1024 frames); MP_ROWS = 4 (row reuse inside a workgroup) measured no better.
#define MP_ROWS 1
template <typename VEC, int NE, typename OP>
__device__ __forceinline__ void maxpool_row(const VEC* x, VEC* y, int H, int W, int CV, int Ho, int Wo, OP vmax) {
  const int groups = (Ho + MP_ROWS - 1) / MP_ROWS;
  const int f = blockIdx.x / groups, hg = blockIdx.x % groups;
  for (int ho = hg * MP_ROWS; ho < min(Ho, (hg + 1) * MP_ROWS); ++ho) {
    const int row = f * Ho + ho;
    for (int idx = threadIdx.x; idx < Wo * CV; idx += blockDim.x) {
      const int wo = idx / CV, c = idx % CV;
      float m[NE];
#pragma unroll
      for (int e = 0; e < NE; ++e) m[e] = -INFINITY;
#pragma unroll
      for (int dh = 0; dh < 3; ++dh) {
        const int hi = ho * 2 - 1 + dh;
        if ((unsigned)hi >= (unsigned)H) continue;
        const VEC* xr = x + ((int64_t)f * H + hi) * W * CV;
#pragma unroll
        for (int dw = 0; dw < 3; ++dw) {
          const int wi = wo * 2 - 1 + dw;
          if ((unsigned)wi >= (unsigned)W) continue;
          vmax(m, xr[wi * CV + c]);
        }
      }
      VEC o;
#pragma unroll
      for (int e = 0; e < NE; ++e) o[e] = (decltype(o[0] + o[0]))m[e];
      y[(int64_t)row * Wo * CV + idx] = o;
    }
  }
}

typedef float mp_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 mp_bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void maxpool_kernel(const float* x, float* y, int H, int W, int C4, int Ho, int Wo) {
  maxpool_row<mp_f32x4, 4>(reinterpret_cast<const mp_f32x4*>(x), reinterpret_cast<mp_f32x4*>(y), H, W, C4, Ho, Wo,
                           [](float* m, const mp_f32x4& v) {
#pragma unroll
                             for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[e]);
                           });
}

__global__ __launch_bounds__(256) void maxpool_bf16_kernel(const __bf16* x, __bf16* y, int H, int W, int C8, int Ho, int Wo) {
  maxpool_row<mp_bf16x8, 8>(reinterpret_cast<const mp_bf16x8*>(x), reinterpret_cast<mp_bf16x8*>(y), H, W, C8, Ho, Wo,
                            [](float* m, const mp_bf16x8& v) {
#pragma unroll
                              for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], (float)v[e]);
                            });
}
extern "C" int cadre_maxpool3x3s2_bf16(const void* x, void* y, int32_t F, int32_t H, int32_t W, int32_t C, void* stream) {
  FAIL_IF(!x || !y || F < 1 || H < 1 || W < 1 || C < 8 || (C & 7), "cadre_maxpool3x3s2_bf16: bad argument");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  FAIL_IF((int64_t)F * Ho > 0x7fffffff, "cadre_maxpool3x3s2_bf16: too many rows");
  hipLaunchKernelGGL(maxpool_bf16_kernel, dim3(F * ((Ho + MP_ROWS - 1) / MP_ROWS)), dim3(256), 0, ST(stream), (const __bf16*)x,
                     (__bf16*)y, H, W, C / 8, Ho, Wo);
  return (int)hipGetLastError();
}

extern "C" int cadre_maxpool3x3s2(const float* x, float* y, int32_t F, int32_t H, int32_t W, int32_t C,
                                  void* stream) {
  FAIL_IF(!x || !y || F < 1 || H < 1 || W < 1 || C < 4 || (C & 3), "cadre_maxpool3x3s2: bad argument");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  FAIL_IF((int64_t)F * Ho > 0x7fffffff, "cadre_maxpool3x3s2: too many rows");
  hipLaunchKernelGGL(maxpool_kernel, dim3(F * ((Ho + MP_ROWS - 1) / MP_ROWS)), dim3(256), 0, ST(stream), x, y, H, W, C / 4, Ho, Wo);
  return (int)hipGetLastError();
}

// ============================================================================ PAM (position attention)
// One workgroup (4 waves) per frame.  qkv [Np][160] = (q 16 | k 16 | v 128) from the merged 1x1-conv GEMM.
// Both products run on the matrix cores (v_mfma_f32_32x32x2_f32, exact fp32 fma chain in k order — the same
// arithmetic as a scalar k-ascending loop): energy = q k^T (da_att.py:43, 96x96 padded, K = 16) as 3x3 tiles,
// out = attention v (da_att.py:47, 96x128, K = 96) as 3x4 tiles; the row softmax in between is a wave-shuffle
// reduction over LDS.  Rows / columns >= Np are zero padding (their energies are -inf before the softmax).
#define PAM_MAXNP 128
typedef float pam_f32x16 __attribute__((ext_vector_type(16)));
// R = Np rounded up to 32 (1 .. 4 row blocks; 128 positions = 149 KB is what one CU's LDS holds): the staged rows, the LDS
// footprint (25 / 58 / 97 / 149 KB) and the tile
// counts follow the map — the reference's native 5 x 8 map (Np = 40) runs 4 + 8 tiles instead of 9 + 12 and two
// workgroups per CU.  The padding contributes exact zeros at the END of every fma chain, so the results do not depend
// on R (same bits as the fixed 96-row form).
// 8 waves per workgroup (two per SIMD): one workgroup per CU fits (97 KB of LDS at Np = 81), and with a single wave
// per SIMD every LDS round trip of the softmax and of the scalar fragment reads was exposed (PMC: 8 k VALU + 0.7 k LDS
// instructions per wave in 129 k wave cycles) — 217 -> 124 us per 1024 frames at Np = 81, 23 -> 14 us for one 5 x 8 frame.
#define PAM_THREADS 512
__global__ __launch_bounds__(PAM_THREADS) void pam_kernel(const float* qkv, const float* x, float gamma, float* y, int Np,
                                                          int out_bf16) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int QP = 17, NW = PAM_THREADS / 64;
  const int RB = (Np + 31) >> 5, R = 32 * RB, AP = R + 1;
  float* q = sm;                    // [R][17]
  float* k = q + R * QP;            // [R][17]
  float* v = k + R * QP;            // [R][128]
  float* att = v + R * 128;         // [R][R + 1]
  const int f = blockIdx.x, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  // staging: the frame's rows as 16-byte pieces (40 per row), ALL requested before the first LDS write
  const float4* src4 = reinterpret_cast<const float4*>(qkv + (int64_t)f * Np * 160);
  constexpr int NST = (PAM_MAXNP * 40 + PAM_THREADS - 1) / PAM_THREADS;
  float4 stg[NST];
#pragma unroll
  for (int j = 0; j < NST; ++j) {
    const int i = tid + PAM_THREADS * j;
    const float4 t = src4[min(i, Np * 40 - 1)];
    stg[j] = i < Np * 40 ? t : float4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int j = 0; j < NST; ++j) {
    const int i = tid + PAM_THREADS * j;
    if (i < R * 40) {
      const int n = i / 40, c4 = i - 40 * n;
      if (c4 < 8) {
        float* d = (c4 < 4 ? q : k) + n * QP + 4 * (c4 & 3);
        d[0] = stg[j].x; d[1] = stg[j].y; d[2] = stg[j].z; d[3] = stg[j].w;
      } else {
        *reinterpret_cast<float4*>(v + n * 128 + 4 * (c4 - 8)) = stg[j];
      }
    }
  }
  __syncthreads();
  // ---- energy[n][m] = q[n] . k[m]: RB x RB tiles of 32x32 over the waves, 8 MFMAs (K = 16) each
  for (int t = wave; t < RB * RB; t += NW) {
    const int n0 = (t / RB) * 32, m0 = (t % RB) * 32;
    pam_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q[(n0 + l31) * QP + 2 * kk + lh], k[(m0 + l31) * QP + 2 * kk + lh], acc, 0, 0, 0);
    const int m = m0 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) att[(n0 + (r & 3) + 8 * (r >> 2) + 4 * lh) * AP + m] = m < Np ? acc[r] : -INFINITY;
  }
  __syncthreads();
  for (int n = wave; n < R; n += NW) {               // row softmax (da_att.py:44); padded rows become zeros
    if (n >= Np) {
      for (int m = lane; m < R; m += 64) att[n * AP + m] = 0.f;
      continue;
    }
    float mx = -INFINITY;
    for (int m = lane; m < R; m += 64) mx = fmaxf(mx, att[n * AP + m]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int m = lane; m < R; m += 64) {
      const float e = expf(att[n * AP + m] - mx);    // exp(-inf) = 0 for the padded columns
      att[n * AP + m] = e;
      s += e;
    }
    s = wave_sum(s);
    for (int m = lane; m < R; m += 64) att[n * AP + m] = att[n * AP + m] / s;
  }
  __syncthreads();
  // ---- out[n][c] = sum_m att[n][m] v[m][c]: wave w owns channel block w & 3 (32 channels) and the row blocks
  // b = (w >> 2) + 2 j (the k order of every output's chain is 0 .. R-1 as before: same bits)
  const int half = wave >> 2;
  pam_f32x16 o[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[j][r] = 0.f;
  const int c = 32 * (wave & 3) + l31;
  const bool two = half + 2 < RB;
  if (half < RB) {
#pragma unroll 8
    for (int kk = 0; kk < R / 2; ++kk) {
      const float bv = v[(2 * kk + lh) * 128 + c];
      o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(att[(32 * half + l31) * AP + 2 * kk + lh], bv, o[0], 0, 0, 0);
      if (two) o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(att[(32 * (half + 2) + l31) * AP + 2 * kk + lh], bv, o[1], 0, 0, 0);
    }
  }
  const float* xf = x + (int64_t)f * Np * 128;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = 32 * (half + 2 * j) + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (n < Np) {
        const float res = gamma * o[j][r] + xf[n * 128 + c];
        if (out_bf16) reinterpret_cast<__bf16*>(y)[(int64_t)f * Np * 128 + n * 128 + c] = (__bf16)res;
        else y[(int64_t)f * Np * 128 + n * 128 + c] = res;
      }
    }
}


// ---- layer-4 maps above PAM_MAXNP positions (round 6: da_att.py:32-51 is size-free; a 384 x 384 input gives 12 x 12 = 144).  One
// workgroup per (frame, block of 32 query rows): the block's energies against ALL keys live in LDS (32 x (R + 1) floats, R = Np
// rounded up to 32: Np <= PAM_BIGNP), keys and values are read from global memory (the frame's qkv rows: L2).  Same arithmetic as the
// small kernel — MFMA fma chains in ascending k, the same strided softmax sums — at the speed of a generality path.
#define PAM_BIGNP 1024
__global__ __launch_bounds__(256) void pam_large_kernel(const float* qkv, const float* x, float gamma, float* y, int Np, int out_bf16) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int QP = 17;
  const int RB = (Np + 31) >> 5, R = 32 * RB, AP = R + 1;
  float* q = sm;                    // [32][17]
  float* att = q + 32 * QP;         // [32][R + 1]
  const int f = blockIdx.x, n0 = 32 * blockIdx.y, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const float* fq = qkv + (int64_t)f * Np * 160;
  for (int i = tid; i < 32 * 16; i += 256) {
    const int r = i >> 4, c = i & 15;
    q[r * QP + c] = n0 + r < Np ? fq[(int64_t)(n0 + r) * 160 + c] : 0.f;
  }
  __syncthreads();
  for (int mb = wave; mb < RB; mb += 4) {                   // energies of the 32 rows against key block mb
    const int m = 32 * mb + l31;
    const float* kr = fq + (int64_t)min(m, Np - 1) * 160 + 16;
    pam_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const float kv = m < Np ? kr[2 * kk + lh] : 0.f;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q[l31 * QP + 2 * kk + lh], kv, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) att[((r & 3) + 8 * (r >> 2) + 4 * lh) * AP + m] = m < Np ? acc[r] : -INFINITY;
  }
  __syncthreads();
  for (int n = wave; n < 32; n += 4) {                      // row softmax (da_att.py:44); rows past the map become zeros
    if (n0 + n >= Np) {
      for (int m = lane; m < R; m += 64) att[n * AP + m] = 0.f;
      continue;
    }
    float mx = -INFINITY;
    for (int m = lane; m < R; m += 64) mx = fmaxf(mx, att[n * AP + m]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int m = lane; m < R; m += 64) {
      const float e = expf(att[n * AP + m] - mx);
      att[n * AP + m] = e;
      sum += e;
    }
    sum = wave_sum(sum);
    for (int m = lane; m < R; m += 64) att[n * AP + m] = att[n * AP + m] / sum;
  }
  __syncthreads();
  // out[n][c] = sum_m att[n][m] v[m][c]: wave w owns channel block w (32 channels); v rows from global (rows past the map: weight 0)
  const int c = 32 * wave + l31;
  pam_f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.f;
  for (int kk = 0; kk < R / 2; ++kk) {
    const int m = 2 * kk + lh;
    const float bv = fq[(int64_t)min(m, Np - 1) * 160 + 32 + c];
    o = __builtin_amdgcn_mfma_f32_32x32x2f32(att[l31 * AP + m], bv, o, 0, 0, 0);
  }
  const float* xf = x + (int64_t)f * Np * 128;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int n = n0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    if (n < Np) {
      const float res = gamma * o[r] + xf[(int64_t)n * 128 + c];
      if (out_bf16) reinterpret_cast<__bf16*>(y)[(int64_t)f * Np * 128 + (int64_t)n * 128 + c] = (__bf16)res;
      else y[(int64_t)f * Np * 128 + (int64_t)n * 128 + c] = res;
    }
  }
}

static int pam_launch(const float* x, const float* qkv, float gamma, void* y, int32_t F, int32_t Np, int out_bf16,
                      void* stream);
extern "C" int cadre_pam(const float* x, const float* qkv, float gamma, float* y, int32_t F, int32_t Np,
                         void* stream) {
  return pam_launch(x, qkv, gamma, y, F, Np, 0, stream);
}
extern "C" int cadre_pam_bf16out(const float* x, const float* qkv, float gamma, void* y, int32_t F, int32_t Np,
                                 void* stream) {
  return pam_launch(x, qkv, gamma, y, F, Np, 1, stream);
}
static int pam_launch(const float* x, const float* qkv, float gamma, void* y, int32_t F, int32_t Np, int out_bf16,
                      void* stream) {
  FAIL_IF(!x || !qkv || !y || F < 1 || Np < 1 || Np > PAM_BIGNP, "cadre_pam: bad argument (Np<=1024)");
  // The row-block kernel at EVERY size (round 6): three workgroups of 15 KB per 9 x 9 frame instead of one of 99 KB — several frames per
  // CU in flight where the one-CU kernel ran its five barrier-separated phases frame after frame: 268 -> 167 us per 2048 frames, 128 ->
  // 87 per 1024 (tools/dbg/pam_cam_time.py), the same bits.  CADRE_PAM_LARGE=0: the one-CU kernel up to 128 positions.
  static const int pam_force_large = [] { const char* e = getenv("CADRE_PAM_LARGE"); return e ? atoi(e) : 1; }();
  if (Np > PAM_MAXNP || pam_force_large) {                 // a workgroup per block of 32 query rows
    const size_t Rb = (size_t)((Np + 31) / 32) * 32;
    const size_t shm_b = sizeof(float) * (32 * 17 + 32 * (Rb + 1));
    static bool attr_b = false;
    if (!attr_b) {
      (void)hipFuncSetAttribute((const void*)pam_large_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr_b = true;
    }
    hipLaunchKernelGGL(pam_large_kernel, dim3(F, (Np + 31) / 32), dim3(256), shm_b, ST(stream), qkv, x, gamma, (float*)y, Np, out_bf16);
    return (int)hipGetLastError();
  }
  const size_t R = (size_t)((Np + 31) / 32) * 32;
  const size_t shm = sizeof(float) * (R * 34 + R * 128 + R * (R + 1));
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)pam_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(pam_kernel, dim3(F), dim3(PAM_THREADS), shm, ST(stream), qkv, x, gamma, (float*)y, Np, out_bf16);
  return (int)hipGetLastError();
}

// ============================================================================ CAM (channel attention)
#define CAM_XP 132                // row pitch of the staged frame [Np][128] (+4: rows 14 apart land on different banks)
#define CAM_EP 132                // row pitch of the energy / attention matrix [128][128]
// Both products on the matrix cores (v_mfma_f32_32x32x2_f32: the fp32 fma chain of a scalar loop in ascending k), 8 waves
// per workgroup: energy = x^T x (da_att.py:74; 128 x 128, K = Np) as 4 x 4 tiles, out^T[n][c] = sum_d x[n][d] att[c][d]
// (:79; Np x 128, K = 128) as 3 x 4 tiles with the channel on the lanes (row-contiguous stores).  The VALU form this
// replaces ran 13 k vector instructions per wave with one wave per SIMD (PMC): 231 -> 119 us per 1024 frames at Np = 81,
// 47 -> 24 us for one 5 x 8 frame.
#define CAM_THREADS 512
__global__ __launch_bounds__(CAM_THREADS) void cam_kernel(const float* x, float gamma, float* y, int Np, int out_bf16) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int NW = CAM_THREADS / 64;
  const int RB = (Np + 31) >> 5, R = 32 * RB;
  float* xs = sm;                 // [R][CAM_XP], rows >= Np zero
  float* E = xs + R * CAM_XP;     // [128][CAM_EP]
  const int f = blockIdx.x, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const float* xf = x + (int64_t)f * Np * 128;
  {  // staging: all 16-byte pieces requested before the first LDS write (one HBM round trip per workgroup)
    constexpr int NST = (PAM_MAXNP * 32 + CAM_THREADS - 1) / CAM_THREADS;
    float4 stg[NST];
#pragma unroll
    for (int j = 0; j < NST; ++j) {
      const int i = tid + CAM_THREADS * j;
      const float4 t = reinterpret_cast<const float4*>(xf)[min(i, Np * 32 - 1)];
      stg[j] = i < Np * 32 ? t : float4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int j = 0; j < NST; ++j) {
      const int i = tid + CAM_THREADS * j;
      if (i < R * 32) *reinterpret_cast<float4*>(xs + (i >> 5) * CAM_XP + (i & 31) * 4) = stg[j];
    }
  }
  __syncthreads();
  {  // energy[c][d] = sum_n x[n][c] x[n][d]: wave w owns tile row w >> 1 and the two tile columns 2 (w & 1), + 1
    const int c0 = 32 * (wave >> 1), d0 = 64 * (wave & 1);
    pam_f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int nk = (Np + 1) >> 1;                    // (row Np is zero when Np is odd)
#pragma unroll 8
    for (int kk = 0; kk < nk; ++kk) {
      const float* row = xs + (2 * kk + lh) * CAM_XP;
      const float a = row[c0 + l31];
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, row[d0 + l31], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, row[d0 + 32 + l31], acc[1], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) E[(c0 + (r & 3) + 8 * (r >> 2) + 4 * lh) * CAM_EP + d0 + 32 * j + l31] = acc[j][r];
  }
  __syncthreads();
  for (int c = wave; c < 128; c += NW) {              // energy_new = rowmax - energy; softmax (:75-76)
    const float e0 = E[c * CAM_EP + lane], e1 = E[c * CAM_EP + lane + 64];
    const float rmax = wave_max(fmaxf(e0, e1));
    const float n0 = rmax - e0, n1 = rmax - e1;
    const float m2 = wave_max(fmaxf(n0, n1));
    const float p0 = expf(n0 - m2), p1 = expf(n1 - m2);
    const float s = wave_sum(p0 + p1);
    E[c * CAM_EP + lane] = p0 / s;
    E[c * CAM_EP + lane + 64] = p1 / s;
  }
  __syncthreads();
  // out[n][c] = sum_d x[n][d] att[c][d] (d ascending): wave w owns channel block w & 3 and position blocks (w >> 2) + 2 j
  const int half = wave >> 2, c = 32 * (wave & 3) + l31;
  const bool two = half + 2 < RB;
  pam_f32x16 o[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[j][r] = 0.f;
  if (half < RB) {
    const float* er = E + c * CAM_EP + lh;
    const float* x0 = xs + (32 * half + l31) * CAM_XP + lh;
    const float* x1 = xs + (32 * (two ? half + 2 : half) + l31) * CAM_XP + lh;
#pragma unroll 8
    for (int kk = 0; kk < 64; ++kk) {
      const float bv = er[2 * kk];
      o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0[2 * kk], bv, o[0], 0, 0, 0);
      if (two) o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1[2 * kk], bv, o[1], 0, 0, 0);
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = 32 * (half + 2 * j) + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (n < Np) {
        const float res = gamma * o[j][r] + xs[n * CAM_XP + c];
        if (out_bf16) reinterpret_cast<__bf16*>(y)[(int64_t)f * Np * 128 + n * 128 + c] = (__bf16)res;
        else y[(int64_t)f * Np * 128 + n * 128 + c] = res;
      }
    }
}


// ---- CAM for maps above PAM_MAXNP positions: the frame is read from global memory (energy: row pairs, coalesced; product: a row per
// lane), only the 128 x 128 attention matrix lives in LDS — any Np.  Same fma chains as the small kernel.
__global__ __launch_bounds__(CAM_THREADS) void cam_large_kernel(const float* x, float gamma, float* y, int Np, int out_bf16) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int NW = CAM_THREADS / 64;
  float* E = sm;                  // [128][CAM_EP]
  const int f = blockIdx.x, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const float* xf = x + (int64_t)f * Np * 128;
  {
    const int c0 = 32 * (wave >> 1), d0 = 64 * (wave & 1);
    pam_f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int nk = (Np + 1) >> 1;
    for (int kk = 0; kk < nk; ++kk) {
      const int n = 2 * kk + lh;
      const float* row = xf + (int64_t)min(n, Np - 1) * 128;
      const bool ok = n < Np;
      const float a = ok ? row[c0 + l31] : 0.f;
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, ok ? row[d0 + l31] : 0.f, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, ok ? row[d0 + 32 + l31] : 0.f, acc[1], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) E[(c0 + (r & 3) + 8 * (r >> 2) + 4 * lh) * CAM_EP + d0 + 32 * j + l31] = acc[j][r];
  }
  __syncthreads();
  for (int c = wave; c < 128; c += NW) {              // energy_new = rowmax - energy; softmax (:75-76)
    const float e0 = E[c * CAM_EP + lane], e1 = E[c * CAM_EP + lane + 64];
    const float rmax = wave_max(fmaxf(e0, e1));
    const float n0 = rmax - e0, n1 = rmax - e1;
    const float m2 = wave_max(fmaxf(n0, n1));
    const float p0 = expf(n0 - m2), p1 = expf(n1 - m2);
    const float s = wave_sum(p0 + p1);
    E[c * CAM_EP + lane] = p0 / s;
    E[c * CAM_EP + lane + 64] = p1 / s;
  }
  __syncthreads();
  // out[n][c] = sum_d x[n][d] att[c][d] (d ascending): wave w owns channel block w & 3 and the position blocks (w >> 2), + 2, + 4, ...
  const int c = 32 * (wave & 3) + l31;
  const int RB = (Np + 31) >> 5;
  const float* er = E + c * CAM_EP + lh;
  for (int nb = wave >> 2; nb < RB; nb += 2) {
    const int n_a = 32 * nb + l31;
    const float* xr = xf + (int64_t)min(n_a, Np - 1) * 128 + lh;
    pam_f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll 8
    for (int kk = 0; kk < 64; ++kk) o = __builtin_amdgcn_mfma_f32_32x32x2f32(xr[2 * kk], er[2 * kk], o, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = 32 * nb + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (n < Np) {
        const float res = gamma * o[r] + xf[(int64_t)n * 128 + c];
        if (out_bf16) reinterpret_cast<__bf16*>(y)[(int64_t)f * Np * 128 + (int64_t)n * 128 + c] = (__bf16)res;
        else y[(int64_t)f * Np * 128 + (int64_t)n * 128 + c] = res;
      }
    }
  }
}


// ---- CAM by output-channel halves (round 6): two workgroups per frame, each with the frame's rows (Np + 1 of them) and ITS 64 rows of
// the channel attention in LDS — 77 KB at Np = 81, two workgroups per CU where cam_kernel's 118 KB ran one frame at a time through its
// four barrier-separated phases.  The same fma chains (ascending k) as cam_kernel: the same bits.
__global__ __launch_bounds__(CAM_THREADS) void cam_split_kernel(const float* x, float gamma, float* y, int Np, int out_bf16) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int NW = CAM_THREADS / 64;
  const int NR = (Np + 2) & ~1;     // rows staged: Np rounded up to even (+ a zero row when Np is odd: the energy's last k pair)
  float* xs = sm;                   // [NR][CAM_XP]
  float* E = xs + NR * CAM_XP;      // [64][CAM_EP]: rows 64 hc .. + 63 of the attention
  const int f = blockIdx.x, hc = blockIdx.y, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const float* xf = x + (int64_t)f * Np * 128;
  for (int i = tid; i < NR * 32; i += CAM_THREADS) {
    const int r = i >> 5;
    const float4 t = r < Np ? reinterpret_cast<const float4*>(xf)[i] : float4{0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<float4*>(xs + r * CAM_XP + (i & 31) * 4) = t;
  }
  __syncthreads();
  {  // energy[c][d] = sum_n x[n][c] x[n][d], c in this half: wave w owns tile (w >> 2, w & 3)
    const int cl = 32 * (wave >> 2), c0 = 64 * hc + cl, d0 = 32 * (wave & 3);
    pam_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int nk = (Np + 1) >> 1;
#pragma unroll 8
    for (int kk = 0; kk < nk; ++kk) {
      const float* row = xs + (2 * kk + lh) * CAM_XP;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(row[c0 + l31], row[d0 + l31], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) E[(cl + (r & 3) + 8 * (r >> 2) + 4 * lh) * CAM_EP + d0 + l31] = acc[r];
  }
  __syncthreads();
  for (int c = wave; c < 64; c += NW) {               // energy_new = rowmax - energy; softmax (:75-76)
    const float e0 = E[c * CAM_EP + lane], e1 = E[c * CAM_EP + lane + 64];
    const float rmax = wave_max(fmaxf(e0, e1));
    const float n0 = rmax - e0, n1 = rmax - e1;
    const float m2 = wave_max(fmaxf(n0, n1));
    const float p0 = expf(n0 - m2), p1 = expf(n1 - m2);
    const float s = wave_sum(p0 + p1);
    E[c * CAM_EP + lane] = p0 / s;
    E[c * CAM_EP + lane + 64] = p1 / s;
  }
  __syncthreads();
  // out[n][c] = sum_d x[n][d] att[c][d] (d ascending): wave w owns channel block w & 1 of the half and the position blocks (w >> 1), + 4
  const int RB = (Np + 31) >> 5;
  const int cl = 32 * (wave & 1) + l31, c = 64 * hc + cl;
  const float* er = E + cl * CAM_EP + lh;
  for (int nb = wave >> 1; nb < RB; nb += 4) {
    const float* x0 = xs + min(32 * nb + l31, Np - 1) * CAM_XP + lh;
    pam_f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll 8
    for (int kk = 0; kk < 64; ++kk) o = __builtin_amdgcn_mfma_f32_32x32x2f32(x0[2 * kk], er[2 * kk], o, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = 32 * nb + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (n < Np) {
        const float res = gamma * o[r] + xs[n * CAM_XP + c];
        if (out_bf16) reinterpret_cast<__bf16*>(y)[(int64_t)f * Np * 128 + n * 128 + c] = (__bf16)res;
        else y[(int64_t)f * Np * 128 + n * 128 + c] = res;
      }
    }
  }
}

static int cam_launch(const float* x, float gamma, void* y, int32_t F, int32_t Np, int out_bf16, void* stream);
extern "C" int cadre_cam(const float* x, float gamma, float* y, int32_t F, int32_t Np, void* stream) {
  return cam_launch(x, gamma, y, F, Np, 0, stream);
}
extern "C" int cadre_cam_bf16out(const float* x, float gamma, void* y, int32_t F, int32_t Np, void* stream) {
  return cam_launch(x, gamma, y, F, Np, 1, stream);
}
static int cam_launch(const float* x, float gamma, void* y, int32_t F, int32_t Np, int out_bf16, void* stream) {
  FAIL_IF(!x || !y || F < 1 || Np < 1 || Np > PAM_BIGNP, "cadre_cam: bad argument (Np<=1024)");
  static const int cam_force_large = [] { const char* e = getenv("CADRE_CAM_LARGE"); return e ? atoi(e) : 0; }();
  if (Np > PAM_MAXNP || cam_force_large) {
    static bool attr_b = false;
    if (!attr_b) {
      (void)hipFuncSetAttribute((const void*)cam_large_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr_b = true;
    }
    hipLaunchKernelGGL(cam_large_kernel, dim3(F), dim3(CAM_THREADS), sizeof(float) * 128 * CAM_EP, ST(stream), x, gamma, (float*)y, Np, out_bf16);
    return (int)hipGetLastError();
  }
  // (round 6) two workgroups per frame, by output-channel halves: 240 -> see tools/dbg/pam_cam_time.py; CADRE_CAM_SPLIT=0: one per frame
  static const int cam_split = [] { const char* e = getenv("CADRE_CAM_SPLIT"); return e ? atoi(e) : 1; }();
  if (cam_split) {
    const size_t shm_s = sizeof(float) * ((size_t)((Np + 2) & ~1) * CAM_XP + 64 * CAM_EP);
    static bool attr_s = false;
    if (!attr_s) {
      (void)hipFuncSetAttribute((const void*)cam_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr_s = true;
    }
    hipLaunchKernelGGL(cam_split_kernel, dim3(F, 2), dim3(CAM_THREADS), shm_s, ST(stream), x, gamma, (float*)y, Np, out_bf16);
    return (int)hipGetLastError();
  }
  const size_t shm = sizeof(float) * ((size_t)((Np + 31) / 32) * 32 * CAM_XP + 128 * CAM_EP);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)cam_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(cam_kernel, dim3(F), dim3(CAM_THREADS), shm, ST(stream), x, gamma, (float*)y, Np, out_bf16);
  return (int)hipGetLastError();
}

// ============================================================================ inter-task attention tail
__global__ __launch_bounds__(256) void intertask_kernel(const float* qkv, float* out, int64_t ldo, float temp) {
  __shared__ float s[6 * 256];
  __shared__ float ext[4][4];                       // per wave: (max, min) of the two key vectors
  const int f = blockIdx.x, i = threadIdx.x;
  const float* src = qkv + (int64_t)f * 1536;
#pragma unroll
  for (int r = 0; r < 6; ++r) s[r * 256 + i] = src[r * 256 + i];
  __syncthreads();
  const float* vq = s, *vk = s + 256, *vv = s + 512, *bq = s + 768, *bk = s + 1024, *bv = s + 1280;
  // The row maximum of the rank-one score matrix q_i * k_j is q_i * max(k) or q_i * min(k) (the product with the
  // extreme of the right sign is the SAME fp32 product the row would have found): two block reductions replace the
  // per-row max pass, and the softmax numerators are computed once — normaliser and weighted sum in one sweep.
  {
    const float a = bk[i], b = vk[i];
    const float amax = wave_max(a), amin = -wave_max(-a), bmax = wave_max(b), bmin = -wave_max(-b);
    if ((i & 63) == 0) { ext[i >> 6][0] = amax; ext[i >> 6][1] = amin; ext[i >> 6][2] = bmax; ext[i >> 6][3] = bmin; }
  }
  __syncthreads();
  float kext[2][2];                                 // [direction][max, min] of that direction's keys
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    kext[d][0] = fmaxf(fmaxf(ext[0][2 * d], ext[1][2 * d]), fmaxf(ext[2][2 * d], ext[3][2 * d]));
    kext[d][1] = fminf(fminf(ext[0][2 * d + 1], ext[1][2 * d + 1]), fminf(ext[2][2 * d + 1], ext[3][2 * d + 1]));
  }
  // direction 0: visual query x bc key -> bc value  (intertask_att.py:137-157)
  // direction 1: bc query x visual key -> visual value (:160-176)
#pragma unroll
  for (int dir = 0; dir < 2; ++dir) {
    const float* Q = dir == 0 ? vq : bq;
    const float* K = dir == 0 ? bk : vk;
    const float* V = dir == 0 ? bv : vv;
    const float qi = Q[i] / temp;
    const float mx = fmaxf(qi * kext[dir][0], qi * kext[dir][1]);
    float sum = 0.f, o = 0.f;
    for (int j = 0; j < 256; ++j) {
      const float e = expf(qi * K[j] - mx);
      sum += e;
      o += V[j] * e;
    }
    o = o / sum + V[i];
    // cat((att_visual, att_bc)) danet.py:232: visual first
    out[(int64_t)f * ldo + (dir == 0 ? 256 : 0) + i] = o;
  }
}

extern "C" int cadre_intertask_att(const float* qkv, float* out, int64_t ldo, int32_t F, float temperature,
                                   void* stream) {
  FAIL_IF(!qkv || !out || F < 1 || ldo < 512, "cadre_intertask_att: bad argument");
  hipLaunchKernelGGL(intertask_kernel, dim3(F), dim3(256), 0, ST(stream), qkv, out, ldo, temperature);
  return (int)hipGetLastError();
}

__global__ void append_meas_kernel(const double* meas, float* feat, int64_t ldo, int F) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= F * 18) return;
  const int f = i / 18, j = i % 18;
  feat[(int64_t)f * ldo + 512 + j] = (float)meas[f * 3 + (j % 3)];
}
extern "C" int cadre_append_measurements(const double* meas, float* feat, int64_t ldo, int32_t F, void* stream) {
  FAIL_IF(!meas || !feat || F < 1 || ldo < 530, "cadre_append_measurements: bad argument");
  hipLaunchKernelGGL(append_meas_kernel, dim3((F * 18 + 255) / 256), dim3(256), 0, ST(stream), meas, feat, ldo, F);
  return (int)hipGetLastError();
}

// ============================================================================ GAE + advantage normalisation
// One 64-lane workgroup per sequence: stage r, V, m in LDS (coalesced), lane 0 runs the strict
// left-to-right scan with explicitly rounded fp32 ops (no FMA contraction), all lanes then
// form the advantages and the double-precision mean / unbiased std.
__global__ __launch_bounds__(64) void gae_kernel(const float* rewards, float* value_preds, const float* masks,
                                                 const float* next_value, float* returns, float* adv, int T,
                                                 float gamma, float gamma_tau, int normalise) {
  // Forbid FMA contraction for the whole body so every product and sum is rounded separately, as
  // torch's one-op-at-a-time CPU code does (HIP's __fmul_rn & co. are plain inline operators that
  // the optimiser still contracts, so the arithmetic is written with bare operators here).
#pragma clang fp contract(off)
  extern __shared__ float sm[];
  float* r = sm, *V = sm + (T + 1), *m = V + (T + 1), *ret = m + (T + 1);
  const int s = blockIdx.x, lane = threadIdx.x;
  const int64_t base = (int64_t)s * (T + 1);
  for (int i = lane; i <= T; i += 64) {
    r[i] = rewards[base + i];
    V[i] = value_preds[base + i];
    m[i] = masks[base + i];
  }
  __syncthreads();
  if (lane == 0) {
    V[T] = next_value[s];                              // storage.py:70
    float gae = 0.f;
    for (int t = T - 1; t >= 0; --t) {                  // storage.py:72-76
      const float t1 = (gamma * V[t + 1]);
      const float t2 = (t1 * m[t]);
      const float t3 = (r[t] + t2);
      const float delta = (t3 - V[t]);
      const float u1 = (gamma_tau * m[t]);
      const float u2 = (u1 * gae);
      gae = (delta + u2);
      ret[t] = (gae + V[t]);
    }
    value_preds[base + T] = V[T];
  }
  __syncthreads();
  double sum = 0.0;
  for (int i = lane; i < T; i += 64) {
    returns[base + i] = ret[i];
    const float a = (ret[i] - V[i]);            // train.py:82
    r[i] = a;                                           // reuse r[] for advantages
    sum += (double)a;
  }
  sum = wave_sum_d(sum);
  if (!normalise) {
    for (int i = lane; i < T; i += 64) adv[(int64_t)s * T + i] = r[i];
    return;
  }
  const double mean = sum / T;
  double sq = 0.0;
  for (int i = lane; i < T; i += 64) {
    const double d = (double)r[i] - mean;
    sq += d * d;
  }
  sq = wave_sum_d(sq);
  const float meanf = (float)mean;
  const float stdf = (float)sqrt(sq / (T - 1));         // torch.std default: unbiased
  const float den = (stdf + 1e-8f);
  for (int i = lane; i < T; i += 64)                    // train.py:87
    adv[(int64_t)s * T + i] = ((r[i] - meanf) / den);
}

extern "C" int cadre_gae(const float* rewards, float* value_preds, const float* masks, const float* next_value,
                         float* returns, float* adv, int32_t nseq, int32_t T, float gamma, float gamma_tau,
                         int32_t normalise, void* stream) {
  FAIL_IF(!rewards || !value_preds || !masks || !next_value || !returns || !adv || nseq < 1 || T < 2 || T > 8000,
          "cadre_gae: bad argument");
  const size_t shm = sizeof(float) * 4 * (T + 1);
  hipLaunchKernelGGL(gae_kernel, dim3(nseq), dim3(64), shm, ST(stream), rewards, value_preds, masks, next_value,
                     returns, adv, T, gamma, gamma_tau, normalise);
  return (int)hipGetLastError();
}

// ============================================================================ minibatch gather (time-major)
__global__ void gather_obs_kernel(const float* obs, int64_t ldo, int S, const int64_t* idx, int B, float* x,
                                  int64_t ldx, int D) {
  const int b = blockIdx.x, s = blockIdx.y;
  const float* src = obs + ((int64_t)idx[b] * S + s) * ldo;
  float* dst = x + ((int64_t)s * B + b) * ldx;
  for (int d = threadIdx.x; d < ldx; d += blockDim.x) dst[d] = d < D ? src[d] : 0.f;
}
extern "C" int cadre_gather_obs(const float* obs, int64_t ldo, int32_t S, const int64_t* idx, int32_t B, float* x,
                                int64_t ldx, int32_t D, void* stream) {
  FAIL_IF(!obs || !idx || !x || S < 1 || B < 1 || D < 1 || ldx < D || ldo < D, "cadre_gather_obs: bad argument");
  hipLaunchKernelGGL(gather_obs_kernel, dim3(B, S), dim3(128), 0, ST(stream), obs, ldo, S, idx, B, x, ldx, D);
  return (int)hipGetLastError();
}

// Fused feed_forward_generator gather + update_policy input packing: one launch moves every field
// of a head's minibatch from the rollout storage into the learner workspace (rows b0..b0+B of a
// Bt-row packed batch, so several workers concatenate without extra copies).
__global__ void gather_minibatch_kernel(const float* obs, int64_t ldo, int S, const float* hn, const float* cn,
                                        int64_t ldh, const int64_t* action, const float* value_preds,
                                        const float* returns, const float* logp, const int32_t* command,
                                        const float* adv, const int64_t* idx, int B, int D, int Hd, int Bt, int b0,
                                        float* X, int64_t ldx, float* h0, float* c0, int64_t ldho, int64_t* actions_o,
                                        int32_t* commands_o, float* old_values_o, float* returns_o, float* old_logp_o,
                                        float* adv_o) {
  const int b = blockIdx.x, s = blockIdx.y;
  const int64_t t = idx[b];
  const int ob = b0 + b;
  if (s < S) {
    const float* src = obs + (t * S + s) * ldo;
    float* dst = X + ((int64_t)s * Bt + ob) * ldx;
    for (int d = threadIdx.x; d < ldx; d += blockDim.x) dst[d] = d < D ? src[d] : 0.f;
    return;
  }
  const float* hs = hn + t * ldh;
  const float* cs = cn + t * ldh;
  for (int d = threadIdx.x; d < ldho; d += blockDim.x) {
    h0[(int64_t)ob * ldho + d] = d < Hd ? hs[d] : 0.f;
    c0[(int64_t)ob * ldho + d] = d < Hd ? cs[d] : 0.f;
  }
  if (threadIdx.x == 0) {
    actions_o[ob] = action[t];
    commands_o[ob] = command[t];
    old_values_o[ob] = value_preds[t];
    returns_o[ob] = returns[t];
    old_logp_o[ob] = logp[t];
    adv_o[ob] = adv[t];
  }
}
extern "C" int cadre_gather_minibatch(const float* obs, int64_t ldo, int32_t S, const float* hn, const float* cn,
                                      int64_t ldh, const int64_t* action, const float* value_preds,
                                      const float* returns, const float* logp, const int32_t* command,
                                      const float* adv, const int64_t* idx, int32_t B, int32_t D, int32_t Hd,
                                      int32_t Bt, int32_t b0, float* X, int64_t ldx, float* h0, float* c0,
                                      int64_t ldho, int64_t* actions_o, int32_t* commands_o, float* old_values_o,
                                      float* returns_o, float* old_logp_o, float* adv_o, void* stream) {
  FAIL_IF(!obs || !hn || !cn || !action || !value_preds || !returns || !logp || !command || !adv || !idx || !X ||
              !h0 || !c0 || !actions_o || !commands_o || !old_values_o || !returns_o || !old_logp_o || !adv_o ||
              S < 1 || B < 1 || D < 1 || Hd < 1 || b0 < 0 || b0 + B > Bt || ldx < D || ldho < Hd,
          "cadre_gather_minibatch: bad argument");
  hipLaunchKernelGGL(gather_minibatch_kernel, dim3(B, S + 1), dim3(128), 0, ST(stream), obs, ldo, S, hn, cn, ldh,
                     action, value_preds, returns, logp, command, adv, idx, B, D, Hd, Bt, b0, X, ldx, h0, c0, ldho,
                     actions_o, commands_o, old_values_o, returns_o, old_logp_o, adv_o);
  return (int)hipGetLastError();
}

// The same gather for all workers and both heads in ONE launch: src[(worker*2 + head)] holds the nine storage pointers
// of that worker's head (stable for the life of the storages: the table is built once), idx [n_src][Bw] the minibatch
// row ids of every (worker, head); outputs are the [2][...] workspace arrays (head stride given).
struct gather_src_t {
  const float* obs; const float* hn; const float* cn; const int64_t* action; const float* value_preds;
  const float* returns; const float* logp; const int32_t* command; const float* adv;
};
__global__ void gather_minibatch_multi_kernel(const gather_src_t* src, int64_t ldo, int S, int64_t ldh, const int64_t* idx,
                                              int Bw, int D, int Hd, int Bt, float* X, int64_t x_hs, int64_t ldx, float* h0,
                                              float* c0, int64_t h_hs, int64_t ldho, int64_t* actions_o, int32_t* commands_o,
                                              float* old_values_o, float* returns_o, float* old_logp_o, float* adv_o) {
  const int b = blockIdx.x, s = blockIdx.y, zi = blockIdx.z;
  const int wi = zi >> 1, hd = zi & 1;
  const gather_src_t g = src[zi];
  const int64_t t = idx[(int64_t)zi * Bw + b];
  const int ob = wi * Bw + b;
  if (s < S) {
    const float* sp = g.obs + (t * S + s) * ldo;
    float* dst = X + hd * x_hs + ((int64_t)s * Bt + ob) * ldx;
    for (int d = threadIdx.x; d < ldx; d += blockDim.x) dst[d] = d < D ? sp[d] : 0.f;
    return;
  }
  const float* hs = g.hn + t * ldh;
  const float* cs = g.cn + t * ldh;
  for (int d = threadIdx.x; d < ldho; d += blockDim.x) {
    h0[hd * h_hs + (int64_t)ob * ldho + d] = d < Hd ? hs[d] : 0.f;
    c0[hd * h_hs + (int64_t)ob * ldho + d] = d < Hd ? cs[d] : 0.f;
  }
  if (threadIdx.x == 0) {
    const int64_t o = (int64_t)hd * Bt + ob;
    actions_o[o] = g.action[t];
    commands_o[o] = g.command[t];
    old_values_o[o] = g.value_preds[t];
    returns_o[o] = g.returns[t];
    old_logp_o[o] = g.logp[t];
    adv_o[o] = g.adv[t];
  }
}
extern "C" int cadre_gather_minibatch_multi(const void* src_table, int32_t n_src, int64_t ldo, int32_t S, int64_t ldh,
                                            const int64_t* idx, int32_t Bw, int32_t D, int32_t Hd, int32_t Bt, float* X,
                                            int64_t x_head_stride, int64_t ldx, float* h0, float* c0, int64_t h_head_stride,
                                            int64_t ldho, int64_t* actions_o, int32_t* commands_o, float* old_values_o,
                                            float* returns_o, float* old_logp_o, float* adv_o, void* stream) {
  FAIL_IF(!src_table || !idx || !X || !h0 || !c0 || !actions_o || !commands_o || !old_values_o || !returns_o || !old_logp_o ||
              !adv_o || n_src < 2 || (n_src & 1) || S < 1 || Bw < 1 || D < 1 || Hd < 1 || (n_src / 2) * Bw > Bt || ldx < D || ldho < Hd,
          "cadre_gather_minibatch_multi: bad argument");
  hipLaunchKernelGGL(gather_minibatch_multi_kernel, dim3(Bw, S + 1, n_src), dim3(128), 0, ST(stream),
                     (const gather_src_t*)src_table, ldo, S, ldh, idx, Bw, D, Hd, Bt, X, x_head_stride, ldx, h0, c0, h_head_stride,
                     ldho, actions_o, commands_o, old_values_o, returns_o, old_logp_o, adv_o);
  return (int)hipGetLastError();
}

// Gather + stable counting sort by command + placement in ONE launch (round 6; until then three: gather into staging rows,
// sort_rows_kernel, permute_minibatch_kernel — 20 us and two launch boundaries of a 580 us minibatch step).  Every workgroup
// (row b of source zi, window step s) re-derives the destination of ITS row from the commands of all Bt rows of its head — Bt
// is 64 .. 256, the commands come out of L2 — and copies the row straight to its sorted place: rank = rows with a smaller command
// + earlier rows with the same command (the order sort_rows_kernel produces: every reduction downstream keeps its order).
// Workgroup (0, S, head's first source) also writes the head's run table seg[2 * (head * C + c)] = (first row, rows) and pos.
__global__ __launch_bounds__(128) void gather_sorted_multi_kernel(const gather_src_t* src, int64_t ldo, int S, int64_t ldh,
                                                                  const int64_t* idx, int Bw, int D, int Hd, int Bt, int C, float* X,
                                                                  int64_t x_hs, int64_t ldx, float* h0, float* c0, int64_t h_hs,
                                                                  int64_t ldho, int64_t* actions_o, int32_t* commands_o,
                                                                  float* old_values_o, float* returns_o, float* old_logp_o,
                                                                  float* adv_o, int32_t* pos, int32_t* seg) {
  extern __shared__ int32_t gs_cmd[];                      // [Bt] commands of this head, then 2 counters
  const int b = blockIdx.x, s = blockIdx.y, zi = blockIdx.z;
  const int wi = zi >> 1, hd = zi & 1;
  for (int r = threadIdx.x; r < Bt; r += blockDim.x) {
    const int w2 = r / Bw, b2 = r - w2 * Bw;
    const int z2 = 2 * w2 + hd;
    gs_cmd[r] = src[z2].command[idx[(int64_t)z2 * Bw + b2]];
  }
  int32_t* cnt = gs_cmd + Bt;
  if (threadIdx.x < 2) cnt[threadIdx.x] = 0;
  __syncthreads();
  const int ob = wi * Bw + b;
  const int cme = gs_cmd[ob];
  int below = 0, before = 0;
  for (int r = threadIdx.x; r < Bt; r += blockDim.x) {
    const int c = gs_cmd[r];
    below += c < cme;
    before += (c == cme) & (r < ob);
  }
  atomicAdd(&cnt[0], below);                               // (integer sums: any order gives the same value)
  atomicAdd(&cnt[1], before);
  __syncthreads();
  const int d = cnt[0] + cnt[1];
  const gather_src_t g = src[zi];
  const int64_t t = idx[(int64_t)zi * Bw + b];
  if (s < S) {
    const float* sp = g.obs + (t * S + s) * ldo;
    float* dst = X + hd * x_hs + ((int64_t)s * Bt + d) * ldx;
    for (int k = threadIdx.x; k < ldx; k += blockDim.x) dst[k] = k < D ? sp[k] : 0.f;
    return;
  }
  const float* hs = g.hn + t * ldh;
  const float* cs = g.cn + t * ldh;
  for (int k = threadIdx.x; k < ldho; k += blockDim.x) {
    h0[hd * h_hs + (int64_t)d * ldho + k] = k < Hd ? hs[k] : 0.f;
    c0[hd * h_hs + (int64_t)d * ldho + k] = k < Hd ? cs[k] : 0.f;
  }
  if (threadIdx.x == 0) {
    const int64_t o = (int64_t)hd * Bt + d;
    actions_o[o] = g.action[t];
    commands_o[o] = cme;
    old_values_o[o] = g.value_preds[t];
    returns_o[o] = g.returns[t];
    old_logp_o[o] = g.logp[t];
    adv_o[o] = g.adv[t];
    pos[hd * Bt + ob] = d;
  }
  if (b == 0 && wi == 0 && (int)threadIdx.x < C) {         // the head's run table
    int off = 0, n = 0;
    for (int r = 0; r < Bt; ++r) { off += gs_cmd[r] < (int)threadIdx.x; n += gs_cmd[r] == (int)threadIdx.x; }
    seg[2 * (hd * C + threadIdx.x)] = off;
    seg[2 * (hd * C + threadIdx.x) + 1] = n;
  }
}
extern "C" int cadre_gather_sorted_multi(const void* src_table, int32_t n_src, int64_t ldo, int32_t S, int64_t ldh,
                                         const int64_t* idx, int32_t Bw, int32_t D, int32_t Hd, int32_t Bt, int32_t C, float* X,
                                         int64_t x_head_stride, int64_t ldx, float* h0, float* c0, int64_t h_head_stride,
                                         int64_t ldho, int64_t* actions_o, int32_t* commands_o, float* old_values_o,
                                         float* returns_o, float* old_logp_o, float* adv_o, int32_t* pos, int32_t* seg, void* stream) {
  FAIL_IF(!src_table || !idx || !X || !h0 || !c0 || !actions_o || !commands_o || !old_values_o || !returns_o || !old_logp_o ||
              !adv_o || !pos || !seg || n_src < 2 || (n_src & 1) || S < 1 || Bw < 1 || D < 1 || Hd < 1 || (n_src / 2) * Bw != Bt ||
              Bt > 8192 || C < 1 || C > 16 || ldx < D || ldho < Hd,
          "cadre_gather_sorted_multi: bad argument (every row of the minibatch comes from the table: (n_src / 2) * Bw == Bt)");
  hipLaunchKernelGGL(gather_sorted_multi_kernel, dim3(Bw, S + 1, n_src), dim3(128), (Bt + 2) * sizeof(int32_t), ST(stream),
                     (const gather_src_t*)src_table, ldo, S, ldh, idx, Bw, D, Hd, Bt, C, X, x_head_stride, ldx, h0, c0, h_head_stride,
                     ldho, actions_o, commands_o, old_values_o, returns_o, old_logp_o, adv_o, pos, seg);
  return (int)hipGetLastError();
}

#ifdef CADRE_AB_KERNELS      // A/B build only (include/cadre_hip_ab.h): the update's cell math now lives in ppo_update.hip
// ============================================================================ LSTM cell pointwise
// Rows sorted by command (row_seg != NULL: net z owns rows [row_seg[2z], +row_seg[2z+1]) of its B): only the rows of
// the 32-row tiles that intersect the run are touched — the same rows the segment-aware GEMMs read and write; the
// others belong to other nets (a net owns a quarter of the minibatch: 4x less traffic for these memory-bound passes).
__device__ __forceinline__ bool lstm_row_outside(const int32_t* row_seg, int z, int b) {
  if (!row_seg) return false;
  const int beg = row_seg[2 * z], cnt = row_seg[2 * z + 1];
  return cnt <= 0 || b < (beg & ~31) || b >= ((beg + cnt + 31) & ~31);
}

__global__ void lstm_fwd_kernel(float* gates, int64_t ldg, int64_t g_str, const float* c_prev, int64_t c_prev_str,
                                int c_prev_div, float* c_out, float* h_out, float* tanh_c, int64_t ldh,
                                int64_t h_str, int B, int Hd, const int32_t* row_seg) {
  const int z = blockIdx.z;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * Hd) return;
  const int b = i / Hd, j = i % Hd;
  if (lstm_row_outside(row_seg, z, b)) return;
  float* g = gates + z * g_str + (int64_t)b * ldg;
  const float ig = sigmoidf_(g[j]);
  const float fg = sigmoidf_(g[Hd + j]);
  const float gg = tanhf(g[2 * Hd + j]);
  const float og = sigmoidf_(g[3 * Hd + j]);
  const float cp = c_prev[(z / c_prev_div) * c_prev_str + (int64_t)b * ldh + j];
  const float c = fg * cp + ig * gg;
  const float tc = tanhf(c);
  g[j] = ig; g[Hd + j] = fg; g[2 * Hd + j] = gg; g[3 * Hd + j] = og;
  const int64_t o = z * h_str + (int64_t)b * ldh + j;
  c_out[o] = c;
  tanh_c[o] = tc;
  h_out[o] = og * tc;
}

extern "C" int cadre_lstm_pointwise_fwd(float* gates, int64_t ldg, int64_t g_str, const float* c_prev,
                                        int64_t c_prev_str, int32_t c_prev_div, float* c_out, float* h_out,
                                        float* tanh_c, int64_t ldh, int64_t h_str, int32_t B, int32_t Hd,
                                        int32_t batch, const int32_t* row_seg, void* stream) {
  FAIL_IF(!gates || !c_prev || !c_out || !h_out || !tanh_c || B < 1 || Hd < 1 || batch < 1 || c_prev_div < 1,
          "cadre_lstm_pointwise_fwd: bad argument");
  dim3 grid((B * Hd + 255) / 256, 1, batch);
  hipLaunchKernelGGL(lstm_fwd_kernel, grid, dim3(256), 0, ST(stream), gates, ldg, g_str, c_prev, c_prev_str,
                     c_prev_div, c_out, h_out, tanh_c, ldh, h_str, B, Hd, row_seg);
  return (int)hipGetLastError();
}

__global__ void lstm_bwd_kernel(const float* gates, float* dgates, int64_t ldg, int64_t g_str, const float* dh,
                                float* dc, int64_t d_str, const float* tanh_c, const float* c_prev,
                                int64_t c_prev_str, int c_prev_div, int64_t ldh, int64_t h_str, int B, int Hd,
                                const int32_t* commands, int C, const int32_t* row_seg) {
  const int z = blockIdx.z;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * Hd) return;
  const int b = i / Hd, j = i % Hd;
  if (lstm_row_outside(row_seg, z, b)) return;           // (inside the tiles of the run, foreign rows still get their zeros)
  if (commands && commands[(z / C) * B + b] != z % C) {      // row of another command net: exact zeros
    float* dgz = dgates + z * g_str + (int64_t)b * ldg;
    dgz[j] = 0.f; dgz[Hd + j] = 0.f; dgz[2 * Hd + j] = 0.f; dgz[3 * Hd + j] = 0.f;
    dc[z * d_str + (int64_t)b * ldh + j] = 0.f;
    return;
  }
  const float* g = gates + z * g_str + (int64_t)b * ldg;
  float* dg = dgates + z * g_str + (int64_t)b * ldg;
  const int64_t o = z * h_str + (int64_t)b * ldh + j;
  const int64_t od = z * d_str + (int64_t)b * ldh + j;
  const float ig = g[j], fg = g[Hd + j], gg = g[2 * Hd + j], og = g[3 * Hd + j];
  const float tc = tanh_c[o];
  const float dht = dh[od];
  const float dct = dc[od] + dht * og * (1.f - tc * tc);
  const float cp = c_prev[(z / c_prev_div) * c_prev_str + (int64_t)b * ldh + j];
  dg[j] = dct * gg * ig * (1.f - ig);
  dg[Hd + j] = dct * cp * fg * (1.f - fg);
  dg[2 * Hd + j] = dct * ig * (1.f - gg * gg);
  dg[3 * Hd + j] = dht * tc * og * (1.f - og);
  dc[od] = dct * fg;
}

extern "C" int cadre_lstm_pointwise_bwd(const float* gates, float* dgates, int64_t ldg, int64_t g_str,
                                        const float* dh, float* dc, int64_t d_str, const float* tanh_c,
                                        const float* c_prev, int64_t c_prev_str, int32_t c_prev_div, int64_t ldh,
                                        int64_t h_str, int32_t B, int32_t Hd, int32_t batch, const int32_t* commands,
                                        int32_t C, const int32_t* row_seg, void* stream) {
  FAIL_IF(!gates || !dgates || !dh || !dc || !tanh_c || !c_prev || B < 1 || Hd < 1 || batch < 1 || c_prev_div < 1,
          "cadre_lstm_pointwise_bwd: bad argument");
  dim3 grid((B * Hd + 255) / 256, 1, batch);
  hipLaunchKernelGGL(lstm_bwd_kernel, grid, dim3(256), 0, ST(stream), gates, dgates, ldg, g_str, dh, dc, d_str,
                     tanh_c, c_prev, c_prev_str, c_prev_div, ldh, h_str, B, Hd, commands, C < 1 ? 1 : C, row_seg);
  return (int)hipGetLastError();
}

#endif

// ============================================================================ column sums / relu backward
__global__ void colsum_kernel(const float* X, int64_t ldx, int64_t x_str, float* out, int64_t o_str, int M, int N,
                              int accumulate, float* out2, const int32_t* row_seg, int period) {
  // 256 threads = 64 columns x 4 row-slices; slices combined through LDS in fixed order
  __shared__ float part[4][64];
  const int z = blockIdx.z;
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), sl = threadIdx.x >> 6;
  float s = 0.f;
  if (col < N) {
    const float* x = X + z * x_str + col;
    // four independent partial sums: the loop is a chain of dependent HBM/L2 loads otherwise (M = 512 rows)
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int m = sl;
    if (row_seg) {
      // rows sorted by command, `period` rows per time step: rows outside the 32-row tiles of this net's run are
      // exact zeros (cadre_lstm_pointwise_bwd) — skipped, every other row added to the partial sum it had before
      const int beg = row_seg[2 * z], cnt = row_seg[2 * z + 1];
      const int lo = beg & ~31, hi = cnt > 0 ? ((beg + cnt + 31) & ~31) : lo;
      for (int t0 = 0; t0 < M; t0 += period)            // (t0, lo, hi are multiples of 32: row sl + 4q -> partial q % 4, as below)
        for (int b = lo + sl; b < hi; b += 16) {
          const float* xr = x + (int64_t)(t0 + b) * ldx;
          s0 += xr[0];
          s1 += xr[4 * ldx];
          s2 += xr[8 * ldx];
          s3 += xr[12 * ldx];
        }
      m = M;
    }
    for (; m + 12 < M; m += 16) {
      s0 += x[(int64_t)m * ldx];
      s1 += x[(int64_t)(m + 4) * ldx];
      s2 += x[(int64_t)(m + 8) * ldx];
      s3 += x[(int64_t)(m + 12) * ldx];
    }
    for (; m < M; m += 4) s0 += x[(int64_t)m * ldx];
    s = (s0 + s1) + (s2 + s3);
  }
  part[sl][threadIdx.x & 63] = s;
  __syncthreads();
  if (sl == 0 && col < N) {
    const float t = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
    float* o = out + z * o_str + col;
    const float v = accumulate ? *o + t : t;
    *o = v;
    if (out2) out2[z * o_str + col] = v;              // b_ih and b_hh enter the gates as a sum: identical gradients
  }
}
extern "C" int cadre_colsum(const float* X, int64_t ldx, int64_t x_str, float* out, int64_t o_str, int32_t M,
                            int32_t N, int32_t batch, int32_t accumulate, void* stream) {
  FAIL_IF(!X || !out || M < 1 || N < 1 || batch < 1, "cadre_colsum: bad argument");
  hipLaunchKernelGGL(colsum_kernel, dim3((N + 63) / 64, 1, batch), dim3(256), 0, ST(stream), X, ldx, x_str, out,
                     o_str, M, N, accumulate, (float*)nullptr, (const int32_t*)nullptr, 0);
  return (int)hipGetLastError();
}
#ifdef CADRE_AB_KERNELS
extern "C" int cadre_colsum2(const float* X, int64_t ldx, int64_t x_str, float* out, float* out2, int64_t o_str, int32_t M,
                             int32_t N, int32_t batch, const int32_t* row_seg, int32_t period, void* stream) {
  FAIL_IF(!X || !out || !out2 || M < 1 || N < 1 || batch < 1, "cadre_colsum2: bad argument");
  FAIL_IF(row_seg && (period < 32 || period % 32 != 0 || M % period != 0), "cadre_colsum2: row segments need period % 32 == 0, M % period == 0");
  hipLaunchKernelGGL(colsum_kernel, dim3((N + 63) / 64, 1, batch), dim3(256), 0, ST(stream), X, ldx, x_str, out,
                     o_str, M, N, 0, out2, row_seg, period);
  return (int)hipGetLastError();
}
#endif

// LSTM state slots: Hs[z][0] <- h0[z / x_div], Cs[z][0] <- c0[z / x_div] for z < Z (rows of ld floats, n_per = B*ld
// floats per net and slot), and dC (Z * n_per floats, may be null) <- 0: the initial hidden state of agent.py:166-175
// for every command net of a head and the zero dL/dc_T of the backward pass, in one launch.
__global__ void lstm_init_kernel(const float* h0, const float* c0, float* Hs, float* Cs, float* dC, int64_t n_per,
                                 int64_t z_str, int x_div, int Z) {
  const int z = blockIdx.y;
  const float4* h4 = reinterpret_cast<const float4*>(h0 + (int64_t)(z / x_div) * n_per);
  const float4* c4 = reinterpret_cast<const float4*>(c0 + (int64_t)(z / x_div) * n_per);
  float4* H4 = reinterpret_cast<float4*>(Hs + (int64_t)z * z_str);
  float4* C4 = reinterpret_cast<float4*>(Cs + (int64_t)z * z_str);
  float4* D4 = dC ? reinterpret_cast<float4*>(dC + (int64_t)z * n_per) : nullptr;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n_per >> 2); i += (int64_t)gridDim.x * blockDim.x) {
    H4[i] = h4[i];
    C4[i] = c4[i];
    if (D4) D4[i] = float4{0.f, 0.f, 0.f, 0.f};
  }
}
extern "C" int cadre_lstm_init(const float* h0, const float* c0, float* Hs, float* Cs, float* dC, int64_t n_per,
                               int64_t z_str, int32_t x_div, int32_t Z, void* stream) {
  FAIL_IF(!h0 || !c0 || !Hs || !Cs || n_per < 4 || (n_per & 3) || (z_str & 3) || x_div < 1 || Z < 1, "cadre_lstm_init: bad argument");
  const int bx = (int)std::min<int64_t>(((n_per >> 2) + 255) / 256, 64);
  hipLaunchKernelGGL(lstm_init_kernel, dim3(bx, Z), dim3(256), 0, ST(stream), h0, c0, Hs, Cs, dC, n_per, z_str, x_div, Z);
  return (int)hipGetLastError();
}

__global__ void relu_bwd_kernel(const float* act, float* dy, int64_t n, const int32_t* commands, int B, int hid, int C) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    bool keep = act[i] > 0.f;
    if (commands) {                                  // [2Z][B][hid]: tower z = 2*net + t, net = head*C + c
      const int64_t row = i / hid;
      const int b = (int)(row % B), net = (int)(row / B) >> 1;
      keep = keep && commands[(net / C) * B + b] == net % C;
    }
    dy[i] = keep ? dy[i] : 0.f;
  }
}
extern "C" int cadre_relu_bwd(const float* act, float* dy, int64_t n, const int32_t* commands, int32_t B, int32_t hid,
                              int32_t C, void* stream) {
  FAIL_IF(!act || !dy || n < 1 || (commands && (B < 1 || hid < 1 || C < 1)), "cadre_relu_bwd: bad argument");
  const int blocks = (int)std::min<int64_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(blocks), dim3(256), 0, ST(stream), act, dy, n, commands, B, hid, C);
  return (int)hipGetLastError();
}

// ============================================================================ rows sorted by command
// One workgroup per head; stable counting sort (rank = earlier rows with the same command), so the
// order inside a command — and with it every reduction order downstream — is deterministic.
__global__ __launch_bounds__(256) void sort_rows_kernel(const int32_t* commands, int B, int C, int32_t* pos, int32_t* seg) {
  extern __shared__ int32_t s_cmd[];
  __shared__ int32_t s_cnt[16];
  const int hd = blockIdx.x;
  const int32_t* cmd = commands + hd * B;
  for (int b = threadIdx.x; b < B; b += blockDim.x) s_cmd[b] = cmd[b];
  if (threadIdx.x < 16) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  if ((int)threadIdx.x < C) {
    int n = 0;
    for (int b = 0; b < B; ++b) n += s_cmd[b] == (int)threadIdx.x;
    s_cnt[threadIdx.x] = n;
  }
  __syncthreads();
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    const int c = s_cmd[b];
    int off = 0, rank = 0;
    for (int k = 0; k < c; ++k) off += s_cnt[k];
    for (int k = 0; k < b; ++k) rank += s_cmd[k] == c;
    pos[hd * B + b] = off + rank;
  }
  if ((int)threadIdx.x < C) {
    int off = 0;
    for (int k = 0; k < (int)threadIdx.x; ++k) off += s_cnt[k];
    seg[2 * (hd * C + threadIdx.x)] = off;
    seg[2 * (hd * C + threadIdx.x) + 1] = s_cnt[threadIdx.x];
  }
}
extern "C" int cadre_sort_rows_by_command(const int32_t* commands, int32_t B, int32_t C, int32_t* pos, int32_t* seg,
                                          void* stream) {
  FAIL_IF(!commands || !pos || !seg || B < 1 || B > 8192 || C < 1 || C > 16, "cadre_sort_rows_by_command: bad argument");
  hipLaunchKernelGGL(sort_rows_kernel, dim3(2), dim3(256), B * sizeof(int32_t), ST(stream), commands, B, C, pos, seg);
  return (int)hipGetLastError();
}

__global__ void permute_minibatch_kernel(const int32_t* pos, int B, int S, const float* X, float* Xo, int64_t ldx,
                                         const float* h0, const float* c0, float* h0o, float* c0o, int64_t ldh,
                                         const int64_t* actions, const int32_t* commands, const float* old_values,
                                         const float* returns, const float* old_logp, const float* adv,
                                         int64_t* actions_o, int32_t* commands_o, float* old_values_o,
                                         float* returns_o, float* old_logp_o, float* adv_o, int64_t x_hstr, int64_t h_hstr) {
  {   // head blockIdx.z: every per-head array is [heads][...] with the strides below
    const int64_t hd = blockIdx.z;
    pos += hd * B;
    X += hd * x_hstr; Xo += hd * x_hstr;
    h0 += hd * h_hstr; c0 += hd * h_hstr; h0o += hd * h_hstr; c0o += hd * h_hstr;
    actions += hd * B; commands += hd * B; old_values += hd * B; returns += hd * B; old_logp += hd * B; adv += hd * B;
    actions_o += hd * B; commands_o += hd * B; old_values_o += hd * B; returns_o += hd * B; old_logp_o += hd * B; adv_o += hd * B;
  }
  const int b = blockIdx.x, s = blockIdx.y, d = pos[b];
  if (s < S) {
    const float4* src = reinterpret_cast<const float4*>(X + ((int64_t)s * B + b) * ldx);
    float4* dst = reinterpret_cast<float4*>(Xo + ((int64_t)s * B + d) * ldx);
    for (int i = threadIdx.x; i < ldx / 4; i += blockDim.x) dst[i] = src[i];
    return;
  }
  for (int i = threadIdx.x; i < ldh; i += blockDim.x) {
    h0o[(int64_t)d * ldh + i] = h0[(int64_t)b * ldh + i];
    c0o[(int64_t)d * ldh + i] = c0[(int64_t)b * ldh + i];
  }
  if (threadIdx.x == 0) {
    actions_o[d] = actions[b]; commands_o[d] = commands[b]; old_values_o[d] = old_values[b];
    returns_o[d] = returns[b]; old_logp_o[d] = old_logp[b]; adv_o[d] = adv[b];
  }
}
extern "C" int cadre_permute_minibatch(const int32_t* pos, int32_t B, int32_t S, const float* X, float* Xo, int64_t ldx,
                                       const float* h0, const float* c0, float* h0o, float* c0o, int64_t ldh,
                                       const int64_t* actions, const int32_t* commands, const float* old_values,
                                       const float* returns, const float* old_logp, const float* adv,
                                       int64_t* actions_o, int32_t* commands_o, float* old_values_o, float* returns_o,
                                       float* old_logp_o, float* adv_o, int32_t heads, int64_t x_hstr, int64_t h_hstr, void* stream) {
  FAIL_IF(!pos || !X || !Xo || !h0 || !c0 || !h0o || !c0o || !actions || !commands || !old_values || !returns ||
              !old_logp || !adv || !actions_o || !commands_o || !old_values_o || !returns_o || !old_logp_o || !adv_o ||
              B < 1 || S < 1 || (ldx & 3) || heads < 1 || (x_hstr & 3),
          "cadre_permute_minibatch: bad argument");
  hipLaunchKernelGGL(permute_minibatch_kernel, dim3(B, S + 1, heads), dim3(128), 0, ST(stream), pos, B, S, X, Xo, ldx, h0, c0,
                     h0o, c0o, ldh, actions, commands, old_values, returns, old_logp, adv, actions_o, commands_o,
                     old_values_o, returns_o, old_logp_o, adv_o, x_hstr, h_hstr);
  return (int)hipGetLastError();
}

// ============================================================================ policy head + PPO loss (fwd+bwd)
#define MAX_NOUT 64
// One wave per row (lane = action index): the log-softmax / entropy reductions are wave shuffles, 16 rows per
// workgroup, B/16 x 2 workgroups.  The three loss sums stay deterministic: every workgroup publishes its partial sums
// (agent-scope stores), takes a ticket, and the workgroup that arrives last adds all partials in index order.
__global__ __launch_bounds__(256) void ppo_loss_kernel(const float* logits, int64_t ldl, int64_t l_ns,
                                                       const float* values, int64_t ldv, int64_t v_ns,
                                                       const int64_t* actions, const int32_t* commands,
                                                       const float* old_values, const float* returns,
                                                       const float* old_logp, const float* adv, int B, int C,
                                                       int n_steer, int n_throttle, float clip, float value_coeff,
                                                       float clip_coeff, float ent_coeff, float inv_b,
                                                       float* losses, float* dlogits, float* dvalues, float* scratch,
                                                       const int32_t* poison) {
  const int hd = blockIdx.y;                       // 0 steer, 1 throttle
  const int K = hd == 0 ? n_steer : n_throttle;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ float red[3][4];
  float s_act = 0.f, s_val = 0.f, s_ent = 0.f;     // lane 0 of each wave
  for (int i = 0; i < 4; ++i) {
    const int b = blockIdx.x * 16 + wave * 4 + i;
    if (b >= B) break;
    const int row = hd * B + b;                    // per-head sample arrays are [2][B]
    const int c = commands[row];
    const bool own_ok = c >= 0 && c < C;
    // the masked-out command nets (agent.py:178-182 multiply by 0) get exact zeros: whole rows of ldl columns
    for (int cc = 0; cc < C; ++cc) {
      if (own_ok && cc == c) continue;
      if (lane < ldl) dlogits[(int64_t)(hd * C + cc) * l_ns + (int64_t)b * ldl + lane] = 0.f;
      if (lane == 0) dvalues[(int64_t)(hd * C + cc) * v_ns + (int64_t)b * ldv] = 0.f;
    }
    if (!own_ok) continue;
    const int a = (int)actions[row];
    const int net = hd * C + c;
    const bool on = lane < K;
    const float x = on ? logits[(int64_t)net * l_ns + (int64_t)b * ldl + lane] : -INFINITY;
    const float mx = wave_max(x);
    const float se = wave_sum(on ? expf(x - mx) : 0.f);
    const float lse = mx + logf(se);               // Categorical(logits=x).logits  distributions.py:80-81
    const float lg = x - lse;
    const float mx2 = wave_max(on ? lg : -INFINITY);
    const float e2 = on ? expf(lg - mx2) : 0.f;
    const float se2 = wave_sum(e2);
    const float pk = e2 / se2;
    const float H = -wave_sum(on ? pk * lg : 0.f); // entropy = -sum p*logp  distributions.py:104
    const float lp = __shfl(lg, a, 64);
    const float v = values[(int64_t)net * v_ns + (int64_t)b * ldv];
    const float A = adv[row], ov = old_values[row], R = returns[row];
    // agent.py:184-187 / 215-218
    const float ratio = expf(lp - old_logp[row]);
    const float s1 = ratio * A;
    const float rc = fminf(fmaxf(ratio, 1.f - clip), 1.f + clip);
    const float s2 = rc * A;
    // agent.py:189-192 / 220-223
    const float dv = v - ov;
    const float dvc = fminf(fmaxf(dv, -clip), clip);
    const float vpc = ov + dvc;
    const float vl = (v - R) * (v - R), vlc = (vpc - R) * (vpc - R);
    s_act += -fminf(s1, s2);
    s_val += fmaxf(vl, vlc);
    s_ent += H;
    // ---- backward of total = vc*0.5*mean(max) + cc*mean(-min) - ec*mean(H)
    const bool in_ratio = ratio >= 1.f - clip && ratio <= 1.f + clip;
    float dmin_dr;                                 // d min(s1,s2) / d ratio (torch ties split 0.5/0.5)
    if (s1 < s2) dmin_dr = A;
    else if (s1 > s2) dmin_dr = in_ratio ? A : 0.f;
    else dmin_dr = 0.5f * A + (in_ratio ? 0.5f * A : 0.f);
    const float dlp = clip_coeff * inv_b * (-dmin_dr) * ratio;
    const bool in_v = dv >= -clip && dv <= clip;
    float dmax_dv;
    const float g1 = 2.f * (v - R), g2 = in_v ? 2.f * (vpc - R) : 0.f;
    if (vl > vlc) dmax_dv = g1;
    else if (vl < vlc) dmax_dv = g2;
    else dmax_dv = 0.5f * g1 + 0.5f * g2;
    if (lane == 0) dvalues[(int64_t)net * v_ns + (int64_t)b * ldv] = value_coeff * inv_b * 0.5f * dmax_dv;
    const float dH = -ent_coeff * inv_b;
    if (lane < ldl)
      dlogits[(int64_t)net * l_ns + (int64_t)b * ldl + lane] =
          on ? dlp * ((lane == a ? 1.f : 0.f) - pk) + dH * (-pk * (lg + H)) : 0.f;
  }
  if (lane == 0) { red[0][wave] = s_val; red[1][wave] = s_act; red[2][wave] = s_ent; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nblk = gridDim.x, me = hd * nblk + blockIdx.x, total = 2 * nblk;
    float* part = scratch + 4;                      // [total][3]; scratch[0] is the arrival counter (zero on entry, reset below)
    const float tv = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    const float ta = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    const float te = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
    __hip_atomic_store(part + 3 * me + 0, tv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(part + 3 * me + 1, ta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(part + 3 * me + 2, te, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned ticket = __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(scratch), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (ticket == (unsigned)(total - 1)) {          // last arriver: every partial has been published
      float sv = 0.f, sa = 0.f, sn = 0.f;
      for (int w = 0; w < total; ++w) {
        sv += __hip_atomic_load(part + 3 * w + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sa += __hip_atomic_load(part + 3 * w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sn += __hip_atomic_load(part + 3 * w + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      // a forward pass whose inter-workgroup wait timed out (cadre_lstm_seq_fwd) is reported where the caller looks
      const float bad = (poison && __hip_atomic_load(poison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) ? __builtin_nanf("") : 0.f;
      losses[0] = value_coeff * 0.5f * sv * inv_b + bad;
      losses[1] = clip_coeff * sa * inv_b + bad;
      losses[2] = ent_coeff * sn * inv_b + bad;
      // the counter goes back to zero for the next launch on this scratch (stream-ordered): no clearing launch per step
      __hip_atomic_store(reinterpret_cast<unsigned*>(scratch), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

__global__ void zero_f32_kernel(float* p, int n) {
  if ((int)threadIdx.x < n) p[threadIdx.x] = 0.f;
}

extern "C" int cadre_ppo_loss(const float* logits, int64_t ldl, int64_t l_ns, const float* values, int64_t ldv,
                              int64_t v_ns, const int64_t* actions, const int32_t* commands, const float* old_values, const float* returns,
                              const float* old_logp, const float* adv, int32_t B, int32_t C, int32_t n_out_steer,
                              int32_t n_out_throttle, float clip, float value_coeff, float clip_coeff,
                              float ent_coeff, float inv_b, float* losses, float* dlogits, float* dvalues,
                              float* scratch, const int32_t* poison, void* stream) {
  FAIL_IF(!logits || !values || !actions || !commands || !old_values || !returns || !old_logp || !adv || !losses ||
              !dlogits || !dvalues || !scratch || B < 1 || C < 1 || n_out_steer < 1 || n_out_steer > MAX_NOUT || n_out_throttle < 1 ||
              n_out_throttle > MAX_NOUT || ldl < n_out_steer || ldl < n_out_throttle || ldl > 64,
          "cadre_ppo_loss: bad argument");
  // scratch[0] (arrival counter) must be zero on entry: zero-initialised by the caller once, reset by the kernel's last
  // arriver after every launch
  hipLaunchKernelGGL(ppo_loss_kernel, dim3((B + 15) / 16, 2), dim3(256), 0, ST(stream), logits, ldl, l_ns, values, ldv, v_ns,
                     actions, commands,
                     old_values, returns, old_logp, adv, B, C, n_out_steer, n_out_throttle, clip, value_coeff,
                     clip_coeff, ent_coeff, inv_b, losses, dlogits, dvalues, scratch, poison);
  return (int)hipGetLastError();
}

// ============================================================================ sampling: argmax(p / q)
__global__ __launch_bounds__(64) void sample_kernel(const float* logits, int64_t ldl, const float* q, int64_t ldq,
                                                    int K, int64_t* action, float* logp) {
  const int r = blockIdx.x, lane = threadIdx.x;
  const float x = lane < K ? logits[(int64_t)r * ldl + lane] : -INFINITY;
  const float mx = wave_max(x);
  const float se = wave_sum(lane < K ? expf(x - mx) : 0.f);
  const float lg = x - (mx + logf(se));                          // normalised logits
  const float mx2 = wave_max(lane < K ? lg : -INFINITY);
  const float e2 = lane < K ? expf(lg - mx2) : 0.f;
  const float p = e2 / wave_sum(e2);                             // F.softmax(self.logits)  distributions.py:97
  const float pn = p / wave_sum(p);                              // Categorical(probs=) renormalises
  float best = lane < K ? pn / q[(int64_t)r * ldq + lane] : -INFINITY;
  int bi = lane;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {                             // argmax, lowest index wins ties
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  const float lsel = __shfl(lg, bi, 64);
  if (lane == 0) { action[r] = bi; logp[r] = lsel; }
}
extern "C" int cadre_sample(const float* logits, int64_t ldl, const float* q, int64_t ldq, int32_t R, int32_t n_out,
                            int64_t* action, float* logp, void* stream) {
  FAIL_IF(!logits || !q || !action || !logp || R < 1 || n_out < 1 || n_out > 64, "cadre_sample: bad argument");
  hipLaunchKernelGGL(sample_kernel, dim3(R), dim3(64), 0, ST(stream), logits, ldl, q, ldq, n_out, action, logp);
  return (int)hipGetLastError();
}

// ============================================================================ categorical evaluate (forward only)
// Model.evaluate_actions / Categorical_1d.log_probs + entropy (models.py:199-208,
// distributions.py:101-105) for a batch of rows; one wave per row.
__global__ __launch_bounds__(64) void categorical_eval_kernel(const float* logits, int64_t ldl, const int64_t* actions,
                                                              int K, float* logp, float* entropy) {
  const int r = blockIdx.x, lane = threadIdx.x;
  const float x = lane < K ? logits[(int64_t)r * ldl + lane] : -INFINITY;
  const float mx = wave_max(x);
  const float se = wave_sum(lane < K ? expf(x - mx) : 0.f);
  const float lg = x - (mx + logf(se));
  const float mx2 = wave_max(lane < K ? lg : -INFINITY);
  const float e2 = lane < K ? expf(lg - mx2) : 0.f;
  const float p = e2 / wave_sum(e2);
  const float h = wave_sum(lane < K ? -p * lg : 0.f);
  const int a = (int)actions[r];
  const float la = __shfl(lg, a & 63, 64);
  if (lane == 0) { logp[r] = la; entropy[r] = h; }
}
extern "C" int cadre_categorical_eval(const float* logits, int64_t ldl, const int64_t* actions, int32_t R, int32_t n_out,
                                      float* logp, float* entropy, void* stream) {
  FAIL_IF(!logits || !actions || !logp || !entropy || R < 1 || n_out < 1 || n_out > 64, "cadre_categorical_eval: bad argument");
  hipLaunchKernelGGL(categorical_eval_kernel, dim3(R), dim3(64), 0, ST(stream), logits, ldl, actions, n_out, logp, entropy);
  return (int)hipGetLastError();
}

// Categorical_1d.forward (distributions.py:66-83): normalised logits, probs and per-row mode.
__global__ __launch_bounds__(64) void categorical_dist_kernel(const float* raw, int64_t ldl, int K, float* logits_out,
                                                              float* probs_out, int64_t* mode_out) {
  const int r = blockIdx.x, lane = threadIdx.x;
  const float x = lane < K ? raw[(int64_t)r * ldl + lane] : -INFINITY;
  const float mx = wave_max(x);
  const float se = wave_sum(lane < K ? expf(x - mx) : 0.f);
  const float lg = x - (mx + logf(se));
  const float mx2 = wave_max(lane < K ? lg : -INFINITY);
  const float e2 = lane < K ? expf(lg - mx2) : 0.f;
  const float p = e2 / wave_sum(e2);
  if (lane < K) {
    if (logits_out) logits_out[(int64_t)r * K + lane] = lg;
    if (probs_out) probs_out[(int64_t)r * K + lane] = p;
  }
  float best = lane < K ? p : -INFINITY;
  int bi = lane;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  if (lane == 0 && mode_out) mode_out[r] = bi;
}
extern "C" int cadre_categorical_dist(const float* raw, int64_t ldl, int32_t R, int32_t n_out, float* logits_out,
                                      float* probs_out, int64_t* mode_out, void* stream) {
  FAIL_IF(!raw || R < 1 || n_out < 1 || n_out > 64 || ldl < n_out, "cadre_categorical_dist: bad argument");
  hipLaunchKernelGGL(categorical_dist_kernel, dim3(R), dim3(64), 0, ST(stream), raw, ldl, n_out, logits_out, probs_out,
                     mode_out);
  return (int)hipGetLastError();
}

// ============================================================================ per-model clip + Adam
// [rlo, rhi): the element range of the arena this call covers (the whole arena, or one rank's shard of the
// reduce-scattered gradient: every rank then holds partial square norms, summed by a 16-double all-reduce)
__global__ void sqnorm_kernel(const float* g, const int64_t* seg_off, double* norms2, int64_t rlo, int64_t rhi) {
  const int mdl = blockIdx.y;
  const int64_t lo = seg_off[mdl] > rlo ? seg_off[mdl] : rlo, hi = seg_off[mdl + 1] < rhi ? seg_off[mdl + 1] : rhi;
  if (hi <= lo) return;
  double s = 0.0;
  if ((lo & 3) == 0) {          // arena segments start on 16-byte boundaries: 16 B per lane
    const int64_t n4 = (hi - lo) >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g + lo);
    // four 16-byte loads in flight per lane (one per iteration left the pass latency-bound at 3 TB/s), four partial sums
    // combined in a fixed order
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    auto sq = [](const float4& v) { return ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w); };
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
      const float4 v0 = g4[i], v1 = g4[i + stride], v2 = g4[i + 2 * stride], v3 = g4[i + 3 * stride];
      s0 += sq(v0); s1 += sq(v1); s2 += sq(v2); s3 += sq(v3);
    }
    for (; i < n4; i += stride) s0 += sq(g4[i]);
    s = (s0 + s1) + (s2 + s3);
    if (blockIdx.x == 0 && threadIdx.x < ((hi - lo) & 3)) {
      const double v = g[lo + (n4 << 2) + threadIdx.x];
      s += v * v;
    }
  } else {
    for (int64_t i = lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < hi; i += (int64_t)gridDim.x * blockDim.x) {
      const double v = g[i];
      s += v * v;
    }
  }
  s = wave_sum_d(s);
  __shared__ double part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(norms2 + mdl, (part[0] + part[1]) + (part[2] + part[3]));
}

__global__ void adam_kernel(float* p, const float* g, float* m, float* v, const int64_t* seg_off,
                            const double* norms2, float max_norm, float step_size, float w1, float beta2,
                            float w2, float bc2_sqrt, float eps) {
  const int mdl = blockIdx.y;
  const int64_t lo = seg_off[mdl], hi = seg_off[mdl + 1];
  const float total = (float)sqrt(norms2[mdl]);
  const float coef = fminf(max_norm / (total + 1e-6f), 1.f);   // clip_grad_norm_: clamp(max=1.0), always applied
  // segment bounds are multiples of 4 floats (arena layout): 16 B per lane
  const int64_t n4 = (hi - lo) >> 2;
  float4* p4 = reinterpret_cast<float4*>(p + lo);
  const float4* g4 = reinterpret_cast<const float4*>(g + lo);
  float4* m4 = reinterpret_cast<float4*>(m + lo);
  float4* v4 = reinterpret_cast<float4*>(v + lo);
  const bool vec = (lo & 3) == 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; vec && i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
    float* pe = &pp.x; float* ge = &gg.x; float* me = &mm.x; float* ve = &vv.x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gi = ge[e] * coef;
      me[e] = me[e] + w1 * (gi - me[e]);                    // exp_avg.lerp_(grad, 1-beta1)
      ve[e] = ve[e] * beta2 + w2 * (gi * gi);               // mul_(beta2).addcmul_(g,g,1-beta2)
      pe[e] = pe[e] - step_size * (me[e] / (sqrtf(ve[e]) / bc2_sqrt + eps));
    }
    p4[i] = pp; m4[i] = mm; v4[i] = vv;
  }
  for (int64_t i = lo + (vec ? (n4 << 2) : 0) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < hi;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[i] * coef;
    const float mi = m[i] + w1 * (gi - m[i]);
    const float vi = v[i] * beta2 + w2 * (gi * gi);
    m[i] = mi;
    v[i] = vi;
    p[i] = p[i] - step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
  }
}

// Graph-capturable variant: the step count lives in device memory so a captured launch sequence
// replays with fresh bias corrections.  scal = norms2 + n_models: {step_size, bc2_sqrt} as doubles.
__global__ void adam_prep_kernel(double* norms2, int n_models, int32_t* step_dev, double lr, double beta1, double beta2) {
  const int i = threadIdx.x;
  if (i < n_models) norms2[i] = 0.0;
  if (i == 0) {
    const int step = ++(*step_dev);
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    norms2[n_models] = lr / bc1;
    norms2[n_models + 1] = sqrt(bc2);
  }
}

__global__ void adam_dev_kernel(float* p, const float* g, float* m, float* v, const int64_t* seg_off,
                                const double* norms2, int n_models, float max_norm, float w1, float beta2, float w2,
                                float eps, int64_t rlo, int64_t rhi) {
  const int mdl = blockIdx.y;
  const int64_t lo = seg_off[mdl] > rlo ? seg_off[mdl] : rlo, hi = seg_off[mdl + 1] < rhi ? seg_off[mdl + 1] : rhi;
  if (hi <= lo) return;
  const float total = (float)sqrt(norms2[mdl]);
  const float coef = fminf(max_norm / (total + 1e-6f), 1.f);
  const float step_size = (float)norms2[n_models], bc2_sqrt = (float)norms2[n_models + 1];
  // segment bounds are multiples of 4 floats (arena layout): 16 B per lane
  const int64_t n4 = (hi - lo) >> 2;
  float4* p4 = reinterpret_cast<float4*>(p + lo);
  const float4* g4 = reinterpret_cast<const float4*>(g + lo);
  float4* m4 = reinterpret_cast<float4*>(m + lo);
  float4* v4 = reinterpret_cast<float4*>(v + lo);
  const bool vec = (lo & 3) == 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; vec && i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
    float* pe = &pp.x; float* ge = &gg.x; float* me = &mm.x; float* ve = &vv.x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gi = ge[e] * coef;
      me[e] = me[e] + w1 * (gi - me[e]);                    // exp_avg.lerp_(grad, 1-beta1)
      ve[e] = ve[e] * beta2 + w2 * (gi * gi);               // mul_(beta2).addcmul_(g,g,1-beta2)
      pe[e] = pe[e] - step_size * (me[e] / (sqrtf(ve[e]) / bc2_sqrt + eps));
    }
    p4[i] = pp; m4[i] = mm; v4[i] = vv;
  }
  for (int64_t i = lo + (vec ? (n4 << 2) : 0) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < hi;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[i] * coef;
    const float mi = m[i] + w1 * (gi - m[i]);
    const float vi = v[i] * beta2 + w2 * (gi * gi);
    m[i] = mi;
    v[i] = vi;
    p[i] = p[i] - step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
  }
}

// ---- Adam with the recurrent weights' fragment-order copies written in the same pass (round 4).  The update's LSTM kernels
// read W_hh of the 8 nets in MFMA fragment order (ppo_update.hip: cadre_pack_lstm_weights: forward [slice][gate][k-block]
// [lane][4], backward the transpose); packing them was its own launch after every optimiser step — 37 MB read twice and 76 MB
// written, 30 us.  Here the thread that steps a 4 x 4 block of a W_hh matrix (rows n .. n+3, columns k .. k+3) also stores
// its four rows into the forward copy and its four columns into the backward copy: 16-byte stores, nothing read twice.
// adam_dev_kernel<true> steps everything else (it skips the W_hh region of the first n_lstm models).
struct whh_pack_t {
  int32_t n_lstm;       // models 0 .. n_lstm-1 are LSTM blocks of lstm_str floats
  int64_t lstm_str;     // (= seg_off[1] - seg_off[0])
  int64_t o_whh;        // offset of W_hh inside a block: [H4][ldw] row-major
  int32_t H4, ldw, D;   // 4 * D gate rows, row pitch (D zero padded to a multiple of 16), hidden width
  float* fwd;           // cadre_pack_lstm_weights' two outputs, net stride p_str
  float* bwd;
  int64_t p_str;
};

// Round 6: ONE WAVE PER 16 x 16 TILE of a W_hh, the copies written as whole KiB.  (Round 4's form gave a thread a 4 x 4 block and
// stored its rows / columns straight into the copies: 64 distinct 64-byte segments per wave instruction — bit-identical, one
// launch and 110 MB less per step, and not faster.)  Lane (r = lane >> 2, cq = lane & 3) steps row 16 mt + r, columns
// 16 jt + 4 cq .. + 3 — 16 rows x 64 contiguous bytes per instruction for p, g, m, v —, leaves the new parameters in a 16 x 16 LDS
// tile of the wave (row pitch 20 floats), and then, as lane (q = lane >> 4, c = lane & 15),
//   forward copy:  reads row c, columns 4 q .. + 3 (one ds_read_b128)   -> [slice][gate][k-block jt][lane][4]: for a gate-aligned tile
//                  one contiguous KiB; D = 530 is 2 mod 16, so gate g's tiles start 2 g rows into a slice: two runs per store;
//   backward copy: reads column c of rows 4 q .. 4 q + 3 (four ds_read_b32) -> [slice jt][quarter][k-block][lane][4]: the row tiles of
//                  the backward layout are the absolute 16-row groups this kernel walks: ONE contiguous KiB per tile.
// Rows past 4 D and hidden units past D are never written: the copies are allocated zeroed and nothing else touches those
// elements (cadre_pack_lstm_weights writes the same zeros).
__global__ __launch_bounds__(256) void adam_whh_pack_kernel(float* p, const float* g, float* m, float* v, const double* norms2,
                                                            int n_models, float max_norm, float w1, float beta2, float w2, float eps,
                                                            whh_pack_t k) {
  constexpr int TW = 2;                                    // tiles per wave and pass: 8 x 16-byte loads in flight per lane
  __shared__ __attribute__((aligned(16))) float tile_s[4][TW][16 * 20];
  const int z = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int CT = k.ldw / 16, RT = (k.H4 + 15) / 16, NB = CT;
  const int ntile = RT * CT;
  const float total = (float)sqrt(norms2[z]);
  const float coef = fminf(max_norm / (total + 1e-6f), 1.f);
  const float step_size = (float)norms2[n_models], bc2_sqrt = (float)norms2[n_models + 1];
  const int64_t base = (int64_t)z * k.lstm_str + k.o_whh;
  const int r = lane >> 2, cq = lane & 3, q = lane >> 4, c = lane & 15;
  for (int t0 = (blockIdx.x * 4 + wave) * TW; t0 < ntile; t0 += gridDim.x * 4 * TW) {
    float4 pp[TW], gg[TW], mm[TW], vv[TW];
    int64_t e[TW];
    bool ok[TW];
#pragma unroll
    for (int i = 0; i < TW; ++i) {                          // consecutive tiles: consecutive column tiles of one row tile
      const int tile = t0 + i, mt = tile / CT, jt = tile - mt * CT;
      const int row = 16 * mt + r;
      ok[i] = tile < ntile && row < k.H4;
      e[i] = base + (int64_t)row * k.ldw + 16 * jt + 4 * cq;
      pp[i] = gg[i] = mm[i] = vv[i] = float4{0.f, 0.f, 0.f, 0.f};
      if (ok[i]) {
        pp[i] = *reinterpret_cast<float4*>(p + e[i]);
        gg[i] = *reinterpret_cast<const float4*>(g + e[i]);
        mm[i] = *reinterpret_cast<float4*>(m + e[i]);
        vv[i] = *reinterpret_cast<float4*>(v + e[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < TW; ++i) {
      float* pe = &pp[i].x; float* ge = &gg[i].x; float* me = &mm[i].x; float* ve = &vv[i].x;
      if (ok[i]) {
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {                    // (the arithmetic of adam_dev_kernel, same order)
          const float gi = ge[cc] * coef;
          me[cc] = me[cc] + w1 * (gi - me[cc]);
          ve[cc] = ve[cc] * beta2 + w2 * (gi * gi);
          pe[cc] = pe[cc] - step_size * (me[cc] / (sqrtf(ve[cc]) / bc2_sqrt + eps));
        }
        *reinterpret_cast<float4*>(p + e[i]) = pp[i];
        *reinterpret_cast<float4*>(m + e[i]) = mm[i];
        *reinterpret_cast<float4*>(v + e[i]) = vv[i];
      }
      *reinterpret_cast<float4*>(tile_s[wave][i] + r * 20 + 4 * cq) = pp[i];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < TW; ++i) {
      const int tile = t0 + i, mt = tile / CT, jt = tile - mt * CT;
      const float* ts = tile_s[wave][i];
      if (tile < ntile) {
        {  // forward copy: row 16 mt + c = gate * D + u
          const int row = 16 * mt + c;
          if (row < k.H4) {
            const int gate = row / k.D, u = row - gate * k.D;
            const float4 val = *reinterpret_cast<const float4*>(ts + c * 20 + 4 * q);
            const int64_t blk = (int64_t)((u >> 4) * 4 + gate) * NB + jt;
            *reinterpret_cast<float4*>(k.fwd + (int64_t)z * k.p_str + (blk * 64 + q * 16 + (u & 15)) * 4) = val;
          }
        }
        {  // backward copy: hidden unit u = 16 jt + c, rows 16 mt + 4 q .. + 3 (zeros past 4 D: the tile's padding rows hold zeros)
          const int u = 16 * jt + c;
          if (u < k.D) {
            const float4 val = {ts[(4 * q) * 20 + c], ts[(4 * q + 1) * 20 + c], ts[(4 * q + 2) * 20 + c], ts[(4 * q + 3) * 20 + c]};
            const int w = mt / NB, j = mt - w * NB;
            const int64_t blk = (int64_t)(jt * 4 + w) * NB + j;
            *reinterpret_cast<float4*>(k.bwd + (int64_t)z * k.p_str + (blk * 64 + lane) * 4) = val;
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// adam_dev_kernel on everything but the W_hh regions of the LSTM models (those: adam_whh_pack_kernel)
__global__ void adam_dev_skip_kernel(float* p, const float* g, float* m, float* v, const int64_t* seg_off, const double* norms2,
                                     int n_models, float max_norm, float w1, float beta2, float w2, float eps, int n_lstm,
                                     int64_t o_whh, int64_t whh_len) {
  const int mdl = blockIdx.y;
  const float total = (float)sqrt(norms2[mdl]);
  const float coef = fminf(max_norm / (total + 1e-6f), 1.f);
  const float step_size = (float)norms2[n_models], bc2_sqrt = (float)norms2[n_models + 1];
  for (int part = 0; part < 2; ++part) {
    int64_t lo = seg_off[mdl], hi = seg_off[mdl + 1];
    if (mdl < n_lstm) {
      if (part == 0) hi = lo + o_whh; else lo = lo + o_whh + whh_len;
    } else if (part == 1) break;
    const int64_t n4 = (hi - lo) >> 2;                      // (every bound is a multiple of 4 floats: arena layout)
    float4* p4 = reinterpret_cast<float4*>(p + lo);
    const float4* g4 = reinterpret_cast<const float4*>(g + lo);
    float4* m4 = reinterpret_cast<float4*>(m + lo);
    float4* v4 = reinterpret_cast<float4*>(v + lo);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
      float4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
      float* pe = &pp.x; float* ge = &gg.x; float* me = &mm.x; float* ve = &vv.x;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float gi = ge[e] * coef;
        me[e] = me[e] + w1 * (gi - me[e]);
        ve[e] = ve[e] * beta2 + w2 * (gi * gi);
        pe[e] = pe[e] - step_size * (me[e] / (sqrtf(ve[e]) / bc2_sqrt + eps));
      }
      p4[i] = pp; m4[i] = mm; v4[i] = vv;
    }
  }
}

extern "C" int cadre_clip_adam_pack_graph(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                                          const int64_t* seg_off, int32_t n_models, double* norms2, double max_norm, double lr,
                                          double beta1, double beta2, double eps, int32_t* step_dev, int32_t n_lstm,
                                          int64_t lstm_str, int64_t o_whh, int32_t H4, int32_t ldw, int32_t D, float* fwd,
                                          float* bwd, int64_t p_str, void* stream) {
  FAIL_IF(!params || !grads || !exp_avg || !exp_avg_sq || !seg_off || !norms2 || !step_dev || !fwd || !bwd || n_models < 1 ||
              n_models > 254 || n_lstm < 1 || n_lstm > n_models,
          "cadre_clip_adam_pack_graph: bad argument");
  FAIL_IF(ldw != 544 || D < 1 || D > ldw || H4 != 4 * D || (H4 & 3) || (o_whh & 3) || (lstm_str & 3) || (p_str & 3) ||
              o_whh + (int64_t)H4 * ldw > lstm_str || p_str < (int64_t)((D + 15) / 16) * 4 * 34 * 256 ||
              (((uintptr_t)params | (uintptr_t)fwd | (uintptr_t)bwd) & 15),
          "cadre_clip_adam_pack_graph: built for W_hh [4 D][544] inside an LSTM block, 16-byte aligned (see cadre_pack_lstm_weights)");
  hipLaunchKernelGGL(adam_prep_kernel, dim3(1), dim3(256), 0, ST(stream), norms2, n_models, step_dev, lr, beta1, beta2);
  const int64_t all = (int64_t)1 << 62;
  hipLaunchKernelGGL(sqnorm_kernel, dim3(64, n_models), dim3(256), 0, ST(stream), grads, seg_off, norms2, (int64_t)0, all);
  const float w1 = (float)(1.0 - beta1), b2 = (float)beta2, w2 = (float)(1.0 - beta2);
  hipLaunchKernelGGL(adam_dev_skip_kernel, dim3(256, n_models), dim3(256), 0, ST(stream), params, grads, exp_avg, exp_avg_sq, seg_off,
                     norms2, n_models, (float)max_norm, w1, b2, w2, (float)eps, n_lstm, o_whh, (int64_t)H4 * ldw);
  whh_pack_t k{n_lstm, lstm_str, o_whh, H4, ldw, D, fwd, bwd, p_str};
  const int blocks = (((H4 + 15) / 16) * (ldw / 16) + 7) / 8;                                 // per pass: two 16 x 16 tiles per wave, four waves per workgroup
  hipLaunchKernelGGL(adam_whh_pack_kernel, dim3(blocks, n_lstm), dim3(256), 0, ST(stream), params, grads, exp_avg, exp_avg_sq,
                     norms2, n_models, (float)max_norm, w1, b2, w2, (float)eps, k);
  return (int)hipGetLastError();
}

extern "C" int cadre_clip_adam_graph(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                                     const int64_t* seg_off, int32_t n_models, double* norms2, double max_norm,
                                     double lr, double beta1, double beta2, double eps, int32_t* step_dev,
                                     void* stream) {
  FAIL_IF(!params || !grads || !exp_avg || !exp_avg_sq || !seg_off || !norms2 || !step_dev || n_models < 1 ||
              n_models > 254,
          "cadre_clip_adam_graph: bad argument");
  hipLaunchKernelGGL(adam_prep_kernel, dim3(1), dim3(256), 0, ST(stream), norms2, n_models, step_dev, lr, beta1, beta2);
  dim3 grid(64, n_models), grid2(256, n_models);
  const int64_t all = (int64_t)1 << 62;
  hipLaunchKernelGGL(sqnorm_kernel, grid, dim3(256), 0, ST(stream), grads, seg_off, norms2, (int64_t)0, all);
  hipLaunchKernelGGL(adam_dev_kernel, grid2, dim3(256), 0, ST(stream), params, grads, exp_avg, exp_avg_sq, seg_off,
                     norms2, n_models, (float)max_norm, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2),
                     (float)eps, (int64_t)0, all);
  return (int)hipGetLastError();
}

// Sharded optimiser step (data-parallel ranks; the gradient arena was reduce-scattered, this rank owns elements
// [rlo, rhi)): cadre_clip_adam_norms leaves this rank's partial per-model square norms in norms2[0..n_models) — the
// caller sums them over the ranks (all-reduce of n_models doubles) — then cadre_clip_adam_apply runs clip + Adam on
// the shard only.  exp_avg / exp_avg_sq point at the SHARD's state: element i of the arena is exp_avg[i - rlo].
extern "C" int cadre_clip_adam_norms(const float* grads, const int64_t* seg_off, int32_t n_models, double* norms2,
                                     double lr, double beta1, double beta2, int32_t* step_dev, int64_t rlo,
                                     int64_t rhi, void* stream) {
  FAIL_IF(!grads || !seg_off || !norms2 || !step_dev || n_models < 1 || n_models > 254 || rlo < 0 || rhi <= rlo ||
              (rlo & 3) || (rhi & 3),
          "cadre_clip_adam_norms: bad argument (shard bounds must be multiples of 4 elements)");
  hipLaunchKernelGGL(adam_prep_kernel, dim3(1), dim3(256), 0, ST(stream), norms2, n_models, step_dev, lr, beta1, beta2);
  hipLaunchKernelGGL(sqnorm_kernel, dim3(64, n_models), dim3(256), 0, ST(stream), grads, seg_off, norms2, rlo, rhi);
  return (int)hipGetLastError();
}

extern "C" int cadre_clip_adam_apply(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                                     const int64_t* seg_off, int32_t n_models, const double* norms2, double max_norm,
                                     double beta1, double beta2, double eps, int64_t rlo, int64_t rhi, void* stream) {
  FAIL_IF(!params || !grads || !exp_avg || !exp_avg_sq || !seg_off || !norms2 || n_models < 1 || n_models > 254 ||
              rlo < 0 || rhi <= rlo || (rlo & 3) || (rhi & 3),
          "cadre_clip_adam_apply: bad argument (shard bounds must be multiples of 4 elements)");
  hipLaunchKernelGGL(adam_dev_kernel, dim3(256, n_models), dim3(256), 0, ST(stream), params, grads, exp_avg - rlo,
                     exp_avg_sq - rlo, seg_off, norms2, n_models, (float)max_norm, (float)(1.0 - beta1), (float)beta2,
                     (float)(1.0 - beta2), (float)eps, rlo, rhi);
  return (int)hipGetLastError();
}

extern "C" int cadre_clip_adam(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                               const int64_t* seg_off, int32_t n_models, double* norms2, double max_norm, double lr,
                               double beta1, double beta2, double eps, int32_t step, void* stream) {
  FAIL_IF(!params || !grads || !exp_avg || !exp_avg_sq || !seg_off || !norms2 || n_models < 1 || step < 1,
          "cadre_clip_adam: bad argument");
  zero_words(norms2, 2 * (int64_t)n_models, ST(stream));
  dim3 grid(64, n_models);
  hipLaunchKernelGGL(sqnorm_kernel, grid, dim3(256), 0, ST(stream), grads, seg_off, norms2, (int64_t)0, (int64_t)1 << 62);
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  hipLaunchKernelGGL(adam_kernel, grid, dim3(256), 0, ST(stream), params, grads, exp_avg, exp_avg_sq, seg_off,
                     norms2, (float)max_norm, (float)(lr / bc1), (float)(1.0 - beta1), (float)beta2,
                     (float)(1.0 - beta2), (float)sqrt(bc2), (float)eps);
  return (int)hipGetLastError();
}
