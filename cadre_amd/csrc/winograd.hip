// winograd.hip — Winograd F(2x2, 3x3) in fp32 for the stride-1 3x3 convolutions of the DANet trunk and head
// (carla_perception/Networks/danet_blocks/resnet.py:26-55, danet.py:21-41), EXPLORATORY (CADRE_WINOGRAD=1, off by
// default): the fp32 encoder sits at 0.84-0.86 of the direct-convolution MFMA roof, and 2.25x fewer
// multiply-accumulates is the only lever above 10 % that does not narrow the arithmetic.
//
//   Y = A^T [ sum_cin (G g G^T) .* (B^T d B) ] A        d: 4x4 input tile (stride 2, halo 1), Y: 2x2 outputs
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]   A^T = [1 1 1 0; 0 1 -1 -1]
//
// Unfused form: cadre_winograd_in writes V[16][T][Cin] (T = F * ceil(H/2) * ceil(W/2) tiles), ONE batched
// cadre_gemm_f32 (batch 16: M[xi] = V[xi] . U[xi]^T, U = G g G^T precomputed by the host in float64) and
// cadre_winograd_out (inverse transform + folded BN + residual + ReLU).  Both transforms are element-wise over
// channels: a thread owns (tile, 4 channels), every load and store is a coalesced 16-byte access.  HBM traffic
// per conv: 4x the input (V written, read by the GEMM) + 4x the output (M written, read back) — which is why the
// form pays only where Cin, Cout >= 256 (layer3 / layer4 / head): DESIGN.md 3.7 has the sizing and the measurement.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/cadre_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

int cadre_fail(const char* msg);

__global__ __launch_bounds__(256) void wino_in_kernel(const float* __restrict__ x, float* __restrict__ V, int F, int H, int W, int C,
                                                       int TH, int TW, long long total) {
  const long long id = (long long)blockIdx.x * 256 + threadIdx.x;
  if (id >= total) return;
  const int C4 = C >> 2;
  const int c4 = (int)(id % C4);
  const long long tile = id / C4;
  const int tx = (int)(tile % TW);
  const long long t2 = tile / TW;
  const int ty = (int)(t2 % TH), f = (int)(t2 / TH);
  const int r0 = 2 * ty - 1, q0 = 2 * tx - 1;
  f32x4 d[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = r0 + i, q = q0 + j;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if ((unsigned)r < (unsigned)H && (unsigned)q < (unsigned)W)
        v = *reinterpret_cast<const f32x4*>(x + (((long long)f * H + r) * W + q) * C + 4 * c4);
      d[i][j] = v;
    }
  f32x4 t[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {                      // B^T d
    t[0][j] = d[0][j] - d[2][j];
    t[1][j] = d[1][j] + d[2][j];
    t[2][j] = d[2][j] - d[1][j];
    t[3][j] = d[1][j] - d[3][j];
  }
  const long long T = (long long)F * TH * TW;
  float* vp = V + tile * C + 4 * c4;
  const long long plane = T * C;
#pragma unroll
  for (int i = 0; i < 4; ++i) {                      // (B^T d) B
    *reinterpret_cast<f32x4*>(vp + (4 * i + 0) * plane) = t[i][0] - t[i][2];
    *reinterpret_cast<f32x4*>(vp + (4 * i + 1) * plane) = t[i][1] + t[i][2];
    *reinterpret_cast<f32x4*>(vp + (4 * i + 2) * plane) = t[i][2] - t[i][1];
    *reinterpret_cast<f32x4*>(vp + (4 * i + 3) * plane) = t[i][1] - t[i][3];
  }
}

// act: 0 none, 1 ReLU; bit 4: the residual is added AFTER the activation (same codes as cadre_gemm_t.act)
__global__ __launch_bounds__(256) void wino_out_kernel(const float* __restrict__ Mx, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, const float* __restrict__ resid,
                                                        float* __restrict__ out, int F, int H, int W, int N, int TH, int TW,
                                                        int act, long long total) {
  const long long id = (long long)blockIdx.x * 256 + threadIdx.x;
  if (id >= total) return;
  const int N4 = N >> 2;
  const int n4 = (int)(id % N4);
  const long long tile = id / N4;
  const int tx = (int)(tile % TW);
  const long long t2 = tile / TW;
  const int ty = (int)(t2 % TH), f = (int)(t2 / TH);
  const long long T = (long long)F * TH * TW;
  const long long plane = T * N;
  const float* mp = Mx + tile * N + 4 * n4;
  f32x4 m[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) m[i][j] = *reinterpret_cast<const f32x4*>(mp + (4 * i + j) * plane);
  f32x4 s[2][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {                      // A^T m
    s[0][j] = m[0][j] + m[1][j] + m[2][j];
    s[1][j] = m[1][j] - m[2][j] - m[3][j];
  }
  const f32x4 sc = scale ? *reinterpret_cast<const f32x4*>(scale + 4 * n4) : f32x4{1.f, 1.f, 1.f, 1.f};
  const f32x4 sh = shift ? *reinterpret_cast<const f32x4*>(shift + 4 * n4) : f32x4{0.f, 0.f, 0.f, 0.f};
  const bool relu = (act & 15) == 1, post = (act & 16) != 0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const f32x4 y2[2] = {s[i][0] + s[i][1] + s[i][2], s[i][1] - s[i][2] - s[i][3]};        // (A^T m) A
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = 2 * ty + i, q = 2 * tx + j;
      if (r < H && q < W) {
        const long long e = (((long long)f * H + r) * W + q) * N + 4 * n4;
        f32x4 y = y2[j] * sc + sh;
        f32x4 rv = {0.f, 0.f, 0.f, 0.f};
        if (resid) rv = *reinterpret_cast<const f32x4*>(resid + e);
        if (!post) y += rv;
        if (relu) {
#pragma unroll
          for (int k = 0; k < 4; ++k) y[k] = fmaxf(y[k], 0.f);
        }
        if (post) y += rv;
        *reinterpret_cast<f32x4*>(out + e) = y;
      }
    }
  }
}

extern "C" int cadre_winograd_in(const float* x, float* V, int32_t F, int32_t H, int32_t W, int32_t C, void* stream) {
  if (!x || !V || F < 1 || H < 1 || W < 1 || C < 4 || (C & 3)) return cadre_fail("cadre_winograd_in: bad argument (C % 4 == 0)");
  if (((uintptr_t)x & 15) || ((uintptr_t)V & 15)) return cadre_fail("cadre_winograd_in: operands must be 16-byte aligned");
  const int TH = (H + 1) / 2, TW = (W + 1) / 2;
  const long long total = (long long)F * TH * TW * (C >> 2);
  if ((total + 255) / 256 > 0x7fffffffLL) return cadre_fail("cadre_winograd_in: too many tiles");
  hipLaunchKernelGGL(wino_in_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, V, F, H, W, C, TH, TW, total);
  return (int)hipGetLastError();
}

extern "C" int cadre_winograd_out(const float* Mx, const float* scale, const float* shift, const float* resid, float* out,
                                  int32_t F, int32_t H, int32_t W, int32_t N, int32_t act, void* stream) {
  if (!Mx || !out || F < 1 || H < 1 || W < 1 || N < 4 || (N & 3)) return cadre_fail("cadre_winograd_out: bad argument (N % 4 == 0)");
  if (((uintptr_t)Mx & 15) || ((uintptr_t)out & 15) || ((uintptr_t)resid & 15) || ((uintptr_t)scale & 15) || ((uintptr_t)shift & 15))
    return cadre_fail("cadre_winograd_out: operands must be 16-byte aligned");
  if ((act & 15) > 1) return cadre_fail("cadre_winograd_out: act must be 0 (none) or 1 (ReLU), bit 4 = residual after the activation");
  const int TH = (H + 1) / 2, TW = (W + 1) / 2;
  const long long total = (long long)F * TH * TW * (N >> 2);
  if ((total + 255) / 256 > 0x7fffffffLL) return cadre_fail("cadre_winograd_out: too many tiles");
  hipLaunchKernelGGL(wino_out_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, Mx, scale, shift, resid, out,
                     F, H, W, N, TH, TW, act, total);
  return (int)hipGetLastError();
}
