// winograd.hip — Winograd F(m x m, 3x3), m = 2 or 3, in fp32 for the stride-1 3x3 convolutions of the DANet trunk and
// head with >= 256 input channels (carla_perception/Networks/danet_blocks/resnet.py:26-55, danet.py:21-41;
// CADRE_WINOGRAD=0 restores direct convolution): the fp32 encoder sits at 0.84-0.86 of the direct-
// convolution MFMA roof, and fewer multiply-accumulates is the only lever above 10 % that does not narrow the arithmetic.
//
//   Y = A^T [ sum_cin (G g G^T) .* (B^T d B) ] A        d: (m+2) x (m+2) input tile (stride m, halo 1), Y: m x m outputs
//
// Unfused form: cadre_winograd_in writes V[(m+2)^2][T][Cin] (T = F * ceil(H/m) * ceil(W/m) tiles), ONE batched
// cadre_gemm_f32 (M[xi] = V[xi] . U[xi]^T, U = G g G^T precomputed by the host in float64) and cadre_winograd_out
// (inverse transform + folded BN + residual + ReLU).  Both transforms are element-wise over channels: a thread owns
// (tile, 4 channels), every load and store is a coalesced 16-byte access.  HBM traffic per conv: (m+2)^2 / m^2 times the
// input (V written, read by the GEMM) and as much of the output (M written, read back) — which is why the form pays only
// where Cin, Cout >= 256 (layer3 / layer4 / head): DESIGN.md 3.7 has the sizing and the measurements.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/cadre_hip.h"
#include "winograd_mats.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

int cadre_fail(const char* msg);

// Transform matrices and the constant-folded dot product: winograd_mats.h (shared with winograd_fused.hip)
// VL: channels per thread (4: 16-byte accesses; 2 for F(6x6), whose 8 x 8 + 8 live pixel vectors would not fit the register
// file as quads)
template <int M, int VL>
__global__ __launch_bounds__(256) void wino_in_kernel(const float* __restrict__ x, float* __restrict__ V, int F, int H, int W, int C,
                                                       int TH, int TW, long long total) {
  typedef float VT __attribute__((ext_vector_type(VL)));
  constexpr int N = wino_mat<M>::N;
  const long long id = (long long)blockIdx.x * 256 + threadIdx.x;
  if (id >= total) return;
  const int CV = C / VL;
  const int cv = (int)(id % CV);
  const long long tile = id / CV;
  const int tx = (int)(tile % TW);
  const long long t2 = tile / TW;
  const int ty = (int)(t2 % TH), f = (int)(t2 / TH);
  const int r0 = M * ty - 1, q0 = M * tx - 1;
  // B^T d one patch COLUMN at a time (column j of B^T d needs column j of d only): N x N + N live pixel vectors instead of
  // 2 x N x N — the 6 x 6 patch of F(4x4) would not fit the register file otherwise.  Same sums in the same order.
  VT t[N * N];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    VT d[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int r = r0 + i, q = q0 + j;
      VT v = VT{};
      if ((unsigned)r < (unsigned)H && (unsigned)q < (unsigned)W)
        v = *reinterpret_cast<const VT*>(x + (((long long)f * H + r) * W + q) * C + VL * cv);
      d[i] = v;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) t[i * N + j] = wino_dot<N, VT>(wino_mat<M>::BT[i], d, 1);
  }
  const long long T = (long long)F * TH * TW;
  float* vp = V + tile * C + VL * cv;
  const long long plane = T * C;
#pragma unroll
  for (int i = 0; i < N; ++i)                        // (B^T d) B
#pragma unroll
    for (int j = 0; j < N; ++j)
      *reinterpret_cast<VT*>(vp + (i * N + j) * plane) = wino_dot<N, VT>(wino_mat<M>::BT[j], t + i * N, 1);
}

// act: 0 none, 1 ReLU; bit 4: the residual is added AFTER the activation (same codes as cadre_gemm_t.act)
template <int M, int VL>
__global__ __launch_bounds__(256) void wino_out_kernel(const float* __restrict__ Mx, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, const float* __restrict__ resid,
                                                        float* __restrict__ out, int F, int H, int W, int Nc, int TH, int TW,
                                                        int act, long long total) {
  typedef float VT __attribute__((ext_vector_type(VL)));
  constexpr int N = wino_mat<M>::N;
  const long long id = (long long)blockIdx.x * 256 + threadIdx.x;
  if (id >= total) return;
  const int NV = Nc / VL;
  const int nv = (int)(id % NV);
  const long long tile = id / NV;
  const int tx = (int)(tile % TW);
  const long long t2 = tile / TW;
  const int ty = (int)(t2 % TH), f = (int)(t2 / TH);
  const long long T = (long long)F * TH * TW;
  const long long plane = T * Nc;
  const float* mp = Mx + tile * Nc + VL * nv;
  VT s[M * N];
#pragma unroll
  for (int j = 0; j < N; ++j) {                      // A^T m, one column of planes at a time (see wino_in_kernel)
    VT m[N];
#pragma unroll
    for (int i = 0; i < N; ++i) m[i] = *reinterpret_cast<const VT*>(mp + (i * N + j) * plane);
#pragma unroll
    for (int i = 0; i < M; ++i) s[i * N + j] = wino_dot<N, VT>(wino_mat<M>::AT[i], m, 1);
  }
  VT sc, sh;
#pragma unroll
  for (int k = 0; k < VL; ++k) { sc[k] = scale ? scale[VL * nv + k] : 1.f; sh[k] = shift ? shift[VL * nv + k] : 0.f; }
  const bool relu = (act & 15) == 1, post = (act & 16) != 0;
#pragma unroll
  for (int i = 0; i < M; ++i)
#pragma unroll
    for (int j = 0; j < M; ++j) {
      const int r = M * ty + i, q = M * tx + j;
      if (r < H && q < W) {
        const long long e = (((long long)f * H + r) * W + q) * Nc + VL * nv;
        VT y = wino_dot<N, VT>(wino_mat<M>::AT[j], s + i * N, 1) * sc + sh;       // (A^T m) A
        VT rv = VT{};
        if (resid) rv = *reinterpret_cast<const VT*>(resid + e);
        if (!post) y += rv;
        if (relu) {
#pragma unroll
          for (int k = 0; k < VL; ++k) y[k] = fmaxf(y[k], 0.f);
        }
        if (post) y += rv;
        *reinterpret_cast<VT*>(out + e) = y;
      }
    }
}

extern "C" int cadre_winograd_in(const float* x, float* V, int32_t F, int32_t H, int32_t W, int32_t C, int32_t m, void* stream) {
  if (!x || !V || F < 1 || H < 1 || W < 1 || C < 4 || (C & 3)) return cadre_fail("cadre_winograd_in: bad argument (C % 4 == 0)");
  if (m != 2 && m != 3 && m != 4 && m != 6) return cadre_fail("cadre_winograd_in: m must be 2 (F(2x2,3x3)), 3 (F(3x3,3x3)), 4 (F(4x4,3x3)) or 6 (F(6x6,3x3))");
  if (((uintptr_t)x & 15) || ((uintptr_t)V & 15)) return cadre_fail("cadre_winograd_in: operands must be 16-byte aligned");
  const int TH = (H + m - 1) / m, TW = (W + m - 1) / m;
  const long long total = (long long)F * TH * TW * (m == 6 ? C >> 1 : C >> 2);
  if ((total + 255) / 256 > 0x7fffffffLL) return cadre_fail("cadre_winograd_in: too many tiles");
  const dim3 grid((unsigned)((total + 255) / 256));
  if (m == 2) hipLaunchKernelGGL((wino_in_kernel<2, 4>), grid, dim3(256), 0, (hipStream_t)stream, x, V, F, H, W, C, TH, TW, total);
  else if (m == 3) hipLaunchKernelGGL((wino_in_kernel<3, 4>), grid, dim3(256), 0, (hipStream_t)stream, x, V, F, H, W, C, TH, TW, total);
  else if (m == 4) hipLaunchKernelGGL((wino_in_kernel<4, 4>), grid, dim3(256), 0, (hipStream_t)stream, x, V, F, H, W, C, TH, TW, total);
  else hipLaunchKernelGGL((wino_in_kernel<6, 2>), grid, dim3(256), 0, (hipStream_t)stream, x, V, F, H, W, C, TH, TW, total);
  return (int)hipGetLastError();
}

extern "C" int cadre_winograd_out(const float* Mx, const float* scale, const float* shift, const float* resid, float* out,
                                  int32_t F, int32_t H, int32_t W, int32_t N, int32_t act, int32_t m, void* stream) {
  if (!Mx || !out || F < 1 || H < 1 || W < 1 || N < 4 || (N & 3)) return cadre_fail("cadre_winograd_out: bad argument (N % 4 == 0)");
  if (m != 2 && m != 3 && m != 4 && m != 6) return cadre_fail("cadre_winograd_out: m must be 2 (F(2x2,3x3)), 3 (F(3x3,3x3)), 4 (F(4x4,3x3)) or 6 (F(6x6,3x3))");
  if (((uintptr_t)Mx & 15) || ((uintptr_t)out & 15) || ((uintptr_t)resid & 15) || ((uintptr_t)scale & 15) || ((uintptr_t)shift & 15))
    return cadre_fail("cadre_winograd_out: operands must be 16-byte aligned");
  if ((act & 15) > 1) return cadre_fail("cadre_winograd_out: act must be 0 (none) or 1 (ReLU), bit 4 = residual after the activation");
  const int TH = (H + m - 1) / m, TW = (W + m - 1) / m;
  const long long total = (long long)F * TH * TW * (m == 6 ? N >> 1 : N >> 2);
  if ((total + 255) / 256 > 0x7fffffffLL) return cadre_fail("cadre_winograd_out: too many tiles");
  const dim3 grid((unsigned)((total + 255) / 256));
  if (m == 2) hipLaunchKernelGGL((wino_out_kernel<2, 4>), grid, dim3(256), 0, (hipStream_t)stream, Mx, scale, shift, resid, out, F, H, W, N, TH, TW, act, total);
  else if (m == 3) hipLaunchKernelGGL((wino_out_kernel<3, 4>), grid, dim3(256), 0, (hipStream_t)stream, Mx, scale, shift, resid, out, F, H, W, N, TH, TW, act, total);
  else if (m == 4) hipLaunchKernelGGL((wino_out_kernel<4, 4>), grid, dim3(256), 0, (hipStream_t)stream, Mx, scale, shift, resid, out, F, H, W, N, TH, TW, act, total);
  else hipLaunchKernelGGL((wino_out_kernel<6, 2>), grid, dim3(256), 0, (hipStream_t)stream, Mx, scale, shift, resid, out, F, H, W, N, TH, TW, act, total);
  return (int)hipGetLastError();
}
