// mfma_shadow.hip — how many independent non-MFMA instructions fit in the shadow of an fp32 MFMA when ONE wave runs per SIMD?
// (tools/dbg/mfma_shadow.py builds and times it.)  Each wave: ITER x 32 x { v_mfma_f32_16x16x4_f32 on 32 independent accumulators ;
// K x <op> } with op = packed add / scalar-lane add / s_nop 0 / ds_read_b64.  The MFMA alone takes 32 cycles: the chain is matrix-pipe bound
// while K x (issue cost of op) fits in the shadow.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// the same question for the bf16 matrix instruction (v_mfma_f32_32x32x16_bf16, 16 passes = 64 cycles... 8 accumulators of 16 registers)
template <int OP, int K>
__global__ __launch_bounds__(256, 1) void shadow_bf16_kernel(int iters, float* sink) {
  __shared__ float lds[4096];
  f32x16 acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  bf16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(1.0f + threadIdx.x * 1e-3f + i); b[i] = (__bf16)(0.5f + i * 0.25f); }
  float af = 1.0f + threadIdx.x * 1e-3f;
  f32x2 x[8], y = {1.f, 2.f};
#pragma unroll
  for (int j = 0; j < 8; ++j) x[j] = f32x2{(float)j, af};
  lds[threadIdx.x] = af;
  __syncthreads();
  const float* lp = lds + (threadIdx.x & 63) * 2;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if constexpr (OP == 0) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[(j + k) & 7]) : "v"(y));
        if constexpr (OP == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[(j + k) & 7][0]) : "v"(y[0]));
        if constexpr (OP == 2) asm volatile("s_nop 0");
        if constexpr (OP == 3) { f32x2 t; asm volatile("ds_read_b64 %0, %1" : "=v"(t) : "v"((unsigned)(uintptr_t)lp)); asm volatile("" :: "v"(t)); }
        if constexpr (OP == 4) asm volatile("v_mov_b32 %0, %1" : "=v"(x[(j + k) & 7][0]) : "v"(y[0]));
      }
    }
    if constexpr (OP == 3) asm volatile("s_waitcnt lgkmcnt(0)");
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += acc[j][0];
#pragma unroll
  for (int j = 0; j < 8; ++j) s += x[j][0] + x[j][1];
  if (s == 12345.678f) sink[0] = s;
}

template <int OP, int K>
__global__ __launch_bounds__(256, 1) void shadow_kernel(int iters, float* sink) {
  __shared__ float lds[4096];
  f32x4 acc[32];
#pragma unroll
  for (int j = 0; j < 32; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f + threadIdx.x * 1e-4f;
  f32x2 x[8], y = {1.f, 2.f};
#pragma unroll
  for (int j = 0; j < 8; ++j) x[j] = f32x2{(float)j, a};
  lds[threadIdx.x] = a;
  __syncthreads();
  const float* lp = lds + (threadIdx.x & 63) * 2;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[j]) : "v"(a), "v"(b));
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if constexpr (OP == 0) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[(j + k) & 7]) : "v"(y));
        if constexpr (OP == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[(j + k) & 7][0]) : "v"(y[0]));
        if constexpr (OP == 2) asm volatile("s_nop 0");
        if constexpr (OP == 3) { f32x2 t; asm volatile("ds_read_b64 %0, %1" : "=v"(t) : "v"((unsigned)(uintptr_t)lp)); asm volatile("" :: "v"(t)); }
        if constexpr (OP == 4) asm volatile("v_mov_b32 %0, %1" : "=v"(x[(j + k) & 7][0]) : "v"(y[0]));
      }
    }
    if constexpr (OP == 3) asm volatile("s_waitcnt lgkmcnt(0)");
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 32; ++j) s += acc[j][0];
#pragma unroll
  for (int j = 0; j < 8; ++j) s += x[j][0] + x[j][1];
  if (s == 12345.678f) sink[0] = s;
}

#define LAUNCH(OP, K) if (op == OP && k == K) { hipLaunchKernelGGL((shadow_kernel<OP, K>), dim3(wgs), dim3(256), 0, (hipStream_t)stream, iters, sink); return (int)hipGetLastError(); }
#define LAUNCH_OP(OP) LAUNCH(OP, 0) LAUNCH(OP, 1) LAUNCH(OP, 2) LAUNCH(OP, 3) LAUNCH(OP, 4) LAUNCH(OP, 6) LAUNCH(OP, 8)
#define LAUNCHB(OP, K) if (op == OP && k == K) { hipLaunchKernelGGL((shadow_bf16_kernel<OP, K>), dim3(wgs), dim3(256), 0, (hipStream_t)stream, iters, sink); return (int)hipGetLastError(); }
#define LAUNCHB_OP(OP) LAUNCHB(OP, 0) LAUNCHB(OP, 1) LAUNCHB(OP, 2) LAUNCHB(OP, 3) LAUNCHB(OP, 4) LAUNCHB(OP, 6) LAUNCHB(OP, 8)
extern "C" int shadow_run_bf16(int op, int k, int wgs, int iters, float* sink, void* stream) {
  LAUNCHB_OP(0) LAUNCHB_OP(1) LAUNCHB_OP(2) LAUNCHB_OP(3) LAUNCHB_OP(4)
  return -1;
}
extern "C" int shadow_run(int op, int k, int wgs, int iters, float* sink, void* stream) {
  LAUNCH_OP(0) LAUNCH_OP(1) LAUNCH_OP(2) LAUNCH_OP(3) LAUNCH_OP(4)
  return -1;
}
