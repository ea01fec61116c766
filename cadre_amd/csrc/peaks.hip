// peaks.hip — sustained matrix-pipe rate of THIS device, the denominator bench.py states next to the datasheet peak
// ("datasheet figures must be re-measured on the box with a stream-copy and an MFMA-peak microbench").
// Every SIMD of every CU runs `waves_per_simd` waves, each a straight chain of register-operand MFMAs on eight
// independent accumulators: no memory, no LDS, no dependency stalls — what the pipe sustains at the clock the chip
// holds under that load.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/cadre_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

int cadre_fail(const char* msg);

template <bool BF16>
__global__ __launch_bounds__(256) void mfma_peak_kernel(int iters, float* sink) {
  f32x16 acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const float a = 1.0f + (float)threadIdx.x * 1e-6f, b = 0.5f;
  bf16x8 a16, b16;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a16[i] = (__bf16)a; b16[i] = (__bf16)b; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if constexpr (BF16) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a16, b16, acc[j], 0, 0, 0);
      else acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += acc[j][0];
  if (s == 12345.678f) sink[0] = s;                          // keeps the chain alive, never taken
}

// Launches `workgroups` x 4 waves, each wave iters x 8 MFMAs (fp32: v_mfma_f32_32x32x2_f32, 4096 FLOP each;
// bf16: v_mfma_f32_32x32x16_bf16, 32768 FLOP each).  The caller times the launch and divides.
extern "C" int cadre_mfma_peak(int32_t bf16, int32_t workgroups, int32_t iters, float* sink, void* stream) {
  if (!sink || workgroups < 1 || iters < 1) return cadre_fail("cadre_mfma_peak: bad argument");
  if (bf16) hipLaunchKernelGGL((mfma_peak_kernel<true>), dim3(workgroups), dim3(256), 0, (hipStream_t)stream, iters, sink);
  else hipLaunchKernelGGL((mfma_peak_kernel<false>), dim3(workgroups), dim3(256), 0, (hipStream_t)stream, iters, sink);
  return (int)hipGetLastError();
}

// bf16 MFMA shape comparison on RANDOM operands (MI355X_MICROARCH.md, DVFS give-back item 7: the chip can hold a higher
// clock on v_mfma_f32_16x16x32_bf16 than on v_mfma_f32_32x32x16_bf16 at equal cycles per FLOP — zeros or constants hide
// it).  Each wave holds 8 + 8 pseudo-random operand fragments and cycles through them; same FLOPs per iteration for both
// shapes (32x32x16: 8 MFMAs on 8 accumulators; 16x16x32: 16 MFMAs on 16 accumulators); static register indices only.
typedef float f32x4p __attribute__((ext_vector_type(4)));
template <int SHAPE>
__global__ __launch_bounds__(256) void mfma_shape_kernel(int iters, float* sink) {
  bf16x8 a[8], b[8];
  unsigned h = (threadIdx.x + 1) * 2654435761u ^ (blockIdx.x * 40503u);
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      h = h * 1664525u + 1013904223u;
      a[i][e] = (__bf16)(((int)(h >> 16 & 0xffff) - 32768) * (1.f / 32768.f));
      h = h * 1664525u + 1013904223u;
      b[i][e] = (__bf16)(((int)(h >> 16 & 0xffff) - 32768) * (1.f / 32768.f));
    }
  float s = 0.f;
  if constexpr (SHAPE == 32) {
    f32x16 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j], b[(j + 3) & 7], acc[j], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) s += acc[j][0];
  } else {
    f32x4p acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = f32x4p{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j & 7], b[(j + 3 + (j >> 3)) & 7], acc[j], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) s += acc[j][0];
  }
  if (s == 12345.678f) sink[0] = s;
}

// fp32 MFMAs on RANDOM operands (round 6): the constant-operand chain of mfma_peak_kernel holds 2.38 GHz; what the chip holds when every
// multiplier input toggles is the ceiling the fp32 GEMM / conv kernels are priced against in DESIGN.md.  SHAPE 2: 8 x
// v_mfma_f32_32x32x2_f32 on 8 accumulators, SHAPE 4: 32 x v_mfma_f32_16x16x4_f32 on 32 accumulators (32768 / 65536 FLOP per wave and
// iteration); 8 + 8 pseudo-random operand registers per lane.
template <int SHAPE>
__global__ __launch_bounds__(256) void mfma_f32_random_kernel(int iters, float* sink) {
  float a[8], b[8];
  unsigned h = (threadIdx.x + 1) * 2654435761u ^ (blockIdx.x * 40503u);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    h = h * 1664525u + 1013904223u;
    a[i] = ((int)(h >> 8 & 0xffffff) - 8388608) * (1.f / 8388608.f);
    h = h * 1664525u + 1013904223u;
    b[i] = ((int)(h >> 8 & 0xffffff) - 8388608) * (1.f / 8388608.f);
  }
  float s = 0.f;
  if constexpr (SHAPE == 2) {
    f32x16 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[(j + 3) & 7], acc[j], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) s += acc[j][0];
  } else {
    f32x4p acc[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) acc[j] = f32x4p{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 32; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j & 7], b[(j + 3 + (j >> 3)) & 7], acc[j], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 32; ++j) s += acc[j][0];
  }
  if (s == 12345.678f) sink[0] = s;
}

// shape 32: 8 x v_mfma_f32_32x32x16_bf16 per iteration, shape 16: 16 x v_mfma_f32_16x16x32_bf16 (262144 FLOP per wave
// and iteration either way), random operands.  The caller times the launch.
extern "C" int cadre_mfma_shape(int32_t shape, int32_t workgroups, int32_t iters, float* sink, void* stream) {
  if (!sink || workgroups < 1 || iters < 1 || (shape != 16 && shape != 32 && shape != 2 && shape != 4)) return cadre_fail("cadre_mfma_shape: bad argument");
  // shapes 2 / 4: the fp32 pipe on random operands (8 x v_mfma_f32_32x32x2_f32 = 32768 FLOP / 32 x v_mfma_f32_16x16x4_f32 = 65536 FLOP per wave and iteration)
  if (shape == 2) { hipLaunchKernelGGL((mfma_f32_random_kernel<2>), dim3(workgroups), dim3(256), 0, (hipStream_t)stream, iters, sink); return (int)hipGetLastError(); }
  if (shape == 4) { hipLaunchKernelGGL((mfma_f32_random_kernel<4>), dim3(workgroups), dim3(256), 0, (hipStream_t)stream, iters, sink); return (int)hipGetLastError(); }
  if (shape == 32) hipLaunchKernelGGL((mfma_shape_kernel<32>), dim3(workgroups), dim3(256), 0, (hipStream_t)stream, iters, sink);
  else hipLaunchKernelGGL((mfma_shape_kernel<16>), dim3(workgroups), dim3(256), 0, (hipStream_t)stream, iters, sink);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// HBM stream peaks: what a kernel that only moves bytes reaches on this device (the denominator for the HBM-bound
// kernels' `frac_of_measured`).  16 B per lane, 8 independent loads in flight per lane, grid-stride over 2048 workgroups
// (8 per CU); the read kernel folds what it loads into one word per lane so nothing is stored.
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void hbm_read_kernel(const f32x4* __restrict__ src, int64_t n16, float* sink) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (; i + 7 * stride < n16; i += 8 * stride) {
    f32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = __builtin_nontemporal_load(src + i + j * stride);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += v[j];
  }
  for (; i < n16; i += stride) acc += __builtin_nontemporal_load(src + i);
  const float s = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  if (s == 12345.678f) sink[0] = s;                          // data-dependent, practically never taken
}

__global__ __launch_bounds__(256) void hbm_copy_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, int64_t n16) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    f32x4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = __builtin_nontemporal_load(src + i + j * stride);
#pragma unroll
    for (int j = 0; j < 4; ++j) __builtin_nontemporal_store(v[j], dst + i + j * stride);
  }
  for (; i < n16; i += stride) dst[i] = src[i];
}

// mode 0: read `bytes` from src (dst unused, may be NULL; sink receives nothing in practice); mode 1: copy src -> dst.
// bytes % 16 == 0, both pointers 16-byte aligned.  The caller times the launch: read = bytes / t, copy = 2 * bytes / t.
extern "C" int cadre_hbm_stream(int32_t mode, const void* src, void* dst, int64_t bytes, float* sink, void* stream) {
  if (!src || bytes < 16 || (bytes & 15) || ((uintptr_t)src & 15) || (mode == 1 && (!dst || ((uintptr_t)dst & 15))) ||
      (mode == 0 && !sink) || mode < 0 || mode > 1)
    return cadre_fail("cadre_hbm_stream: bad argument");
  const int64_t n16 = bytes >> 4;
  if (mode == 0)
    hipLaunchKernelGGL(hbm_read_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, (const f32x4*)src, n16, sink);
  else
    hipLaunchKernelGGL(hbm_copy_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, (const f32x4*)src, (f32x4*)dst, n16);
  return (int)hipGetLastError();
}
