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
