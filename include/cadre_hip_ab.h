/*
 * cadre_hip_ab.h — entry points of the A/B build only (CADRE_BUILD_AB=1 python -m cadre_amd.build): kernels the
 * product dispatch superseded, kept as measured comparison points.  The default
 * libcadre_hip.so neither compiles nor exports them.
 *   conv_stream_f32.hip / conv_stream_bf16.hip  cadre_gemm_t.tile 12: 64x64 conv, several M-tiles per workgroup
 *   gemm_f32_skinny.hip                         cadre_gemm_t.tile 11 (fp32): register-direct skinny GEMM
 *   conv3x3_c64_bf16.hip                        cadre_conv3x3_c64_bf16 below
 */
#ifndef CADRE_HIP_AB_H
#define CADRE_HIP_AB_H
#include "cadre_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
/* Stage-1 convolutions of the bf16 encoder (resnet.py:26-55, layer1): 3x3 / stride 1 / pad 1, 64 -> 64 channels
 * on dense NHWC bf16 x [F][H][W][64], w bf16 [64][576] (k = (kh*3 + kw)*64 + ci), y = act(conv * scale + shift
 * (+ resid)), bf16 out.  HBM-bound layer: weights resident in LDS, autonomous waves fed by LDS-DMA rings
 * (conv3x3_c64_bf16.hip).  Same k order, MFMA and epilogue arithmetic as cadre_gemm_bf16 a_mode 2. */
int cadre_conv3x3_c64_bf16(const void* x, const void* w, const float* scale, const float* shift,
                           const void* resid, void* out, int32_t F, int32_t H, int32_t W, int32_t relu,
                           void* stream);
#ifdef __cplusplus
}
#endif
#endif
