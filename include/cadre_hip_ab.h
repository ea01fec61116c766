/*
 * cadre_hip_ab.h — entry points of the A/B build only (CADRE_BUILD_AB=1 python -m cadre_amd.build): kernels the
 * product dispatch superseded, kept as measured comparison points.  The default
 * libcadre_hip.so neither compiles nor exports them.
 *   conv_stream_f32.hip / conv_stream_bf16.hip  cadre_gemm_t.tile 12: 64x64 conv, several M-tiles per workgroup
 *   gemm_f32_skinny.hip                         cadre_gemm_t.tile 11 (fp32): register-direct skinny GEMM
 *   gemm_stream_f32.hip                         cadre_gemm_t.tile 13 (fp32): short-K dense NT product, several M-tiles per workgroup
 *   conv3x3_c64_bf16.hip                        cadre_conv3x3_c64_bf16 below
 *   conv3x3_w128.hip                            cadre_conv3x3_w128 below: 128 x 128 wave tile, weights streamed to registers (round 5)
 *   ppo_update.hip (under CADRE_AB_KERNELS)     cadre_lstm_seq_fwd below: the persistent forward LSTM (208 vs 110 us)
 *   conv3x3_ring.hip (under CADRE_AB_KERNELS)   the 16x16x32-MFMA instantiations of the window conv (CADRE_RING_M16=1)
 *   cadre_kernels.hip (under CADRE_AB_KERNELS)  the unfused LSTM cell passes and the two-output column sum of the
 *                                               round-2 update (superseded by ppo_update.hip)
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
/* 3x3 / stride 1 / pad 1 convolution on dense bf16 NHWC with a 128-position x 128-channel WAVE tile (resnet.py:26-55 conv1 / conv2 of
 * layer2 .. layer4, danet.py:21-41 head convs), bf16 model: y = act(conv(x; w) + shift[n] + resid) as bf16 [F*H*W][N]; x [F][H][W][Cin],
 * Cin % 64 == 0, Cin >= 128, N % 128 == 0; scale must be NULL (folded into the weight rows by the caller); resid bf16 [F*H*W][N] or
 * NULL (added before the activation); act 0 / 1 (ReLU).  w in FRAGMENT order [N/128][Cin/64][9 taps kh*3+kw][4 k-steps][4 blocks of
 * 32 channels][64 lanes][8]: lane (l31, lh) of block cb of k-step s holds channel 128 g + 32 cb + l31, input channels
 * 64 c + 16 s + 8 lh .. + 7 (cadre_amd/encoder.py _w128_w) — the weights go from memory straight to the matrix cores' operand
 * registers, only the pixel window is staged in LDS (ab/conv3x3_w128.hip; a tie with the ping-pong kernel: profiles/r05_w128_wave_tile_vs_ping_pong.txt).  _supported: host logic, no launch. */
int cadre_conv3x3_w128(const void* x, const void* w, const float* scale, const float* shift, const void* resid, void* out, int32_t F,
                       int32_t H, int32_t W, int32_t Cin, int32_t N, int32_t act, void* stream);
int cadre_conv3x3_w128_supported(int32_t F, int32_t H, int32_t W, int32_t Cin, int32_t N);
/* ---------------------------------------------------------------- LSTM cell pointwise
 * nn.LSTMCell gate math (models.py:130-152).  gates [B][ldg] pre-activations (i,f,g,o blocks
 * of Hd), batched over `batch` nets with strides; c_prev of net z is read at
 * c_prev + (z / c_prev_div) * c_prev_str (step 0 shares the head's c0 across its 4 command
 * nets).  Forward overwrites gates with the activated values (kept for backward), writes
 * c_out, h_out, tanh_c. */
int cadre_lstm_pointwise_fwd(float* gates, int64_t ldg, int64_t g_str, const float* c_prev,
                             int64_t c_prev_str, int32_t c_prev_div, float* c_out, float* h_out,
                             float* tanh_c, int64_t ldh, int64_t h_str, int32_t B, int32_t Hd,
                             int32_t batch, const int32_t* row_seg, void* stream);
/* row_seg (may be NULL; [batch][2] = (first row, count) of the run of rows net z owns when the minibatch is sorted by
 * command, cadre_sort_rows_by_command): only rows of the 32-row tiles that intersect the run are touched — the rows the
 * segment-aware GEMMs (cadre_gemm_t.seg_mode) read and write. */
/* backward of one step: dh (in: upstream+recurrent grad of h_t), dc (in/out: grad of c_t ->
 * grad of c_{t-1}), both [batch][B][ldh] with net stride d_str; activated gates, tanh_c,
 * c_prev -> dgates [B][ldg] */
int cadre_lstm_pointwise_bwd(const float* gates, float* dgates, int64_t ldg, int64_t g_str,
                             const float* dh, float* dc, int64_t d_str, const float* tanh_c,
                             const float* c_prev,
                             int64_t c_prev_str, int32_t c_prev_div, int64_t ldh, int64_t h_str,
                             int32_t B, int32_t Hd, int32_t batch, const int32_t* commands, int32_t C,
                             const int32_t* row_seg, void* stream);
/* (commands != NULL: net z = head*C + c only keeps rows whose command is c; the other rows get
 * dgates = 0 and dc = 0 whatever dh holds — see cadre_gemm_t.seg_mode.) */
/* db_ih = db_hh = colsum(dG) in one pass (both biases enter the LSTM gates as a sum): out and out2 [batch][o_str] */
int cadre_colsum2(const float* X, int64_t ldx, int64_t x_str, float* out, float* out2, int64_t o_str, int32_t M,
                  int32_t N, int32_t batch, const int32_t* row_seg, int32_t period, void* stream);
/* (row_seg != NULL: rows sorted by command, `period` rows per time step — rows outside the 32-row tiles of net z's run
 * are exact zeros by construction and are skipped; the sum is bit-identical to the full one.) */
/* All S forward steps of `Z` nets in ONE persistent launch (ppo_update.hip, lstm_seq_fwd_kernel): a workgroup = (net, 16
 * hidden units) keeps its packed recurrent weights in registers for the whole launch, the workgroups of a net exchange
 * the activation rows h_t through L2 (write-through stores, one arrival counter per (net, step), one agent-scope
 * acquire per step: cdna_hip_programming.md Guideline 16).  G [S][B][ldg], Hs / Cs / TC [S+1][B][ldh] per net, slot 0 =
 * initial state (in), slots 1..S written.  sync_ws: Z*S + 1 int32 of device memory — the counters (zeroed here) and a
 * status word that a timed-out wait (bounded spins: a launch can never hang) sets to 1; pass it to cadre_ppo_loss, which
 * then returns NaN losses.  Same results, bit for bit, as S calls of cadre_lstm_step_fwd. */
int cadre_lstm_seq_fwd(const float* Wp, int64_t wp_str, const float* bias, int64_t b_str, float* G, int32_t ldg,
                       int64_t g_str, float* Hs, float* Cs, float* TC, int32_t ldh, int64_t h_str, int32_t B, int32_t D,
                       int32_t S, int32_t Z, const int32_t* row_seg, int32_t* sync_ws, void* stream);
#ifdef __cplusplus
}
#endif
#endif