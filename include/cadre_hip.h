/*
 * cadre_hip.h — C ABI of libcadre_hip.so, the MI355X (gfx950) kernels behind the Cadre PPO
 * learner hot path.  Plain pointers + sizes + a hipStream_t (passed as void*),
 * int status return (0 = ok, <0 = bad argument, >0 = hipError_t), never throws, never owns
 * caller memory, no torch types.  All pointers are DEVICE pointers unless noted.
 *
 * The reference (BIT-MCS/Cadre) is pure Python; the "FFI" a maintainer binds is ctypes
 * (see INTEGRATION.md).  Each entry point cites the reference code it replaces
 * (paths relative to /root/reference).
 */
#ifndef CADRE_HIP_H
#define CADRE_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define CADRE_ABI_VERSION 15
int cadre_abi_version(void);
/* human-readable last argument error of the calling thread ("" if none) */
const char* cadre_last_error(void);

/* ---------------------------------------------------------------- generic tiled GEMM / conv
 * C[z][M,N] = act( (sum_k A(m,k) B(n,k)) * scale[n] + shift[n] + resid[m,n] )
 * on v_mfma_f32_32x32x2_f32 (exact f32 fma chain).  Replaces every nn.Linear / nn.Conv2d /
 * nn.LSTMCell matmul on the path: carla_perception/Networks/danet_blocks/resnet.py:111-181,
 * danet.py:21-41,96,108, intertask_att.py:39-80, ppo_agent/models.py:130-177,
 * ppo_agent/distributions.py:34-40 (forward) and their autograd backward (dX, dW).
 */
typedef struct {
  const float* A;      /* a_mode 0: [M][lda] (k contiguous); 1: [K][lda] (m contiguous);
                          2: NHWC activation, implicit-GEMM conv gather (Cin%32==0);
                          3: NHWC4 activation (Cin==4 stem), k-tile = one kernel row:
                             K = KH*32, B rows laid out [KH][32] with zeros past KW*4       */
  const float* B;      /* b_mode 0: [N][ldb] (k contiguous); 1: [K][ldb] (n contiguous)  */
  float* C;            /* [M][ldc] row-major (== NHWC for convs)                         */
  const float* scale;  /* per-n multiplier or NULL (=1)   (folded eval BatchNorm)        */
  const float* shift;  /* per-n addend or NULL (=0)       (bias / folded BN)             */
  const float* resid;  /* [M][ldr] added before the activation, or NULL; may alias C     */
  int64_t lda, ldb, ldc, ldr;
  int32_t M, N, K;
  int32_t a_mode, b_mode;
  int32_t act;         /* 0 none, 1 ReLU, 2 LeakyReLU(slope); |16: add resid AFTER the act   */
  float slope;
  /* batch (grid.z): operand offset = ((z / div) % mod) * stride elements                */
  int32_t batch;
  int32_t a_div, a_mod, b_div, b_mod, c_div, c_mod, s_div, s_mod, r_div, r_mod;
  int64_t a_str, b_str, c_str, s_str, r_str;
  /* conv geometry (a_mode 2/3): input [Nimg][H][W][Cin], output [Nimg][Ho][Wo][N]       */
  int32_t H, W, Cin, Ho, Wo, KH, KW, stride, pad;
  /* split-K: >1 writes raw partial sums, slice s to `C + s*M*ldc` (batch 1) or to
     `C + s*slots*c_str + z-offset` (batched; slots = distinct C z-slots); finish with
     cadre_splitk_reduce.  With split_k>1 scale/shift/resid/act are ignored here.        */
  int32_t split_k;
  int32_t tile;        /* 0 auto, 1 = 128x128, 2 = 128x64, 3 = 64x64, 4 = 256x128,
                          5 = 128x256, 6 = 256x64 (4-6: one workgroup per CU);
                          7 = 256x256 on 8 waves (cadre_gemm_bf16 only);
                          8 = 128x128 on 8 waves, 9 = 32x128 on 4 waves (row-sorted
                          minibatches: skips in 32-row steps), 10 = 128x64 on 8 waves
                          (8, 9: cadre_gemm_f32 only; 10 also cadre_gemm_bf16, which adds
                          11 = 256x64 on 8 waves).  Tiles 11 (fp32) / 12 exist only in the A/B
                          build (include/cadre_hip_ab.h)                                    */
  int32_t flags;       /* bit 1: C is bf16; bit 2: resid is bf16 (cadre_gemm_bf16; bit 1 also
                          honoured by cadre_gemm_f32's vector epilogue); others must be 0   */
  /* Row segments (cadre_gemm_f32 only; PPO update with the minibatch rows sorted by command,
     agent.py:170-182 evaluates every command net on every row and masks 3 of 4): batch entry z
     owns rows [row_seg[2*(z/seg_div)], +row_seg[2*(z/seg_div)+1]) of every period of
     `seg_period` rows (seg_period % 32 == 0; the M tile must divide it: 32- or 64-row tiles).
     seg_mode 1: M-tiles that own no row of their period return without writing; seg_mode 2: k-tiles (k = row index, K % seg_period == 0) outside the segment are
     not multiplied.  Rows of other nets inside a computed tile are still multiplied — callers
     keep their gradients exactly zero (cadre_relu_bwd / cadre_lstm_pointwise_bwd masks).
     seg_mode 3 (NT products, no split-K / residual): the M index runs over the entry's OWN rows
     only, period after period (compact row mc = row (mc / cnt) * seg_period + beg + mc % cnt
     of A and C): no row of another net is read, multiplied or written, whatever the run's
     alignment (the LSTM input projection of the update: [S*B] rows, period B).              */
  int32_t seg_mode;    /* 0 off                                                               */
  const int32_t* row_seg;
  int32_t seg_period, seg_div;
} cadre_gemm_t;
int cadre_gemm_f32(const cadre_gemm_t* p, void* stream);
/* the tile id cadre_gemm_f32 would use for this descriptor (p->tile, or the auto choice) */
int cadre_gemm_pick_tile(const cadre_gemm_t* p);
/* Same contract with bf16 A/B operands (element counts/strides in bf16 elements), fp32
 * accumulation on v_mfma_f32_32x32x16_bf16 and an fp32 epilogue; a_mode 0, 2 (Cin%64==0) or
 * 4 (Cin==4 stem on a zero-padded NHWC4 image: H, W = padded sizes, pad = 0, K = ceil(KH/2)*64,
 * B rows [ceil(KH/2)][64] = two kernel rows of 8 pixels x 4 channels, zeros elsewhere), b_mode 0; C bf16 or f32 (flags bit 1), resid bf16 or f32 (flags bit 2).  BASELINE config C3
 * ("bf16 encoder / fp32 losses"). */
int cadre_gemm_bf16(const cadre_gemm_t* p, void* stream);
/* tile id cadre_gemm_bf16 would launch for this descriptor (host logic, no launch) */
int cadre_gemm_bf16_pick_tile(const cadre_gemm_t* p);
/* 3x3 / stride 1 / pad 1 convolution on dense NHWC (resnet.py:26-55 conv1/conv2, danet.py:21-41 conv5a/5c/51/52):
 * x [F][H][W][Cin], w [N][NC][9][128 B] (NC = Cin*elem/128 channel chunks; chunk-major, tap = kh*3+kw, then the
 * chunk's channels), y = act(conv*scale[n] + shift[n] (+ resid)) (+ resid after act when act & 16), out [F*H*W][N].
 * flags bit 0: bf16 operands (else fp32); bit 1: out bf16; bit 2: resid bf16.  Each input pixel goes through LDS once
 * per channel chunk (conv3x3_ring.hip).  cadre_conv3x3_ring_supported: host logic, no launch; it takes the same flags
 * word plus 8 = a residual is present and bounds every tensor (< 2 GiB: 32-bit buffer offsets) at its own element size,
 * so a 1 from it means cadre_conv3x3_ring with those flags does not fail on geometry. */
int cadre_conv3x3_ring(const void* x, const void* w, const float* scale, const float* shift, const void* resid,
                       void* out, int32_t F, int32_t H, int32_t W, int32_t Cin, int32_t N, int32_t act,
                       int32_t flags, void* stream);
int cadre_conv3x3_ring_supported(int32_t F, int32_t H, int32_t W, int32_t Cin, int32_t N, int32_t flags);
/* tile configuration as ntile (64 / 128) + 1000 * WVM (waves along the positions: 4 = 256-position tile on 8 waves, one
 * workgroup per CU) + 100000 when an 8-wave ping-pong kernel runs + 1000000 * G when it is the G-k-tiles-per-slot form
 * (bf16): names the instantiation conv3x3_ring_kernel<bf16, ntile, res, out_bf16, WVM> /
 * conv3x3_ring_pp_kernel<bf16, ntile, res, out_bf16, false> / conv3x3_ring_pp2_kernel<ntile, res, out_bf16, G> */
int cadre_conv3x3_ring_ntile(int32_t F, int32_t H, int32_t W, int32_t Cin, int32_t N, int32_t bf16);
/* 3x3 / STRIDE 2 / pad 1 convolution on dense bf16 NHWC (resnet.py:26-55 conv1 of layer2.0 / layer3.0 / layer4.0, stride 2:
 * resnet.py:152-158), bf16 model: x [F][H][W][Cin] with even H, W; w [N][Cin/64][9][64] with the nine taps of a 64-channel
 * chunk in PLANE order (kh,kw) = (0,0) (0,2) (2,0) (2,2) (1,0) (1,2) (0,1) (2,1) (1,1); y = act(conv + shift[n]) as bf16
 * [F*(H/2)*(W/2)][N]; scale must be NULL (the caller folds the BN scale into the weight rows: the kernel starts its sums at
 * the shift and has no epilogue multiply); act 0 / 1 (ReLU).  The input is staged as four parity planes in LDS, each pixel once per (M tile,
 * channel chunk, N tile) instead of once per tap (conv3x3_s2.hip).  cadre_conv3x3_s2_supported: host logic, no launch. */
int cadre_conv3x3_s2(const void* x, const void* w, const float* scale, const float* shift, void* out, int32_t F, int32_t H,
                     int32_t W, int32_t Cin, int32_t N, int32_t act, void* stream);
int cadre_conv3x3_s2_supported(int32_t F, int32_t H, int32_t W, int32_t Cin, int32_t N);
/* Second conv of a down-sampling BasicBlock with the block's shortcut folded in (resnet.py:40-55 with downsample = resnet.py:152-158),
 * bf16 model: out = act( conv3x3_s1_p1(x; W2) + conv1x1_s2(x2; Wd) + shift[n] ) as bf16 [F*H*W][N], one fp32 accumulation —
 * x [F][H][W][C1], x2 [F][2H][2W][Cd] (the shortcut reads pixel (2h, 2w)), both folded-BN scales already in the weight rows,
 * shift = shift2 + shift_d.  w [N][9*C1/64 + Cd/64][64]: per 64-channel chunk c of C1 the nine taps kh*3+kw of W2, then (c < Cd/64)
 * chunk c of Wd (cadre_amd/encoder.py _s1x_w).  Needs Cd <= C1.  conv3x3_s1x.hip; _supported: host logic, no launch. */
int cadre_conv3x3_s1x(const void* x, const void* x2, const void* w, const float* shift, void* out, int32_t F, int32_t H, int32_t W,
                      int32_t C1, int32_t Cd, int32_t N, int32_t act, void* stream);
int cadre_conv3x3_s1x_supported(int32_t F, int32_t H, int32_t W, int32_t C1, int32_t Cd, int32_t N);
/* weight stages of the launch at this map width / channel count (2, or 3 under CADRE_S1X_STAGES=3 where they fit): the second
 * template argument of conv3x3_s1x_kernel<nps, stages> */
int cadre_conv3x3_s1x_stages(int32_t W, int32_t N);
/* Dense bf16 NT product with split-K into raw fp32 partial sums (the inter-task attention's first layers, intertask_att.py:39-80, bf16
 * model): slab[s][M][ldc] = A[M][k in slice s] . B[N][k in slice s]^T, slice s = 64-element k-tiles [s * per, (s + 1) * per), per =
 * ceil(K / 64 / split_k) — the slices, the k order and therefore every partial sum of cadre_gemm_bf16 with split_k (reduce with
 * cadre_splitk_reduce).  A [M][lda] bf16; B in FRAGMENT order [N/128][K/16][4 blocks of 32 columns][64 lanes][8]: lane (l31, lh) of
 * block cb of 16-element k-step q holds B[128 g + 32 cb + l31][16 q + 8 lh .. + 7] (cadre_amd/encoder.py _w128_dense_b) — it goes from
 * memory straight to the matrix cores' operand registers, only A is staged in LDS (gemm_bf16_w128.hip).  N % 256 == 0, K % 64 == 0.
 * _supported: host logic, no launch. */
int cadre_gemm_bf16_w128(const void* A, const void* B, float* C, int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldc,
                         int32_t split_k, void* stream);
int cadre_gemm_bf16_w128_supported(int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldc, int32_t split_k);
/* Sustained matrix-pipe rate of this device (peaks.hip; SURVEY.md 8d asks for the measured peak next to the datasheet
 * one): workgroups x 4 waves, each iters x 8 register-operand MFMAs on independent accumulators (fp32:
 * v_mfma_f32_32x32x2_f32 = 4096 FLOP, bf16: v_mfma_f32_32x32x16_bf16 = 32768 FLOP).  The caller times the launch. */
int cadre_mfma_peak(int32_t bf16, int32_t workgroups, int32_t iters, float* sink, void* stream);
/* bf16 MFMA shape comparison on pseudo-random register operands (peaks.hip): shape 32 = v_mfma_f32_32x32x16_bf16,
 * shape 16 = v_mfma_f32_16x16x32_bf16; 262144 FLOP per wave and iteration either way; workgroups x 4 waves.
 * shape 2 = v_mfma_f32_32x32x2_f32, shape 4 = v_mfma_f32_16x16x4_f32 on random fp32 operands: 8 x 4096 / 32 x 2048 FLOP per wave and iteration. */
int cadre_mfma_shape(int32_t shape, int32_t workgroups, int32_t iters, float* sink, void* stream);
/* HBM stream peaks of this device (peaks.hip): mode 0 reads `bytes` from src with 16-B loads, 8 in flight per lane
 * (nothing stored), mode 1 copies src -> dst.  The caller times the launch (read: bytes / t, copy: 2 * bytes / t). */
int cadre_hbm_stream(int32_t mode, const void* src, void* dst, int64_t bytes, float* sink, void* stream);
/* C[M][ldc] = act(sum_s slab[s][M][lds] * scale + shift + resid) */
int cadre_splitk_reduce(const float* slabs, int32_t split_k, int64_t slab_stride, int64_t lds,
                        float* C, int64_t ldc, int32_t M, int32_t N, const float* scale,
                        const float* shift, int32_t act, float slope, const int32_t* row_seg, int32_t period,
                        void* stream);
/* (row_seg != NULL: the M rows are [batch][period] sorted by command, row_seg [batch][2]; only the rows of the 32-row
 * tiles of each batch entry's run — the ones a seg_mode 1 GEMM wrote — are reduced, the others left as they are.) */

/* ---------------------------------------------------------------- encoder pieces
 * pre_process: ppo_agent/agent.py:43-75.  rgb u8 [F][H][W][3], route u8 [F][W][H] (stored
 * transposed like the reference) -> out f32 NHWC [F][H][W][4].  rgb/255 via the 256-entry
 * LUT `lut255` (host computes float32(i/255.) in double, bit-exact with agent.py:46).
 * Route quirk agent.py:51-54: per-frame max, value stored back into uint8 -> {0,1}.
 * `route_norm` (u8 [F][W][H], may be NULL) receives the mutated route the reference leaves
 * in the caller's dict.  `frame_max` is u32 [F] scratch. */
int cadre_preprocess(const uint8_t* rgb, const uint8_t* route, const float* lut255,
                     float* out, uint8_t* route_norm, uint32_t* frame_max,
                     int32_t F, int32_t H, int32_t W, void* stream);
/* Same conversion into the interior (pad_t, pad_l) of a zero-padded bf16 NHWC4 image
 * [F][Hp][Wp][4] whose border the caller keeps zero: input of the bf16 stem (cadre_gemm_bf16
 * a_mode 4), which then needs no halo masks. */
int cadre_preprocess_bf16pad(const uint8_t* rgb, const uint8_t* route, const float* lut255, void* out,
                             uint8_t* route_norm, uint32_t* frame_max, int32_t F, int32_t H, int32_t W,
                             int32_t Hp, int32_t Wp, int32_t pad_t, int32_t pad_l, void* stream);
/* The same pre_process as ONE packed dword per pixel for the fused front (cadre_stem_pool):
 * out u32 [F][H][W] = R | G<<8 | B<<16 | route<<24, the normalised route {0,1} stored as byte 0 / 255 so
 * that every byte takes the same /255 conversion (255/255 == 1.0f exactly).  route_norm / frame_max as above.
 * frame_idx (i64 [F], may be NULL): output frame f is source frame frame_idx[f] — the sliding 8-frame windows of
 * train.py:50-75 repeat each camera frame 8 times, the gather rides on the packing pass. */
int cadre_pack_obs(const uint8_t* rgb, const uint8_t* route, uint32_t* out, uint8_t* route_norm,
                   uint32_t* frame_max, int32_t F, int32_t H, int32_t W, const int64_t* frame_idx, int32_t n_src,
                   void* stream);
/* (with frame_idx the route maxima are taken once per SOURCE frame: n_src = number of frames in rgb / route, frame_max
 * u32 scratch of n_src entries; without it n_src is ignored and frame_max holds F entries.) */
/* Fused encoder front: packed observation -> /255 -> conv1 7x7/s2/p3 (4 -> 64) + folded BN + ReLU ->
 * MaxPool2d(3,2,1)  (agent.py:46, resnet.py:111-115,168-172) in one kernel; the stem map never reaches HBM.
 * wt: tap-major weights [64][taps][4]; fp32: tap = ky*7 + kx, 50 taps (one zero tap), BN scale in `scale`;
 *     bf16: tap = ky*8 + kx, 56 taps (kx = 7: zeros), ALREADY multiplied by the BN scale and `scale` = NULL
 *     (the kernel starts its sums at the shift).
 * out: pooled map, element (f, p, c, ch) at out_off + f*out_frame + p*out_row + c*out_px + ch (f32, or bf16 when
 * bf16 == 1, which also selects bf16 MFMA).  bf16 == 2: the fp32 front (fp32 `scale`, fp32 out) computed on the bf16 matrix cores
 * with exact products — wt = [3 pieces][64][216] bf16 with piece1 + piece2 + piece3 == the fp32 weight exactly, k = tap*4 + channel,
 * tap = ky*7 + kx, zeros past tap 48; pixel bytes are exact in bf16, sums are fp32.  Geometries: cadre_stem_pool_supported(H, W)
 * (host logic). */
int cadre_stem_pool(const uint32_t* img, const void* wt, const float* scale, const float* shift,
                    void* out, int32_t F, int32_t H, int32_t W, int32_t bf16,
                    int64_t out_frame, int64_t out_row, int32_t out_px, int64_t out_off, void* stream);
int cadre_stem_pool_supported(int32_t H, int32_t W);

/* FUSED Winograd F(2x2, 3x3) for 64 -> 64 stride-1 / pad-1 3x3 convs in fp32 (csrc/winograd_c64.hip; the fp32 model's layer1,
 * resnet.py:26-55): input transform, 16 plane products and inverse transform in one kernel, nothing of the transform
 * domain leaves the CU.  U: the transformed weights (G g G^T)[xi][cout][cin] laid out [8 chunks of 8 cin][16 planes][64 positions][8],
 * position 16 b + n = output channel 4 n + b, cin of chunk 2d + e at index 2q + s = input channel 16 d + 4 q + 2 e + s
 * (cadre_amd/encoder.py _winograd_u_c64).  out = act(conv * scale + shift (+ resid)), act 0 none / 1 ReLU; x, resid, out
 * [F][H][W][64] below 2 GiB. */
int cadre_winograd_c64(const float* x, const float* U, const float* scale, const float* shift, const float* resid, float* out,
                       int32_t F, int32_t H, int32_t W, int32_t act, void* stream);

/* Winograd F(m x m, 3x3) transforms, m = 2, 3 or 4, fp32, NHWC (csrc/winograd.hip).  A stride-1 / pad-1 3x3 convolution
 * (resnet.py:26-55) = cadre_winograd_in -> ONE cadre_gemm_f32 with batch (m+2)^2 (M[xi] = V[xi] . U[xi]^T,
 * U[xi][cout][cin] = (G g G^T)[xi] prepared by the host; Cook-Toom points 0, 1, -1, inf (m = 2) / 0, 3/4, -3/4, 2, inf (m = 3) /
 * 0, +-3/4, +-3/2, inf (m = 4):
 * cadre_amd/encoder.py _WINO_G) -> cadre_winograd_out.
 * T = F * ceil(H/m) * ceil(W/m) tiles.  V: [(m+2)^2][T][C], Mx: [(m+2)^2][T][N], x / resid / out: [F][H][W][C or N].
 * out = act(M . scale + shift (+ resid)) with cadre_gemm_t's act codes (0 none, 1 ReLU, bit 4: residual after act). */
int cadre_winograd_in(const float* x, float* V, int32_t F, int32_t H, int32_t W, int32_t C, int32_t m, void* stream);
int cadre_winograd_out(const float* Mx, const float* scale, const float* shift, const float* resid, float* out,
                       int32_t F, int32_t H, int32_t W, int32_t N, int32_t act, int32_t m, void* stream);
/* Winograd F(m x m, 3x3), m = 2, 3 or 4, fp32, with the plane products and the inverse transform in ONE kernel (csrc/winograd_fused.hip;
 * the stride-1 / pad-1 3x3 convs of resnet.py:26-55 and danet.py:21-41 with >= 128 channels): the (m+2)^2 product planes M never
 * reach HBM.  cadre_winograd_in_frag writes V = B^T d B in MFMA-fragment order [(m+2)^2][C/16][Tpad/16][4 kk][16 tiles][4]
 * (channel 16 c + 4 kk + e of tile 16 tb + t at float ((((xi * C/16 + c) * Tpad/16 + tb) * 4 + kk) * 16 + t) * 4 + e; Tpad = T rounded up
 * to 64; cadre_winograd_frag_elems floats); cadre_winograd_gemm_out multiplies every plane by U — (G g G^T)[xi][cout][cin] packed
 * [N/32][C/16][(m+2)^2][2 nb][4 kk][16 couts][4] (cadre_amd/encoder.py _winograd_u_frag) — and stores
 * out = act(A^T M A * scale + shift (+ resid)), act as in cadre_winograd_out.  The kernels take m 2..4, C % 32 == 0, N % 32 == 0, every
 * tensor below 2 GiB; cadre_winograd_fused_capable says so; cadre_winograd_fused_supported is the encoder's dispatch question — capability AND policy (default: m == 4, where
 * the fused form measured faster; CADRE_WINOGRAD_FUSED=2: every geometry the kernels take, 0: never). */
int cadre_winograd_fused_supported(int32_t F, int32_t H, int32_t W, int32_t Cin, int32_t N, int32_t m);
int cadre_winograd_fused_capable(int32_t F, int32_t H, int32_t W, int32_t Cin, int32_t N, int32_t m);
/* tile blocks per workgroup of the cadre_winograd_gemm_out launch for T tiles x N channels: 4 (items of 64 tiles) or 1 (few tiles: items
 * of 16 tiles on 128-thread workgroups) — the same bits either way; wino_gemm_out_kernel<m, NTB> */
int cadre_winograd_fused_ntb(int32_t T, int32_t N);
int64_t cadre_winograd_frag_elems(int32_t F, int32_t H, int32_t W, int32_t Cin, int32_t m);
int cadre_winograd_in_frag(const float* x, float* V, int32_t F, int32_t H, int32_t W, int32_t C, int32_t m, void* stream);
int cadre_winograd_gemm_out(const float* V, const float* U, const float* scale, const float* shift, const float* resid, float* out,
                            int32_t F, int32_t H, int32_t W, int32_t Cin, int32_t N, int32_t act, int32_t m, void* stream);
/* The fused front converts bytes arithmetically (fma(x, r_hi, x * r_lo), r_hi + r_lo = 1/255 split in two floats) instead of through the
 * table: counts, into *mismatches (device int32), the i in 0..255 for which that differs from lut255[i]; must be 0. */
int cadre_div255_selfcheck(const float* lut255, int32_t* mismatches, void* stream);
/* nn.MaxPool2d(3,2,1) resnet.py:114 on NHWC [F][H][W][C] (C%4==0) -> [F][Ho][Wo][C] */
int cadre_maxpool3x3s2(const float* x, float* y, int32_t F, int32_t H, int32_t W, int32_t C,
                       void* stream);
/* bf16 NHWC variant of cadre_maxpool3x3s2 (C%8==0) */
int cadre_maxpool3x3s2_bf16(const void* x, void* y, int32_t F, int32_t H, int32_t W, int32_t C,
                            void* stream);
/* PAM_Module.forward da_att.py:32-51 after the three 1x1 convs: qkv [F][Np][160] =
 * (query 16 | key 16 | value 128) comes from ONE cadre_gemm_f32 over the concatenated
 * query/key/value conv weights; this kernel does energy = q.k^T, row softmax,
 * out = att.v, y = gamma*out + x on NHWC x [F][Np][128].  Np <= 128: one workgroup per frame (the attention matrix of a frame
 * lives in one CU's LDS); 128 < Np <= 1024: a workgroup per block of 32 query rows, keys / values from global memory. */
int cadre_pam(const float* x, const float* qkv, float gamma, float* y, int32_t F, int32_t Np,
              void* stream);
/* CAM_Module.forward da_att.py:63-83 on NHWC x [F][Np][128] */
int cadre_cam(const float* x, float gamma, float* y, int32_t F, int32_t Np, void* stream);
/* same kernels writing y as bf16 (fp32 math; feeds the bf16 conv51/conv52) */
int cadre_pam_bf16out(const float* x, const float* qkv, float gamma, void* y, int32_t F, int32_t Np,
                      void* stream);
int cadre_cam_bf16out(const float* x, float gamma, void* y, int32_t F, int32_t Np, void* stream);
/* InterTaskAtt 'transformer' tail intertask_att.py:137-176: qkv [F][6][256] ordered
 * (vis_q, vis_k, vis_v, bc_q, bc_k, bc_v) -> out [F][ldo] = cat(att_visual, att_bc) */
int cadre_intertask_att(const float* qkv, float* out, int64_t ldo, int32_t F, float temperature,
                        void* stream);
/* agent.py:105-111: feat[f][512..529] = float(measurements[f][j%3]) (f64 -> f32), ld = ldo */
int cadre_append_measurements(const double* meas, float* feat, int64_t ldo, int32_t F, void* stream);

/* ---------------------------------------------------------------- storage math
 * RolloutStorage.compute_returns GAE branch (ppo_agent/storage.py:69-76), strict fp32
 * left-to-right, no FMA contraction; one lane per sequence.  Arrays are [nseq][T+1]
 * (value_preds[.][T] is overwritten with next_value like storage.py:70).  Then
 * train.py:82-88: adv = ret[:-1]-V[:-1], optional (adv-mean)/(std_unbiased+1e-8). */
int cadre_gae(const float* rewards, float* value_preds, const float* masks, const float* next_value,
              float* returns, float* adv, int32_t nseq, int32_t T, float gamma, float gamma_tau,
              int32_t normalise, void* stream);
/* feed_forward_generator gather (storage.py:99-120): obs [T+1][S][ldo] rows idx[B] ->
 * time-major x [S][B][ldx]; hn/cn/scalars gathered likewise. */
int cadre_gather_obs(const float* obs, int64_t ldo, int32_t S, const int64_t* idx, int32_t B,
                     float* x, int64_t ldx, int32_t D, void* stream);

/* storage.py:99-120 + the unpacking at agent.py:167-168 in one launch: every field of one head's
 * minibatch (rows idx[0..B) of the storage) is written into rows b0..b0+B of the Bt-row packed
 * update workspace: X [S][Bt][ldx] time-major, h0/c0 [Bt][ldho], scalars [Bt]. */
int cadre_gather_minibatch(const float* obs, int64_t ldo, int32_t S, const float* hn, const float* cn,
                           int64_t ldh, const int64_t* action, const float* value_preds,
                           const float* returns, const float* logp, const int32_t* command,
                           const float* adv, const int64_t* idx, int32_t B, int32_t D, int32_t Hd,
                           int32_t Bt, int32_t b0, float* X, int64_t ldx, float* h0, float* c0,
                           int64_t ldho, int64_t* actions_o, int32_t* commands_o, float* old_values_o,
                           float* returns_o, float* old_logp_o, float* adv_o, void* stream);

/* The same gather for every worker and both heads in one launch (train.py:93-102 iterates workers; each worker's two
 * feed_forward_generators storage.py:93-120).  src_table: n_src = 2 * workers records of nine device pointers
 * {obs, hn, cn, action, value_preds, returns, action_log_probs, command, adv} for (worker, head) = (i / 2, i % 2);
 * idx [n_src][Bw] row ids; worker w fills rows w*Bw .. of the Bt-row batch; outputs are [2][...] arrays of the two
 * heads (X / h0 / c0 with the given head strides, the scalars [2][Bt]). */
int cadre_gather_minibatch_multi(const void* src_table, int32_t n_src, int64_t ldo, int32_t S, int64_t ldh,
                                 const int64_t* idx, int32_t Bw, int32_t D, int32_t Hd, int32_t Bt, float* X,
                                 int64_t x_head_stride, int64_t ldx, float* h0, float* c0, int64_t h_head_stride,
                                 int64_t ldho, int64_t* actions_o, int32_t* commands_o, float* old_values_o,
                                 float* returns_o, float* old_logp_o, float* adv_o, void* stream);
/* The same gather with the stable counting sort by command and the placement of cadre_sort_rows_by_command +
 * cadre_permute_minibatch in the SAME launch: row (worker, b) of head hd lands at its sorted position in the [2][...] workspace
 * arrays, pos[hd * Bt + row] gets that position and seg[2 * (hd * C + c)] = (first row, rows) of command c (agent.py:166-237's
 * per-command nets each read one run of rows).  Needs (n_src / 2) * Bw == Bt. */
int cadre_gather_sorted_multi(const void* src_table, int32_t n_src, int64_t ldo, int32_t S, int64_t ldh,
                              const int64_t* idx, int32_t Bw, int32_t D, int32_t Hd, int32_t Bt, int32_t C, float* X,
                              int64_t x_head_stride, int64_t ldx, float* h0, float* c0, int64_t h_head_stride,
                              int64_t ldho, int64_t* actions_o, int32_t* commands_o, float* old_values_o,
                              float* returns_o, float* old_logp_o, float* adv_o, int32_t* pos, int32_t* seg, void* stream);

/* Recurrent weights W [4*D][ldw = 544] of `Z` nets (net z at + z * w_str) -> MFMA FRAGMENT ORDER for the fused LSTM
 * steps, both directions (ppo_update.hip): `fwd` for cadre_lstm_step_fwd, `bwd` (the transpose: the backward reduces
 * over the gate axis) for cadre_lstm_step_bwd; ceil(D / 16) * 4 * 34 * 256 floats per net each, net stride p_str.
 * Once per optimiser step — a fragment read from the row-major matrix is 16 rows x 64 bytes per wave instruction, from
 * these copies one contiguous KiB. */
int cadre_pack_lstm_weights(const float* W, int64_t w_str, int32_t ldw, int32_t D, int32_t Z, float* fwd, float* bwd,
                            int64_t p_str, void* stream);
/* One LSTM time step of `Z` nets, product AND cell math in one launch (ppo_update.hip; models.py:139-152 nn.LSTMCell):
 * gates = G (x-projection + b_ih, [B][ldg] per net) + h_{t-1} W^T + bias; G <- (i, f, g, o) activated; c_t, h_t,
 * tanh(c_t) written.  Wp: the packed weights (`fwd` of cadre_pack_lstm_weights); K = ldh = 544 (hidden D zero padded);
 * net z at + z * *_str.  row_seg [Z][2] (may be NULL): (first row, count) of net z's run of rows — exactly those rows
 * are read and written (16-row tiles from the run's first row).
 * `rev` (0/1): order in which the unit slices are walked — alternate it from step to step so that each XCD's L2
 * re-uses the most recently read weights first (speed only). */
int cadre_lstm_step_fwd(const float* Wp, int64_t wp_str, const float* bias, int64_t b_str, float* G, int32_t ldg,
                        int64_t g_str, const float* Hprev, const float* Cprev, float* Hout, float* Cout, float* TCout,
                        int32_t ldh, int64_t h_str, int32_t B, int32_t D, int32_t Z, const int32_t* row_seg,
                        int32_t rev, void* stream);
/* The MLP towers of update_policy (models.py:171-177 critic, distributions.py:34-40 actor; Linear(530 -> 128) ReLU
 * Linear(128 -> 128) ReLU Linear(128 -> n_out), n_out rows padded to 64, input rows [B][ldh = 544]) for Z2 = 2 * nets
 * towers in three launches (ppo_update.hip).  Tower z2 = 2*net + {actor, critic} at P + z2 * t_str; offs[6] = float
 * offsets of (W1, b1, W2, b2, W3, b3) inside a tower, weights [out][in] row-major; A1 / A2 / dA1 / dA2 [Z2][B][128],
 * O3 / dO3 [Z2][B][64]; net z's input rows at Hin + z * h_str.  row_seg [Z2/2][2] (may be NULL = every row): (first
 * row, count) of the run of rows net z owns in the row-sorted minibatch — no other row is read or written.
 * cadre_mlp_fwd: A1, A2 (post-ReLU) and O3.  cadre_mlp_bwd: dA2 = (dO3 W3) [A2 > 0], dA1 = (dA2 W2) [A1 > 0] and
 * dH[net] = sum over the net's two towers of dA1 W1 ([B][ldh] at dH + net * d_str).  cadre_mlp_dw: the gradients of the
 * six parameter blocks of every tower at G + z2 * t_str + offs[...] (written, not accumulated). */
int cadre_mlp_fwd(const float* P, int64_t t_str, const int32_t* offs, const float* Hin, int32_t ldh, int64_t h_str,
                  float* A1, float* A2, float* O3, int32_t B, int32_t Z2, const int32_t* row_seg, void* stream);
int cadre_mlp_bwd(const float* P, int64_t t_str, const int32_t* offs, const float* dO3, const float* A1, const float* A2,
                  float* dA1, float* dA2, float* dH, int32_t ldh, int64_t d_str, int32_t B, int32_t Z2,
                  const int32_t* row_seg, void* stream);
int cadre_mlp_dw(const float* dO3, const float* dA2, const float* dA1, const float* A2, const float* A1, const float* Hin,
                 int32_t ldh, int64_t h_str, float* G, int64_t t_str, const int32_t* offs, int32_t B, int32_t Z2,
                 const int32_t* row_seg, void* stream);
/* One backward time step: dh_{t-1} = dG_t W (+ dh_in) on the packed transposed weights (`bwd` of
 * cadre_pack_lstm_weights) and dG_t in fragment order (dGp_in: what the previous call left in its dGp_out), then the
 * cell backward of step t-1 in the same launch: dG_{t-1} from the activated gates G_act, tanh(c_{t-1}), c_{t-2}, written
 * row-major (dG_out, [B][ldg]) AND in fragment order (dGp_out: ceil(B / 16) tiles of 16 x 2176 floats per net, net stride
 * gp_str; zero-initialised by the caller, pad never written); dC in/out.  dGp_in == NULL: no product (first step:
 * dh = dh_in, the gradient of the MLP towers).  commands [2][B] (may be NULL): net z = head*C + c keeps only rows whose
 * command is c, the others get dG = 0 and dc = 0 (unsorted minibatch); row_seg [Z][2] (may be NULL): (first row, count) of
 * net z's run in the row-sorted minibatch — exactly those rows are read and written, and the 16-row tiles of dGp count
 * from the run's first row. */
int cadre_lstm_step_bwd(const float* Wp, int64_t wp_str, const float* dGp_in, float* dGp_out, int64_t gp_str,
                        float* dG_out, const float* G_act, int32_t ldg, int64_t g_str, const float* dh_in, float* dC,
                        int64_t d_str, const float* TC, const float* Cprev, int32_t ldh, int64_t h_str, int32_t B, int32_t D,
                        int32_t Z, const int32_t* commands, int32_t C, const int32_t* row_seg, int32_t rev, void* stream);
/* The LSTM weight gradients of `Z` nets in one launch (ppo_update.hip): dWhh = sum_t dG_t^T h_{t-1}, dWih = sum_t
 * dG_t^T x_t ([H4][ldw] each, columns 0..N-1 written), dbih = dbhh = column sums of dG.  dG [S][B][ldg] (gate axis zero
 * padded to a multiple of 64), Hs [S+1][B][ldh] (slot t = h_{t-1}), X [S][B][ldh] (net z reads input slot z / x_div).
 * row_seg: only rows [beg & ~3, beg + cnt) of every time step are multiplied — the rows of other nets hold exact zeros. */
int cadre_lstm_dw(const float* dG, int32_t ldg, int64_t g_str, const float* Hs, const float* X, int32_t ldh,
                  int64_t h_str, int64_t x_str, int32_t x_div, float* dWhh, float* dWih, float* dbih, float* dbhh,
                  int32_t ldw, int64_t w_str, int32_t B, int32_t S, int32_t H4, int32_t N, int32_t Z,
                  const int32_t* row_seg, void* stream);
/* column sums: out[z][n] (+)= sum_m X[z][m][n]  (bias gradients) */
int cadre_colsum(const float* X, int64_t ldx, int64_t x_str, float* out, int64_t o_str,
                 int32_t M, int32_t N, int32_t batch, int32_t accumulate, void* stream);
/* y = relu'(act) * dy elementwise, in place on dy.  Optional ownership mask (commands != NULL): the
 * buffers are [2*Z][B][hid] (tower z = 2*net + t, net = head*C + c); rows whose command differs from
 * the net's c are set to exactly 0. */
int cadre_relu_bwd(const float* act, float* dy, int64_t n, const int32_t* commands, int32_t B,
                   int32_t hid, int32_t C, void* stream);
/* Stable sort of each head's B minibatch rows by command: pos[hd][b] = sorted position of row b,
 * seg[hd*C + c] = (begin, count) of command c.  commands/pos are [2][B] i32, seg is [2*C][2] i32. */
int cadre_sort_rows_by_command(const int32_t* commands, int32_t B, int32_t C, int32_t* pos, int32_t* seg,
                               void* stream);
/* Apply pos to the packed minibatch (cadre_gather_minibatch outputs) of `heads` heads in one launch: dst row pos[b] = src
 * row b for X [S][B][ldx], h0/c0 [B][ldh] and the six per-row scalars; head h reads and writes X + h * x_hstr,
 * h0 / c0 + h * h_hstr, pos and the scalars + h * B. */
int cadre_permute_minibatch(const int32_t* pos, int32_t B, int32_t S, const float* X, float* Xo, int64_t ldx,
                            const float* h0, const float* c0, float* h0o, float* c0o, int64_t ldh,
                            const int64_t* actions, const int32_t* commands, const float* old_values,
                            const float* returns, const float* old_logp, const float* adv, int64_t* actions_o,
                            int32_t* commands_o, float* old_values_o, float* returns_o, float* old_logp_o,
                            float* adv_o, int32_t heads, int64_t x_hstr, int64_t h_hstr, void* stream);

/* ---------------------------------------------------------------- policy head + PPO loss
 * Replaces Model.evaluate_actions (models.py:199-208), Categorical_1d (distributions.py:
 * 66-105) and the loss of CadreAgent.update_policy (agent.py:166-229) forward AND backward.
 * Per head hd in {0:steer,1:throttle}, command net c in 0..C-1 (net = hd*C+c; C = command_num, the reference ships 4):
 *   logits: net n row b at logits + n*l_ns + b*ldl (first n_out[hd] columns valid),
 *   values: net n row b at values + n*v_ns + b*ldv.  Sample arrays are [2 heads][B].
 * Outputs: losses[3] = (value_loss*value_coeff, action_loss*clip_coeff, ent*ent_coeff),
 *   dlogits / dvalues (same addressing) = d total_loss / d(raw logits | critic output);
 *   whole dlogits rows (ldl columns) are written, only column 0 of a dvalues row.
 * `inv_b` = 1/(rows per worker minibatch) (sum of per-worker means).
 * `scratch`: 4 + 6 * ceil(B / 16) floats of device memory (arrival counter + per-workgroup partial sums: the losses are
 * summed in a fixed order whatever the order the workgroups finish in; scratch[0] must be ZERO before the first launch on a
 * scratch buffer — every launch leaves it zero again, so consecutive launches need no clearing).  `poison` (may be NULL): device int32; when
 * nonzero (the status word of the A/B build's cadre_lstm_seq_fwd) the three losses come out NaN. */
int cadre_ppo_loss(const float* logits, int64_t ldl, int64_t l_ns, const float* values, int64_t ldv,
                   int64_t v_ns, const int64_t* actions, const int32_t* commands, const float* old_values, const float* returns,
                   const float* old_logp, const float* adv, int32_t B, int32_t C, int32_t n_out_steer,
                   int32_t n_out_throttle, float clip, float value_coeff, float clip_coeff,
                   float ent_coeff, float inv_b, float* losses, float* dlogits, float* dvalues,
                   float* scratch, const int32_t* poison, void* stream);
/* Model.act sampling (models.py:184-189, distributions.py:96-99) == argmax(p/q), q supplied
 * by the host from the torch CPU generator.  logits [R][ldl] raw; outputs action i64 [R],
 * log_prob f32 [R] of the sampled action. */
int cadre_sample(const float* logits, int64_t ldl, const float* q, int64_t ldq, int32_t R,
                 int32_t n_out, int64_t* action, float* logp, void* stream);

/* Model.evaluate_actions forward (models.py:199-208): per row log-prob of `actions` and entropy
 * of Categorical(logits=raw logits) */
int cadre_categorical_eval(const float* logits, int64_t ldl, const int64_t* actions, int32_t R,
                           int32_t n_out, float* logp, float* entropy, void* stream);
/* Categorical_1d.forward (distributions.py:66-83): what Categorical(logits=raw) exposes — normalised
 * `logits` = raw - logsumexp(raw) and `probs` = softmax(logits), [R][n_out] each (dense rows), plus the
 * per-row argmax of probs (first maximum, `mode`).  Any of the three outputs may be NULL. */
int cadre_categorical_dist(const float* raw, int64_t ldl, int32_t R, int32_t n_out, float* logits_out,
                           float* probs_out, int64_t* mode_out, void* stream);

/* Initial LSTM state of every net z < Z: Hs[z*z_str ..] <- h0[(z / x_div)*n_per ..], same for Cs <- c0 (n_per floats
 * each, agent.py:166-175 hidden_state_batch shared by a head's command nets) and dC (Z*n_per floats, may be NULL) <- 0 */
int cadre_lstm_init(const float* h0, const float* c0, float* Hs, float* Cs, float* dC, int64_t n_per,
                    int64_t z_str, int32_t x_div, int32_t Z, void* stream);

/* ---------------------------------------------------------------- optimiser
 * chief.py:13-21 + main.py:55: per-model clip_grad_norm_(max_norm) then Adam (torch defaults)
 * over a flat parameter arena made of `n_models` segments [seg_off[i], seg_off[i+1]).
 * norms2 is f64 [n_models] scratch (zeroed inside). */
int cadre_clip_adam(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                    const int64_t* seg_off, int32_t n_models, double* norms2, double max_norm,
                    double lr, double beta1, double beta2, double eps, int32_t step, void* stream);

/* Same math, hipGraph-capturable: the 1-based step counter is *step_dev (device int32,
 * incremented by the call); norms2 must hold n_models + 2 doubles. */
int cadre_clip_adam_graph(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                          const int64_t* seg_off, int32_t n_models, double* norms2, double max_norm,
                          double lr, double beta1, double beta2, double eps, int32_t* step_dev,
                          void* stream);


/* cadre_clip_adam_graph that ALSO maintains the fragment-order copies of the recurrent weights (cadre_pack_lstm_weights'
 * `fwd` / `bwd`, net stride p_str): models 0 .. n_lstm-1 are LSTM blocks of lstm_str floats whose W_hh [H4 = 4 D][ldw = 544]
 * starts o_whh floats into the block.  The thread that steps a 4 x 4 block of W_hh stores it into both copies, so the
 * update needs no packing launch after an optimiser step (chief.py:13-21 + models.py:139-152's weights for the next step).
 * Same parameters, bit for bit, as cadre_clip_adam_graph followed by cadre_pack_lstm_weights. */
int cadre_clip_adam_pack_graph(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                               const int64_t* seg_off, int32_t n_models, double* norms2, double max_norm, double lr,
                               double beta1, double beta2, double eps, int32_t* step_dev, int32_t n_lstm,
                               int64_t lstm_str, int64_t o_whh, int32_t H4, int32_t ldw, int32_t D, float* fwd,
                               float* bwd, int64_t p_str, void* stream);

/* Data-parallel ranks with a reduce-scattered gradient arena (reference semantics chief.py:13-21: SUM
 * over workers, per-model clip, Adam): this rank owns arena elements [rlo, rhi) (multiples of 4).
 * cadre_clip_adam_norms: increments *step_dev, stores the step's bias-correction scalars in norms2[n_models..+2) and
 * leaves the PARTIAL per-model square norms of grads[rlo..rhi) in norms2[0..n_models) — the caller all-reduces those
 * n_models doubles (SUM) over the ranks.  cadre_clip_adam_apply: clip + Adam on the shard; exp_avg / exp_avg_sq hold
 * the shard's state only (rhi - rlo floats each, element i of the arena at [i - rlo]); params / grads are the full
 * arenas.  The caller then all-gathers the parameter shards.  With [0, total) the pair equals cadre_clip_adam_graph. */
int cadre_clip_adam_norms(const float* grads, const int64_t* seg_off, int32_t n_models, double* norms2, double lr,
                          double beta1, double beta2, int32_t* step_dev, int64_t rlo, int64_t rhi, void* stream);
int cadre_clip_adam_apply(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                          const int64_t* seg_off, int32_t n_models, const double* norms2, double max_norm,
                          double beta1, double beta2, double eps, int64_t rlo, int64_t rhi, void* stream);

#ifdef __cplusplus
}
#endif
#endif
