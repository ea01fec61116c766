// winograd_c64.hip — FUSED Winograd F(2x2, 3x3) for the fp32 model's 64 -> 64 stride-1 3x3 convs (layer1:
// carla_perception/Networks/danet_blocks/resnet.py:26-55 — conv3x3 + folded BN (+ residual) + ReLU), gfx950.
//
// The unfused form (winograd.hip) does not pay at 64 channels: the transform-domain planes are 4x (F(2x2)) / 2.78x
// (F(3x3)) the activations and a 64-channel layer has too few FLOPs per byte to carry them through HBM.  Here nothing
// of the transform domain leaves the CU:
//   * workgroup = 4 waves, ONE per SIMD with 512 registers each; a wave owns 16 tiles (2x2 outputs each) x all 64 output
//     channels x all 16 transform planes = 16 x 4 v_mfma_f32_16x16x4_f32 accumulator blocks = 256 registers (the AGPR
//     file) — the inverse transform is register-local: a lane holds every plane of its (tile, channel);
//   * the 64 input channels go by in eight chunks (steps) of 8: a lane loads the 4x4 input patch of ITS tile for FOUR
//     channels every other step (16 loads of 16 bytes; the four k-lanes of a tile read 64 contiguous bytes), transforms two
//     channels per step in registers (B^T d B: 32 adds per channel) and feeds the MFMAs' A operand directly;
//   * the transformed weights U = G g G^T (host, float64 -> fp32; [chunk][plane][position][8 cin], 32 KB per chunk; position
//     16 b + n = output channel 4 n + b) stream through four LDS buffers by LDS-DMA, two chunks ahead, shared by the four waves;
//     a B fragment is one ds_read_b64 of 512 contiguous bytes (conflict-free);
//   * the memory requests of step t+2 (8 weight pieces, 16 patch pixels per wave) are issued ONE PER PLANE between the MFMAs
//     of step t: one counted s_waitcnt vmcnt + one raw barrier per 512 MFMAs (128 per wave);
//   * the epilogue (inverse transform, BN, residual, ReLU) stores 16 bytes per lane: MFMA block b, column n is output
//     channel 4n + b, so a lane holds four consecutive channels of a pixel.
// 2.25x fewer multiplies than the direct conv, rounding error BELOW the direct conv's.
//
// Built with -mllvm -enable-misched=0 -mllvm -pragma-unroll-threshold=262144 (cadre_amd/build.py EXTRA_FLAGS): the source is
// written in issue order (the machine scheduler's reordering cost 40-100 VGPR spills), and the step loop must be unrolled
// (register sets indexed by step parity).  Measured per 1024 frames of 72 x 72, without / with residual, against 2.97 ms
// for the direct window kernel (tools/wino_c64_ab.py, tools/wino_c64_ablate.py, profiles/r04_wino_c64_*.txt):
//   3.09 / 3.20 ms  with the machine scheduler
//   2.50 / 2.62     without it, requests issued as a burst of 24 per wave in front of the MFMA block
//   1.92 / 2.33     requests one per plane between the MFMAs: the burst kept all four waves of the CU in the address queue —
//                   not issuing MFMAs — while the texture addresser worked through 96 requests ("no patch loads" had run 1.65)
//   1.89 / 1.97     accumulators pinned to the AGPR file (inline-asm MFMAs, pinned epilogue reads): hipcc had moved ~150
//                   v_accvgpr_read of later epilogue groups to the top of the epilogue and spilled patch registers to make room
//                   (a spill store waits for the load that fills the register: a memory round trip inside the MFMA block);
//                   residuals of tiles 0, 1 requested inside the last MFMA block; 256 -> 139 VGPRs, no spills
//   1.84 / 1.90     16-byte stores and residual loads (the cout permutation above): 128 instead of 512 requests per item and CU
//   1.74 / 1.80     16-byte patch requests (four channels of a pixel = two steps' worth, every other step; U's cin axis permuted to
//                   match): half the requests and cache lines touched per byte
//   1.66 / 1.75     no control flow in the item loop: max(x, floor) instead of a uniform `if (relu)` per store (32 basic blocks in
//                   the epilogue), the tile / offset arithmetic on selects, the inverse transform on packed pairs (rocprof in
//                   the bench: 1.75 / 1.79 ms per launch)
// The MFMAs with their LDS fragment reads alone run 1.33 ms; what is left (profiles/r04_wino_c64_ablation_branch_free.txt): patch
// requests 0.18 (every pixel is requested by the four tiles that overlap it: FETCH 1.92 GB against 1.36 algorithmic), epilogue
// 0.15, weight DMA 0.06, wait + barrier 0.07.  Lessons in the code: separate LDS objects per DMA buffer (one object = s_waitcnt vmcnt(0) before
// every fragment read), no branch around loads (PHI copies wait for memory on the spot), scheduling fences around the MFMA
// block, contiguous item ranges per workgroup (halo rows from the workgroup's own L1 / L2).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/cadre_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

int cadre_fail(const char* msg);

#ifndef W2_ASM_MFMA
#define W2_ASM_MFMA 1
#endif
// W2_PIPE = 1: the NEXT step's input transform rides between this step's MFMAs, one packed add behind an MFMA (and the patch requests
// move a step earlier).  MEASURED AND NOT ADOPTED (round 6, tools/w2_ablate.py, bit-identical): 1.72 / 1.82 ms against 1.66 / 1.74 without /
// with residual.  The reason is a property of the fp32 matrix instructions (tools/dbg/mfma_shadow.py, profiles/r06_mfma_shadow.txt): a
// vector-ALU instruction does NOT run in the shadow of a v_mfma_f32_16x16x4_f32 — the first one behind an MFMA costs 13 cycles of the
// 32, every further one 4 (8 for v_mov_b32), with one wave per SIMD and with two; s_nop and ds_read are free.  fp32 MFMA and fp32 VALU
// share the multipliers (the datasheet's vector and matrix fp32 rates are the same number).  So the 32 packed adds cost what they cost
// at the top of the step, plus the 13-cycle switch sixteen times instead of once.
#ifndef W2_PIPE
#define W2_PIPE 0
#endif
// W2_SOFF = 1 (round 6): no vector instruction per patch request inside the MFMA blocks — the 16 patch offsets of a lane (with the
// out-of-range select) are formed ONCE per item, the double-step's channel offset rides in the load's scalar offset (and the per-pixel
// masks no longer sit in 32 SGPRs: 75 -> 15 scalars spilled to VGPR lanes).  Bit-identical, 1.66 -> 1.57 / 1.75 -> 1.66 ms without / with
// residual (tools/w2_ablate.py).  0: the round-4 form (add + select per request).
#ifndef W2_SOFF
#define W2_SOFF 1
#endif
#ifndef W2_ABL
#define W2_ABL 0      // tools/w2_ablate.py: 1 no MFMA, 2 no patch loads, 4 no weight DMA, 8 no epilogue, 16 no wait + barrier, 64 no input transform, 128 no fragment reads after a step's first
#endif

struct w2_args {
  const float* x;        // [F][H][W][64]
  const float* U;        // [8 chunks][16 planes][64 positions: 16 b + n = cout 4 n + b][8 cin]
  const float* scale;    // [64] folded BN (may be null: 1)
  const float* shift;    // [64] (may be null: 0)
  const float* resid;    // [F][H][W][64] or null: added before the activation
  float* out;            // [F][H][W][64]
  int F, H, W, TH, TW;
  int ntiles;            // F * TH * TW
  int ngroups;           // ceil(ntiles / 16)
  int relu;
};

// The 64 accumulator blocks are pinned to the AGPR file ("+a"): left to the register allocator, hipcc moved blocks between
// AGPRs and VGPRs inside the steps around the epilogue (v_accvgpr_read + s_nop 8 behind every second MFMA, the patch
// registers spilled to make room).  Inline asm hides the MFMA from the hazard recognizer: the only read of an accumulator by
// the vector ALU is the epilogue's, behind mfma_drain().
__device__ __forceinline__ void mfma_acc(f32x4& acc, float av, float bv, bool first) {
#if W2_ASM_MFMA && defined(__HIP_DEVICE_COMPILE__)
  if (first) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=a"(acc) : "v"(av), "v"(bv));
  else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(av), "v"(bv));
#else
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, first ? zero4 : acc, 0, 0, 0);
#endif
}
// (an empty asm that "rewrites" the block right where a group reads it: hipcc otherwise moves ~150 v_accvgpr_read of LATER
// groups to the top of the epilogue — and spills patch registers, whose spill stores wait for the loads that fill them)
__device__ __forceinline__ void acc_pin(f32x4& acc) {
#if W2_ASM_MFMA && defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+a"(acc));
#endif
}
__device__ __forceinline__ void mfma_drain() {
#if W2_ASM_MFMA && defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");     // (an 8-pass MFMA's result is readable by the vector ALU 18 wait states later at most)
#endif
}

template <bool RES>
__global__ __launch_bounds__(256, 1) void wino2_c64_kernel(w2_args a) {
  // four 32 KB weight-chunk buffers as SEPARATE objects: hipcc then knows that the LDS-DMA into one does not alias the
  // fragment reads of another and does not put s_waitcnt vmcnt(0) in front of every step's first ds_read.  The chunk of
  // step t+2 is requested at step t (a step is 128 MFMAs per wave = 1.8 us: less than an L2 round trip under load)
  __shared__ __attribute__((aligned(16))) char ubuf0[32768];
  __shared__ __attribute__((aligned(16))) char ubuf1[32768];
  __shared__ __attribute__((aligned(16))) char ubuf2[32768];
  __shared__ __attribute__((aligned(16))) char ubuf3[32768];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, q = lane >> 4;
  constexpr unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)((long long)a.F * a.H * a.W * 256), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsU = __builtin_amdgcn_make_buffer_rsrc((void*)a.U, 0, 8 * 32768, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, (int)((long long)a.F * a.H * a.W * 256), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void*)(RES ? a.resid : a.x), 0, (int)((long long)a.F * a.H * a.W * 256), 0x00020000);
  // a workgroup walks a CONTIGUOUS range of items (64 tiles each): the input rows its tiles share with the tile row above
  // were fetched by itself half an item earlier — they come from its own L1 / its XCD's L2, not from HBM again
  const int nitems = (a.ngroups + 3) >> 2;
  const int per_wg = (nitems + (int)gridDim.x - 1) / (int)gridDim.x;
  const int item0 = (int)blockIdx.x * per_wg;
  const int my_items = min(per_wg, nitems - item0);
  if (my_items <= 0) return;
  const int thw = a.TH * a.TW;

  // ---- per-item lane state of the A side: tile = group * 16 + n, patch origin (2ty-1, 2tx-1), channel slice 4q of a chunk
  // tile -> (frame, tile row, tile column) by float reciprocals + one correction step each (tile < 2^23: exact in a float; the
  // quotient estimate is off by at most one): 9 vector instructions per division where the integer division takes 17 — five tile
  // decodes per item sit inside the MFMA blocks, and vector instructions are paid in matrix cycles (round 6)
  const float inv_thw = 1.0f / (float)thw, inv_tw = 1.0f / (float)a.TW;
  auto decode = [&](int t, int& f, int& ty, int& tx) {
    int qf = (int)((float)t * inv_thw);
    int rem = t - qf * thw;
    qf += (rem >= thw) ? 1 : 0; qf -= (rem < 0) ? 1 : 0;
    rem = t - qf * thw;
    int qy = (int)((float)rem * inv_tw);
    int rx = rem - qy * a.TW;
    qy += (rx >= a.TW) ? 1 : 0; qy -= (rx < 0) ? 1 : 0;
    rx = rem - qy * a.TW;
    f = qf; ty = qy; tx = rx;
  };
  int a_base = 0;                 // byte offset of patch pixel (0, 0), channel 4q of double-step 0 (may be negative: masked pixels only)
  unsigned a_rows = 0, a_cols = 0;
  unsigned a_voff[16];            // (W2_SOFF) byte offset of each patch pixel at double-step 0, or the out-of-range bit
  auto plan_a = [&](int item_l) {      // (branch-free: it runs in front of step 6's MFMA block)
    const int item = item0 + item_l;
    const int tile = (item * 4 + wave) * 16 + n;
    const bool live = (item_l < my_items) & (tile < a.ntiles);
    const int tl = live ? tile : 0;
    int f, ty, tx;
    decode(tl, f, ty, tx);
    const int r0 = 2 * ty - 1, c0 = 2 * tx - 1;
    unsigned rows = 0, cols = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      rows |= ((unsigned)(r0 + i) < (unsigned)a.H) ? 1u << i : 0u;
      cols |= ((unsigned)(c0 + i) < (unsigned)a.W) ? 1u << i : 0u;
    }
    a_rows = live ? rows : 0u;
    a_cols = live ? cols : 0u;
    a_base = ((f * a.H + r0) * a.W + c0) * 256 + 16 * q;
    if constexpr (W2_SOFF != 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          a_voff[4 * i + j] = ((a_rows >> i) & (a_cols >> j) & 1u) ? (unsigned)(a_base + (i * a.W + j) * 256) : OOB;
    }
  };
  // patches are requested by DOUBLE-STEP d (16 input channels): a lane asks for the four channels 16d + 4q .. + 3 of each of its
  // 16 patch pixels at once (16 bytes: half the requests — and cache lines touched — per byte of the 8-byte form, measured
  // 1.84 / 1.94 -> 1.77 / 1.82 ms); components 2e, 2e + 1 feed step 2d + e (U's cin axis is laid out to match).  Two register sets by
  // parity of d: the set of d + 1 is requested during step 2d (one request per plane between the MFMAs), a full step ahead of its use
  f32x4 dq2[2][16];
  auto request_q1 = [&](int d, f32x4* dq, int i, int j) {
    const unsigned okm = (a_rows >> i) & (a_cols >> j) & 1u;
    unsigned off = okm ? (unsigned)(a_base + (i * a.W + j) * 256 + d * 64) : OOB;
    if constexpr ((W2_ABL & 32) != 0) off = (unsigned)(((i * 4 + j) * 256 + d * 64 + 16 * q) + n * 4096);
    if constexpr (W2_SOFF != 0 && (W2_ABL & 34) == 0) {
      dq[4 * i + j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, (int)a_voff[4 * i + j], d * 64, 0));
      return;
    }
    if constexpr ((W2_ABL & 2) == 0) dq[4 * i + j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, (int)off, 0, 0));
    else dq[4 * i + j] = f32x4{(float)off, (float)d, (float)off, (float)d};
  };
  auto request_u1 = [&](int chunk, int buf, int i) {      // one 1 KB piece of this wave's quarter of the chunk
    const int piece = wave * 8 + i;
    if constexpr ((W2_ABL & 4) == 0)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsU, (__attribute__((address_space(3))) void*)((buf == 0 ? ubuf0 : buf == 1 ? ubuf1 : buf == 2 ? ubuf2 : ubuf3) + piece * 1024), 16,
                                             lane * 16, chunk * 32768 + piece * 1024, 0, 0);
  };
  auto request_u = [&](int chunk, int buf) {      // this wave's quarter (8 KB) of the chunk
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int piece = wave * 8 + i;              // 32 pieces of 1 KB
      if constexpr ((W2_ABL & 4) == 0)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsU, (__attribute__((address_space(3))) void*)((buf == 0 ? ubuf0 : buf == 1 ? ubuf1 : buf == 2 ? ubuf2 : ubuf3) + piece * 1024), 16,
                                               lane * 16, chunk * 32768 + piece * 1024, 0, 0);
    }
  };

  f32x4 acc[16][4];               // [plane][16-channel block]: rows = tiles 4q + r, column = channel 16*blk + n
  float sc[4], sh[4];
  const float act_floor = a.relu ? 0.f : -__builtin_inff();
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    sc[b] = a.scale ? a.scale[4 * n + b] : 1.f;
    sh[b] = a.shift ? a.shift[4 * n + b] : 0.f;
  }

  // ---- prologue: weights chunk 0 and the first item's first two patches
  plan_a(0);
  request_u(0, 0);
  request_u(1, 1);
#pragma unroll
  for (int k = 0; k < 16; ++k) request_q1(0, dq2[0], k >> 2, k & 3);
  if constexpr (W2_PIPE != 0) {                    // (the transform of step 2 runs during step 1: double-step 1 is due a step earlier)
#pragma unroll
    for (int k = 0; k < 16; ++k) request_q1(1, dq2[1], k >> 2, k & 3);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // Input transform V = B^T d B of one step's two channels (16 planes, packed pairs).  W2_PIPE: V lives across the steps — the 32 packed
  // adds of step t+1's transform are issued ONE OR TWO PER PLANE behind the MFMAs of step t (row `i` of the column pass in front of
  // plane 4i, plane k's value into V[k] once plane k's MFMAs have been issued and the next plane's are in the pipe): with one wave
  // per SIMD a transform at the top of the step ran with the matrix pipe empty (0.10-0.12 ms of 1.84 per 1024 frames).
  f32x2 V[16];
  f32x2 tt[2][4];                  // two rows of the column pass
  auto dn_of = [&](int cn, int k) -> f32x2 {       // patch pixel k, the two channels of step cn (cn = 8: the next item's step 0)
    const f32x4& d4 = dq2[((cn & 7) >> 1) & 1][k];
    return (cn & 1) ? f32x2{d4[2], d4[3]} : f32x2{d4[0], d4[1]};
  };
  // (packed adds as asm volatile: they stay exactly where the source puts them — ONE behind an MFMA, in its 32-cycle shadow; left to
  // hipcc the twelve of a row came out scalar and in one run between two MFMAs: slower than the transform at the top of the step)
  auto pk_add = [](f32x2 x, f32x2 y) -> f32x2 {
    f32x2 r;
    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
  };
  auto pk_sub = [](f32x2 x, f32x2 y) -> f32x2 {
    f32x2 r;
    asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y));
    return r;
  };
  auto tt_one = [&](int cn, int i, int j) {       // element j of row i of tt = B^T d
    if (i == 0) tt[0][j] = pk_sub(dn_of(cn, 0 + j), dn_of(cn, 8 + j));
    if (i == 1) tt[1][j] = pk_add(dn_of(cn, 4 + j), dn_of(cn, 8 + j));
    if (i == 2) tt[0][j] = pk_sub(dn_of(cn, 8 + j), dn_of(cn, 4 + j));
    if (i == 3) tt[1][j] = pk_sub(dn_of(cn, 4 + j), dn_of(cn, 12 + j));
  };
  auto tt_row = [&](int cn, int i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) tt_one(cn, i, j);
  };
  auto v_of = [&](int k) -> f32x2 {               // V[k] from row k >> 2 of tt
    const f32x2* t = tt[(k >> 2) & 1];
    return (k & 3) == 0 ? pk_sub(t[0], t[2]) : (k & 3) == 1 ? pk_add(t[1], t[2]) : (k & 3) == 2 ? pk_sub(t[2], t[1]) : pk_sub(t[1], t[3]);
  };
  if constexpr (W2_PIPE != 0) {                   // step 0 of the first item
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      tt_row(0, i);
#pragma unroll
      for (int j = 0; j < 4; ++j) V[4 * i + j] = v_of(4 * i + j);
    }
  }

  for (int item_l = 0; item_l < my_items; ++item_l) {
    // (byte offsets of the item's output pixels: formed inside step 6's MFMA block — two integer divisions per tile that would
    // otherwise run in the epilogue, where nothing hides them)
    unsigned eo[4][4];
    auto offsets = [&](int r, unsigned (&o)[4]) {
      const int tile = ((item0 + item_l) * 4 + wave) * 16 + 4 * q + r;
      int f, ty, tx;
      decode(tile, f, ty, tx);
      const int y0 = 2 * ty, x0 = 2 * tx;
      const bool tv = tile < a.ntiles;
      const unsigned e00 = (unsigned)((((f * a.H + y0) * a.W + x0) * 64 + 4 * n) * 4);
#pragma unroll
      for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx)
          o[2 * dy + dx] = ((int)tv & (int)(y0 + dy < a.H) & (int)(x0 + dx < a.W)) ? e00 + (unsigned)((dy * a.W + dx) * 256) : OOB;      // (bitwise: no control flow inside the MFMA block)
    };
#pragma unroll
    for (int c = 0; c < 8; ++c) {                 // (unrolled: chunk, LDS buffer c & 1 and "first chunk" are compile-time)
      // ---- input transform of this step's patch: V = B^T d B (per channel), 16 planes x 4 channels
      if constexpr (W2_PIPE == 0) {
      f32x2 dn[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) dn[k] = f32x2{dq2[(c >> 1) & 1][k][2 * (c & 1)], dq2[(c >> 1) & 1][k][2 * (c & 1) + 1]};
      if constexpr ((W2_ABL & 64) != 0) {
#pragma unroll
        for (int k = 0; k < 16; ++k) V[k] = dn[k];
      } else {
        // (32 packed adds, as asm: hipcc split half of them into scalar pairs — 72 vector instructions per step where 32 do, and a
        // vector instruction is paid in matrix cycles here)
        f32x2 tt[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          tt[0 + j] = pk_sub(dn[0 + j], dn[8 + j]);
          tt[4 + j] = pk_add(dn[4 + j], dn[8 + j]);
          tt[8 + j] = pk_sub(dn[8 + j], dn[4 + j]);
          tt[12 + j] = pk_sub(dn[4 + j], dn[12 + j]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          V[4 * i + 0] = pk_sub(tt[4 * i + 0], tt[4 * i + 2]);
          V[4 * i + 1] = pk_add(tt[4 * i + 1], tt[4 * i + 2]);
          V[4 * i + 2] = pk_sub(tt[4 * i + 2], tt[4 * i + 1]);
          V[4 * i + 3] = pk_sub(tt[4 * i + 1], tt[4 * i + 3]);
        }
      }
      }
      // (scheduling fences: hipcc otherwise hoists the NEXT step's transform up to its loads — in front of this step's
      // MFMAs — and waits for memory there, and sinks the B-fragment reads down to their first use)
      __builtin_amdgcn_sched_barrier(0);
      // ---- requests of step t+1: weights chunk into the other buffer (its readers finished at the last barrier), patch
      // (unconditional — past the workgroup's last step the weights land in a buffer nobody reads and the patch offsets
      // are out of bounds: a branch around the loads would put register copies, and with them a wait for memory, right here)
      // (W2_PIPE: a double-step's patch is consumed a step earlier — by the transform riding in the step before — so it is requested a
      // step earlier too: double-step (c + 3) / 2 during ODD step c, into the set whose last reader was step c - 1; steps 5 and 7 ask for
      // the next item's double-steps 0 and 1)
      if (c == (W2_PIPE ? 4 : 6)) plan_a(item_l + 1);            // (steps t+2, t+3 of chunks 0, 1 belong to the next item)
      // (epilogue state of the item's last step: byte offsets of tile r's four pixels at channel 4n, residual values.  MFMA
      // block b, column n is output channel 4n + b — the host lays U out that way, cadre_amd/encoder.py _winograd_u_c64 — so a
      // lane ends up with FOUR CONSECUTIVE channels of a pixel: 16-byte stores and residual loads, 16 of each per lane and
      // item instead of 64.  The epilogue was bound by the address path: 512 four-byte requests per item and CU)
      f32x4 rv[4][4];                            // residuals by (tile, pixel): channels 4n .. 4n+3
      auto req_res1 = [&](int r, int px) {       // residual of tile r, pixel px
        rv[r][px] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsR, (int)eo[r][px], 0, 0));
      };
      __builtin_amdgcn_sched_barrier(0);
      // ---- 16 planes x 4 channel blocks x 4 k-steps of MFMAs; B fragments one plane ahead
      const char* ub = ((c & 3) == 0 ? ubuf0 : (c & 3) == 1 ? ubuf1 : (c & 3) == 2 ? ubuf2 : ubuf3) + (n * 32 + q * 8);
      f32x2 bf[2][4];
#pragma unroll
      for (int b = 0; b < 4; ++b) bf[0][b] = *reinterpret_cast<const f32x2*>(ub + b * 512);
      if constexpr ((W2_ABL & 128) != 0) {
#pragma unroll
        for (int b = 0; b < 4; ++b) { bf[1][b] = bf[0][b] + f32x2{1.f, 2.f}; asm volatile("" : "+v"(bf[1][b])); }
      }
#pragma unroll
      for (int p = 0; p < 16; ++p) {
        if (p + 1 < 16 && (W2_ABL & 128) == 0) {
#pragma unroll
          for (int b = 0; b < 4; ++b) bf[(p + 1) & 1][b] = *reinterpret_cast<const f32x2*>(ub + (p + 1) * 2048 + b * 512);
        }
        {
          // one memory request per plane, between the MFMAs: a burst of 24 per wave in front of the block keeps all four
          // waves of the CU in the address queue — not issuing MFMAs — while the texture addresser works through 96 requests
          if (p < 8) request_u1((c + 2) & 7, (c + 2) & 3, p);
          if (c == 6 && (p & 3) == 1) offsets(p >> 2, eo[p >> 2]);
          if (W2_PIPE == 0 && (c & 1) == 0) request_q1(((c >> 1) + 1) & 3, dq2[((c >> 1) + 1) & 1], p >> 2, p & 3);
          if (W2_PIPE != 0 && (c & 1) == 1 && (c != 7 || (W2_ABL & 8) != 0)) request_q1(((c + 3) >> 1) & 3, dq2[((c + 3) >> 1) & 1], p >> 2, p & 3);      // (step 7: in the epilogue, below)
          if (c == 7 && RES && (W2_ABL & 8) == 0 && (p & 1)) req_res1(p >> 3, (p >> 1) & 3);      // (tiles 0, 1: 8 requests)
        }
        // step c+1's transform (W2_PIPE): one packed add behind an MFMA — row p/4 of the column pass behind the first four MFMAs of
        // planes 0, 4, 8, 12; V[p-1] of the next step behind the sixth, once plane p-1's MFMAs are a plane behind (the operand
        // registers of an issued MFMA are free: it reads them as it starts)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            if constexpr ((W2_ABL & 1) == 0)
              mfma_acc(acc[p][b], V[p][s], bf[p & 1][b][s], c == 0 && s == 0);
            else acc[p][b][s] = ((c == 0 && s == 0) ? 0.f : acc[p][b][s]) + V[p][s] * bf[p & 1][b][s];
            if constexpr (W2_PIPE != 0) {
              if ((p & 3) == 0 && s == 0) tt_one(c + 1, p >> 2, b);
              if (p >= 1 && s == 1 && b == 1) V[p - 1] = v_of(p - 1);
            }
          }
        __builtin_amdgcn_sched_barrier(0);          // plane p+1's fragment reads stay in front of plane p's MFMAs
      }
      if constexpr (W2_PIPE != 0) V[15] = v_of(15);
      __builtin_amdgcn_sched_barrier(0);
      // ---- end of an item: inverse transform (register-local), BN, residual, ReLU, stores
      if (c == 7 && (W2_ABL & 8) != 0) {
#pragma unroll
        for (int p = 0; p < 16; ++p)
#pragma unroll
          for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float kv = acc[p][b][r]; asm volatile("" :: "v"(kv)); }
      }
      if (c == 7) mfma_drain();
      if (c == 7 && (W2_ABL & 8) == 0) {
        // lane (n, q): tiles 4q + r of the wave's 16, channels 4n .. 4n+3.  Memory requests ride a few at a time between the
        // (tile, channel block) groups: the residuals of tiles 0, 1 were requested inside the MFMA block, those of tiles 2, 3
        // go out beside tiles 0, 1, the next item's chunk-1 patch (into the set this step's transform freed) beside tiles
        // 2, 3 — no group waits for a request issued right in front of it
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          __builtin_amdgcn_sched_barrier(0);
          float v[4][4];                           // [pixel][channel 4n + b]
          // two channel blocks at a time on packed fp32 pairs (no MFMA runs beside the epilogue: v_pk_add_f32 is two adds per slot;
          // the 256 accumulators are read a group at a time, not all up front)
#pragma unroll
          for (int bp = 0; bp < 2; ++bp) {
            if (RES && r < 2) { req_res1(r + 2, 2 * bp); req_res1(r + 2, 2 * bp + 1); }
            f32x2 m[16];
#pragma unroll
            for (int p = 0; p < 16; ++p) { acc_pin(acc[p][2 * bp]); acc_pin(acc[p][2 * bp + 1]); m[p] = f32x2{acc[p][2 * bp][r], acc[p][2 * bp + 1][r]}; }
            f32x2 s0[4], s1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              s0[j] = pk_add(pk_add(m[0 + j], m[4 + j]), m[8 + j]);
              s1[j] = pk_sub(pk_sub(m[4 + j], m[8 + j]), m[12 + j]);
            }
            f32x2 y[4];
            y[0] = pk_add(pk_add(s0[0], s0[1]), s0[2]); y[1] = pk_sub(pk_sub(s0[1], s0[2]), s0[3]);
            y[2] = pk_add(pk_add(s1[0], s1[1]), s1[2]); y[3] = pk_sub(pk_sub(s1[1], s1[2]), s1[3]);
            const f32x2 sc2 = {sc[2 * bp], sc[2 * bp + 1]}, sh2 = {sh[2 * bp], sh[2 * bp + 1]};
#pragma unroll
            for (int px = 0; px < 4; ++px) {
              const f32x2 t = y[px] * sc2 + sh2;
              v[px][2 * bp] = t[0]; v[px][2 * bp + 1] = t[1];
            }
            __builtin_amdgcn_sched_barrier(0);
          }
#pragma unroll
          for (int px = 0; px < 4; ++px) {
            f32x4 o = {v[px][0], v[px][1], v[px][2], v[px][3]};
            if constexpr (RES) o += rv[r][px];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], act_floor);      // (no branch: the epilogue stays one basic block per tile)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rsO, (int)eo[r][px], 0, 0);
          }
          if constexpr (W2_PIPE != 0) {
            // the next item's double-step 1 (due at ITS step 1), four pixels beside each tile's stores: the set's registers carry the
            // residuals of tiles 0, 1 through the MFMA block of this step (requested there, they would be 64 registers too many)
#pragma unroll
            for (int j = 0; j < 4; ++j) request_q1(1, dq2[1], r, j);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr ((W2_ABL & 16) == 0) {
        // in-order completion: step t+1's weights and patch (both requested a step ago) have landed once at most THIS step's
        // requests are still in flight
        // even steps: 8 weight pieces + the 16 patch requests of the next double-step stay in flight; odd steps: the weight pieces
        // (+ the epilogue's 16 residual loads and 16 stores at the end of an item)
        if constexpr (W2_PIPE != 0) {                     // (the patch requests ride in the odd steps: 16 more in flight there, 16 fewer in the even ones)
          if (c == 7 && (W2_ABL & 8) == 0) { if (RES) asm volatile("s_waitcnt vmcnt(56)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); }
          else if (c & 1) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else
        if (c == 7 && (W2_ABL & 8) == 0) { if (RES) asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); }
        else if (c & 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        __builtin_amdgcn_s_barrier();                       // ... everybody's; and everybody is done reading buffer c & 1
        asm volatile("" ::: "memory");
      }
    }
  }
}

// U must be laid out [8][16][64][8] (chunk c = 2d + e of 8 input channels, plane xi = 4i + j, position 16 b + n = output channel 4 n + b,
// index 2q + s in the chunk = input channel 16 d + 4 q + 2 e + s):
// cadre_amd/encoder.py _winograd_u_c64.  out = act(conv * scale + shift (+ resid)), act: 0 none, 1 ReLU.
extern "C" int cadre_winograd_c64(const float* x, const float* U, const float* scale, const float* shift, const float* resid, float* out,
                                  int32_t F, int32_t H, int32_t W, int32_t act, void* stream) {
  if (!x || !U || !out || F < 1 || H < 1 || W < 1) return cadre_fail("cadre_winograd_c64: bad argument");
  if (act != 0 && act != 1) return cadre_fail("cadre_winograd_c64: act must be 0 (none) or 1 (ReLU after the residual)");
  if ((((uintptr_t)x | (uintptr_t)U | (uintptr_t)out | (uintptr_t)resid) & 15) || (((uintptr_t)scale | (uintptr_t)shift) & 3))
    return cadre_fail("cadre_winograd_c64: operands must be 16-byte aligned");
  if ((long long)F * H * W * 256 >= (1ll << 31)) return cadre_fail("cadre_winograd_c64: the activation tensor must stay below the 2 GiB buffer window: chunk the batch");
  w2_args a;
  a.x = x; a.U = U; a.scale = scale; a.shift = shift; a.resid = resid; a.out = out;
  a.F = F; a.H = H; a.W = W; a.TH = (H + 1) / 2; a.TW = (W + 1) / 2;
  const long long nt = (long long)F * a.TH * a.TW;
  if (nt > 0x7fffffff - 64) return cadre_fail("cadre_winograd_c64: too many tiles");
  a.ntiles = (int)nt;
  a.ngroups = (int)((nt + 15) / 16);
  a.relu = act;
  const int nitems = (a.ngroups + 3) / 4;
  const int grid = nitems < 256 ? nitems : 256;
  if (resid) hipLaunchKernelGGL(wino2_c64_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(wino2_c64_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}
