// stem_pool.hip — the front of the DANet encoder as ONE kernel for gfx950:
//   u8 observation (packed RGB + route, cadre_pack_obs) -> /255 LUT (agent.py:46) -> 7x7/s2 conv +
//   folded eval-BatchNorm + ReLU -> 3x3/s2 max-pool (resnet.py:111-115, 168-172) -> NHWC pooled map.
// Replaces preprocess (1.36 GB f32 write at 1024 x 288x288), the stem conv's 5.4 GB output and the
// max-pool's re-read of it: HBM traffic of the front drops from ~14 GB to ~1.7 GB per 1024 frames, and the
// fp32 stem runs its MFMAs with K = 200 (49 taps x 4 channels, padded by ONE tap).
//
// Work decomposition: a PAIR of waves (one per 32-channel half) owns (frame, band of pooled rows) and walks down
//   the band one pooled row per iteration; four pairs per workgroup (2 waves per SIMD), each pair with its own ring.
//   Iteration p computes stem rows 2p, 2p+1 as NT MFMA tiles of "2 stem rows x 16 stem columns" x 64
//   channels (v_mfma_f32_32x32x2_f32 / v_mfma_f32_32x32x16_bf16).  Tile row i = (dy, dx) with
//   dy = (i>>2)&1, dx = (i&3) + 4*(i>>3): in the accumulator layout (row = (reg&3) + 8*(reg>>2) + 4*(lane>>5),
//   col = lane&31) lane half h then holds stem row 2p+h, columns x0 .. x0+15 in its 16 registers — the
//   horizontal 3-max is register-local, the vertical one a lane-half exchange (v_permlane32_swap) plus the
//   previous iteration's row kept in registers.  The A operand is formed from the pair's LDS ring of 13 input
//   rows.  fp32: even / odd pixel planes (stride-2 taps become unit stride, bank-conflict free), the ring holds the
//   PACKED u8 pixels and each fragment element is converted by div255() — bit-identical to the reference's
//   float32(rgb / 255.) for all 256 inputs (cadre_div255_selfcheck).  bf16: converted once per pixel at staging, the two
//   planes interleaved ([pixel pair][plane][4 channels] = 16 bytes), taps ordered 7 rows x 8 (kx = 7: zero weights, K = 224):
//   a lane's fragment of a k-step is ONE ds_read_b128 at a fixed offset of its row base.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/cadre_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

int cadre_fail(const char* msg);

#define SP_TAPS32 50      // fp32: 49 taps + 1 zero tap  -> K = 200
#define SP_WP32 204       // fp32 weight row pitch in floats ((pitch/4) odd: conflict-free ds_read_b128 over rows)
#define SP_TAPS16 56      // bf16: 7 kernel rows x 8 taps (kx = 7: zero weights) -> K = 224 (14 k-steps of 16)
#define SP_WP16 232       // bf16 weight row pitch in bf16 elements (464 B: (pitch_bytes/16) odd)
#define SP_TAPSX 52       // exact-bf16 form of the fp32 front: 49 taps + 3 zero taps -> K = 208 (13 k-steps of 16), THREE weight pieces
#define SP_WPX 216        // its weight row pitch in bf16 elements (432 B: (pitch_bytes/16) odd)

struct stem_args {
  const uint32_t* img;    // [F][H][W] packed pixels: R | G<<8 | B<<16 | route<<24 (route byte 0 or 255)
  const void* wt;         // fp32 [64][50][4]: tap-major (tap = ky*7 + kx), zero padded; bf16 [64][7][8][4] x BN scale (kx = 7: zeros)
  const float* scale;     // folded BN, [64] (fp32; bf16: NULL, folded into wt)
  const float* shift;
  void* out;              // pooled map
  int F, H, W, Hs, Ws, Hp, Wp;
  int PB, NB;             // pooled rows per band, bands per frame
  int PP;                 // ring plane pitch in pixels
  int total;              // F * NB (frame, band) units; one PAIR of waves per unit
  long long out_frame, out_row;      // output strides in elements: frame, pooled row
  int out_px;                        // pooled pixel stride in elements (>= 64)
  long long out_off;                 // element offset of pooled (0, 0, 0, ch 0)
};

// float32(i / 255.) for an integer-valued float i in [0, 255], EXACTLY the reference's value (agent.py:46 divides in
// double and stores float32): fl(1/255) split into a high and a low part — checked against the host table for
// all 256 inputs by cadre_div255_selfcheck (tests/test_kernels_gpu.py).  3 VALU per value (convert, multiply, fma), no table lookup, no LDS.
__device__ __forceinline__ float div255(float f) {
  // f / 255 = f * (r_hi + r_lo) with r_hi = fl(1/255), r_lo = fl(1/255 - r_hi): the product f * r_hi is exact inside the fma
  // (8-bit f), so fma(f, r_hi, fl(f * r_lo)) is the correctly rounded quotient for all 256 inputs — one multiply + one fma
  // (round 4; until then a multiply and a Newton step: two fmas)
  const float r_hi = 0x1.010102p-8f, r_lo = -0x1.fdfdfep-33f;
  return __builtin_fmaf(f, r_hi, f * r_lo);
}

__global__ void div255_check_kernel(const float* lut, int* bad) {
  const int i = threadIdx.x;
  if (div255((float)i) != lut[i]) atomicAdd(bad, 1);
}
extern "C" int cadre_div255_selfcheck(const float* lut255, int32_t* mismatches, void* stream) {
  if (!lut255 || !mismatches) return cadre_fail("cadre_div255_selfcheck: bad argument");
  hipError_t e = hipMemsetAsync(mismatches, 0, sizeof(int32_t), (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(div255_check_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, lut255, mismatches);
  return (int)hipGetLastError();
}

// The kernel (second form, round 4; the first form's decomposition — header above — with its two barriers per iteration,
// whole-chunk epilogues and per-fetch address arithmetic measured 0.83 vs 0.58 ms per 1024 frames in bf16, 4.58 vs 4.55 in
// fp32: tools/stem_ablate.py):
//   * ring of 13 input rows per pair: iteration p reads rows 4p-3 .. 4p+5 while the four rows the NEXT iteration adds
//     (4p+6 .. 4p+9) are converted and written into the four slots iteration p-1 released — ONE workgroup barrier per
//     iteration, staging rides in the MFMA shadow of chunk 0;
//   * the epilogue of chunk j-1 (BN, horizontal 3-max, lane-half exchange, vertical 3-max, stores) is issued in pieces
//     between the k-steps of chunk j (two accumulator sets); only the last chunk's epilogue of an iteration is exposed;
//   * ReLU moved behind the pool (max(0, max(a, b, c)) == max(relu a, relu b, relu c)), BN on packed fp32 pairs, the
//     vertical max on BOTH lane halves (one v_permlane32_swap of (h[k], h[k+4]) hands columns k to the lower half and
//     k+4 to the upper: 4 swaps / 4 max3 / 4 full-wave stores per tile instead of 8 / 8 / 8 half-wave ones);
//   * every ring address of an iteration is formed once (one VGPR per (k-step, tap pair)); the k-loop carries no
//     address arithmetic, stores go through a per-frame buffer descriptor with scalar offsets.
#define SP_RING 13        // input rows in a pair's ring
#ifndef STEM_ABL
#define STEM_ABL 0       // tools/stem_ablate.py: 1 no MFMA, 2 no u8 -> float conversion (fp32), 4 no staging, 8 no epilogue, 16 no ring reads, 32 no weight reads, 64 no barrier
#endif

// X3 (round 6, opt-in: cadre_stem_pool mode 2): the fp32 front on the bf16 matrix cores with EXACT products.  The A operand is an 8-bit
// integer — a pixel byte is exact in bf16 — and an fp32 weight is exactly the sum of three bf16 pieces (8 + 8 + 8 significand bits:
// p1 = bf16(w), p2 = bf16(w - p1), p3 = w - p1 - p2), so every product byte x piece is exact in fp32 and the sums are fp32: the
// arithmetic of v_mfma_f32_32x32x2_f32 up to the order of the additions, in 3 x 13 v_mfma_f32_32x32x16_bf16 per tile instead of 100
// fp32 MFMAs (1248 against 6400 matrix cycles), and the byte -> bf16 conversions ride free beside bf16 MFMAs (profiles/
// r06_mfma_shadow.txt).  Everything else is the fp32 kernel: packed-u8 ring, fp32 accumulators, BN (with the /255) + ReLU + pool in
// fp32, fp32 output.  wt: [3 pieces][64][SP_WPX] bf16, k = tap * 4 + channel (tap = ky * 7 + kx; taps 49 .. 51 zero).
template <int NT, int CH, bool RAGGED, bool BF16, bool X3 = false>
__global__ __launch_bounds__(512, 2) void stem_pool_kernel(stem_args a) {
  static_assert(NT % CH == 0, "tiles per row pair must split into whole chunks");
  static_assert(!(BF16 && X3), "X3 is a form of the fp32 front");
  constexpr int NCH = NT / CH;
  constexpr int NLD = (NT + 3) / 4;
  constexpr int WROW = BF16 ? SP_WP16 / 2 : X3 ? SP_WPX / 2 : SP_WP32;
  constexpr int WROWS = X3 ? 3 * 64 : 64;          // weight rows in LDS (X3: piece-major)
  constexpr int NK = BF16 ? SP_TAPS16 / 4 : X3 ? SP_TAPSX / 4 : SP_TAPS32 / 2;
  constexpr int PXD = BF16 ? 2 : 1;
  constexpr int PP = 16 * NT + 4;                   // ring plane pitch in pixels (cadre_stem_pool checks a.PP == PP)
  constexpr int RP = 2 * PP * PXD;                  // ring row pitch in dwords
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* wts = smem;
  uint32_t* ring0 = reinterpret_cast<uint32_t*>(smem + WROWS * WROW);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // (provably uniform: unit, frame, band live in SGPRs)
  const int l31 = lane & 31, lh = lane >> 5;
  const int pair = wave >> 1, hN = wave & 1;
  {
    constexpr int CPR = (BF16 ? SP_TAPS16 * 8 : X3 ? SP_WPX * 2 : SP_TAPS32 * 16) / 16;
    const char* src = reinterpret_cast<const char*>(a.wt);
    for (int i = tid; i < WROWS * CPR; i += 512) {
      const int n = i / CPR, c = i - n * CPR;
      *reinterpret_cast<f32x4*>(wts + n * WROW + c * 4) = *reinterpret_cast<const f32x4*>(src + ((size_t)n * CPR + c) * 16);
    }
  }
  const int u = blockIdx.x * 4 + pair;
  const bool valid = u < a.total;
  const int f = valid ? u / a.NB : 0, band = valid ? u - f * a.NB : 0;
  const int p0 = band * a.PB, p1 = valid ? min(a.Hp, p0 + a.PB) : 0;
  uint32_t* ring = ring0 + pair * (SP_RING * RP);
  for (int i = hN * 64 + lane; i < SP_RING * RP; i += 128) ring[i] = 0u;
  const uint32_t* frame = a.img + (size_t)f * a.H * a.W;
  const int W4 = a.W >> 2;

  u32x4 pre[NLD];
  // group n = input rows 4n+2 .. 4n+5 (the rows iteration n adds to iteration n-1's); outside the image: zeros
  auto load_group = [&](int n) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int c = i * 128 + hN * 64 + lane;
      const int j = (c >= W4) + (c >= 2 * W4) + (c >= 3 * W4), r = 4 * n + 2 + j;       // c / W4 for c < 4 * W4
      u32x4 v = {0u, 0u, 0u, 0u};
      if (c < a.W && r >= 0 && r < a.H) v = *reinterpret_cast<const u32x4*>(frame + (size_t)r * a.W + (c - j * W4) * 4);
      pre[i] = v;
    }
  };
  auto put = [&](uint32_t* dst, uint32_t px) {
    if constexpr (BF16) {
      typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
      bf16x4 o;
#pragma unroll
      for (int s = 0; s < 4; ++s) o[s] = (__bf16)div255((float)((px >> (8 * s)) & 255u));
      *reinterpret_cast<u32x2*>(dst) = __builtin_bit_cast(u32x2, o);
    } else {
      *dst = px;
    }
  };
  // ring slot of input row r: (r + 13) % 13, r >= -6.  store piece (i, e): pixel e of chunk i of the group
  auto store_piece = [&](int n, int i, int e) {
    const int c = i * 128 + hN * 64 + lane;
    if (c < a.W) {
      const int j = (c >= W4) + (c >= 2 * W4) + (c >= 3 * W4), hx = (c - j * W4) * 2;
      int sl = (4 * n + 2 + 13) % SP_RING + j;          // (uniform part: SALU)
      sl = sl >= SP_RING ? sl - SP_RING : sl;
      uint32_t* row = ring + sl * RP;
      // pixel x = 2*hx + e -> ring pixel x + 3: plane (x + 3) & 1, index (x + 3) >> 1.  fp32: two planes of PP dwords;
      // bf16: the planes interleaved, [index][plane][4 channels] = 16 bytes per index (one ds_read_b128 per fragment)
      const int pl = (e & 1) ^ 1, ix = hx + 1 + ((e + 1) >> 1);
      put(row + (BF16 ? ix * 4 + pl * 2 : pl * PP + ix), pre[i][e]);
    }
  };
  auto store_group = [&](int n) {
#pragma unroll
    for (int i = 0; i < NLD; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) store_piece(n, i, e);
  };

  const int pstart = p0 > 0 ? p0 - 1 : 0;
  __syncthreads();                                   // weights + zeroed rings
  if (valid) {
    for (int n = pstart - 2; n <= pstart; ++n) { load_group(n); store_group(n); }       // rows 4ps-6 .. 4ps+5
  }
  __syncthreads();

  // fp32 (round 6): the A operand is the pixel byte as a float, the /255 of agent.py:46 rides in the folded BN scale — ONE vector
  // instruction per fragment element (v_cvt_f32_ubyteN) instead of three (convert, multiply, fma of div255()).  A vector instruction is
  // paid in matrix cycles in fp32 (profiles/r06_mfma_shadow.txt): 12 of them per four 64-cycle MFMAs were a quarter of this kernel.  The
  // products byte x weight are exact to 1 ulp like before; the one rounding of x / 255 the reference has per input is gone and one of
  // scale / 255 per channel is new (the goldens move in the eighth digit).  -DSTEM_DIV255_A: the element-wise form.
#ifdef STEM_DIV255_A
  const float sc = BF16 ? 1.f : a.scale[32 * hN + l31], sh = a.shift[32 * hN + l31];
#else
  const float sc = BF16 ? 1.f : a.scale[32 * hN + l31] / 255.f, sh = a.shift[32 * hN + l31];
#endif
  constexpr int PHN = 4;                             // previous stem row's horizontal maxima, columns k + 4*lh
  float prevH[NT][PHN];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int k = 0; k < PHN; ++k) prevH[t][k] = 0.f;

  const int dy = (l31 >> 2) & 1, dx = (l31 & 3) + 4 * (l31 >> 3);
  const int esz = BF16 ? 2 : 4;
  // stores: per-frame descriptor, lane offset = (column 4*lh, channel) and a scalar offset per (row, pooled column)
  const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(reinterpret_cast<char*>(a.out) + (a.out_off + (long long)f * a.out_frame) * esz), 0, 0x7ffffffc, 0x00020000);
  const unsigned lane_o = (unsigned)((4 * lh * a.out_px + 32 * hN + l31) * esz);
  const unsigned wlane = (unsigned)(((32 * hN + l31) * WROW + lh * 4) * 4);              // byte offset of this lane's weight row (+ lane half)
  const char* wbase = reinterpret_cast<const char*>(wts);
  const char* rbase = reinterpret_cast<const char*>(ring);

  f32x16 acc[2][CH];
  float hk[CH][8];                                   // horizontal maxima of the chunk in its epilogue
  float carry = 0.f;
  // ring byte offsets of the iteration: fp32 one per k-quad (tap 2q + lane half; the two halves' taps may lie in different
  // kernel rows), bf16 one per kernel row (k-step q = row q>>1, taps 4(q&1) + 2*half + {0,1}: plane = e, pair index + half)
  unsigned aoff[BF16 ? 7 : X3 ? 2 * NK : NK];      // (X3: two taps per lane and k-step: tap 4 jj + 2 * lane half + e)

  // ---- epilogue of tile t of the chunk held in acc[b], in four pieces
  auto epi = [&](int b, int t, int T, int piece, int p, unsigned lane_e) {
    f32x16& v = acc[b][t];
    if (piece == 0 && !BF16) {
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      const f32x2 sc2 = {sc, sc}, sh2 = {sh, sh};
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        f32x2 y = {v[r], v[r + 1]};
        y = __builtin_elementwise_fma(y, sc2, sh2);
        v[r] = y[0]; v[r + 1] = y[1];
      }
      if constexpr (RAGGED) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (16 * T + r >= a.Ws || 2 * p + lh >= a.Hs) v[r] = 0.f;
      }
    } else if (piece == 0) {                         // bf16: the scale is folded into the weights, the sums started at the shift
      if constexpr (RAGGED) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (16 * T + r >= a.Ws || 2 * p + lh >= a.Hs) v[r] = 0.f;
      }
    } else if (piece == 1) {
      hk[t][0] = fmaxf(fmaxf(carry, v[0]), v[1]);
#pragma unroll
      for (int k = 1; k < 8; ++k) hk[t][k] = fmaxf(fmaxf(v[2 * k - 1], v[2 * k]), v[2 * k + 1]);
      carry = v[15];
    } else if (piece == 2) {
      // (h[k], h[k+4]) -> lower half: rows 2p, 2p+1 of column k; upper half: rows 2p, 2p+1 of column k+4
      asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %4\n\tv_permlane32_swap_b32 %1, %5\n\tv_permlane32_swap_b32 %2, %6\n\t"
          "v_permlane32_swap_b32 %3, %7\n\ts_nop 1"
          : "+v"(hk[t][0]), "+v"(hk[t][1]), "+v"(hk[t][2]), "+v"(hk[t][3]), "+v"(hk[t][4]), "+v"(hk[t][5]), "+v"(hk[t][6]), "+v"(hk[t][7]));
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float pv = prevH[T][k];
        float o;                                     // max(0, pv, a, b) (asm: the compiler would canonicalise the exchanged values first)
        asm("v_max3_f32 %0, %1, %2, %3\n\tv_max_f32 %0, 0, %0" : "=&v"(o) : "v"(pv), "v"(hk[t][k]), "v"(hk[t][k + 4]));
        prevH[T][k] = hk[t][k + 4];
        const int c = 8 * T + k;                                   // (+ 4*lh in lane_o)
        {
          unsigned vo = lane_e;                      // (0x80000000 while the band rebuilds the row above it: dropped by the bounds check)
          if constexpr (RAGGED) vo = (c + 4 * lh < a.Wp) ? lane_e : 0x80000000u;
          const int so = (int)(((long long)p * a.out_row + (long long)c * a.out_px) * esz);      // scalar
          if constexpr (BF16) {
            const __bf16 ob = (__bf16)o;
            __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(short, ob), rsO, (int)vo, so, 0);
          } else {
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), rsO, (int)vo, so, 0);
          }
        }
      }
    }
  };

  int srow = (4 * pstart - 3 + 13) % SP_RING;       // slot of input row 4p-3 (uniform)
  const int iters = a.PB + 1;
  for (int it = 0; it < iters; ++it) {
    const int p = pstart + it;
    const bool act = p < p1;
    if (act) {
      const unsigned lane_e = p >= p0 ? lane_o : 0x80000000u;
      // ---- ring byte offsets of the iteration: row slots for ky = 0..7 on both pixel planes, then one per fetch
      {
        unsigned ro[8];                              // plane-0 byte offset of (row slot of ky, column dx)
#pragma unroll
        for (int ky = 0; ky < 8; ++ky) {
          int sl = srow + 2 * dy + ky;
          sl = sl >= SP_RING ? sl - SP_RING : sl;
          ro[ky] = (unsigned)((sl * RP + dx * (BF16 ? 4 : 1)) * 4);
        }
        if constexpr (BF16) {
#pragma unroll
          for (int ky = 0; ky < 7; ++ky) aoff[ky] = ro[ky] + (unsigned)(lh * 16);
        } else if constexpr (X3) {
#pragma unroll
          for (int jj = 0; jj < NK; ++jj)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const int tA = 4 * jj + e, tB = tA + 2;             // lane half 0 / 1 (taps 49 .. 51: zero weights, any mapped LDS address)
              const int kyA = tA / 7, kxA = tA % 7, kyB = tB / 7, kxB = tB % 7;
              const unsigned xa = ro[kyA] + (unsigned)(((kxA & 1) * PP + (kxA >> 1)) * 4);
              const unsigned xb = ro[kyB] + (unsigned)(((kxB & 1) * PP + (kxB >> 1)) * 4);
              aoff[2 * jj + e] = lh ? xb : xa;
            }
        } else {
#pragma unroll
          for (int q = 0; q < NK; ++q) {
            const int tA = 2 * q, tB = tA + 1;
            const int kyA = tA / 7, kxA = tA % 7, kyB = tB / 7, kxB = tB % 7;
            const unsigned xa = ro[kyA] + (unsigned)(((kxA & 1) * PP + (kxA >> 1)) * 4);
            const unsigned xb = ro[kyB] + (unsigned)(((kxB & 1) * PP + (kxB >> 1)) * 4);
            aoff[q] = lh ? xb : xa;
          }
        }
      }
      if constexpr ((STEM_ABL & 4) == 0) load_group(p + 1);      // rows 4p+6 .. 4p+9: written into the ring behind the last chunk's k-steps
      carry = 0.f;
#pragma unroll
      for (int j = 0; j < NCH; ++j) {
        const int b = j & 1;
        // first k-step's C operand: fp32 inline 0; bf16 the BN shift (scale folded into the weights by the host)
        const float c0 = BF16 ? sh : 0.f;
        const f32x16 zero16 = {c0, c0, c0, c0, c0, c0, c0, c0, c0, c0, c0, c0, c0, c0, c0, c0};
        constexpr int CHB = CH * 16 * (BF16 ? 16 : 4);       // ring bytes per chunk of tiles (bf16: 16 bytes per pixel-pair index)
        // side work of k-step q: chunk 0 = staging of the next iteration's rows; chunk j > 0 = epilogue pieces of chunk j-1
        auto side = [&](int q) {
          if (j == NCH - 1) {
            constexpr int NP = NLD * 4;
            const int lo = (q * NP) / NK, hi = ((q + 1) * NP) / NK;
#pragma unroll
            for (int s = 0; s < NP; ++s)
              if (s >= lo && s < hi && (STEM_ABL & 4) == 0) store_piece(p + 1, s >> 2, s & 3);
          }
          if (j > 0) {
            constexpr int NP = CH * 4;
            const int lo = (q * NP) / NK, hi = ((q + 1) * NP) / NK;
#pragma unroll
            for (int s = 0; s < NP; ++s)
              if (s >= lo && s < hi && (STEM_ABL & 8) == 0) epi(b ^ 1, s >> 2, (j - 1) * CH + (s >> 2), s & 3, p, lane_e);
          }
        };
        if constexpr (X3) {
          auto fetch = [&](int jj, uint32_t (*px)[2], f32x4* bq) {
#pragma unroll
            for (int t = 0; t < CH; ++t)
#pragma unroll
              for (int e = 0; e < 2; ++e) px[t][e] = *reinterpret_cast<const uint32_t*>(rbase + aoff[2 * jj + e] + j * CHB + t * 64);
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) bq[pc] = *reinterpret_cast<const f32x4*>(wbase + wlane + pc * (64 * WROW * 4) + jj * 32);
          };
          uint32_t pxn[CH][2];
          f32x4 bn[3];
          fetch(0, pxn, bn);
#pragma unroll
          for (int jj = 0; jj < NK; ++jj) {
            bf16x8 av[CH];
#pragma unroll
            for (int t = 0; t < CH; ++t)
#pragma unroll
              for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) av[t][4 * e + s2] = (__bf16)(float)((pxn[t][e] >> (8 * s2)) & 255u);
            bf16x8 bw[3];
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) bw[pc] = __builtin_bit_cast(bf16x8, bn[pc]);
            if (jj + 1 < NK) fetch(jj + 1, pxn, bn);
            // (smallest piece first: the sum of a tap's three products is exact either way, the running sum meets the large term last)
#pragma unroll
            for (int pc = 2; pc >= 0; --pc)
#pragma unroll
              for (int t = 0; t < CH; ++t)
                acc[b][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[t], bw[pc], (jj == 0 && pc == 2) ? zero16 : acc[b][t], 0, 0, 0);
            side(jj);
            __builtin_amdgcn_sched_barrier(0);
          }
        } else if constexpr (!BF16) {
          auto fetch = [&](int q, uint32_t* px) {
#pragma unroll
            for (int t = 0; t < CH; ++t) {
              if constexpr ((STEM_ABL & 16) == 0) px[t] = *reinterpret_cast<const uint32_t*>(rbase + aoff[q] + j * CHB + t * 64);
              else px[t] = 0x10203040u + q + t + lane;
            }
          };
          auto conv = [&](const uint32_t* px, float (*af)[4]) {
#pragma unroll
            for (int t = 0; t < CH; ++t)
#pragma unroll
              for (int s = 0; s < 4; ++s) {
#ifdef STEM_DIV255_A
                if constexpr ((STEM_ABL & 2) == 0) af[t][s] = div255((float)((px[t] >> (8 * s)) & 255u));
#else
                if constexpr ((STEM_ABL & 2) == 0) af[t][s] = (float)((px[t] >> (8 * s)) & 255u);
#endif
                else af[t][s] = __builtin_bit_cast(float, px[t] + s);
              }
          };
#ifndef STEM_UNGROUPED
          // Vector instructions in ONE run per PAIR of k-steps, behind the pair's first MFMA (round 6): in fp32 the first vector
          // instruction behind an MFMA costs 13 matrix cycles, every further one of the run 4 (profiles/r06_mfma_shadow.txt) — left to
          // the scheduler the conversions sat one or two behind each MFMA (four switches per k-step).  The run: the 8 conversions of
          // k-steps q+2, q+3 (into the other register set), the ring addresses of q+4, q+5 and the side work of the two steps.
          static_assert(NK >= 2, "two k-steps in the prologue");
          uint32_t pxn[2][CH];
          float af[2][2][CH][4];
          fetch(0, pxn[0]);
          fetch(1, pxn[1]);
          conv(pxn[0], af[0][0]);
          conv(pxn[1], af[0][1]);
          if (2 < NK) fetch(2, pxn[0]);
          if (3 < NK) fetch(3, pxn[1]);
          f32x4 bw = *reinterpret_cast<const f32x4*>(wbase + wlane);
#pragma unroll
          for (int q = 0; q < NK; ++q) {
            const int set = (q >> 1) & 1, h = q & 1;
            f32x4 bn = bw;
            if (q + 1 < NK) {
              if constexpr ((STEM_ABL & 32) == 0) bn = *reinterpret_cast<const f32x4*>(wbase + wlane + (q + 1) * 32);
              else bn = f32x4{1.f + q, 2.f, 3.f + lane, 4.f};
            }
            __builtin_amdgcn_sched_barrier(0);
            auto mm = [&](int s) {
#pragma unroll
              for (int t = 0; t < CH; ++t) {
                if constexpr ((STEM_ABL & 1) == 0) acc[b][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][h][t][s], bw[s], (q == 0 && s == 0) ? zero16 : acc[b][t], 0, 0, 0);
                else acc[b][t][s] = ((q == 0 && s == 0) ? 0.f : acc[b][t][s]) + af[set][h][t][s] * bw[s];
              }
            };
            mm(0);
            if (h == 0) {
              __builtin_amdgcn_sched_barrier(0);
              if (q + 2 < NK) conv(pxn[0], af[set ^ 1][0]);
              if (q + 3 < NK) conv(pxn[1], af[set ^ 1][1]);
              if (q + 4 < NK) fetch(q + 4, pxn[0]);
              if (q + 5 < NK) fetch(q + 5, pxn[1]);
              side(q);
              if (q + 1 < NK) side(q + 1);
              __builtin_amdgcn_sched_barrier(0);
            }
            mm(1); mm(2); mm(3);
            __builtin_amdgcn_sched_barrier(0);
            bw = bn;
          }
#else
          uint32_t pxn[CH];
          float af[CH][4];
          fetch(0, pxn);
          conv(pxn, af);
          fetch(1, pxn);
          f32x4 bw = *reinterpret_cast<const f32x4*>(wbase + wlane);
#pragma unroll
          for (int q = 0; q < NK; ++q) {
            f32x4 bn = bw;
            if (q + 1 < NK) {
              if constexpr ((STEM_ABL & 32) == 0) bn = *reinterpret_cast<const f32x4*>(wbase + wlane + (q + 1) * 32);
              else bn = f32x4{1.f + q, 2.f, 3.f + lane, 4.f};
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
              for (int t = 0; t < CH; ++t) {
                if constexpr ((STEM_ABL & 1) == 0) acc[b][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[t][s], bw[s], (q == 0 && s == 0) ? zero16 : acc[b][t], 0, 0, 0);
                else acc[b][t][s] = ((q == 0 && s == 0) ? 0.f : acc[b][t][s]) + af[t][s] * bw[s];
              }
            if (q + 1 < NK) conv(pxn, af);
            if (q + 2 < NK) fetch(q + 2, pxn);
            side(q);
            __builtin_amdgcn_sched_barrier(0);
            bw = bn;
          }
#endif
        } else {
          unsigned cb[7];                            // the chunk's row bases (keeps every fragment offset inside ds_read2_b64's range)
#pragma unroll
          for (int ky = 0; ky < 7; ++ky) cb[ky] = aoff[ky] + j * CHB;
          auto fetch = [&](int q, u32x4* px, f32x4& bq) {
#pragma unroll
            for (int t = 0; t < CH; ++t) {
              if constexpr ((STEM_ABL & 16) == 0) px[t] = *reinterpret_cast<const u32x4*>(rbase + cb[q >> 1] + ((q & 1) * 32 + t * 256));
              else px[t] = u32x4{0x3f803f80u + q, 0x3f803f80u + t + lane, 0x3f803f80u, 0x3f803f80u};
            }
            if constexpr ((STEM_ABL & 32) == 0) bq = *reinterpret_cast<const f32x4*>(wbase + wlane + q * 32);
            else bq = f32x4{1.f + q, 2.f, 3.f + lane, 4.f};
          };
          u32x4 pxn[CH];
          f32x4 bn;
          fetch(0, pxn, bn);
#pragma unroll
          for (int q = 0; q < NK; ++q) {
            u32x4 av[CH];
#pragma unroll
            for (int t = 0; t < CH; ++t) av[t] = pxn[t];
            const bf16x8 bw = __builtin_bit_cast(bf16x8, bn);
            if (q + 1 < NK) fetch(q + 1, pxn, bn);
#pragma unroll
            for (int t = 0; t < CH; ++t) {
              if constexpr ((STEM_ABL & 1) == 0) acc[b][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[t]), bw, q == 0 ? zero16 : acc[b][t], 0, 0, 0);
              else acc[b][t][q & 15] = (q == 0 ? 0.f : acc[b][t][q & 15]) + __builtin_bit_cast(float, av[t][q & 3]) * bn[q & 3];
            }
            side(q);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        if constexpr ((STEM_ABL & 8) != 0) {         // keep the sums alive without the epilogue
          if (j > 0) {
#pragma unroll
            for (int t = 0; t < CH; ++t)
#pragma unroll
              for (int r = 0; r < 16; ++r) { float kv = acc[b ^ 1][t][r]; asm volatile("" :: "v"(kv)); }
          }
        }
      }
      // the last chunk's epilogue (nothing left to hide it behind)
#pragma unroll
      for (int t = 0; t < CH; ++t)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          if constexpr ((STEM_ABL & 8) == 0) epi((NCH - 1) & 1, t, (NCH - 1) * CH + t, s, p, lane_e);
          else { float kv = acc[(NCH - 1) & 1][t][s]; asm volatile("" :: "v"(kv)); }
        }
    }
    srow += 4;
    srow = srow >= SP_RING ? srow - SP_RING : srow;
    if constexpr ((STEM_ABL & 64) == 0) __syncthreads();     // rows of iteration p+1 written, rows of iteration p read
  }
}

template <int NT, int CH32, int CH16, bool RAGGED>
static int launch_stem(const stem_args& a, int mode, size_t lds, hipStream_t st) {
  const dim3 grid((a.total + 3) / 4), block(512);
  const bool bf16 = mode == 1;
  if (mode == 2) {
    auto k = stem_pool_kernel<NT, CH32, RAGGED, false, true>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(k, grid, block, lds, st, a);
  } else if (bf16) {
    auto k = stem_pool_kernel<NT, CH16, RAGGED, true>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(k, grid, block, lds, st, a);
  } else {
    auto k = stem_pool_kernel<NT, CH32, RAGGED, false>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(k, grid, block, lds, st, a);
  }
  return (int)hipGetLastError();
}

// Geometry supported by the fused front: NT = ceil(Ws / 16) in {3, 8, 9} (84x84, 144x256, 288x288 and anything
// with the same tile count), W % 4 == 0, W <= 32 * NT.  Returns 1 if supported (host logic, no launch).
extern "C" int cadre_stem_pool_supported(int32_t H, int32_t W) {
  if (H < 7 || W < 7 || (W & 3)) return 0;
  const int Ws = (W + 6 - 7) / 2 + 1;
  const int NT = (Ws + 15) / 16;
  if (!(NT == 3 || NT == 8 || NT == 9)) return 0;
  if (W > 32 * NT) return 0;
  return 1;
}

extern "C" int cadre_stem_pool(const uint32_t* img, const void* wt, const float* scale, const float* shift,
                               void* out, int32_t F, int32_t H, int32_t W, int32_t bf16,
                               int64_t out_frame, int64_t out_row, int32_t out_px, int64_t out_off, void* stream) {
  if (!img || !wt || !shift || !out || F < 1) return cadre_fail("cadre_stem_pool: bad argument");
  if (bf16 < 0 || bf16 > 2) return cadre_fail("cadre_stem_pool: mode must be 0 (fp32), 1 (bf16) or 2 (fp32 result from three exact bf16 weight pieces)");
  if (bf16 == 1 ? scale != nullptr : scale == nullptr)
    return cadre_fail("cadre_stem_pool: fp32 takes the folded-BN scale, bf16 takes weights already multiplied by it (scale = NULL)");
  if (!cadre_stem_pool_supported(H, W)) return cadre_fail("cadre_stem_pool: unsupported geometry (see cadre_stem_pool_supported)");
  if (out_px < 64 || ((uintptr_t)img & 15) || ((uintptr_t)wt & 15)) return cadre_fail("cadre_stem_pool: bad output stride / alignment");
  stem_args a;
  a.img = img; a.wt = wt; a.scale = scale; a.shift = shift; a.out = out;
  a.F = F; a.H = H; a.W = W;
  a.Hs = (H + 6 - 7) / 2 + 1; a.Ws = (W + 6 - 7) / 2 + 1;
  a.Hp = (a.Hs + 2 - 3) / 2 + 1; a.Wp = (a.Ws + 2 - 3) / 2 + 1;
  const int NT = (a.Ws + 15) / 16;
  // bands: enough wave units to fill 256 CUs x 4 waves; a band start costs one extra iteration
  int NB = (1024 + F - 1) / F;
  if (NB < 1) NB = 1;
  if (NB > a.Hp) NB = a.Hp;
  a.PB = (a.Hp + NB - 1) / NB;
  a.NB = (a.Hp + a.PB - 1) / a.PB;
  if ((long long)F * a.NB > 0x7fffffffLL) return cadre_fail("cadre_stem_pool: too many frames");
  a.total = F * a.NB;
  int PP = (W + 8 + 1) / 2;
  if (PP < 16 * NT + 4) PP = 16 * NT + 4;
  PP = ((PP + 3) / 8) * 8 + 4;                        // == 4 (mod 8): the two stem rows of a tile read disjoint banks
  a.PP = PP;
  a.out_frame = out_frame; a.out_row = out_row; a.out_px = out_px; a.out_off = out_off;
  const int wrow = bf16 == 1 ? SP_WP16 / 2 : bf16 == 2 ? 3 * (SP_WPX / 2) : SP_WP32;
  if (PP != 16 * NT + 4) return cadre_fail("cadre_stem_pool: ring plane pitch");       // (always: W <= 32 * NT)
  const size_t lds = (size_t)64 * wrow * 4 + (size_t)4 * SP_RING * 2 * PP * (bf16 == 1 ? 8 : 4);
  if (lds > 160 * 1024) return cadre_fail("cadre_stem_pool: frame too wide for the LDS ring");
  hipStream_t st = (hipStream_t)stream;
  const bool ragged = (a.Ws % 16) != 0 || (a.Hs & 1) || a.Wp * 2 != a.Ws;
  // tiles per MFMA chunk (CH): fp32 1 (two accumulator sets + the previous row's maxima: 204 VGPRs), bf16 3 / 2 (B fragment
  // shared by the chunk; 244 VGPRs, no spill at 2 waves per SIMD)
  if (NT == 9) return ragged ? launch_stem<9, 1, 3, true>(a, bf16, lds, st) : launch_stem<9, 1, 3, false>(a, bf16, lds, st);
  if (NT == 8) return ragged ? launch_stem<8, 1, 2, true>(a, bf16, lds, st) : launch_stem<8, 1, 2, false>(a, bf16, lds, st);
  return launch_stem<3, 1, 3, true>(a, bf16, lds, st);
}
