// gemm_bf16_w128.hip — dense bf16 NT product with split-K, raw fp32 partial sums: slab[s][M][ldc] = A[M][k in slice s] . B[N][k in slice s]^T,
// for the inter-task attention's first layers (intertask_att.py:39-80 `*_query/key/value_layer.1`: [F][Np*512] x [1536][Np*512]^T per
// branch — the bf16 model's largest dense product, 261 GFLOP per 2048 frames and branch).  Round 5 ("bf16 tile
// GEMMs on LDS-DMA ... inter-task GEMM >= 1.1 PFLOP/s"); the tile kernel (gemm_bf16.hip: global -> VGPR -> ds_write, two
// __syncthreads() per k-tile) ran it at 0.81-0.83 PFLOP/s.
//
// The scheme of ab/conv3x3_w128.hip, which on this shape has no epilogue to speak of (one 256 x 256 tile of output per 2592 k):
//   * workgroup = 4 waves, one per SIMD, 512 registers: 256 x 256 tile, wave (wm, wn) owns 128 rows x 128 columns = 16 accumulator
//     blocks of 32 x 32 pinned to the AGPR file (inline-asm v_mfma_f32_32x32x16_bf16); per k-step of 16 a wave issues 16 MFMAs on
//     four A fragments and four B fragments, every memory request between two MFMAs (file built without the machine scheduler);
//   * A (the activations): 64-k chunks of the tile's 256 rows (32 KB) go through TWO LDS buffers by LDS-DMA (source-side XOR swizzle,
//     conflict-free ds_read_b128); the chunk after next is requested — all eight pieces of a wave — in the k-step that follows the
//     chunk's barrier, so between a wave's last piece and the next barrier lie exactly the twelve B loads of three k-steps
//     (in-order completion: ONE counted vmcnt per chunk);
//   * B (the weights) never touches LDS: stored in FRAGMENT order by the host (cadre_amd/encoder.py _w128_dense_b), streamed by each
//     wave straight into a ring of eight register sets, seven k-steps ahead;
//   * ONE barrier per chunk of 64 MFMAs per wave, in front of its last k-step (behind it the next chunk's rows have landed and
//     nobody reads the current buffer any more: the first fragments of the next chunk are read under the last 16 MFMAs);
//   * k order inside a slice and the slice boundaries (ceil(K/64 / split) chunks each) are those of cadre_gemm_bf16 with split_k:
//     every output element is the same sequence of MFMA accumulations — bit-identical partial sums, so the encoder's results stay
//     independent of the frame batch (latent cache) whichever kernel a batch size selects.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include <utility>
#include "../../include/cadre_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

int cadre_fail(const char* msg);

#ifndef GW_ABL          // ablation builds: 1 no MFMAs, 2 no A-fragment reads, 4 no A DMA, 8 no B loads, 16 no stores
#define GW_ABL 0
#endif

struct gw_args {
  const void* A;         // [M][lda] bf16
  const void* B;         // [N/128][K/16][4 blocks][64 lanes][8] bf16 (fragment order)
  float* C;              // [split][M][ldc] fp32
  int M, N, K, lda, ldc;
  int nck;               // K / 64
  int per;               // chunks per slice
  int mtiles, ntiles;
};

template <class F, int... I>
__device__ __forceinline__ void gw_static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void gw_static_for(F&& f) { gw_static_for_impl(f, std::make_integer_sequence<int, N>{}); }

template <int N>
__device__ __forceinline__ void gw_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ void gw_mfma(f32x16& acc, const f32x4& bf, const f32x4& af) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr ((GW_ABL & 1) != 0) { asm volatile("" : "+a"(acc) : "v"(bf), "v"(af)); }
  else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(bf), "v"(af));
#endif
}
__device__ __forceinline__ void gw_mfma0(f32x16& acc, const f32x4& bf, const f32x4& af) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(acc) : "v"(bf), "v"(af));
#endif
}

__global__ __launch_bounds__(256, 1) void gemm_bf16_w128_kernel(gw_args a) {
  constexpr int BUF_B = 256 * 128;                         // one A buffer: 256 rows x 128 B
  constexpr int NSL = 8, D = 7;                            // B register sets, prefetch depth in k-steps
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF_B];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  // tile order: the N tiles of an (M tile, slice) are neighbours (they share the A rows: L2 hits), XCD-contiguous
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int nt = bid % a.ntiles, rest = bid / a.ntiles;
  const int mt = rest % a.mtiles, sl = rest / a.mtiles;
  const int c_begin = sl * a.per, c_end = min(a.nck, c_begin + a.per);
  const int nchunks = c_end - c_begin;
  float* Cs = a.C + (int64_t)sl * a.M * a.ldc;
  if (nchunks <= 0) {                                      // (a slice past the end of K: its slab is zeros)
    for (int i = tid; i < 256 * 64; i += 256) {
      const int r = i >> 6, c4 = (i & 63) << 2;
      const int m = mt * 256 + r, n = nt * 256 + c4;
      if (m < a.M && n < a.N) *reinterpret_cast<f32x4*>(Cs + (int64_t)m * a.ldc + n) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    return;
  }
  const int lda_b = a.lda * 2;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.A, 0, a.M * lda_b, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)a.B, 0, (a.N / 128) * (a.K / 16) * 4096, 0x00020000);
  auto swz = [](int row) constexpr -> int { return (row >> 1) & 7; };
  // ---- A DMA: piece j = 4 n + wave of a chunk = rows 8 j .. 8 j + 7; this lane brings row 8 j + (lane >> 3), LDS chunk (lane & 7)
  // <- source chunk (lane & 7) ^ swz(row); swz(row) = (lane >> 4) ^ 4 (j & 1), j & 1 = wave & 1.  A row past M lies past
  // num_records (zeros).
  const int a_lane = (mt * 256 + 8 * wave + (lane >> 3)) * lda_b + ((((lane & 7) ^ (lane >> 4) ^ (4 * (wave & 1)))) << 4);
  auto send_a = [&](int c, int buf, int n) __attribute__((always_inline)) {
    int al = a_lane;
    asm volatile("" : "+v"(al));
    unsigned voff = (unsigned)((c * 128 + n * 32 * lda_b) + al);
    char* dst = smem + buf * BUF_B + (4 * n + wave) * 1024;
    if constexpr ((GW_ABL & 4) != 0) { voff = 0x80000000u; }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)dst, 16, (int)voff, 0, 0, 0);
  };
  // ---- B stream: k-step q of the slice = 4 blocks x 1 KB at byte (16-k index) * 4096 of this wave's 128-column group
  const int b_lane = lane * 16;
  const int gb = (nt * 2 + wn) * (a.K / 16) * 4096 + c_begin * (4 * 4096);
  const int b_last = (a.N / 128) * (a.K / 16) * 4096 - 4096;      // (the prefetch runs up to seven k-steps past a slice: clamped to the tensor)
  auto load_b = [&](int soff_, int cb) __attribute__((always_inline)) -> f32x4 {
    if constexpr ((GW_ABL & 8) != 0) return f32x4{0.01f * lane, -2.5f + cb, 0.125f, 1.f};
    const int soff = min(soff_, b_last);
    int bl = b_lane;
    asm volatile("" : "+v"(bl));
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, bl + cb * 1024, soff, 0));
  };
  // A fragment of block mb (rows 128 wm + 32 mb + l31), k-step s: chunk 2 s + lh of the row, swizzled
  unsigned afa[4];
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
    const int row = 128 * wm + 32 * mb + l31;
    afa[mb] = (unsigned)(row * 128) ^ (unsigned)(swz(row) << 4) ^ ((unsigned)lh << 4);
  }
  auto lds_read = [&](unsigned off) __attribute__((always_inline)) -> f32x4 {
    if constexpr ((GW_ABL & 2) != 0) return f32x4{(float)(lane * 3), 1.5f, -0.75f * lane, 0.3f + (float)off};
    return *reinterpret_cast<const f32x4*>(smem + off);
  };

  // ---- prologue: chunks 0 and 1 of A, k-steps 0 .. 6 of B, the first A fragments
  f32x16 acc[4][4];
  f32x4 breg[NSL][4], afr[2][4];
#pragma unroll
  for (int n = 0; n < 8; ++n) send_a(c_begin, 0, n);
#pragma unroll
  for (int n = 0; n < 8; ++n) send_a(c_begin + 1, 1, n);   // (past the slice: the next slice's rows or zeros, never read)
#pragma unroll
  for (int q = 0; q < D; ++q)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) breg[q][cb] = load_b(gb + q * 4096, cb);
  gw_wait_vm<4 * D>();                                     // the A pieces (older than the B loads) have landed
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) afr[0][mb] = lds_read(afa[mb]);

  using std::integral_constant;
  // NCH chunks (1 or 2) starting at chunk c (buffer parity PAR of the first): 4 NCH k-steps, fully unrolled.  FIRST: the slice's
  // first k-step (the accumulators start from the product)
  auto chunks = [&](auto nch_c, auto first_c, int c) __attribute__((always_inline)) {
    constexpr int NCH = decltype(nch_c)::value;
    constexpr bool FIRST = decltype(first_c)::value;
    const int kq = (c - c_begin) * 4;                      // k-step index of the body's first k-step inside the slice
    gw_static_for<4 * NCH>([&](auto q_c) __attribute__((always_inline)) {
      constexpr int q = decltype(q_c)::value;
      constexpr int ch = q / 4, s = q % 4;                 // chunk of the body (buffer ch & 1: bodies start on buffer 0), k-step in it
      constexpr int sq = q % NSL, sb = (q + D) % NSL;
      if constexpr (s == 3) {
        // the chunk's barrier: my pieces of the next chunk have landed (requested four k-steps ago; since the last of them only
        // the twelve B loads of this chunk's k-steps 0 .. 2), my reads of this buffer are done
        gw_wait_vm<12>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      // A fragments of the next k-step: this chunk's (s < 3) or k-step 0 of the next chunk in the other buffer
      const unsigned rbase = (s < 3) ? (unsigned)((ch & 1) * BUF_B) : (unsigned)(((ch + 1) & 1) * BUF_B);
      const unsigned rx = (s < 3) ? (unsigned)((s + 1) << 5) : 0u;
      const int soff = gb + (kq + q + D) * 4096;
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
          if constexpr (FIRST && q == 0) gw_mfma0(acc[mb][cb], breg[sq][cb], afr[q & 1][mb]);
          else gw_mfma(acc[mb][cb], breg[sq][cb], afr[q & 1][mb]);
          const int slot = 4 * cb + mb;                    // a request behind every MFMA in a chunk's last k-step, every second one else
          if constexpr (s == 3) {
            if (slot < 4) afr[(q + 1) & 1][slot] = lds_read(rbase + (afa[slot] ^ rx));
            else if (slot < 8) breg[sb][slot - 4] = load_b(soff, slot - 4);
            else send_a(c + ch + 2, ch & 1, slot - 8);      // the chunk after next, into the buffer this barrier freed
          } else if (mb == 1 || mb == 3) {
            const int r = 2 * cb + (mb >> 1);
            if (r < 4) afr[(q + 1) & 1][r] = lds_read(rbase + (afa[r] ^ rx));
            else breg[sb][r - 4] = load_b(soff, r - 4);
          }
        }
      }
    });
  };

  int c = c_begin;
  if (nchunks >= 2) {
    chunks(integral_constant<int, 2>{}, integral_constant<bool, true>{}, c);
    c += 2;
    for (; c + 2 <= c_end; c += 2) chunks(integral_constant<int, 2>{}, integral_constant<bool, false>{}, c);
    if (c < c_end) chunks(integral_constant<int, 1>{}, integral_constant<bool, false>{}, c);
  } else {
    chunks(integral_constant<int, 1>{}, integral_constant<bool, true>{}, c);
  }

  // ---- epilogue: raw fp32 partial sums.  Accumulator block (mb, cb): lane (l31, lh), register r holds row 32 mb + l31, column
  // 32 cb + 8 (r >> 2) + 4 lh + (r & 3); the lane-half exchange leaves a lane with eight consecutive columns: two 16-byte stores
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc((void*)Cs, 0, a.M * a.ldc * 4, 0x00020000);
  const int e_lane = ((mt * 256 + 128 * wm + l31) * a.ldc + nt * 256 + 128 * wn + 8 * lh) * 4;
  gw_static_for<32>([&](auto p_c) __attribute__((always_inline)) {
    constexpr int p = decltype(p_c)::value;
    constexpr int mb = p >> 3, cb = (p >> 1) & 3, h = p & 1;
    if constexpr ((p & 7) == 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) asm volatile("" : "+a"(acc[mb][k]));
    }
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = acc[mb][cb][8 * h + e];
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %4\n\tv_permlane32_swap_b32 %1, %5\n\tv_permlane32_swap_b32 %2, %6\n\t"
        "v_permlane32_swap_b32 %3, %7\n\ts_nop 1"
        : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
    int el = e_lane;
    asm volatile("" : "+v"(el));
    const int row = mt * 256 + 128 * wm + 32 * mb;         // (+ l31: rows past M lie past num_records only if the LAST row does — test the lane's row)
    const bool ok = row + l31 < a.M;
    const int bo = ok ? (32 * mb * a.ldc + 32 * cb + 16 * h) * 4 + el : (int)0x80000000u;
    if constexpr ((GW_ABL & 16) != 0) {
#pragma unroll
      for (int e = 0; e < 8; ++e) asm volatile("" :: "v"(v[e]), "v"(bo));
    } else {
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{v[0], v[1], v[2], v[3]}), rsC, bo, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{v[4], v[5], v[6], v[7]}), rsC, bo + 16, 0, 0);
    }
  });
}

// ---------------------------------------------------------------------------------------------------------------
static const int g_gw_on = [] { const char* e = getenv("CADRE_GEMM_W128"); return e ? atoi(e) : 1; }();

static int gw_capable(int M, int N, int K, int lda, int ldc, int split) {
  if (M < 1 || N < 256 || N % 256 != 0 || K < 128 || K % 64 != 0 || split < 1 || lda < K || ldc < N) return 0;
  if ((lda & 7) || (ldc & 3)) return 0;
  const long long lim = 1ll << 31;
  if ((long long)M * lda * 2 >= lim || (long long)N * K * 2 >= lim || (long long)M * ldc * 4 >= lim) return 0;
  return 1;
}

extern "C" int cadre_gemm_bf16_w128_supported(int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldc, int32_t split_k) {
  return g_gw_on && gw_capable(M, N, K, lda, ldc, split_k);
}

extern "C" int cadre_gemm_bf16_w128(const void* A, const void* B, float* C, int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldc,
                                    int32_t split_k, void* stream) {
  if (!A || !B || !C) return cadre_fail("cadre_gemm_bf16_w128: null operand");
  if (!gw_capable(M, N, K, lda, ldc, split_k))
    return cadre_fail("cadre_gemm_bf16_w128: unsupported shape (N % 256 == 0, K % 64 == 0, K >= 128, lda % 8 == 0, ldc % 4 == 0, every operand < 2 GiB)");
  if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) return cadre_fail("cadre_gemm_bf16_w128: operands must be 16-byte aligned");
  gw_args a;
  a.A = A; a.B = B; a.C = C; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldc = ldc;
  a.nck = K / 64;
  a.per = (a.nck + split_k - 1) / split_k;                  // (cadre_gemm_bf16's split: ceil(k-tiles / split) per slice)
  a.mtiles = (M + 255) / 256; a.ntiles = N / 256;
  const int grid = a.mtiles * a.ntiles * split_k;
  hipLaunchKernelGGL(gemm_bf16_w128_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}
