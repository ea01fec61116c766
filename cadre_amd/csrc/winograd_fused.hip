// winograd_fused.hip — Winograd F(m x m, 3x3), m = 2, 3, 4, in fp32 with the plane products AND the inverse transform in ONE
// kernel (round 6; the stride-1 3x3 convs of the fp32 DANet trunk and head with >= 128 channels:
// carla_perception/Networks/danet_blocks/resnet.py:26-55, danet.py:21-41).
//
// The three-launch form (winograd.hip: input transform -> batched cadre_gemm_f32 -> output transform) moves the transform-domain
// tensors through HBM twice each: V = B^T d B written and read, M = V U^T written and read — (m+2)^2 / m^2 = 2.25 .. 2.78 times
// the activation each way.  On layer2 (128 channels, 36 x 36 maps) that is 7.5 GB per conv against 1.4 GB of activations: the
// batched GEMM runs at 103 TFLOP/s because it is HBM-bound, and the two transform launches cost as much again.
// Here M never exists: one workgroup owns ALL (m+2)^2 planes of a (64 tiles x 32 output channels) block, so the inverse transform
// A^T M A, the folded BN, the residual and the ReLU run on the accumulators.
//
//   cadre_winograd_in_frag   x [F][H][W][C]  ->  V in MFMA-FRAGMENT order  [plane][C/16][Tpad/16][4 kk][16 tiles][4 channels]:
//                            the 1 KB block (plane, 16-channel chunk, 16-tile block) is exactly what the 64 lanes of a wave read as
//                            ONE ds_read_b128 per lane — lane (kk = l >> 4, tile = l & 15) holds channels 16 c + 4 kk .. + 3 =
//                            the A operands of four v_mfma_f32_16x16x4_f32 k-steps — and what ONE buffer_load ... lds instruction
//                            copies (lane-linear, 16 B per lane): no swizzle, no bank conflict, no address arithmetic.
//   cadre_winograd_gemm_out  for every plane xi: M[xi] = V[xi] U[xi]^T, then out = act(A^T M A * scale + shift (+ resid)).
//                            U comes packed by the host in the same fragment order [N/32][C/16][plane][2 nb][4 kk][16 couts][4]
//                            (cadre_amd/encoder.py _winograd_u_frag).
//
// Kernel structure (wino_gemm_out_kernel<m>): 8 waves = (4 tile blocks of 16) x (2 channel blocks of 16), two waves per SIMD;
// a wave keeps the 16 x 16 accumulator block of EVERY plane (P x 4 registers: 144 for F(4x4)), so the inverse transform is
// lane-local (no LDS exchange).  k order: (16-channel chunk, group of PG planes); per group a SLOT of LDS holds the V blocks of the
// four tile blocks and the U blocks of the two channel blocks (6 PG KB), filled by LDS-DMA D slots ahead through a ring of
// R = D + 1 slots.  Step q: every wave confirms its own requests of slot q + 1 (counted s_waitcnt vmcnt), one raw s_barrier, the
// requests of slot q + D go out into the buffer slot q - 1 was read from, the fragments of slot q + 1 are read into the second
// register set, and the 4 PG MFMAs of slot q run from the first — fragment reads and DMA issue lie under the MFMAs of the
// wave and of its SIMD partner.  Persistent workgroups walk items (64-tile block, 32-channel block), the requests run on across
// item boundaries: the next item's first slots land while the epilogue runs.
//
// Arithmetic: exact fp32 (v_mfma_f32_16x16x4_f32 is a k-ordered fp32 fma chain); per (tile, channel, plane) the channel sum
// runs over chunks in ascending order and inside a chunk in the order e = 0..3 (k-step), kk = 0..3 (MFMA k index) of channel
// 16 c + 4 kk + e — the same for every tile wherever it sits in the batch: per-frame results are bit-identical across batch sizes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "../../include/cadre_hip.h"
#include "winograd_mats.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

int cadre_fail(const char* msg);

// -DWGO_ABL=<bits>: timing ablations (results wrong by construction, never in the product build): 1 no MFMAs, 2 no fragment
// reads, 4 no DMA after the prologue, 8 no epilogue, 16 no stores (the epilogue's arithmetic and residual loads stay), 32 no inverse
// transform / transposes, 64 no drain in front of the epilogue
#ifndef WGO_ABL
#define WGO_ABL 0
#endif
// A/B knobs (tools/wgo_ablate.py --variants): request distance in slots (default: ring - 1), cache policy bits of the V / U requests
// (aux of buffer_load ... lds: 1 sc0, 2 nt, 16 sc1)
#ifndef WGO_VAUX
#define WGO_VAUX 0
#endif
#ifndef WGO_UAUX
#define WGO_UAUX 0
#endif

// ---------------------------------------------------------------------------------------------------------------
// input transform into fragment order
// A wave owns 8 tiles x 32 channels: per patch pixel its 64 lanes read 8 x 128 contiguous bytes (whole cache lines), per plane
// they write, for each of the 8 (16-channel chunk, kk) rows of the fragment layout, 8 tiles x 16 B = 128 contiguous bytes.
// (First build: 16 tiles x 16 channels per wave — one contiguous KiB per plane store, but 16 half lines per load: 0.49 ms on
// layer2 against 0.40 for the row-major transform of winograd.hip.)  WIF_LANES: 0 lane = 8 q + t (a lane quad = four tiles of
// one channel quad: contiguous in the store), 1 lane = 8 t + q (a lane quad = four channel quads of one tile: contiguous in the load).
#ifndef WIF_LANES
#define WIF_LANES 0
#endif
template <int M>
__global__ __launch_bounds__(256) void wino_in_frag_kernel(const float* __restrict__ x, float* __restrict__ V, int F, int H, int W, int C,
                                                            int TH, int TW, int T, int TB16, int KC16) {
  constexpr int N = wino_mat<M>::N;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c32 = blockIdx.y * 2 + (wave >> 1);
  if (c32 * 2 >= KC16) return;
  const int t8 = WIF_LANES ? lane >> 3 : lane & 7, q8 = WIF_LANES ? lane & 7 : lane >> 3;
  const int tbg = blockIdx.x;
  const int t16 = (wave & 1) * 8 + t8;
  const int tile = tbg * 16 + t16;
  const int c16 = 2 * c32 + (q8 >> 2), kk = q8 & 3;
  const bool live = tile < T;
  const int tx = tile % TW;
  const int t2 = tile / TW;
  const int ty = t2 % TH, f = t2 / TH;
  const int r0 = M * ty - 1, q0 = M * tx - 1;
  const int ch = 32 * c32 + 4 * q8;
  // B^T d one patch COLUMN at a time (winograd.hip wino_in_kernel: same sums in the same order)
  f32x4 t[N * N];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    f32x4 d[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int r = r0 + i, q = q0 + j;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (live && (unsigned)r < (unsigned)H && (unsigned)q < (unsigned)W)
        v = *reinterpret_cast<const f32x4*>(x + (((long long)f * H + r) * W + q) * C + ch);
      d[i] = v;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) t[i * N + j] = wino_dot<N, f32x4>(wino_mat<M>::BT[i], d, 1);
  }
  const long long plane = (long long)KC16 * TB16 * 256;          // floats per plane
  float* vp = V + ((long long)c16 * TB16 + tbg) * 256 + (kk * 16 + t16) * 4;
#pragma unroll
  for (int i = 0; i < N; ++i)                                    // (B^T d) B
#pragma unroll
    for (int j = 0; j < N; ++j)
      *reinterpret_cast<f32x4*>(vp + (i * N + j) * plane) = wino_dot<N, f32x4>(wino_mat<M>::BT[j], t + i * N, 1);
}

// ---------------------------------------------------------------------------------------------------------------
struct wgo_args {
  const float* V;         // [P][KC16][TB16][4][16][4]
  const float* U;         // [N/32][KC16][P][2][4][16][4]
  const float* scale;     // [N] or null
  const float* shift;     // [N] or null
  const float* resid;     // [F][H][W][N] or null
  float* out;             // [F][H][W][N]
  int F, H, W, N;
  int TH, TW, T, TB16, KC16;
  int ntiles, items;
  int act;
  unsigned v_bytes, u_bytes, o_bytes;
};

template <int N, int I = 0, typename F>
__device__ __forceinline__ void wgo_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    wgo_static_for<N, I + 1>(f);
  }
}

template <int N>
__device__ __forceinline__ void wgo_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// inverse transform of one (tile, channel): y[M][M] = A^T m A, m[N][N] (plane xi = N i + j)
template <int M>
__device__ __forceinline__ void wgo_inverse(const float (&m)[wino_mat<M>::N * wino_mat<M>::N], float (&y)[M * M]) {
  constexpr int N = wino_mat<M>::N;
  float s[M * N];
#pragma unroll
  for (int j = 0; j < N; ++j)                          // A^T m: column j of planes
#pragma unroll
    for (int i = 0; i < M; ++i) s[i * N + j] = wino_dot<N, float>(wino_mat<M>::AT[i], &m[j], N);
#pragma unroll
  for (int i = 0; i < M; ++i)
#pragma unroll
    for (int j = 0; j < M; ++j) y[i * M + j] = wino_dot<N, float>(wino_mat<M>::AT[j], &s[i * N], 1);
}

// NTB: 16-tile blocks per workgroup (waves = 2 NTB: NTB tile blocks x the two channel blocks).  4: the learner's shape (items of
// 64 tiles x 32 channels); 1: few tiles — an 8-frame act() pass has 648 tiles on a 36 x 36 map: 44 items of 64 tiles leave 212 CUs
// idle, 164 items of 16 tiles (two 128-thread workgroups per CU) do not.  Every (tile, channel) is summed in the same order by
// either shape: the choice may depend on the batch size without touching batch invariance.
template <int MT, int NTB>
__global__ __launch_bounds__(128 * NTB, (NTB == 4 ? 2 : 1)) void wino_gemm_out_kernel(wgo_args a) {
  constexpr int NN = wino_mat<MT>::N, P = NN * NN;
  constexpr int PG = (P == 25) ? 5 : 4;                 // planes per slot
  constexpr int NG = P / PG;                            // slots per 16-channel chunk
  constexpr int R = (MT == 4) ? 6 : (MT == 3 ? 5 : 4);  // ring slots
#ifdef WGO_D
  constexpr int D = WGO_D;
#else
  constexpr int D = R - 1;                              // request distance in slots
#endif
  constexpr int CU = (MT == 2) ? 1 : 2;                 // chunks per unrolled block: NG * CU is even and a multiple of R
  static_assert((NG * CU) % R == 0 && (NG * CU) % 2 == 0, "slot buffer and register set must be static per unrolled step");
  constexpr int NW = 2 * NTB;                           // waves
  constexpr int SLOT_B = (NTB + 2) * PG * 1024;         // V: PG x NTB tile blocks x 1 KB, U: PG x 2 channel blocks x 1 KB
  constexpr int U_OFF = NTB * PG * 1024;
  constexpr int NI_V = (NTB * PG + NW - 1) / NW, NI_U = (2 * PG + NW - 1) / NW, NI = NI_V + NI_U;       // DMA instructions per wave and slot
  constexpr int WAITN = NI * (D - 2);                   // younger requests when slot q + 1 must have landed (slots q + 2 .. q + D - 1)
  static_assert(WAITN < 64, "vmcnt is a 6-bit field");
  constexpr unsigned OOB = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tb = wave % NTB, nb = wave / NTB;           // (NTB = 4: waves w and w + 4 share a SIMD — the same tile block, the two channel blocks)
  const unsigned dump_off = (unsigned)(R * SLOT_B);

  // ---- items of this workgroup.  Workgroups are dealt round-robin over the 8 XCDs: give each XCD a contiguous run of item ids
  // (bijective remap, cdna_hip_programming.md T1), item = (tile block of 64) * ntiles + channel block: the channel blocks of one tile
  // block run side by side on one XCD and share its V lines in that L2; every workgroup streams the same U.
  const int G = gridDim.x;
  int bid = blockIdx.x;
  {
    const int xcd = bid & 7, q = G >> 3, r = G & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  if (bid >= a.items) return;
  const int nitems = (a.items - bid + G - 1) / G;
  const int KC16 = a.KC16, TB16 = a.TB16;

  const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)a.V, 0, (int)a.v_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsU = __builtin_amdgcn_make_buffer_rsrc((void*)a.U, 0, (int)a.u_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsNull = __builtin_amdgcn_make_buffer_rsrc((void*)a.V, 0, 0, 0x00020000);      // zero records: every request out of range

  // ---- DMA duties of this wave.  V instruction i: block jv = wave + NW i of the slot's NTB PG (plane pp = jv / NTB, tile block jv % NTB);
  // U instruction i: KB ju = wave + NW i of the slot's 2 PG contiguous KB.  Surplus instructions (jv >= NTB PG / ju >= 2 PG: F(3x3)
  // only) request out of range and land their zeros in the dump KB — every wave issues NI per slot, one vmcnt count fits all.
  unsigned vlane[NI_V], vdst[NI_V], ulane[NI_U], udst[NI_U];
#pragma unroll
  for (int i = 0; i < NI_V; ++i) {
    const int jv = wave + NW * i;
    const bool real = jv < NTB * PG;
    vlane[i] = real ? (unsigned)(((jv / NTB) * KC16 * TB16 + (jv % NTB)) * 1024 + lane * 16) : OOB;
    vdst[i] = real ? (unsigned)(jv * 1024) : dump_off;
  }
#pragma unroll
  for (int i = 0; i < NI_U; ++i) {
    const int ju = wave + NW * i;
    const bool real = ju < 2 * PG;
    ulane[i] = real ? (unsigned)(ju * 1024 + lane * 16) : OOB;
    udst[i] = real ? (unsigned)(U_OFF + ju * 1024) : dump_off;
  }
  bool abl_pro = true;
  // requests of slot (item (mt_x, nt_x), chunk c_x, plane group g) into ring buffer buf; live = false (past the workgroup's last
  // item): the offsets get the out-of-range bit, zeros land in a buffer nobody reads
  auto send = [&](int buf, int mt_x, int nt_x, int c_x, int g, bool live) {
    if ((WGO_ABL & 4) && !abl_pro) live = false;
    const unsigned dead = live ? 0u : OOB;
    const unsigned sV = (unsigned)(((PG * g * KC16 + c_x) * TB16 + mt_x * NTB) * 1024) | dead;
    const unsigned sU = (unsigned)(((nt_x * KC16 + c_x) * P + PG * g) * 2048) | dead;
#pragma unroll
    for (int i = 0; i < NI_V; ++i) {
      const unsigned dst = vdst[i] == dump_off ? dump_off : (unsigned)(buf * SLOT_B) + vdst[i];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (__attribute__((address_space(3))) void*)(smem + dst), 16, (int)(vlane[i] + sV), 0, 0, WGO_VAUX);
    }
#pragma unroll
    for (int i = 0; i < NI_U; ++i) {
      const unsigned dst = udst[i] == dump_off ? dump_off : (unsigned)(buf * SLOT_B) + udst[i];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsU, (__attribute__((address_space(3))) void*)(smem + dst), 16, (int)(ulane[i] + sU), 0, 0, WGO_UAUX);
    }
  };

  // one DMA instruction of a slot (i < NI_V: V instruction i, else U instruction i - NI_V): issued BETWEEN the MFMAs of a step, one
  // at a time (-DWGO_BURST: all NI right behind the barrier — 24 requests at once keep every wave of the CU in the texture
  // addresser's queue with its MFMAs behind them in program order, the lesson of winograd_c64.hip: 1.211 vs 1.158 ms on layer2,
  // 1.301 / 1.145 vs 1.175 / 1.046 ms on the F(3x3) shapes)
  // (The slot's offset goes in the instruction's SCALAR offset, the lane's in its vector offset — both fixed registers: no vector
  // instruction per request.  A v_add_u32 behind an MFMA is not free in fp32: it costs 13 matrix cycles, DESIGN.md 3.4 / profiles/
  // r06_mfma_shadow.txt.  The scalar offset is outside the range check, so a dead slot — past the workgroup's last item — cannot
  // carry the out-of-range bit there: it swaps in a descriptor of zero records instead, one s_cselect.  -DWGO_VADDR: the round-6 form.)
  auto send_one = [&](auto i_c, int buf, unsigned sV, unsigned sU) {
    constexpr int i = decltype(i_c)::value;
#ifdef WGO_VADDR
    if constexpr (i < NI_V) {
      const unsigned dst = vdst[i] == dump_off ? dump_off : (unsigned)(buf * SLOT_B) + vdst[i];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (__attribute__((address_space(3))) void*)(smem + dst), 16, (int)(vlane[i] + sV), 0, 0, WGO_VAUX);
    } else {
      const unsigned dst = udst[i - NI_V] == dump_off ? dump_off : (unsigned)(buf * SLOT_B) + udst[i - NI_V];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsU, (__attribute__((address_space(3))) void*)(smem + dst), 16, (int)(ulane[i - NI_V] + sU), 0, 0, WGO_UAUX);
    }
#else
    // (a dead slot — past the workgroup's last item — arrives here with scalar offset 0: it re-reads the first KB of V / U into a
    //  buffer nobody reads; the ablation build without DMA swaps in a descriptor of zero records)
    // (-DWGO_OPAQUE: LDS addresses and per-slot offsets formed where they are used, behind opaque copies — hoisted out of the item loop
    //  they are ~60 SGPRs too many and come back through one v_readlane per step.  MEASURED AND NOT ADOPTED: no spills, no readlanes,
    //  and 1.145 against 1.073 ms on layer2 — the scalar chains in front of each request cost more than the readlane.)
    if constexpr (i < NI_V) {
      unsigned vd = vdst[i];
#ifdef WGO_OPAQUE
      asm volatile("" : "+s"(vd));
#endif
      const unsigned dst = vd == dump_off ? dump_off : (unsigned)(buf * SLOT_B) + vd;
      __builtin_amdgcn_raw_ptr_buffer_load_lds((WGO_ABL & 4) ? rsNull : rsV, (__attribute__((address_space(3))) void*)(smem + dst), 16, (int)vlane[i], (int)sV, 0, WGO_VAUX);
    } else {
      unsigned ud = udst[i - NI_V];
#ifdef WGO_OPAQUE
      asm volatile("" : "+s"(ud));
#endif
      const unsigned dst = ud == dump_off ? dump_off : (unsigned)(buf * SLOT_B) + ud;
      __builtin_amdgcn_raw_ptr_buffer_load_lds((WGO_ABL & 4) ? rsNull : rsU, (__attribute__((address_space(3))) void*)(smem + dst), 16, (int)ulane[i - NI_V], (int)sU, 0, WGO_UAUX);
    }
#endif
  };

  // ---- fragments: lane l reads bytes 16 l of its 1 KB block.  The reads are inline asm with the waits placed by hand: ONE
  // s_waitcnt lgkmcnt(0) per step, behind the barrier (the reads went out a whole step earlier) — hipcc's own counted waits sit in
  // front of each MFMA and count only ITS loads; with LDS-DMA requests of the same wave in flight they made every step wait for the
  // requests it had just issued (measured: fragment reads + DMA + MFMAs ran the SUM of their times, any two of them the maximum).
  // ds offsets are 16 bits: three base registers 64 KB apart, the slot / plane offset is the literal.
  unsigned a_base[3], b_base[3];
#pragma unroll
  for (int h = 0; h < 3; ++h) {
    a_base[h] = (unsigned)(tb * 1024 + lane * 16 + h * 65536);
    b_base[h] = (unsigned)(nb * 1024 + lane * 16 + h * 65536);
  }
  f32x4 fa[2][PG], fb[2][PG];
  auto read_frag = [&fa, &fb, &a_base, &b_base, lane](auto set_c, auto buf_c, auto pp_c) {
    constexpr int set = decltype(set_c)::value, buf = decltype(buf_c)::value, pp = decltype(pp_c)::value;
    if constexpr ((WGO_ABL & 2) != 0) {
      fa[set][pp] = f32x4{(float)lane, 1.5f, -0.25f * pp, 0.3f};
      fb[set][pp] = f32x4{0.01f * lane, -2.5f, 0.125f * pp, 1.f};
    } else {
      constexpr unsigned ta = (unsigned)(buf * SLOT_B + pp * NTB * 1024), tbo = (unsigned)(buf * SLOT_B + U_OFF + pp * 2048);
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[set][pp]) : "v"(a_base[ta >> 16]), "n"(ta & 0xffffu));
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[set][pp]) : "v"(b_base[tbo >> 16]), "n"(tbo & 0xffffu));
    }
  };

  f32x4 acc[P];
#pragma unroll
  for (int p = 0; p < P; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- epilogue of one item.  Accumulators: lane (l15 = l & 15, lq = l >> 4), register r of plane p = M[p][tile 16 tb + 4 lq + r]
  // [channel 16 nb + l15].  The inverse transform and the folded BN run in that layout (one channel per lane: one scale / shift);
  // then, per output pixel, the four tiles of a lane meet the four channels of its lane quad in a 4 x 4 transpose over (register,
  // lane & 3) — two DPP exchange stages — after which lane (lq, a = l15 >> 2, b = l15 & 3) holds channels 4 a .. 4 a + 3 of tile
  // 4 lq + b: 16-byte residual loads and stores (first build: 4-byte accesses, 128 memory instructions per wave and item — the
  // epilogue was 23 % of the launch).
  const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, (int)a.o_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void*)(a.resid ? a.resid : a.out), 0, a.resid ? (int)a.o_bytes : 0, 0x00020000);
  const float act_floor = (a.act & 15) == 1 ? 0.f : -__builtin_inff();
  const bool post = (a.act & 16) != 0;
  const float r_pre = post ? 0.f : 1.f, r_post = post ? 1.f : 0.f;
  const float inv_tw = 1.0f / (float)a.TW, inv_th = 1.0f / (float)a.TH;
  auto xch = [](float v, auto ctrl_c) {
    constexpr int ctrl = decltype(ctrl_c)::value;
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, true));
  };
  float bn_sc = 1.f, bn_sh = 0.f;
  auto epilogue = [&](int mt_e, int nt_e) {
    if constexpr ((WGO_ABL & 8) != 0) {
#pragma unroll
      for (int p = 0; p < P; ++p) { asm volatile("" :: "v"(acc[p])); }
    } else {
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      constexpr int NPX = MT * MT, HALF = (NPX + 1) / 2;
      // the tile this lane stores: (f, ty, tx) by a float reciprocal + one correction step (t < 2^23)
      const int t = mt_e * (16 * NTB) + tb * 16 + 4 * (lane >> 4) + (lane & 3);
      int q1 = (int)((float)t * inv_tw);
      int tx = t - q1 * a.TW;
      q1 += (tx >= a.TW) ? 1 : 0; q1 -= (tx < 0) ? 1 : 0;
      tx = t - q1 * a.TW;
      int f = (int)((float)q1 * inv_th);
      int ty = q1 - f * a.TH;
      f += (ty >= a.TH) ? 1 : 0; f -= (ty < 0) ? 1 : 0;
      ty = q1 - f * a.TH;
      const int row0 = MT * ty, col0 = MT * tx;
      const int n4 = nt_e * 32 + nb * 16 + ((lane >> 2) & 3) * 4;
      const unsigned base = (unsigned)((((f * a.H + row0) * a.W + col0) * a.N + n4) * 4);
      const int nrow = t < a.T ? a.H - row0 : 0, ncol = a.W - col0;      // rows / columns of the tile inside the map
      const int rowb = a.W * a.N * 4, colb = a.N * 4;
      auto px_off = [&](int e) -> unsigned {
        const int i = e / MT, j = e % MT;
        return (i < nrow && j < ncol) ? base + (unsigned)(i * rowb + j * colb) : OOB;
      };
      // residual of the tile's pixels, requested in two halves AHEAD of their use (a descriptor with zero records when there is
      // none: zeros) — first build: each load in front of its own use, sixteen memory round trips in a row per item
      f32x4 rr[NPX];
#pragma unroll
      for (int e = 0; e < HALF; ++e) rr[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsR, (int)px_off(e), 0, 0));
      // (the asm MFMAs are invisible to the hazard recognizer: the vector ALU reads accumulators below — 48 wait states cover the
      //  40-cycle result latency of the item's last v_mfma_f32_16x16x4_f32)
      asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
      // inverse transform on register QUADS (the four tiles of a lane at once): A^T m column by column — the six planes of a column
      // die as its four sums are formed — then (A^T m) A row by row, and the folded BN of this lane's channel
      f32x4 y[NPX];
      if constexpr ((WGO_ABL & 32) != 0) {
#pragma unroll
        for (int e = 0; e < NPX; ++e) y[e] = acc[e] * bn_sc + bn_sh;
      } else {
        f32x4 s4[MT * NN];
#pragma unroll
        for (int j = 0; j < NN; ++j)
#pragma unroll
          for (int i = 0; i < MT; ++i) s4[i * NN + j] = wino_dot<NN, f32x4>(wino_mat<MT>::AT[i], &acc[j], NN);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < MT; ++j) y[i * MT + j] = wino_dot<NN, f32x4>(wino_mat<MT>::AT[j], &s4[i * NN], 1) * bn_sc + bn_sh;
      }
#pragma unroll
      for (int e = HALF; e < NPX; ++e) rr[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsR, (int)px_off(e), 0, 0));
      const bool odd = (lane & 1) != 0, hi = (lane & 2) != 0;
#pragma unroll
      for (int e = 0; e < NPX; ++e) {
        const float x0 = y[e][0], x1 = y[e][1], x2 = y[e][2], x3 = y[e][3];
        // stage A: 2 x 2 blocks over (register pair, lane pair)
        const float t0 = xch(x0, std::integral_constant<int, 0xB1>{}), t1 = xch(x1, std::integral_constant<int, 0xB1>{});
        const float t2 = xch(x2, std::integral_constant<int, 0xB1>{}), t3 = xch(x3, std::integral_constant<int, 0xB1>{});
        const float n0 = odd ? t1 : x0, n1 = odd ? x1 : t0, n2 = odd ? t3 : x2, n3 = odd ? x3 : t2;
        // stage B: the off-diagonal 2 x 2 blocks trade places between lanes b and b ^ 2
        const float u0 = xch(n0, std::integral_constant<int, 0x4E>{}), u1 = xch(n1, std::integral_constant<int, 0x4E>{});
        const float u2 = xch(n2, std::integral_constant<int, 0x4E>{}), u3 = xch(n3, std::integral_constant<int, 0x4E>{});
        f32x4 v = {hi ? u2 : n0, hi ? u3 : n1, hi ? n2 : u0, hi ? n3 : u1};
        if constexpr ((WGO_ABL & 32) != 0) v = y[e];
        // (the residual in front of or behind the activation by two uniform multipliers, 1 and 0: x + 1 r and x + 0 r are the sums
        //  the selects formed — three vector instructions per element where the selects made five; vector instructions are paid
        //  in the other wave's matrix cycles)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float w = __builtin_fmaf(rr[e][c], r_pre, v[c]);
          w = fmaxf(w, act_floor);
          v[c] = __builtin_fmaf(rr[e][c], r_post, w);
        }
        if constexpr ((WGO_ABL & 16) != 0) { asm volatile("" :: "v"(v)); }
        else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsO, (int)px_off(e), 0, 0);
      }
    }
#if defined(WGO_STAGGER) || WGO_ABL || !defined(WGO_ZEROC)
#pragma unroll
    for (int p = 0; p < P; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
#endif
  };

  // (Measured and not kept: start delays of (a hash of the workgroup id) / 64 of an item, so that the 256 workgroups do not reach
  //  their epilogues — 67 MB of residual lines and stores chip-wide — together: 1.381 vs 1.377 ms on layer2, no effect.)
  // ---- prologue: slots 0 .. D - 1 of the first item requested, landed, published; fragments of slot 0 in register set 0
  int mt = bid / a.ntiles, nt = bid - mt * a.ntiles;
  {
    int c_x = 0, g_x = 0;
#pragma unroll
    for (int s = 0; s < D; ++s) {
      send(s, mt, nt, c_x, g_x, true);
      if (++g_x == NG) { g_x = 0; ++c_x; }
    }
  }
  abl_pro = false;
  wgo_wait_vm<0>();

  using std::integral_constant;
  // (Measured and not kept: the two waves of a SIMD HALF A STEP apart — waves 4 - 7 taking a step's barrier, DMA issue and fragment
  //  reads between the second and third k-step of the slot's MFMAs, MI355X_MICROARCH.md "two waves per SIMD" item 9.  F(3x3):
  //  1.169 / 1.154 ms against 1.152 / 1.128 in lockstep on layer3 / layer4; F(4x4): the second code path cost 193 spilled registers.
  //  The LATE role stays a compile-time parameter of the item loop for that measurement: -DWGO_STAGGER.)
  unsigned sV_cur = 0, sU_cur = 0;
  int tgt_cur = 0;
  (void)sV_cur; (void)sU_cur; (void)tgt_cur;
  auto run = [&](auto late_c) {
    constexpr bool LATE = decltype(late_c)::value;
    for (int li = 0; li < nitems; ++li) {
      const int idx_n = bid + (li + 1) * G;
      const bool more = li + 1 < nitems;
      const int mt_n = idx_n / a.ntiles, nt_n = idx_n - mt_n * a.ntiles;
      // every request of this item's first D slots has landed: this wave's share was drained in front of the previous item's
      // epilogue (in front of the loop for the first item), everybody's is published by the barrier; the epilogue's stores stay in
      // flight.  Fragments of slot 0 into register set 0.
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      {                                                    // folded BN of this lane's accumulator channel (used by the epilogue: requested here)
        const int n = nt * 32 + nb * 16 + (lane & 15);
        bn_sc = a.scale ? a.scale[n] : 1.f;
        bn_sh = a.shift ? a.shift[n] : 0.f;
      }
      wgo_static_for<PG>([&](auto pp_c) { read_frag(integral_constant<int, 0>{}, integral_constant<int, 0>{}, pp_c); });
      // one unrolled block = CU chunks x NG slots.  FIRST: the block behind the drain — the slots its first D - 1 steps would
      // confirm are confirmed; LAST: its last step reads no fragments ahead (the next item's slot 0 is read behind the epilogue)
      auto block = [&](auto first_c, auto last_c, int c0) {
        constexpr bool FIRST = decltype(first_c)::value, LAST = decltype(last_c)::value;
        wgo_static_for<NG * CU>([&](auto s_c) {
          constexpr int s = decltype(s_c)::value;
          constexpr int cu = s / NG, g = s % NG;
          constexpr int nxt = (s + 1) % R, tgt = (s + D) % R; // (slot q itself sits in buffer s % R: its fragments were read a step ago)
          constexpr bool ahead = !(LAST && s == NG * CU - 1);
          // MFMAs of slot q, k-steps [E0, E1), k-step outer, plane inner (consecutive MFMAs are independent); behind the MFMAs of
          // k-step ER the fragments of slot q + 1 go into the other register set
          auto mfmas = [&](auto e0_c, auto e1_c, auto er_c) {
            constexpr int E0 = decltype(e0_c)::value, E1 = decltype(e1_c)::value, ER = decltype(er_c)::value;
            wgo_static_for<E1 - E0>([&](auto e_c) {
              constexpr int e = E0 + decltype(e_c)::value;
              wgo_static_for<PG>([&](auto pp_c) {
                constexpr int pp = decltype(pp_c)::value;
                if constexpr ((WGO_ABL & 1) != 0) {
                  asm volatile("" :: "v"(fa[s & 1][pp]), "v"(fb[s & 1][pp]));
                } else {
                  // (inline asm with the accumulator tied: left to the register allocator the builtin's three-address form gets a
                  //  fresh destination block per MFMA — 144 accumulators do not survive that.  Same-register SrcC chains need no
                  //  wait states.)
                  // (-DWGO_ZEROC: an item's first k-step starts its plane from the inline constant 0 instead of 144 cleared registers —
                  //  measured 1.091 against 1.073 ms on layer2: not adopted)
#ifdef WGO_ZEROC
                  if constexpr (FIRST && cu == 0 && e == 0 && !LATE)
#else
                  if constexpr (false)
#endif
                    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=v"(acc[PG * g + pp]) : "v"(fa[s & 1][pp][e]), "v"(fb[s & 1][pp][e]));
                  else
                    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[PG * g + pp]) : "v"(fa[s & 1][pp][e]), "v"(fb[s & 1][pp][e]));
                }
                if constexpr (e == ER && ahead) read_frag(integral_constant<int, (s + 1) & 1>{}, integral_constant<int, nxt>{}, pp_c);
#ifndef WGO_BURST
                {   // DMA instruction i behind MFMA number PG + 1 + i * stride of the step (behind the fragment reads of k-step 0)
                  constexpr int idx = e * PG + pp, STR = (3 * PG - 1) / NI > 0 ? (3 * PG - 1) / NI : 1;
                  if constexpr (!LATE && idx >= PG && (idx - PG) % STR == 0 && (idx - PG) / STR < NI)
                    send_one(integral_constant<int, (idx - PG) / STR>{}, tgt_cur, sV_cur, sU_cur);
                }
#endif
              });
            });
          };
          if constexpr (LATE) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (the fragments of slot q: read behind the previous barrier)
            mfmas(integral_constant<int, 0>{}, integral_constant<int, 2>{}, integral_constant<int, -1>{});
          }
          // slot q + 1 landed (this wave's share), then published; the fragments of slot q (read a step ago) are in their registers
          if constexpr (!(FIRST && s <= D - 2)) wgo_wait_vm<WAITN>();
          __builtin_amdgcn_s_barrier();
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          // requests of slot q + D: (chunk, group) = (c0 + cu, g) + D slots, possibly in the next item
          constexpr int gd = (g + D) % NG;
          const int cd = c0 + cu + (g + D) / NG;
          const bool wrap = cd >= KC16;
#ifndef WGO_BURST
          {
            const bool live = (!wrap || more) && !((WGO_ABL & 4) != 0);
            const int mt_x = wrap ? mt_n : mt, nt_x = wrap ? nt_n : nt, c_x = wrap ? cd - KC16 : cd;
#ifdef WGO_VADDR
            const unsigned dead = live ? 0u : OOB;
            sV_cur = (unsigned)(((PG * gd * KC16 + c_x) * TB16 + mt_x * NTB) * 1024) | dead;
            sU_cur = (unsigned)(((nt_x * KC16 + c_x) * P + PG * gd) * 2048) | dead;
#else
            int kc = KC16, tb16 = TB16;
#ifdef WGO_OPAQUE
            asm volatile("" : "+s"(kc), "+s"(tb16));
#endif
            sV_cur = live ? (unsigned)(((PG * gd * kc + c_x) * tb16 + mt_x * NTB) * 1024) : 0u;
            sU_cur = live ? (unsigned)(((nt_x * kc + c_x) * P + PG * gd) * 2048) : 0u;
#endif
            tgt_cur = tgt;
          }
#else
          send(tgt, wrap ? mt_n : mt, wrap ? nt_n : nt, wrap ? cd - KC16 : cd, gd, !wrap || more);
#endif
          if constexpr (LATE) mfmas(integral_constant<int, 2>{}, integral_constant<int, 4>{}, integral_constant<int, 2>{});
          else mfmas(integral_constant<int, 0>{}, integral_constant<int, 4>{}, integral_constant<int, 0>{});
        });
      };
      if (KC16 == CU) {
        block(integral_constant<bool, true>{}, integral_constant<bool, true>{}, 0);
      } else {
        block(integral_constant<bool, true>{}, integral_constant<bool, false>{}, 0);
        for (int c0 = CU; c0 + CU < KC16; c0 += CU) block(integral_constant<bool, false>{}, integral_constant<bool, false>{}, c0);
        block(integral_constant<bool, false>{}, integral_constant<bool, true>{}, KC16 - CU);
      }
      if constexpr ((WGO_ABL & 64) == 0)
      wgo_wait_vm<0>();                                  // (the next item's first D slots: requested up to a step ago; the epilogue's
                                                         //  own loads and stores count from zero, and nothing waits for the stores
                                                         //  before step D - 1 of the next item)
      epilogue(mt, nt);
      mt = mt_n; nt = nt_n;
    }
  };
#ifndef WGO_STAGGER
  run(integral_constant<bool, false>{});
#else
  if (nb == 0) run(integral_constant<bool, false>{});
  else run(integral_constant<bool, true>{});
#endif
}

// ---------------------------------------------------------------------------------------------------------------
static int wf_capable(int F, int H, int W, int Cin, int N, int m) {
  if (m != 2 && m != 3 && m != 4) return 0;
  if (F < 1 || H < 1 || W < 1) return 0;
  if (Cin < 32 || (Cin % 32) || N < 32 || (N % 32)) return 0;          // two 16-channel chunks per unrolled block; 32-channel items
  const long long TH = (H + m - 1) / m, TW = (W + m - 1) / m, T = (long long)F * TH * TW;
  const long long Tpad = (T + 63) / 64 * 64, P = (m + 2) * (m + 2), lim = 1ll << 31;
  if (T >= (1 << 23)) return 0;                                        // tile coordinates by float reciprocal
  if (P * Cin * Tpad * 4 >= lim || P * Cin * N * 4 >= lim || (long long)F * H * W * N * 4 >= lim || (long long)F * H * W * Cin * 4 >= lim) return 0;
  return 1;
}

// CADRE_WINOGRAD_FUSED: 0 never, 1 (default) where the fused form measured faster than the three launches — F(4x4), i.e. the 36 x 36
// maps of layer2: 1.56 against 1.86 ms per conv at 1024 frames; on the F(3x3) layers the plane GEMM of the three-launch form already
// runs at 124 - 137 TFLOP/s and its transforms are small: fused 1.48 / 1.24 against 1.24 / 1.15 ms —, 2 every geometry
// the kernels take
static const int g_wf_on = [] { const char* e = getenv("CADRE_WINOGRAD_FUSED"); return e ? atoi(e) : 1; }();

// 1: the kernels take this geometry (whatever the policy says)
extern "C" int cadre_winograd_fused_capable(int32_t F, int32_t H, int32_t W, int32_t Cin, int32_t N, int32_t m) {
  return wf_capable(F, H, W, Cin, N, m);
}

// 1: the encoder should run this conv as cadre_winograd_in_frag + cadre_winograd_gemm_out (host logic: capability AND policy)
extern "C" int cadre_winograd_fused_supported(int32_t F, int32_t H, int32_t W, int32_t Cin, int32_t N, int32_t m) {
  if (!g_wf_on || !wf_capable(F, H, W, Cin, N, m)) return 0;
  return (g_wf_on >= 2 || m == 4) ? 1 : 0;
}

// floats of the V workspace cadre_winograd_in_frag writes and cadre_winograd_gemm_out reads (tiles padded to 64)
extern "C" int64_t cadre_winograd_frag_elems(int32_t F, int32_t H, int32_t W, int32_t Cin, int32_t m) {
  const long long TH = (H + m - 1) / m, TW = (W + m - 1) / m, T = (long long)F * TH * TW;
  return (long long)(m + 2) * (m + 2) * Cin * ((T + 63) / 64 * 64);
}

extern "C" int cadre_winograd_in_frag(const float* x, float* V, int32_t F, int32_t H, int32_t W, int32_t C, int32_t m, void* stream) {
  if (!x || !V) return cadre_fail("cadre_winograd_in_frag: null operand");
  if (!wf_capable(F, H, W, C, 32, m)) return cadre_fail("cadre_winograd_in_frag: unsupported geometry (m 2..4, C % 32 == 0, every tensor < 2 GiB)");
  if (((uintptr_t)x & 15) || ((uintptr_t)V & 15)) return cadre_fail("cadre_winograd_in_frag: operands must be 16-byte aligned");
  const int TH = (H + m - 1) / m, TW = (W + m - 1) / m, T = F * TH * TW;
  const int TB16 = (T + 63) / 64 * 4, KC16 = C / 16;
  const dim3 grid((unsigned)TB16, (unsigned)((KC16 + 3) / 4));      // a workgroup: 16 tiles x 64 channels
  hipStream_t st = (hipStream_t)stream;
  if (m == 2) hipLaunchKernelGGL(wino_in_frag_kernel<2>, grid, dim3(256), 0, st, x, V, F, H, W, C, TH, TW, T, TB16, KC16);
  else if (m == 3) hipLaunchKernelGGL(wino_in_frag_kernel<3>, grid, dim3(256), 0, st, x, V, F, H, W, C, TH, TW, T, TB16, KC16);
  else hipLaunchKernelGGL(wino_in_frag_kernel<4>, grid, dim3(256), 0, st, x, V, F, H, W, C, TH, TW, T, TB16, KC16);
  return (int)hipGetLastError();
}

// 16-tile blocks per workgroup cadre_winograd_gemm_out runs with for T tiles and N output channels (the second template argument of
// wino_gemm_out_kernel<m, NTB>: the profiling key / kernel name of the launch): 4 where items of 64 tiles fill the chip, else 1
extern "C" int cadre_winograd_fused_ntb(int32_t T, int32_t N) {
  static const int ntb_env = [] { const char* e = getenv("CADRE_WINOGRAD_FUSED_NTB"); return e ? atoi(e) : 0; }();
  if (ntb_env == 1 || ntb_env == 4) return ntb_env;
  return ((T + 63) / 64) * (N / 32) < 256 ? 1 : 4;
}

template <int MT, int NTB>
static int wgo_launch(const wgo_args& a, hipStream_t st) {
  constexpr int NN = wino_mat<MT>::N, P = NN * NN, PG = (P == 25) ? 5 : 4, R = (MT == 4) ? 6 : (MT == 3 ? 5 : 4);
  const size_t lds = (size_t)R * (NTB + 2) * PG * 1024 + 1024;
  (void)hipFuncSetAttribute((const void*)wino_gemm_out_kernel<MT, NTB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int slots = NTB == 4 ? 256 : 512;                  // persistent workgroups: one per CU (two of the 128-thread shape)
  const int wgs = a.items < slots ? a.items : slots;
  hipLaunchKernelGGL((wino_gemm_out_kernel<MT, NTB>), dim3(wgs), dim3(128 * NTB), lds, st, a);
  return (int)hipGetLastError();
}

extern "C" int cadre_winograd_gemm_out(const float* V, const float* U, const float* scale, const float* shift, const float* resid,
                                       float* out, int32_t F, int32_t H, int32_t W, int32_t Cin, int32_t N, int32_t act, int32_t m,
                                       void* stream) {
  if (!V || !U || !out) return cadre_fail("cadre_winograd_gemm_out: null operand");
  if (!wf_capable(F, H, W, Cin, N, m))
    return cadre_fail("cadre_winograd_gemm_out: unsupported geometry (m 2..4, Cin % 32 == 0, N % 32 == 0, every tensor < 2 GiB: chunk the batch)");
  if ((act & 15) > 1) return cadre_fail("cadre_winograd_gemm_out: act must be 0 (none) or 1 (ReLU), bit 4 = residual after the activation");
  if (((uintptr_t)V | (uintptr_t)U | (uintptr_t)out | (uintptr_t)resid | (uintptr_t)scale | (uintptr_t)shift) & 15)
    return cadre_fail("cadre_winograd_gemm_out: operands must be 16-byte aligned");
  wgo_args a;
  a.V = V; a.U = U; a.scale = scale; a.shift = shift; a.resid = resid; a.out = out;
  a.F = F; a.H = H; a.W = W; a.N = N; a.act = act;
  a.TH = (H + m - 1) / m; a.TW = (W + m - 1) / m; a.T = F * a.TH * a.TW;
  const int mtiles64 = (a.T + 63) / 64;
  a.TB16 = mtiles64 * 4; a.KC16 = Cin / 16;               // (the V layout pads the tiles to 64 whatever the item shape)
  a.ntiles = N / 32;
  // items of 64 tiles where they fill the chip, of 16 tiles (128-thread workgroups, two per CU) where they would not — the same
  // bits either way (CADRE_WINOGRAD_FUSED_NTB=1|4 forces a shape)
  const bool small = cadre_winograd_fused_ntb(a.T, N) == 1;
  a.items = (small ? (a.T + 15) / 16 : mtiles64) * a.ntiles;
  const long long P = (m + 2) * (m + 2);
  a.v_bytes = (unsigned)(P * Cin * (long long)mtiles64 * 64 * 4);
  a.u_bytes = (unsigned)(P * Cin * (long long)N * 4);
  a.o_bytes = (unsigned)((long long)F * H * W * N * 4);
  hipStream_t st = (hipStream_t)stream;
  if (small) {
    if (m == 2) return wgo_launch<2, 1>(a, st);
    if (m == 3) return wgo_launch<3, 1>(a, st);
    return wgo_launch<4, 1>(a, st);
  }
  if (m == 2) return wgo_launch<2, 4>(a, st);
  if (m == 3) return wgo_launch<3, 4>(a, st);
  return wgo_launch<4, 4>(a, st);
}
