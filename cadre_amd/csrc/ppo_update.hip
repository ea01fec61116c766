// ppo_update.hip — fused kernels of the PPO minibatch update (reference ppo_agent/agent.py:166-237 through
// ppo_agent/models.py:139-152 nn.LSTMCell and models.py:171-177 / distributions.py:34-40 MLP towers).
//
// The update's recurrent products are SKINNY: a command net owns 16 (minibatch 64) to ~64 (minibatch 256) rows of the
// row-sorted minibatch, against 2120 x 544 weights.  On the general tile kernel (gemm_f32.hip) such a product is one
// 272-deep chain of v_mfma_f32_32x32x2_f32 per wave (17-45 us per launch) followed by a pointwise pass and, backward, a
// split-K reduction pass: 3 x 8 launches each way.  Here one launch per time step does the product AND the cell math:
//
//   * v_mfma_f32_16x16x4_f32: 16 rows x 16 columns x 4 k per instruction at 32 cycles — for <= 16-row tiles twice the
//     useful rate of the 32x32x2 form, and a K = 544 chain is 136 instructions;
//   * the weights go from L2 / Infinity Cache STRAIGHT into MFMA fragments: lane (c = lane & 15, q = lane >> 4) holds
//     k 16j + 4q .. +3 of column c = the operand of the four MFMAs of k-block j (the k order inside a block is permuted
//     identically on A and B, so the products pair correctly).  A weight is used by exactly one wave — nothing to share
//     through LDS.  Read from the row-major matrices such a fragment is 16 rows x 64 bytes per wave instruction, and the
//     texture addresser serves scattered 64-byte pieces at a quarter of its streaming rate (measured: 76 % of the wave
//     cycles stalled on instruction issue, 20 GB/s per CU) — so once per update cadre_pack_lstm_weights lays the
//     recurrent weights out in FRAGMENT ORDER for both directions (the backward needs the transpose anyway: its
//     reduction index is the gate axis): every weight load is then one contiguous KiB;
//   * forward: workgroup = (net, 16 hidden units, chunk of 16*RT rows), wave g = gate g (i, f, g, o) of those units over
//     the full K; the four gate tiles meet in LDS and the workgroup finishes c_t, h_t, tanh(c_t) and the activated gates;
//   * backward: workgroup = (net, 16 hidden units, chunk of rows), wave w = quarter w of the (zero padded) gate axis of
//     dG_t; the four partial tiles meet in LDS = dh_{t-1}, and the same workgroup turns it into dG_{t-1} and dc_{t-2} —
//     no split-K slabs, no reduction pass, no pointwise pass.  dG_{t-1} is written twice: row-major (the weight-gradient
//     kernel reads whole rows) and in fragment order for the next step's A operand (contiguous KiB loads again).
//
// Rows sorted by command (row_seg): every kernel here works on exactly a net's run of rows [first, first + count) —
// 16-row tiles from the run's first row; no row of another net is read or written (see step_item).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
#include "../../include/cadre_hip.h"
#ifdef CADRE_AB_KERNELS
#include "../../include/cadre_hip_ab.h"
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));

int cadre_fail(const char* msg);
#define ST(s) ((hipStream_t)(s))
#define FAIL_IF(cond, msg) \
  if (cond) return cadre_fail(msg)

namespace {

__device__ __forceinline__ float sigmoid_(float x) { return 1.f / (1.f + expf(-x)); }

// -DLSTM_TRACE (tools/lstm_trace.py builds its own library; the product build has no stamp): wave 0 of every workgroup
// of the step kernels stamps the shader clock at section boundaries and the 100 MHz wall clock at entry and exit
#ifdef LSTM_TRACE
__device__ long long* g_lstm_trace_dev = nullptr;
#define LS_DECL unsigned long long ls_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}; const unsigned long long ls_r0 = __builtin_amdgcn_s_memrealtime()
#define LS_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ls_t[k]) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define LS_FLUSH() do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); LS_STAMP(6); if (g_lstm_trace_dev && threadIdx.x == 0) { \
    long long* tr_ = g_lstm_trace_dev + (size_t)blockIdx.x * 16; for (int k_ = 0; k_ < 7; ++k_) tr_[k_] = (long long)ls_t[k_]; \
    tr_[8] = (long long)ls_r0; tr_[9] = (long long)__builtin_amdgcn_s_memrealtime(); tr_[10] = 1; } } while (0)
#else
#define LS_DECL do { } while (0)
#define LS_STAMP(k) do { } while (0)
#define LS_FLUSH() do { } while (0)
#endif

struct fwd_args {
  const float* Wp;      // recurrent weights in fragment order (cadre_pack_lstm_weights), net stride wp_str
  const float* bias;    // [H4] of net 0, net stride b_str (may be null)
  float* G;             // this step's gates [B][ldg] of net 0, net stride g_str: in = x-projection, out = activated gates
  const float* Hprev;   // h_{t-1} [B][ldh], net stride h_str
  const float* Cprev;   // c_{t-1}
  float* Hout;          // h_t
  float* Cout;          // c_t
  float* TCout;         // tanh(c_t)
  const int32_t* row_seg;
  int64_t wp_str, b_str, g_str, h_str;
  int ldg, ldh, B, D, Z, NS, rev;
};

// Work item = (net, 16 hidden units, chunk of 16*RT rows), from a 1-D grid with the NET as the fastest index: workgroups
// are dealt round-robin over the 8 XCDs, so net z's workgroups share one XCD and its 4.6 MB of recurrent weights live
// in that XCD's 4 MB L2 between time steps instead of coming from the Infinity Cache every step; `rev` walks the unit
// slices in the opposite order on alternate steps (the most recently used weights are re-used first: an LRU cache
// slightly smaller than the set it cycles through would otherwise miss every time).  Placement only affects speed.
// Rows sorted by command: the chunks start AT the net's first row and cover exactly its run [r_lo, r_hi) — traced
// (tools/lstm_trace.py) the MFMA loop runs at 93 % of the fp32 matrix-pipe rate, so every 16-row tile of other nets' rows
// inside an aligned cover was pipe time on the step's critical path (a 16-row run unaligned in 32-row tiles: 64 rows).
__device__ __forceinline__ bool step_item(int Z, int NS, int rev, const int32_t* row_seg, int B, int rows, int& z, int& slice,
                                          int& row0, int& r_lo, int& r_hi) {
  const int id = blockIdx.x;
  z = id % Z;
  const int rest = id / Z;
  slice = rest % NS;
  if (rev) slice = NS - 1 - slice;
  r_lo = 0;
  r_hi = B;
  if (row_seg) {
    const int beg = row_seg[2 * z], cnt = row_seg[2 * z + 1];
    if (cnt <= 0) return false;
    r_lo = beg;
    r_hi = min(B, beg + cnt);
  }
  row0 = r_lo + (rest / NS) * rows;
  return row0 < r_hi;
}

template <int V>
struct int_c { static constexpr int value = V; };

// 16*RT rows x 16 units x 4 gates per workgroup (the last chunk of a run: only its live 16-row tiles); NB = K / 16
// k-blocks (K = ldh, zero padded past D).
// LDS: the h_{t-1} rows of the chunk (shared by the four gate waves), pitch 552 floats = 138 16-byte slots:
// conflict-free for the fragment reads (lane (c, q) reads slot 138*row + 4j + q: 16 distinct slots mod 16 per lane group).
template <int RT, int NB>
__global__ __launch_bounds__(256) void lstm_step_fwd_kernel(fwd_args p) {
  constexpr int PD = NB / 2;                              // k-blocks of weights in flight per wave (17 KiB)
  constexpr int AP = 552;                                 // LDS row pitch of the activation rows (floats)
  constexpr int K = 16 * NB;
  __shared__ __attribute__((aligned(16))) float ah[16 * RT * AP];
  __shared__ float xg[4 * 16 * RT * 16];
  const int tid = threadIdx.x, lane = tid & 63, g = tid >> 6;
  const int c = lane & 15, q = lane >> 4;
  int z, slice, row0, r_lo, r_hi;
  if (!step_item(p.Z, p.NS, p.rev, p.row_seg, p.B, 16 * RT, z, slice, row0, r_lo, r_hi)) return;
  LS_DECL;
  LS_STAMP(0);
  const int D = p.D, u = slice * 16 + c;
  const int uc = u < D ? u : D - 1;                       // units past D: zero weights, result discarded
  const int64_t hz = (int64_t)z * p.h_str;
  float* gz = p.G + (int64_t)z * p.g_str;
  auto body = [&](auto lt_) {
    constexpr int LT = decltype(lt_)::value;              // live 16-row tiles of this chunk
    constexpr int ROWS = 16 * LT, NCH = ROWS * (K / 4), NLD = (NCH + 255) / 256;
    // Issue order = arrival order (vmcnt retires in order): the activation rows first — they go to LDS while the weights
    // are still in flight — then the accumulator seeds and the cell state, then the weight ring.
    // activation rows -> LDS (every thread 16-byte chunks, coalesced along the rows; rows past the chunk repeat the last)
    const float* hp = p.Hprev + hz;
    f32x4 st[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = min(tid + 256 * i, NCH - 1), r = idx / (K / 4), ch = idx - r * (K / 4);
      st[i] = *reinterpret_cast<const f32x4*>(hp + (int64_t)min(row0 + r, r_hi - 1) * p.ldh + 4 * ch);
    }
    __builtin_amdgcn_sched_barrier(0);
    // accumulators start from the x-projection (+ b_ih, folded there) and b_hh: D[row = 4q + r][col = c]
    // (loaded unconditionally from clamped addresses: a load under a per-element condition is a branch + vmcnt(0) each)
    const float bv = p.bias ? p.bias[(int64_t)z * p.b_str + g * D + uc] : 0.f;
    // (two accumulators per row tile, even / odd k-steps: a single chain of this MFMA is latency- not issue-paced)
    f32x4 acc[LT][2];
#pragma unroll
    for (int rt = 0; rt < LT; ++rt) {
      acc[rt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 4; ++r)
        acc[rt][0][r] = gz[(int64_t)min(row0 + 16 * rt + 4 * q + r, r_hi - 1) * p.ldg + g * D + uc];
    }
    float cprev[LT];                                      // c_{t-1} of the elements this thread finishes (clamped: discarded past the run)
#pragma unroll
    for (int e = 0; e < LT; ++e) {
      const int pr = tid + 256 * e, rl = pr >> 4, uu = pr & 15;
      cprev[e] = p.Cprev[hz + (int64_t)min(row0 + rl, r_hi - 1) * p.ldh + min(slice * 16 + uu, D - 1)];
    }
    __builtin_amdgcn_sched_barrier(0);
    // fragment order: [slice][gate][k-block][lane][4] — one contiguous KiB per wave and k-block
    const float* wp = p.Wp + (int64_t)z * p.wp_str + ((int64_t)(slice * 4 + g) * NB * 64 + lane) * 4;
    f32x4 bq[PD];
#pragma unroll
    for (int j = 0; j < PD; ++j) bq[j] = *reinterpret_cast<const f32x4*>(wp + 256 * j);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = tid + 256 * i, r = idx / (K / 4), ch = idx - r * (K / 4);
      if (idx < NCH) *reinterpret_cast<f32x4*>(ah + r * AP + 4 * ch) = st[i];
    }
#pragma unroll
    for (int rt = 0; rt < LT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[rt][0][r] += bv;
    LS_STAMP(1);
    __syncthreads();
    LS_STAMP(2);
    const float* arow = ah + c * AP + 4 * q;
    // (the scheduler sinks loads towards their use to save registers; the order is pinned so that PD blocks of weights
    //  stay in flight — one wave per SIMD has only its own loads to hide the L2 / Infinity Cache latency — and the
    //  activation fragments are read from LDS one k-block ahead of the MFMAs that use them)
    f32x4 an[LT];
#pragma unroll
    for (int rt = 0; rt < LT; ++rt) an[rt] = *reinterpret_cast<const f32x4*>(arow + 16 * rt * AP);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int s = j % PD;
      const f32x4 b = bq[s];
      f32x4 a[LT];
#pragma unroll
      for (int rt = 0; rt < LT; ++rt) a[rt] = an[rt];
      if (j + 1 < NB) {
#pragma unroll
        for (int rt = 0; rt < LT; ++rt) an[rt] = *reinterpret_cast<const f32x4*>(arow + 16 * rt * AP + 16 * (j + 1));
      }
      if (j + PD < NB) bq[s] = *reinterpret_cast<const f32x4*>(wp + 256 * (j + PD));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int rt = 0; rt < LT; ++rt)
          acc[rt][i & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt][i], b[i], acc[rt][i & 1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    LS_STAMP(3);
    // the four gate tiles meet in LDS: xg[gate][row][unit]
#pragma unroll
    for (int rt = 0; rt < LT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) xg[(g * ROWS + 16 * rt + 4 * q + r) * 16 + c] = acc[rt][0][r] + acc[rt][1][r];
    __syncthreads();
    LS_STAMP(4);
#pragma unroll
    for (int e = 0; e < LT; ++e) {
      const int pr = tid + 256 * e, rl = pr >> 4, uu = pr & 15;
      const int row = row0 + rl, un = slice * 16 + uu;
      if (row >= r_hi || un >= D) continue;
      const float ig = sigmoid_(xg[(0 * ROWS + rl) * 16 + uu]);
      const float fg = sigmoid_(xg[(1 * ROWS + rl) * 16 + uu]);
      const float gg = tanhf(xg[(2 * ROWS + rl) * 16 + uu]);
      const float og = sigmoid_(xg[(3 * ROWS + rl) * 16 + uu]);
      const int64_t o = hz + (int64_t)row * p.ldh + un;
      const float cn = __builtin_fmaf(fg, cprev[e], ig * gg);      // (spelled out: the persistent kernel must round the same way)
      const float tc = tanhf(cn);
      float* gr = gz + (int64_t)row * p.ldg + un;
      gr[0] = ig; gr[D] = fg; gr[2 * D] = gg; gr[3 * D] = og;
      p.Cout[o] = cn;
      p.TCout[o] = tc;
      p.Hout[o] = og * tc;
    }
  };
  if constexpr (RT == 2) {
    if (r_hi - row0 > 16) body(int_c<2>{}); else body(int_c<1>{});
  } else {
    body(int_c<RT>{});
  }
  LS_STAMP(5);
  LS_FLUSH();
}

#ifdef CADRE_AB_KERNELS      // A/B build only (include/cadre_hip_ab.h): measured slower than the per-step launches
// ---------------------------------------------------------------------------------------------------------------------
// The S forward steps of all nets in ONE launch (the update's minibatch: Z = 8 nets, S = 8 steps).  A launch per step
// re-streams its 139 KB of recurrent weights per workgroup from the Infinity Cache (37 MB per step over the chip: the 8
// nets' weights do not fit the 32 MB of L2) behind a chain of dependent round trips (kernel arguments -> row segment ->
// operands -> c_{t-1}); measured 19-24 us per step against 3.6 us of MFMA time.  Here a workgroup = (net, 16 hidden
// units) keeps ITS weights in registers for all S steps (34 KiB per wave) and only the activation rows h_{t-1} move:
// the workgroups of a net exchange them through L2 with the publish / consume protocol of cdna_hip_programming.md
// Guideline 16 (recipe R1): h_t is stored write-through (agent-scope atomic stores), every storing wave drains its
// stores, the workgroup's barrier, ONE lane adds to the counter of (net, step); a consumer polls that counter (relaxed,
// one lane), ONE agent-scope acquire, then plain loads.  Results never depend on placement or timing; all Z * ceil(D/16)
// workgroups must be resident together (2 per CU fit: <= 256 VGPRs, 78 KB of LDS — 272 of 512 slots), every spin is
// bounded by the shader clock and reports through `status` instead of hanging.
struct seq_fwd_args {
  const float* Wp;      // packed recurrent weights (forward copy), net stride wp_str
  const float* bias;    // b_hh
  float* G;             // [S][B][ldg] per net: in x-projection, out activated gates
  float* Hs;            // [S+1][B][ldh] per net: slot 0 = h_{-1} (in), slots 1..S written
  float* Cs;
  float* TC;
  const int32_t* row_seg;
  int32_t* counters;    // [Z][S] arrivals, zeroed by the launch function
  int32_t* status;      // [0] != 0: a wait timed out (results invalid)
  int64_t wp_str, b_str, g_str, h_str;
  int ldg, ldh, B, D, S, Z, NS;
};

template <int NB>
__global__ __launch_bounds__(256, 2) void lstm_seq_fwd_kernel(seq_fwd_args p) {
  constexpr int AP = 552, ROWS = 32, K = 16 * NB;
  __shared__ __attribute__((aligned(16))) float ah[ROWS * AP + 256];     // (+ the tail of the last 1-KiB piece)
  __shared__ float xg[4][ROWS][16];
  __shared__ int s_ok;
  const int tid = threadIdx.x, lane = tid & 63;
  const int g = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;
  const int z = blockIdx.x % p.Z, slice = blockIdx.x / p.Z;
  int r_lo = 0, r_hi = p.B;
  if (p.row_seg) {
    const int beg = p.row_seg[2 * z], cnt = p.row_seg[2 * z + 1];
    r_lo = beg & ~31;
    r_hi = cnt > 0 ? min(p.B, (beg + cnt + 31) & ~31) : r_lo;      // a net without rows still takes part in the counters
  }
  const int D = p.D, u = slice * 16 + c, uc = u < D ? u : D - 1;
  // this wave's weights: gate g of the slice, all K — resident for the whole launch
  const float* wp = p.Wp + (int64_t)z * p.wp_str + ((int64_t)(slice * 4 + g) * NB * 64 + lane) * 4;
  f32x4 bq[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) bq[j] = *reinterpret_cast<const f32x4*>(wp + 256 * j);
  const float bv = p.bias ? p.bias[(int64_t)z * p.b_str + g * D + uc] : 0.f;
  float* gz = p.G + (int64_t)z * p.g_str;
  float* hz = p.Hs + (int64_t)z * p.h_str;
  float* cz = p.Cs + (int64_t)z * p.h_str;
  float* tz = p.TC + (int64_t)z * p.h_str;
  const int64_t slot_g = (int64_t)p.B * p.ldg, slot_h = (int64_t)p.B * p.ldh;
  const float* arow = ah + c * AP + 4 * q;
  const __amdgpu_buffer_rsrc_t rsH = __builtin_amdgcn_make_buffer_rsrc((void*)hz, 0, (int)((int64_t)(p.S + 1) * p.B * p.ldh * 4), 0x00020000);
  for (int t = 0; t < p.S; ++t) {
    if (t > 0) {
      // ---- consume: every slice of net z has published h_{t-1} (slot t).  One lane polls, one acquire, then plain loads.
      if (tid == 0) {
        int ok = 1;
        const long long t0 = __builtin_amdgcn_s_memtime();
        while (__hip_atomic_load(p.counters + z * p.S + (t - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < p.NS) {
          __builtin_amdgcn_s_sleep(2);
          if (__builtin_amdgcn_s_memtime() - t0 > 400000000ll) { ok = 0; break; }      // ~0.2 s: report, never hang
        }
        if (!ok) __hip_atomic_store(p.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_ok = ok;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();
      if (!s_ok) return;                                    // uniform over the workgroup
    }
    for (int row0 = r_lo; row0 < r_hi; row0 += ROWS) {
      __syncthreads();                                      // the previous chunk's fragment and gate-tile reads are done
      // h_{t-1} rows of the chunk -> LDS by LDS-DMA (no staging registers: the weights own the register file): piece pi =
      // 64 consecutive 16-byte slots of the padded image, lane l brings slot 64*pi + l = (row, chunk) = divmod(slot, 138)
      for (int pi = g; pi < (ROWS * (AP / 4) + 63) / 64; pi += 4) {
        const int sl = min(64 * pi + lane, ROWS * (AP / 4) - 1), r = sl / (AP / 4), ch = min(sl - r * (AP / 4), K / 4 - 1);
        const int voff = ((t * p.B + min(row0 + r, r_hi - 1)) * p.ldh + 4 * ch) * 4;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsH, (__attribute__((address_space(3))) void*)(ah + pi * 256), 16, voff, 0, 0, 0);
      }
      f32x4 acc[2][2];
      const float* gt = gz + (int64_t)t * slot_g;
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        acc[rt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r)
          acc[rt][0][r] = gt[(int64_t)min(row0 + 16 * rt + 4 * q + r, r_hi - 1) * p.ldg + g * D + uc] + bv;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces have landed
      __syncthreads();
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const f32x4 b = bq[j];
        f32x4 a[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) a[rt] = *reinterpret_cast<const f32x4*>(arow + 16 * rt * AP + 16 * j);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int rt = 0; rt < 2; ++rt)
            acc[rt][i & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt][i], b[i], acc[rt][i & 1], 0, 0, 0);
      }
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) xg[g][16 * rt + 4 * q + r][c] = acc[rt][0][r] + acc[rt][1][r];
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int pr = tid + 256 * e, rl = pr >> 4, uu = pr & 15;
        const int row = row0 + rl, un = slice * 16 + uu;
        if (row >= r_hi || un >= D) continue;
        const float ig = sigmoid_(xg[0][rl][uu]);
        const float fg = sigmoid_(xg[1][rl][uu]);
        const float gg = tanhf(xg[2][rl][uu]);
        const float og = sigmoid_(xg[3][rl][uu]);
        const int64_t o = (int64_t)row * p.ldh + un;
        const float cn = __builtin_fmaf(fg, cz[(int64_t)t * slot_h + o], ig * gg);      // c_{t-1}: this workgroup's own store of the last step
        const float tc = tanhf(cn);
        float* gr = gz + (int64_t)t * slot_g + (int64_t)row * p.ldg + un;
        gr[0] = ig; gr[D] = fg; gr[2 * D] = gg; gr[3 * D] = og;
        cz[(int64_t)(t + 1) * slot_h + o] = cn;
        tz[(int64_t)(t + 1) * slot_h + o] = tc;
        // h_t is what the other workgroups of the net read: write-through store (agent scope)
        __hip_atomic_store(hz + (int64_t)(t + 1) * slot_h + o, og * tc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    // ---- publish: every storing wave drains its stores, the workgroup's barrier, one lane signals
    if (t + 1 < p.S) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(p.counters + z * p.S + t, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
#endif


__global__ void zero_i32_kernel(int32_t* p, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = 0;
}

struct bwd_args {
  const float* Wp;      // transposed recurrent weights in fragment order (cadre_pack_lstm_weights), net stride wp_str
  const float* dGp_in;  // dG_t in fragment order (null: no product, dh = dh_in only), net stride gp_str
  float* dGp_out;       // dG_{t-1} in fragment order: [16-row tile][k-block][lane][4], for the next step
  float* dG_out;        // dG_{t-1} row-major [B][ldg] of net 0, net stride g_str
  const float* G_act;   // activated gates of step t-1
  const float* dh_in;   // upstream dL/dh_{t-1} [B][ldh] added to the product (may be null), net stride d_str
  float* dC;            // in: dL/dc_{t-1}; out: dL/dc_{t-2}; [B][ldh], net stride d_str
  const float* TC;      // tanh(c_{t-1})
  const float* Cprev;   // c_{t-2}
  const int32_t* commands;   // [2][B] (null: no ownership mask)
  const int32_t* row_seg;
  int64_t wp_str, gp_str, g_str, h_str, d_str;
  int ldg, ldh, B, D, C, Z, NS, rev;
};

// wave w multiplies k-blocks [w*NB, (w+1)*NB) of the 4*NB blocks of the (zero padded) gate axis.
// The fragment-order copies of dG index their 16-row tiles RELATIVE to the net's first row (r_lo): producer (step t) and
// consumer (step t-1) see the same row segment.
template <int RT, int NB, bool GEMM>
__global__ __launch_bounds__(256) void lstm_step_bwd_kernel(bwd_args p) {
  __shared__ float xs[4 * 16 * RT * 16];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int c = lane & 15, q = lane >> 4;
  int z, slice, row0, r_lo, r_hi;
  if (!step_item(p.Z, p.NS, p.rev, p.row_seg, p.B, 16 * RT, z, slice, row0, r_lo, r_hi)) return;
  LS_DECL;
  LS_STAMP(0);
  const int D = p.D;
  const int64_t gzo = (int64_t)z * p.g_str, hz = (int64_t)z * p.h_str, dz = (int64_t)z * p.d_str;
  auto body = [&](auto lt_) {
    constexpr int LT = decltype(lt_)::value;              // live 16-row tiles of this chunk
    constexpr int PD = LT == 1 ? 16 : 11;                 // k-blocks ((1 + LT) KiB each) in flight per wave
    constexpr int ROWS = 16 * LT;
    // the cell-backward operands of the elements this thread finishes: requested before the product, used after it
    // (clamped addresses, discarded past the run — no load under a per-element condition)
    float e_ga[LT][4], e_tc[LT], e_dc[LT], e_cp[LT], e_dh[LT];
    int e_cmd[LT];
#pragma unroll
    for (int e = 0; e < LT; ++e) {
      const int pr = tid + 256 * e, rl = pr >> 4, uu = pr & 15;
      const int row = min(row0 + rl, r_hi - 1), un = min(slice * 16 + uu, D - 1);
      const float* ga = p.G_act + gzo + (int64_t)row * p.ldg + un;
      e_ga[e][0] = ga[0]; e_ga[e][1] = ga[D]; e_ga[e][2] = ga[2 * D]; e_ga[e][3] = ga[3 * D];
      e_tc[e] = p.TC[hz + (int64_t)row * p.ldh + un];
      e_cp[e] = p.Cprev[hz + (int64_t)row * p.ldh + un];
      e_dc[e] = p.dC[dz + (int64_t)row * p.ldh + un];
      e_dh[e] = p.dh_in ? p.dh_in[dz + (int64_t)row * p.ldh + un] : 0.f;
      e_cmd[e] = p.commands ? p.commands[(z / p.C) * p.B + row] : z % p.C;
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (GEMM) {
      // fragment order: weights [slice][K quarter][k-block][lane][4], dG_t [16-row tile][k-block of the whole gate axis][lane][4]
      const float* wp = p.Wp + (int64_t)z * p.wp_str + ((int64_t)(slice * 4 + w) * NB * 64 + lane) * 4;
      const float* ap[LT];
#pragma unroll
      for (int rt = 0; rt < LT; ++rt)
        ap[rt] = p.dGp_in + (int64_t)z * p.gp_str + (((int64_t)((row0 - r_lo) / 16 + rt) * 4 + w) * NB * 64 + lane) * 4;
      // two accumulators per row tile (even / odd k-blocks): a single chain of this MFMA is latency- not issue-paced
      f32x4 acc[LT][2];
#pragma unroll
      for (int rt = 0; rt < LT; ++rt) acc[rt][0] = acc[rt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 bq[PD], aq[PD][LT];
#pragma unroll
      for (int j = 0; j < PD; ++j) {
        bq[j] = *reinterpret_cast<const f32x4*>(wp + 256 * j);
#pragma unroll
        for (int rt = 0; rt < LT; ++rt) aq[j][rt] = *reinterpret_cast<const f32x4*>(ap[rt] + 256 * j);
      }
      __builtin_amdgcn_sched_barrier(0);
      LS_STAMP(1);
      LS_STAMP(2);
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int s = j % PD;
        const f32x4 b = bq[s];
        f32x4 a[LT];
#pragma unroll
        for (int rt = 0; rt < LT; ++rt) a[rt] = aq[s][rt];
        if (j + PD < NB) {
          bq[s] = *reinterpret_cast<const f32x4*>(wp + 256 * (j + PD));
#pragma unroll
          for (int rt = 0; rt < LT; ++rt) aq[s][rt] = *reinterpret_cast<const f32x4*>(ap[rt] + 256 * (j + PD));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int rt = 0; rt < LT; ++rt)
            acc[rt][i & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt][i], b[i], acc[rt][i & 1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      LS_STAMP(3);
#pragma unroll
      for (int rt = 0; rt < LT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) xs[(w * ROWS + 16 * rt + 4 * q + r) * 16 + c] = acc[rt][0][r] + acc[rt][1][r];
      __syncthreads();
      LS_STAMP(4);
    }
#pragma unroll
    for (int e = 0; e < LT; ++e) {
      const int pr = tid + 256 * e, rl = pr >> 4, uu = pr & 15;
      const int row = row0 + rl, un = slice * 16 + uu;
      if (row >= r_hi || un >= D) continue;
      float* dg = p.dG_out + gzo + (int64_t)row * p.ldg + un;
      const int64_t od = dz + (int64_t)row * p.ldh + un;
      // fragment-order copy: element (row, n) at [((row - r_lo) / 16) * 4*NB + n / 16][((n % 16) / 4) * 16 + (row - r_lo) % 16][n % 4]
      const int rr = row - r_lo;
      float* dgp = p.dGp_out + (int64_t)z * p.gp_str + ((int64_t)(rr >> 4) * 4 * NB * 64 + (rr & 15)) * 4;
      auto put = [&](int gate, float v) {
        const int n = gate * D + un;
        dg[gate * D] = v;
        dgp[(n >> 4) * 256 + ((n >> 2) & 3) * 64 + (n & 3)] = v;
      };
      if (e_cmd[e] != z % p.C) {                           // row of another command net (unsorted minibatch): exact zeros
        put(0, 0.f); put(1, 0.f); put(2, 0.f); put(3, 0.f);
        p.dC[od] = 0.f;
        continue;
      }
      float dht = e_dh[e];
      if constexpr (GEMM)
        dht += (xs[(0 * ROWS + rl) * 16 + uu] + xs[(1 * ROWS + rl) * 16 + uu]) + (xs[(2 * ROWS + rl) * 16 + uu] + xs[(3 * ROWS + rl) * 16 + uu]);
      const float ig = e_ga[e][0], fg = e_ga[e][1], gg = e_ga[e][2], og = e_ga[e][3];
      const float tc = e_tc[e];
      const float dct = e_dc[e] + dht * og * (1.f - tc * tc);
      const float cp = e_cp[e];
      put(0, dct * gg * ig * (1.f - ig));
      put(1, dct * cp * fg * (1.f - fg));
      put(2, dct * ig * (1.f - gg * gg));
      put(3, dht * tc * og * (1.f - og));
      p.dC[od] = dct * fg;
    }
  };
  if constexpr (RT == 2) {
    if (r_hi - row0 > 16) body(int_c<2>{}); else body(int_c<1>{});
  } else {
    body(int_c<RT>{});
  }
  LS_STAMP(5);
  LS_FLUSH();
}

struct dw_args {
  const float* dG;      // [S][B][ldg] gate gradients of net 0, net stride g_str
  const float* Hs;      // h_{t-1} of step t at Hs + t*B*ldh: [S+1][B][ldh], net stride h_str
  const float* X;       // x_t: [S][B][ldh] of input slot 0; net z reads slot z / x_div, slot stride x_str
  float* dWhh;          // [H4][ldw] of net 0, net stride w_str (gradient arena)
  float* dWih;
  float* dbih;          // [H4]
  float* dbhh;
  const int32_t* row_seg;
  int64_t g_str, h_str, x_str, w_str;
  int ldg, ldh, ldw, B, S, H4, N, Z, x_div, MG, NG;
};

// dW_hh = sum_t dG_t^T h_{t-1}, dW_ih = sum_t dG_t^T x_t, db_ih = db_hh = column sums of dG (autograd of
// models.py:139-152 over the S steps) in ONE launch.  Both operands are "k-major" — the reduction index is the row
// (sample) index, rows are contiguous in the output index — so an MFMA fragment is a plain coalesced load: lane
// (c = lane & 15, q = lane >> 4) of k-step s loads 16 bytes of row 4s + q: dG[row][m0 + 4c ..+3] feeds the A operand of
// FOUR row tiles (output rows m0 + 4c + i, i = 0..3, interleaved), Y[row][n0 + 4c ..+3] the B operand of four column
// tiles: a wave owns a 64*TM x 64 output tile = 16*TM v_mfma_f32_16x16x4_f32 per 1 + TM KiB loads, no LDS, and its results
// are 16 (m) x 4 (n) blocks per lane: 16-byte stores, 256 B per output row.  Only a net's own run of rows is
// multiplied (4-row k-steps from its first row; the tail of the last step is masked).  Work item = wave tile; the net is the
// fastest index of the 1-D grid (one XCD per net: its dG / h / x rows stay in that XCD's L2).
// three waves per SIMD with four k-steps in flight each (157 VGPRs) rather than two with eight (201): the launch's
// ramp, tile epilogues and tail overlap with more MFMA work — 81 -> 76 us (minibatch 64), 229 -> 221 us (256)
#ifndef DW_OCC
#define DW_OCC 3
#endif
#ifndef DW_PD
#define DW_PD 4
#endif
template <int TM>
__global__ __launch_bounds__(256, TM == 1 ? DW_OCC : 1) void lstm_dw_kernel(dw_args p) {
  constexpr int PD = TM == 1 ? DW_PD : 6;                     // k-steps ((1 + TM) KiB per wave) in flight
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably wave-uniform: the buffer descriptors built from it
                                                                 // stay in scalar registers (no waterfall loop per load)
  const int c = lane & 15, q = lane >> 4;
  const int z = blockIdx.x % p.Z;
  const int tile = (blockIdx.x / p.Z) * 4 + wave;         // 0 .. 2*MG*NG: (kind, m-group, n-group), n fastest
  const int per_kind = p.MG * p.NG;
  if (tile >= 2 * per_kind) return;
  const int kind = tile / per_kind, tk = tile - kind * per_kind;
  const int mg = tk / p.NG, ng = tk - mg * p.NG;
  int lo = 0, hi = p.B;
  if (p.row_seg) {
    const int beg = p.row_seg[2 * z], cnt = p.row_seg[2 * z + 1];
    lo = beg;                                             // exactly the net's run: the step kernels write no other row
    hi = cnt > 0 ? min(p.B, beg + cnt) : lo;
  }
  const int nb = (hi - lo + 3) >> 2;                      // k-steps per time step
  const int KT = nb * p.S;
  const int m0 = 64 * TM * mg, n0 = 64 * ng;
  const int nc = (n0 + 4 * c < p.N) ? c : 0;              // column chunk past the row pitch: a valid one, discarded
  // Operands through buffer descriptors: the per-lane part of an address (row q of the k-step, 16-byte column chunk c) is
  // a constant VGPR, the k-step's row offset a SCALAR — no vector ALU work per k-step (64-bit row * pitch products per
  // lane cost as much issue time as a third of the MFMAs) — and rows past the end of the tensor read as zeros.
  const float* gbase = p.dG + (int64_t)z * p.g_str + m0;
  const float* ybase = (kind == 0 ? p.Hs + (int64_t)z * p.h_str : p.X + (int64_t)(z / p.x_div) * p.x_str) + n0;
  const unsigned g_bytes = (unsigned)((int64_t)p.S * p.B * p.ldg * 4 - (int64_t)m0 * 4);
  const unsigned y_bytes = (unsigned)((int64_t)p.S * p.B * p.ldh * 4 - (int64_t)n0 * 4);
  const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc((void*)gbase, 0, (int)g_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc((void*)ybase, 0, (int)y_bytes, 0x00020000);
  const int voff_g = (q * p.ldg + 4 * c) * 4, voff_y = (q * p.ldh + 4 * nc) * 4;
  f32x4 acc[TM][4][4];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[tm][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 bsum[TM];                                         // column sums of dG (this lane's rows 4s + q)
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) bsum[tm] = f32x4{0.f, 0.f, 0.f, 0.f};
  // k-step ks -> rows t*B + lo + 4*(ks % nb) + q, walked incrementally in scalar registers (state of the next k-step to
  // request).  Requests past the last k-step go out of bounds: zeros, nothing to undo.  Rows of the run's last 4-row step
  // past its end belong to another net (never written in this net's dG: stale) or to the next time step: the dG operand
  // and h / x operands of those rows are masked to zero when they are used (MASK).
  int t_n = 0, b_n = 0, ks_n = 0;
  const bool MASK = ((hi - lo) & 3) != 0;
  const bool want_bsum = kind == 0 && ng == 0;
  f32x4 aq[PD][TM], yq[PD];
  auto request = [&](int slot) {
    const int row = t_n * p.B + lo + 4 * b_n;             // scalar
    const int so_g = ks_n < KT ? row * p.ldg * 4 : 0x7fffffff;
    const int so_y = ks_n < KT ? row * p.ldh * 4 : 0x7fffffff;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
      aq[slot][tm] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsG, voff_g + 256 * tm, so_g, 0));
    yq[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsY, voff_y, so_y, 0));
    ++ks_n;
    const int b1 = b_n + 1;
    const int wrap = b1 == nb ? 1 : 0;
    b_n = wrap ? 0 : b1;
    t_n += wrap;
  };
  // (KT == 0 — a net without rows — runs no iteration: every prologue request is out of bounds)
#pragma unroll
  for (int s = 0; s < PD; ++s) request(s);
  __builtin_amdgcn_sched_barrier(0);                     // (PD k-steps stay in flight: the loads are not sunk to their use)
  int b_c = 0;                                            // k-step inside its time step, consumer side
  for (int ks = 0; ks < KT; ks += PD) {
#pragma unroll
    for (int s = 0; s < PD; ++s) {
      f32x4 av[TM];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) av[tm] = aq[s][tm];
      f32x4 yv = yq[s];
      request(s);
      __builtin_amdgcn_sched_barrier(0);
      // (only the run's LAST 4-row step of a time step can hold foreign rows: the selects — eight vector instructions, paid in matrix
      //  cycles in fp32 — run there and nowhere else; the column sums only in the waves that store them.  Round 6.)
#ifdef DW_MASK_ALWAYS
      if (MASK) {
#else
      if (MASK && b_c == nb - 1) {
#endif
        const bool dead = lo + 4 * b_c + q >= hi;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int i = 0; i < 4; ++i) av[tm][i] = dead ? 0.f : av[tm][i];
#pragma unroll
        for (int i = 0; i < 4; ++i) yv[i] = dead ? 0.f : yv[i];
      }
      const int b1 = b_c + 1;
      b_c = b1 == nb ? 0 : b1;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
#ifdef DW_MASK_ALWAYS
        bsum[tm] += av[tm];
#else
        if (want_bsum) bsum[tm] += av[tm];
#endif
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[tm][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tm][i], yv[j], acc[tm][i][j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // ---- store: lane (c, q) holds rows m0 + 64*tm + 16q + 4r + i, columns n0 + 4c .. +3
  float* out = (kind == 0 ? p.dWhh : p.dWih) + (int64_t)z * p.w_str;
  if (n0 + 4 * c < p.N) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int m = m0 + 64 * tm + 16 * q + 4 * r + i;
          if (m < p.H4)
            *reinterpret_cast<f32x4*>(out + (int64_t)m * p.ldw + n0 + 4 * c) =
                f32x4{acc[tm][i][0][r], acc[tm][i][1][r], acc[tm][i][2][r], acc[tm][i][3][r]};
        }
  }
  if (kind == 0 && ng == 0) {                              // bias gradients: sum the four row quarters (q) of the wave
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v = bsum[tm][i];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        bsum[tm][i] = v;
      }
      if (q == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int m = m0 + 64 * tm + 4 * c + i;
          if (m < p.H4) {
            p.dbih[(int64_t)z * p.w_str + m] = bsum[tm][i];
            p.dbhh[(int64_t)z * p.w_str + m] = bsum[tm][i];
          }
        }
      }
    }
  }
}

#ifdef CADRE_AB_KERNELS
// The same products with the operands of a workgroup's four wave tiles SHARED through LDS: the
// workgroup owns a 128 x 128 output tile (2 x 2 waves of 64 x 64); per k-step (4 rows) the two dG blocks (4 rows x 64
// gate columns = 1 KiB each) and the two h / x blocks come in ONCE by LDS-DMA — a wave issues all loads of one block
// kind, lane-linear: lane (c, q) of a DMA instruction writes exactly the 16 bytes lane (c, q) of a consumer reads — and
// each is read by two waves: half the L2 traffic of lstm_dw_kernel<1> (1.3 GB per launch at minibatch 256, 66 % of its
// wave cycles waiting on loads).  KS k-steps per stage, two stages, counted vmcnt + two barriers per stage.
// MEASURED AND NOT ADOPTED (A/B build, CADRE_DW_LDS=4|8): bit-level parity with the tests, 243-253 vs 218 us at minibatch
// 256 and 92-99 vs 84 us at 64 (profiles/r04_lstm_dw_lds_sharing.txt) — like the 128 x 64 wave tile before it: operand
// BYTES are not what the launch waits for; the barriers and one wave less per SIMD cost more than the halved traffic saves.
template <int KS>
__global__ __launch_bounds__(256, 2) void lstm_dw_lds_kernel(dw_args p, int MG2, int NG2) {
  __shared__ __attribute__((aligned(16))) char lds[2 * KS * 4096];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int c = lane & 15, q = lane >> 4;
  const int z = blockIdx.x % p.Z;
  const int t2 = blockIdx.x / p.Z;                        // (kind, mg2, ng2), ng2 fastest
  const int per_kind = MG2 * NG2;
  const int kind = t2 / per_kind, tk = t2 - kind * per_kind;
  const int mg2 = tk / NG2, ng2 = tk - mg2 * NG2;
  const int mg = 2 * mg2 + wm, ng = 2 * ng2 + wn;
  const bool live = mg < p.MG && ng < p.NG;               // (a wave tile past the matrix: takes part in staging and barriers only)
  int lo = 0, hi = p.B;
  if (p.row_seg) {
    const int beg = p.row_seg[2 * z], cnt = p.row_seg[2 * z + 1];
    lo = beg;
    hi = cnt > 0 ? min(p.B, beg + cnt) : lo;
  }
  const int nb = (hi - lo + 3) >> 2;
  const int KT = nb * p.S;
  const int m0 = 128 * mg2, n0 = 128 * ng2;
  // this wave stages block kind `wave`: 0 / 1 = dG columns m0 + 64*{0,1}, 2 / 3 = h|x columns n0 + 64*{0,1}
  const bool stage_g = wave < 2;
  const int half = wave & 1;
  const float* gbase = p.dG + (int64_t)z * p.g_str + m0;
  const float* ybase = (kind == 0 ? p.Hs + (int64_t)z * p.h_str : p.X + (int64_t)(z / p.x_div) * p.x_str) + n0;
  const unsigned g_bytes = (unsigned)((int64_t)p.S * p.B * p.ldg * 4 - (int64_t)m0 * 4);
  const unsigned y_bytes = (unsigned)((int64_t)p.S * p.B * p.ldh * 4 - (int64_t)n0 * 4);
  const __amdgpu_buffer_rsrc_t rsS = stage_g ? __builtin_amdgcn_make_buffer_rsrc((void*)gbase, 0, (int)g_bytes, 0x00020000)
                                             : __builtin_amdgcn_make_buffer_rsrc((void*)ybase, 0, (int)y_bytes, 0x00020000);
  const int ld_s = stage_g ? p.ldg : p.ldh;
  // column chunk past the row pitch (last n-half of h / x): a valid chunk instead, its products are never stored
  const int col = 64 * half + 4 * c;
  const int colv = (!stage_g && n0 + col >= p.ldh) ? 4 * c : col;
  const int voff = (q * ld_s + colv) * 4;
  int t_n = 0, b_n = 0, ks_n = 0;                         // producer side: next k-step to request
  auto issue_stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < KS; ++i) {
      const int row = t_n * p.B + lo + 4 * b_n;
      const int so = ks_n < KT ? row * ld_s * 4 : 0x7fffffff;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsS, (__attribute__((address_space(3))) void*)(lds + (buf * KS + i) * 4096 + wave * 1024), 16, voff, so, 0, 0);
      ++ks_n;
      const int b1 = b_n + 1;
      const int wrap = b1 == nb ? 1 : 0;
      b_n = wrap ? 0 : b1;
      t_n += wrap;
    }
  };
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
  const bool MASK = ((hi - lo) & 3) != 0;
  int b_c = 0;
  const int NS = (KT + KS - 1) / KS;
  issue_stage(0);
  for (int s = 0; s < NS; ++s) {
    issue_stage((s + 1) & 1);                             // (past the end: out-of-bounds requests, zeros)
    // raw barriers + this wave's own counted vmcnt (a __syncthreads() fence would drain vmcnt to 0 and with it the prefetch)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KS) : "memory");       // this wave's loads of stage s have landed
    __builtin_amdgcn_s_barrier();                         // ... and everybody else's
    asm volatile("" ::: "memory");
    const char* sb = lds + (s & 1) * KS * 4096;
    if (live) {
#pragma unroll
      for (int i = 0; i < KS; ++i) {
        f32x4 av = *reinterpret_cast<const f32x4*>(sb + i * 4096 + wm * 1024 + lane * 16);
        f32x4 yv = *reinterpret_cast<const f32x4*>(sb + i * 4096 + 2048 + wn * 1024 + lane * 16);
        if (MASK) {
          const bool dead = lo + 4 * b_c + q >= hi;
#pragma unroll
          for (int e = 0; e < 4; ++e) { av[e] = dead ? 0.f : av[e]; yv[e] = dead ? 0.f : yv[e]; }
        }
        const int b1 = b_c + 1;
        b_c = b1 == nb ? 0 : b1;
        bsum += av;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[a][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], yv[j], acc[a][j], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // (fragment reads done: they fed the MFMAs above)
    __builtin_amdgcn_s_barrier();                         // buffer s & 1 is free: the next iteration refills it
    asm volatile("" ::: "memory");
  }
  if (!live) return;
  const int mw = m0 + 64 * wm, nw = n0 + 64 * wn;
  float* out = (kind == 0 ? p.dWhh : p.dWih) + (int64_t)z * p.w_str;
  if (nw + 4 * c < p.N) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = mw + 16 * q + 4 * r + i;
        if (m < p.H4)
          *reinterpret_cast<f32x4*>(out + (int64_t)m * p.ldw + nw + 4 * c) = f32x4{acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
      }
  }
  if (kind == 0 && ng == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v = bsum[i];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      bsum[i] = v;
    }
    if (q == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = mw + 4 * c + i;
        if (m < p.H4) {
          p.dbih[(int64_t)z * p.w_str + m] = bsum[i];
          p.dbhh[(int64_t)z * p.w_str + m] = bsum[i];
        }
      }
    }
  }
}

#endif

// Recurrent weights W [4*D][ldw] (k contiguous) of `Z` nets -> fragment order for both directions, NB = ldw / 16:
//   fwd[z][slice][gate g][k-block j][lane (q, c)][i] = W[g*D + 16*slice + c][16j + 4q + i]          (0 past unit D)
//   bwd[z][slice][quarter w][k-block j][lane (q, c)][i] = W[n = 16*(NB*w + j) + 4q + i][16*slice + c]  (0 past 4*D / D)
// one workgroup per (slice, gate | quarter, net) and direction, one KiB per wave and k-block.
__global__ __launch_bounds__(256) void pack_lstm_weights_kernel(const float* W, int64_t w_str, int ldw, int D, int NB, float* fwd,
                                                                float* bwd, int64_t p_str) {
  const int slice = blockIdx.x, gw = blockIdx.y >> 1, dir = blockIdx.y & 1, z = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, q = lane >> 4;
  const float* w = W + (int64_t)z * w_str;
  const int u = 16 * slice + c;
  const int64_t blk = (int64_t)(slice * 4 + gw) * NB;
  for (int j = wave; j < NB; j += 4) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (dir == 0) {
      if (u < D) v = *reinterpret_cast<const f32x4*>(w + (int64_t)(gw * D + u) * ldw + 16 * j + 4 * q);
      *reinterpret_cast<f32x4*>(fwd + (int64_t)z * p_str + ((blk + j) * 64 + lane) * 4) = v;
    } else {
      const int n = 16 * (NB * gw + j) + 4 * q;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (n + i < 4 * D && u < D) v[i] = w[(int64_t)(n + i) * ldw + u];
      *reinterpret_cast<f32x4*>(bwd + (int64_t)z * p_str + ((blk + j) * 64 + lane) * 4) = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// MLP towers of the update (models.py:171-177 critic, distributions.py:34-40 actor): Linear(530 -> 128) ReLU
// Linear(128 -> 128) ReLU Linear(128 -> n_out), 16 towers (tower z2 = 2*net + {actor, critic}), each on its net's run of
// rows.  Per tower the products are a few MFLOP: as 17 tile-GEMM / column-sum / mask launches of 5-9 us they cost 98 us
// of a 650 us minibatch step.  Three launches instead: the three layers forward; the dX chain backward (dO3 -> dA2 ->
// dA1 -> dh_S, both towers of a net summed in registers); the three weight gradients + bias gradients.
// Weights are read in place (row-major, k contiguous): forward B fragments are 16 rows x 64 bytes per wave instruction,
// backward ones 4 rows x 64 bytes of single floats — the slow scattered pattern the LSTM kernels avoid by packing, but a
// tower's weights are 377 KB read by a handful of workgroups once per step: not worth a packed copy.
constexpr int MLP_HID = 128, MLP_NP = 64, MLP_K1 = 544;

struct mlp_fwd_args {
  const float* P;        // tower z2 at P + z2 * t_str; weights [out][in] row-major at the offsets below
  const float* Hin;      // h_S rows [B][ldh] of net z2 / 2 at Hin + (z2 / 2) * h_str
  float* A1;             // [Z2][B][128] relu(layer 1)
  float* A2;             // [Z2][B][128] relu(layer 2)
  float* O3;             // [Z2][B][64]  layer 3 (+ bias)
  const int32_t* row_seg;
  int64_t t_str, h_str;
  int o_w1, o_b1, o_w2, o_b2, o_w3, o_b3;
  int ldh, B, Z2;
};

__device__ __forceinline__ bool mlp_item(int n_z, const int32_t* row_seg, int B, int& z, int& net_of_seg, int& row0, int& r_hi,
                                         int seg_shift) {
  z = blockIdx.x % n_z;
  const int chunk = blockIdx.x / n_z;
  net_of_seg = z >> seg_shift;
  int r_lo = 0;
  r_hi = B;
  if (row_seg) {
    const int beg = row_seg[2 * net_of_seg], cnt = row_seg[2 * net_of_seg + 1];
    if (cnt <= 0) return false;
    r_lo = beg;
    r_hi = min(B, beg + cnt);
  }
  row0 = r_lo + 16 * chunk;
  return row0 < r_hi;
}

// workgroup = (tower, 16 rows of its run), 8 waves: wave w owns hidden units 16w .. 16w+15 of layers 1 and 2, output
// columns 16w .. of layer 3 (w < 4).  Activations meet in LDS between the layers.
__global__ __launch_bounds__(512) void mlp_fwd_kernel(mlp_fwd_args p) {
  constexpr int AP = 552, HP = 132, PD = 8;               // (all 34 layer-1 fragments up front measured slower: 19.2 vs 15.6 us —
                                                           //  the scattered 64-byte pieces are paced by the texture addresser)
  __shared__ __attribute__((aligned(16))) float hs[16 * AP];
  __shared__ __attribute__((aligned(16))) float a1s[16 * HP];
  __shared__ __attribute__((aligned(16))) float a2s[16 * HP];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int c = lane & 15, q = lane >> 4;
  int z2, net, row0, r_hi;
  if (!mlp_item(p.Z2, p.row_seg, p.B, z2, net, row0, r_hi, 1)) return;
  const float* P = p.P + (int64_t)z2 * p.t_str;
  // h rows -> LDS (16 rows x 136 chunks of 16 bytes; rows past the run repeat the last)
  const float* hp = p.Hin + (int64_t)net * p.h_str;
  constexpr int NCH = 16 * (MLP_K1 / 4), NLD = (NCH + 511) / 512;
  f32x4 st[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int idx = min(tid + 512 * i, NCH - 1), r = idx / (MLP_K1 / 4), ch = idx - r * (MLP_K1 / 4);
    st[i] = *reinterpret_cast<const f32x4*>(hp + (int64_t)min(row0 + r, r_hi - 1) * p.ldh + 4 * ch);
  }
  __builtin_amdgcn_sched_barrier(0);
  const int n = 16 * w + c;
  const float* w1 = P + p.o_w1 + (int64_t)n * MLP_K1 + 4 * q;
  f32x4 bq[PD];
#pragma unroll
  for (int j = 0; j < PD; ++j) bq[j] = *reinterpret_cast<const f32x4*>(w1 + 16 * j);
  // layer 2 / 3 fragments of this wave: requested now, used after layer 1
  f32x4 b2q[MLP_HID / 16], b3q[MLP_HID / 16];
  const float* w2 = P + p.o_w2 + (int64_t)n * MLP_HID + 4 * q;
  const float* w3 = P + p.o_w3 + (int64_t)(n & (MLP_NP - 1)) * MLP_HID + 4 * q;
#pragma unroll
  for (int j = 0; j < MLP_HID / 16; ++j) {
    b2q[j] = *reinterpret_cast<const f32x4*>(w2 + 16 * j);
    b3q[j] = *reinterpret_cast<const f32x4*>(w3 + 16 * j);
  }
  const float bias1 = P[p.o_b1 + n], bias2 = P[p.o_b2 + n], bias3 = P[p.o_b3 + (n & (MLP_NP - 1))];
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int idx = tid + 512 * i, r = idx / (MLP_K1 / 4), ch = idx - r * (MLP_K1 / 4);
    if (idx < NCH) *reinterpret_cast<f32x4*>(hs + r * AP + 4 * ch) = st[i];
  }
  __syncthreads();
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  const float* arow = hs + c * AP + 4 * q;
#pragma unroll
  for (int j = 0; j < MLP_K1 / 16; ++j) {
    const int s = j % PD;
    const f32x4 b = bq[s];
    const f32x4 a = *reinterpret_cast<const f32x4*>(arow + 16 * j);
    if (j + PD < MLP_K1 / 16) bq[s] = *reinterpret_cast<const f32x4*>(w1 + 16 * (j + PD));
    __builtin_amdgcn_sched_barrier(0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], acc1, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  const int64_t ob = (int64_t)z2 * p.B;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * q + r;
    const float v = fmaxf((acc0[r] + acc1[r]) + bias1, 0.f);
    a1s[row * HP + n] = v;
    if (row0 + row < r_hi) p.A1[(ob + row0 + row) * MLP_HID + n] = v;
  }
  __syncthreads();
  acc0 = f32x4{0.f, 0.f, 0.f, 0.f}; acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < MLP_HID / 16; ++j) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(a1s + c * HP + 16 * j + 4 * q);
    const f32x4 b = b2q[j];
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], acc1, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * q + r;
    const float v = fmaxf((acc0[r] + acc1[r]) + bias2, 0.f);
    a2s[row * HP + n] = v;
    if (row0 + row < r_hi) p.A2[(ob + row0 + row) * MLP_HID + n] = v;
  }
  __syncthreads();
  if (w < MLP_NP / 16) {
    acc0 = f32x4{0.f, 0.f, 0.f, 0.f}; acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < MLP_HID / 16; ++j) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(a2s + c * HP + 16 * j + 4 * q);
      const f32x4 b = b3q[j];
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], acc1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * q + r;
      if (row0 + row < r_hi) p.O3[(ob + row0 + row) * MLP_NP + n] = (acc0[r] + acc1[r]) + bias3;
    }
  }
}

struct mlp_bwd_args {
  const float* P;
  const float* dO3;      // [Z2][B][64]
  const float* A1;       // [Z2][B][128] (ReLU masks)
  const float* A2;
  float* dA1;            // [Z2][B][128] out: dL/d(pre-activation of layer 1)
  float* dA2;
  float* dH;             // [Z][B][ldh] out: dL/dh_S, both towers of the net summed
  const int32_t* row_seg;
  int64_t t_str, d_str;
  int o_w1, o_w2, o_w3;
  int ldh, B, Z, NG;     // NG column groups of dh: workgroup (net, 16 rows, g) owns the 16-column tiles t % NG == g
};

// B operand of a dX product: b[k][j] = W[k0 + k][j0 + j] with W row-major [k][ld] — lane (c = j, q): rows 4q .. 4q+3 of
// k-block jb, one float each (64 contiguous bytes per row and wave instruction)
__device__ __forceinline__ f32x4 ld_kmajor(const float* W, int ld, int jb, int q, int col) {
  const float* s = W + (int64_t)(16 * jb + 4 * q) * ld + col;
  return f32x4{s[0], s[ld], s[2 * ld], s[3 * ld]};
}

__global__ __launch_bounds__(512) void mlp_bwd_kernel(mlp_bwd_args p) {
  constexpr int HP = 132, OP = 68, NT = MLP_K1 / 16;      // 34 column tiles of dh
  __shared__ __attribute__((aligned(16))) float d3s[16 * OP];
  __shared__ __attribute__((aligned(16))) float d2s[16 * HP];
  __shared__ __attribute__((aligned(16))) float d1s[16 * HP];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int c = lane & 15, q = lane >> 4;
  const int zg = blockIdx.x % (p.Z * p.NG);
  const int z = zg % p.Z, g = zg / p.Z;
  const int chunk = blockIdx.x / (p.Z * p.NG);
  int r_lo = 0, r_hi = p.B;
  if (p.row_seg) {
    const int beg = p.row_seg[2 * z], cnt = p.row_seg[2 * z + 1];
    if (cnt <= 0) return;
    r_lo = beg;
    r_hi = min(p.B, beg + cnt);
  }
  const int row0 = r_lo + 16 * chunk;
  if (row0 >= r_hi) return;
  // dh tiles of this wave: t = g + NG * (w + 8 m), m = 0 .. : at most MT
  constexpr int MT = 2;
  f32x4 acc_h[MT][2];
#pragma unroll
  for (int m = 0; m < MT; ++m) acc_h[m][0] = acc_h[m][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int n = 16 * w + c;
  for (int tw = 0; tw < 2; ++tw) {
    const int z2 = 2 * z + tw;
    const float* P = p.P + (int64_t)z2 * p.t_str;
    const int64_t ob = (int64_t)z2 * p.B;
    if (tw) __syncthreads();                               // the previous tower's tiles have been read
    if (tid < 256) {
      const int r = tid >> 4, ch = tid & 15;
      *reinterpret_cast<f32x4*>(d3s + r * OP + 4 * ch) =
          *reinterpret_cast<const f32x4*>(p.dO3 + (ob + min(row0 + r, r_hi - 1)) * MLP_NP + 4 * ch);
    }
    // ReLU masks of this lane's elements (rows 4q + r, column n)
    float m2[4], m1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t o = (ob + min(row0 + 4 * q + r, r_hi - 1)) * MLP_HID + n;
      m2[r] = p.A2[o];
      m1[r] = p.A1[o];
    }
    f32x4 b3[MLP_NP / 16], b2[MLP_HID / 16], b1[MT][MLP_HID / 16];
#pragma unroll
    for (int j = 0; j < MLP_NP / 16; ++j) b3[j] = ld_kmajor(P + p.o_w3, MLP_HID, j, q, n);
#pragma unroll
    for (int j = 0; j < MLP_HID / 16; ++j) b2[j] = ld_kmajor(P + p.o_w2, MLP_HID, j, q, n);
    // (the W1 fragments of this wave's dh tiles too: their round trip runs under the dA2 / dA1 phases)
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int t = min(g + p.NG * (w + 8 * m), NT - 1);
#pragma unroll
      for (int j = 0; j < MLP_HID / 16; ++j) b1[m][j] = ld_kmajor(P + p.o_w1, MLP_K1, j, q, 16 * t + c);
    }
    __syncthreads();
    // dA2 = (dO3 W3) masked by A2 > 0: tile w
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < MLP_NP / 16; ++j) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(d3s + c * OP + 16 * j + 4 * q);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b3[j][0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b3[j][1], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b3[j][2], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b3[j][3], acc1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * q + r;
      const float v = m2[r] > 0.f ? acc0[r] + acc1[r] : 0.f;
      d2s[row * HP + n] = v;
      if (g == 0 && row0 + row < r_hi) p.dA2[(ob + row0 + row) * MLP_HID + n] = v;
    }
    __syncthreads();
    // dA1 = (dA2 W2) masked by A1 > 0
    acc0 = f32x4{0.f, 0.f, 0.f, 0.f}; acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < MLP_HID / 16; ++j) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(d2s + c * HP + 16 * j + 4 * q);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b2[j][0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b2[j][1], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b2[j][2], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b2[j][3], acc1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * q + r;
      const float v = m1[r] > 0.f ? acc0[r] + acc1[r] : 0.f;
      d1s[row * HP + n] = v;
      if (g == 0 && row0 + row < r_hi) p.dA1[(ob + row0 + row) * MLP_HID + n] = v;
    }
    __syncthreads();
    // dh += dA1 W1: this wave's column tiles
    f32x4 a1f[MLP_HID / 16];
#pragma unroll
    for (int j = 0; j < MLP_HID / 16; ++j) a1f[j] = *reinterpret_cast<const f32x4*>(d1s + c * HP + 16 * j + 4 * q);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int t = g + p.NG * (w + 8 * m);
      if (t >= NT) continue;                               // wave-uniform
#pragma unroll
      for (int j = 0; j < MLP_HID / 16; ++j) {
        acc_h[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1f[j][0], b1[m][j][0], acc_h[m][0], 0, 0, 0);
        acc_h[m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1f[j][1], b1[m][j][1], acc_h[m][1], 0, 0, 0);
        acc_h[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1f[j][2], b1[m][j][2], acc_h[m][0], 0, 0, 0);
        acc_h[m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1f[j][3], b1[m][j][3], acc_h[m][1], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int t = g + p.NG * (w + 8 * m);
    if (t >= NT) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = row0 + 4 * q + r;
      if (row < r_hi) p.dH[(int64_t)z * p.d_str + (int64_t)row * p.ldh + 16 * t + c] = acc_h[m][0][r] + acc_h[m][1][r];
    }
  }
}

struct mlp_dw_args {
  const float* dO3;      // [Z2][B][64]
  const float* dA2;      // [Z2][B][128]
  const float* dA1;
  const float* A2;
  const float* A1;
  const float* Hin;      // [B][ldh] of net z2 / 2
  float* G;              // gradient arena of the towers: tower z2 at G + z2 * t_str, same offsets as the parameters
  const int32_t* row_seg;
  int64_t t_str, h_str;
  int o_w1, o_b1, o_w2, o_b2, o_w3, o_b3;
  int ldh, B, Z2;
};

// dW_l = dY_l^T X_l and db_l = colsum dY_l of the three layers of a tower over its run of rows: both operands k-major
// (the reduction index is the row), fragments are coalesced 16-byte loads through buffer descriptors, a wave owns a
// 64 x 64 output tile (lstm_dw_kernel's scheme): 18 + 4 + 2 wave tiles per tower, 6 workgroups.
__global__ __launch_bounds__(256, 2) void mlp_dw_kernel(mlp_dw_args p) {
  constexpr int PD = 4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;
  const int z2 = blockIdx.x % p.Z2;
  const int tile = (blockIdx.x / p.Z2) * 4 + wave;        // 0 .. 23
  if (tile >= 24) return;
  int lo = 0, hi = p.B;
  if (p.row_seg) {
    const int beg = p.row_seg[2 * (z2 >> 1)], cnt = p.row_seg[2 * (z2 >> 1) + 1];
    lo = beg;
    hi = cnt > 0 ? min(p.B, beg + cnt) : lo;
  }
  // layer of the tile: 1 (tiles 0..17: 2 m-groups x 9 n-groups), 2 (18..21: 2 x 2), 3 (22..23: 1 x 2)
  const float *A, *Y;
  int lda, ldy, M, N, mg, ng, o_w, o_b;
  const int64_t ob = (int64_t)z2 * p.B;
  if (tile < 18) {
    A = p.dA1 + ob * MLP_HID; lda = MLP_HID; M = MLP_HID; Y = p.Hin + (int64_t)(z2 >> 1) * p.h_str; ldy = p.ldh; N = MLP_K1;
    mg = tile / 9; ng = tile - 9 * mg; o_w = p.o_w1; o_b = p.o_b1;
  } else if (tile < 22) {
    A = p.dA2 + ob * MLP_HID; lda = MLP_HID; M = MLP_HID; Y = p.A1 + ob * MLP_HID; ldy = MLP_HID; N = MLP_HID;
    mg = (tile - 18) >> 1; ng = (tile - 18) & 1; o_w = p.o_w2; o_b = p.o_b2;
  } else {
    A = p.dO3 + ob * MLP_NP; lda = MLP_NP; M = MLP_NP; Y = p.A2 + ob * MLP_HID; ldy = MLP_HID; N = MLP_HID;
    mg = 0; ng = tile - 22; o_w = p.o_w3; o_b = p.o_b3;
  }
  const int m0 = 64 * mg, n0 = 64 * ng;
  const int nc = (n0 + 4 * c < N) ? c : 0;                 // column chunk past the row: a valid one, discarded
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(A + m0), 0, (int)(((int64_t)p.B * lda - m0) * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc((void*)(Y + n0), 0, (int)(((int64_t)p.B * ldy - n0) * 4), 0x00020000);
  const int voff_a = (q * lda + 4 * c) * 4, voff_y = (q * ldy + 4 * nc) * 4;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
  const int KT = (hi - lo + 3) >> 2;                       // 4-row k-steps from the run's first row; the tail is masked
  f32x4 aq[PD], yq[PD];
  auto request = [&](int slot, int ks) {
    const int so_a = ks < KT ? (lo + 4 * ks) * lda * 4 : 0x7fffffff;
    const int so_y = ks < KT ? (lo + 4 * ks) * ldy * 4 : 0x7fffffff;
    aq[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, voff_a, so_a, 0));
    yq[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsY, voff_y, so_y, 0));
  };
#pragma unroll
  for (int s = 0; s < PD; ++s) request(s, s);
  __builtin_amdgcn_sched_barrier(0);
  for (int ks = 0; ks < KT; ks += PD) {
#pragma unroll
    for (int s = 0; s < PD; ++s) {
      f32x4 av = aq[s];
      f32x4 yv = yq[s];
      request(s, ks + s + PD);
      __builtin_amdgcn_sched_barrier(0);
      const bool dead = lo + 4 * (ks + s) + q >= hi;       // (k-steps past KT: zeros from the bounds check; the tail row by row,
#pragma unroll                                             //  both operands: a row past the run may hold anything)
      for (int i = 0; i < 4; ++i) { av[i] = dead ? 0.f : av[i]; yv[i] = dead ? 0.f : yv[i]; }
      bsum += av;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], yv[j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float* G = p.G + (int64_t)z2 * p.t_str;
  if (n0 + 4 * c < N) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + 16 * q + 4 * r + i;
        if (m < M) *reinterpret_cast<f32x4*>(G + o_w + (int64_t)m * N + n0 + 4 * c) = f32x4{acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
      }
  }
  if (ng == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v = bsum[i];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      bsum[i] = v;
    }
    if (q == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + 4 * c + i;
        if (m < M) G[o_b + m] = bsum[i];
      }
    }
  }
}

int rt_for(int B) {
  static const int forced = [] { const char* e = getenv("CADRE_LSTM_RT"); return e ? atoi(e) : 0; }();
  if (forced == 1 || forced == 2) return forced;
  // minibatch 64 (a net owns ~16 rows): one 16-row tile per workgroup — ~1.5 tiles per (net, slice), at most two
  // workgroups on a CU; bigger minibatches: two tiles per workgroup (each weight KiB feeds both)
  return B <= 64 ? 1 : 2;
}

}  // namespace

#ifdef LSTM_TRACE
extern "C" int cadre_lstm_set_trace(void* ptr) {
  long long* v = (long long*)ptr;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_lstm_trace_dev), &v, sizeof(v));
}
#endif

extern "C" int cadre_lstm_step_fwd(const float* Wp, int64_t wp_str, const float* bias, int64_t b_str, float* G, int32_t ldg,
                                   int64_t g_str, const float* Hprev, const float* Cprev, float* Hout, float* Cout,
                                   float* TCout, int32_t ldh, int64_t h_str, int32_t B, int32_t D, int32_t Z,
                                   const int32_t* row_seg, int32_t rev, void* stream) {
  FAIL_IF(!Wp || !G || !Hprev || !Cprev || !Hout || !Cout || !TCout || B < 1 || D < 1 || Z < 1, "cadre_lstm_step_fwd: bad argument");
  FAIL_IF(ldh != 544 || D > ldh || ldg < 4 * D, "cadre_lstm_step_fwd: built for K = ldh = 544 (hidden 530 zero padded), ldg >= 4*D");
  FAIL_IF((((uintptr_t)Wp | (uintptr_t)Hprev) & 15) || (wp_str & 3) || (h_str & 3), "cadre_lstm_step_fwd: operands must be 16-byte aligned");
  const int rt = rt_for(B), rows = 16 * rt, NS = (D + 15) / 16;
  fwd_args a{Wp, bias, G, Hprev, Cprev, Hout, Cout, TCout, row_seg, wp_str, b_str, g_str, h_str, ldg, ldh, B, D, Z, NS, rev & 1};
  dim3 grid(Z * NS * ((B + rows - 1) / rows));          // row chunks: workgroups past a net's cover of rows return at once
  if (rt == 1) hipLaunchKernelGGL((lstm_step_fwd_kernel<1, 34>), grid, dim3(256), 0, ST(stream), a);
  else hipLaunchKernelGGL((lstm_step_fwd_kernel<2, 34>), grid, dim3(256), 0, ST(stream), a);
  return (int)hipGetLastError();
}

extern "C" int cadre_lstm_step_bwd(const float* Wp, int64_t wp_str, const float* dGp_in, float* dGp_out, int64_t gp_str,
                                   float* dG_out, const float* G_act, int32_t ldg, int64_t g_str, const float* dh_in,
                                   float* dC, int64_t d_str, const float* TC, const float* Cprev, int32_t ldh, int64_t h_str,
                                   int32_t B, int32_t D, int32_t Z, const int32_t* commands, int32_t C,
                                   const int32_t* row_seg, int32_t rev, void* stream) {
  FAIL_IF(!dG_out || !dGp_out || !G_act || !dC || !TC || !Cprev || B < 1 || D < 1 || Z < 1 || (!dGp_in && !dh_in) || (commands && C < 1),
          "cadre_lstm_step_bwd: bad argument");
  FAIL_IF(ldh != 544 || D > ldh || ldg < 4 * D || 4 * D > 2176, "cadre_lstm_step_bwd: built for ldh = 544 (hidden 530 zero padded), ldg >= 4*D");
  FAIL_IF(gp_str < (int64_t)((B + 15) / 16) * 2176 * 16 || (gp_str & 3) || ((uintptr_t)dGp_out & 15),
          "cadre_lstm_step_bwd: the fragment-order copies hold ceil(B / 16) tiles of 16 x 2176 floats per net");
  if (dGp_in) FAIL_IF(!Wp || (((uintptr_t)Wp | (uintptr_t)dGp_in) & 15) || (wp_str & 3), "cadre_lstm_step_bwd: operands must be 16-byte aligned");
  const int rt = rt_for(B), rows = 16 * rt, NS = (D + 15) / 16;
  bwd_args a{Wp, dGp_in, dGp_out, dG_out, G_act, dh_in, dC, TC, Cprev, commands, row_seg, wp_str, gp_str, g_str, h_str, d_str,
             ldg, ldh, B, D, C < 1 ? 1 : C, Z, NS, rev & 1};
  dim3 grid(Z * NS * ((B + rows - 1) / rows));
#define LB(RT_)                                                                                                  \
  do {                                                                                                           \
    if (dGp_in) hipLaunchKernelGGL((lstm_step_bwd_kernel<RT_, 34, true>), grid, dim3(256), 0, ST(stream), a);   \
    else hipLaunchKernelGGL((lstm_step_bwd_kernel<RT_, 34, false>), grid, dim3(256), 0, ST(stream), a);         \
  } while (0)
  if (rt == 1) LB(1); else LB(2);
#undef LB
  return (int)hipGetLastError();
}

extern "C" int cadre_pack_lstm_weights(const float* W, int64_t w_str, int32_t ldw, int32_t D, int32_t Z, float* fwd, float* bwd,
                                       int64_t p_str, void* stream) {
  FAIL_IF(!W || !fwd || !bwd || D < 1 || Z < 1 || ldw != 544 || D > ldw || (w_str & 3) || (p_str & 3) ||
              (((uintptr_t)W | (uintptr_t)fwd | (uintptr_t)bwd) & 15) || p_str < (int64_t)((D + 15) / 16) * 4 * 34 * 256,
          "cadre_pack_lstm_weights: bad argument (ldw = 544, 16-byte aligned, p_str >= ceil(D/16) * 4 * 34 * 256 floats)");
  hipLaunchKernelGGL(pack_lstm_weights_kernel, dim3((D + 15) / 16, 8, Z), dim3(256), 0, ST(stream), W, w_str, ldw, D, ldw / 16,
                     fwd, bwd, p_str);
  return (int)hipGetLastError();
}

extern "C" int cadre_lstm_dw(const float* dG, int32_t ldg, int64_t g_str, const float* Hs, const float* X, int32_t ldh,
                             int64_t h_str, int64_t x_str, int32_t x_div, float* dWhh, float* dWih, float* dbih,
                             float* dbhh, int32_t ldw, int64_t w_str, int32_t B, int32_t S, int32_t H4, int32_t N,
                             int32_t Z, const int32_t* row_seg, void* stream) {
  FAIL_IF(!dG || !Hs || !X || !dWhh || !dWih || !dbih || !dbhh || B < 1 || S < 1 || H4 < 1 || N < 1 || Z < 1 || x_div < 1,
          "cadre_lstm_dw: bad argument");
  FAIL_IF((ldg & 3) || (ldh & 3) || (ldw & 3) || (N & 3) || N > ldh || N > ldw || ldg < ((H4 + 63) & ~63) ||
              (((uintptr_t)dG | (uintptr_t)Hs | (uintptr_t)X | (uintptr_t)dWhh | (uintptr_t)dWih) & 15) ||
              ((g_str | h_str | x_str | w_str) & 3),
          "cadre_lstm_dw: 16-byte aligned operands, N % 4 == 0, N <= ldh / ldw, ldg >= H4 rounded up to 64 (zero padded)");
  FAIL_IF((int64_t)S * B * ldg * 4 >= (1ll << 31) || (int64_t)(S + 1) * B * ldh * 4 >= (1ll << 31),
          "cadre_lstm_dw: a net's dG / h rows must stay below the 2 GiB buffer window");
  // wave tile 64 x 64; 128 x 64 (CADRE_DW_TM=2: a third less operand traffic per MFMA, one wave per SIMD) measured slower:
  // 277 vs 218 us at minibatch 256 — operand traffic is not what bounds the launch
  static const int tm_env = [] { const char* e = getenv("CADRE_DW_TM"); return e ? atoi(e) : 1; }();
  const int TM = (tm_env == 2 && ldg >= ((H4 + 127) & ~127)) ? 2 : 1;
  const int MG = (H4 + 64 * TM - 1) / (64 * TM), NG = (N + 63) / 64;
  dw_args a{dG, Hs, X, dWhh, dWih, dbih, dbhh, row_seg, g_str, h_str, x_str, w_str, ldg, ldh, ldw, B, S, H4, N, Z, x_div, MG, NG};
#ifdef CADRE_AB_KERNELS
  static const int lds_env = [] { const char* e = getenv("CADRE_DW_LDS"); return e ? atoi(e) : 0; }();     // operands shared through LDS (A/B build)
  if (lds_env && TM == 1) {
    const int MG2 = (MG + 1) / 2, NG2 = (NG + 1) / 2;
    if (lds_env == 4) hipLaunchKernelGGL(lstm_dw_lds_kernel<4>, dim3(Z * 2 * MG2 * NG2), dim3(256), 0, ST(stream), a, MG2, NG2);
    else hipLaunchKernelGGL(lstm_dw_lds_kernel<8>, dim3(Z * 2 * MG2 * NG2), dim3(256), 0, ST(stream), a, MG2, NG2);
    return (int)hipGetLastError();
  }
#endif
  const int wgs = (2 * MG * NG + 3) / 4;
  if (TM == 1) hipLaunchKernelGGL(lstm_dw_kernel<1>, dim3(Z * wgs), dim3(256), 0, ST(stream), a);
  else hipLaunchKernelGGL(lstm_dw_kernel<2>, dim3(Z * wgs), dim3(256), 0, ST(stream), a);
  return (int)hipGetLastError();
}

// The MLP towers of the update in three launches (mlp_fwd_kernel / mlp_bwd_kernel / mlp_dw_kernel above): hidden width
// 128, output rows padded to 64, input width ldh = 544.  Towers z2 = 2*net + {0, 1} at P + z2 * t_str with the weight /
// bias offsets o_* (floats) inside a tower; row_seg [Z2/2][2] (may be null): only each net's run of rows.
static int mlp_check(const char* who, int64_t t_str, const int32_t* o, int32_t ldh, int32_t B, int32_t Z2) {
  static thread_local char msg[160];
  if (ldh != MLP_K1 || B < 1 || Z2 < 2 || (Z2 & 1) || (t_str & 3) || ((o[0] | o[2] | o[4]) & 3) || o[0] < 0 || o[1] < 0 || o[2] < 0 || o[3] < 0 ||
      o[4] < 0 || o[5] < 0) {
    snprintf(msg, sizeof msg, "%s: built for towers 544 -> 128 -> 128 -> 64 (ldh = 544), 16-byte aligned weight offsets, an even tower count", who);
    return cadre_fail(msg);
  }
  return 0;
}

extern "C" int cadre_mlp_fwd(const float* P, int64_t t_str, const int32_t* offs, const float* Hin, int32_t ldh, int64_t h_str,
                             float* A1, float* A2, float* O3, int32_t B, int32_t Z2, const int32_t* row_seg, void* stream) {
  FAIL_IF(!P || !offs || !Hin || !A1 || !A2 || !O3, "cadre_mlp_fwd: null operand");
  if (int rc = mlp_check("cadre_mlp_fwd", t_str, offs, ldh, B, Z2)) return rc;
  FAIL_IF((((uintptr_t)P | (uintptr_t)Hin) & 15) || (h_str & 3), "cadre_mlp_fwd: operands must be 16-byte aligned");
  mlp_fwd_args a{P, Hin, A1, A2, O3, row_seg, t_str, h_str, offs[0], offs[1], offs[2], offs[3], offs[4], offs[5], ldh, B, Z2};
  hipLaunchKernelGGL(mlp_fwd_kernel, dim3(Z2 * ((B + 15) / 16)), dim3(512), 0, ST(stream), a);
  return (int)hipGetLastError();
}

extern "C" int cadre_mlp_bwd(const float* P, int64_t t_str, const int32_t* offs, const float* dO3, const float* A1, const float* A2,
                             float* dA1, float* dA2, float* dH, int32_t ldh, int64_t d_str, int32_t B, int32_t Z2,
                             const int32_t* row_seg, void* stream) {
  FAIL_IF(!P || !offs || !dO3 || !A1 || !A2 || !dA1 || !dA2 || !dH, "cadre_mlp_bwd: null operand");
  if (int rc = mlp_check("cadre_mlp_bwd", t_str, offs, ldh, B, Z2)) return rc;
  FAIL_IF(((uintptr_t)dO3 & 15), "cadre_mlp_bwd: dO3 must be 16-byte aligned");
  constexpr int NG = 4;                                   // column groups of dh: 8 waves x 2 tiles x 4 groups >= 34 tiles
  mlp_bwd_args a{P, dO3, A1, A2, dA1, dA2, dH, row_seg, t_str, d_str, offs[0], offs[2], offs[4], ldh, B, Z2 / 2, NG};
  hipLaunchKernelGGL(mlp_bwd_kernel, dim3((Z2 / 2) * NG * ((B + 15) / 16)), dim3(512), 0, ST(stream), a);
  return (int)hipGetLastError();
}

extern "C" int cadre_mlp_dw(const float* dO3, const float* dA2, const float* dA1, const float* A2, const float* A1, const float* Hin,
                            int32_t ldh, int64_t h_str, float* G, int64_t t_str, const int32_t* offs, int32_t B, int32_t Z2,
                            const int32_t* row_seg, void* stream) {
  FAIL_IF(!dO3 || !dA2 || !dA1 || !A2 || !A1 || !Hin || !G || !offs, "cadre_mlp_dw: null operand");
  if (int rc = mlp_check("cadre_mlp_dw", t_str, offs, ldh, B, Z2)) return rc;
  FAIL_IF((((uintptr_t)dO3 | (uintptr_t)dA2 | (uintptr_t)dA1 | (uintptr_t)A2 | (uintptr_t)A1 | (uintptr_t)Hin | (uintptr_t)G) & 15) || (h_str & 3),
          "cadre_mlp_dw: operands must be 16-byte aligned");
  FAIL_IF((int64_t)B * ldh * 4 >= (1ll << 31), "cadre_mlp_dw: a net's rows must stay below the 2 GiB buffer window");
  mlp_dw_args a{dO3, dA2, dA1, A2, A1, Hin, G, row_seg, t_str, h_str, offs[0], offs[1], offs[2], offs[3], offs[4], offs[5], ldh, B, Z2};
  hipLaunchKernelGGL(mlp_dw_kernel, dim3(Z2 * 6), dim3(256), 0, ST(stream), a);
  return (int)hipGetLastError();
}

#ifdef CADRE_AB_KERNELS
// All S forward steps of `Z` nets in one persistent launch (see lstm_seq_fwd_kernel): G [S][B][ldg], Hs / Cs / TC
// [S+1][B][ldh] per net with slot 0 = the initial state.  `sync_ws`: Z * S + 1 int32 of device memory (arrival counters,
// zeroed here, + a status word that a timed-out wait sets to 1 — check it with the results; it is never cleared here).
// Needs all Z * ceil(D/16) workgroups resident (<= 512): launched on an otherwise idle stream order, as the update does.
extern "C" int cadre_lstm_seq_fwd(const float* Wp, int64_t wp_str, const float* bias, int64_t b_str, float* G, int32_t ldg,
                                  int64_t g_str, float* Hs, float* Cs, float* TC, int32_t ldh, int64_t h_str, int32_t B,
                                  int32_t D, int32_t S, int32_t Z, const int32_t* row_seg, int32_t* sync_ws, void* stream) {
  FAIL_IF(!Wp || !G || !Hs || !Cs || !TC || !sync_ws || B < 1 || D < 1 || S < 1 || Z < 1, "cadre_lstm_seq_fwd: bad argument");
  FAIL_IF(ldh != 544 || D > ldh || ldg < 4 * D, "cadre_lstm_seq_fwd: built for K = ldh = 544 (hidden 530 zero padded), ldg >= 4*D");
  FAIL_IF((((uintptr_t)Wp | (uintptr_t)Hs) & 15) || (wp_str & 3) || (h_str & 3), "cadre_lstm_seq_fwd: operands must be 16-byte aligned");
  const int NS = (D + 15) / 16;
  FAIL_IF(Z * NS > 512, "cadre_lstm_seq_fwd: more workgroups than can be resident together (2 per CU)");
  FAIL_IF((int64_t)(S + 1) * B * ldh * 4 >= (1ll << 31), "cadre_lstm_seq_fwd: a net's h rows must stay below the 2 GiB buffer window");
  hipLaunchKernelGGL(zero_i32_kernel, dim3(1), dim3(256), 0, ST(stream), sync_ws, Z * S);
  seq_fwd_args a{Wp, bias, G, Hs, Cs, TC, row_seg, sync_ws, sync_ws + Z * S, wp_str, b_str, g_str, h_str, ldg, ldh, B, D, S, Z, NS};
  hipLaunchKernelGGL((lstm_seq_fwd_kernel<34>), dim3(Z * NS), dim3(256), 0, ST(stream), a);
  return (int)hipGetLastError();
}
#endif

