// conv3x3_c64_bf16.hip — the stage-1 convolutions of the bf16 encoder (BASELINE config C3): 3x3, stride 1,
// pad 1, 64 -> 64 channels on NHWC bf16 (resnet.py:26-55 BasicBlock, layer1), folded eval BatchNorm,
// optional residual, ReLU.  At N = 64 / K = 576 these layers sit below the bf16 ridge (288 FLOP per HBM
// byte): the bound is HBM, and what a tile-per-workgroup GEMM loses on them is not arithmetic but turnover —
// nine k-tiles per tile, then a prologue and an epilogue nothing overlaps.  Structure here:
//   * the whole 64 x 576 weight matrix stays RESIDENT in LDS (73 KB) for the life of the workgroup;
//   * every WAVE is autonomous: it walks its own contiguous run of 32-position M-tiles and feeds itself
//     through a private ring of LDS stages filled by LDS-DMA (buffer_load ... lds, no VGPR round trip,
//     hardware zero fill for the halo taps), three stages ahead, ordered by its own counted vmcnt —
//     no workgroup barrier and no cross-wave hand-off anywhere in the main loop, so the four waves of a CU
//     drift apart and one wave's epilogue runs under the others' loads and MFMAs;
//   * the (tile, tap) sequence is one flat stream: the prefetch runs straight across tile boundaries;
//   * 128-byte pixel rows are XOR-swizzled on the SOURCE address (chunk ^ ((row>>1)&7)), the LDS image
//     stays lane-linear as LDS-DMA requires, ds_read_b128 fragment reads are conflict free;
//   * the epilogue goes through a private 4 KB LDS slab per wave so that residual reads and output stores
//     are 16 bytes per lane.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../../include/cadre_hip_ab.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

int cadre_fail(const char* msg);

#define C64_WPITCH 1168            // weight row pitch in bytes (1152 + 16: (pitch/16) odd -> conflict-free rows)
#define C64_NST 4                  // ring stages per wave (3 in flight + the one being read)
#define C64_STAGE 4096             // 32 positions x 128 B
#define C64_SLAB 4608               // epilogue slab: 32 rows x 36 floats
#define C64_WAVE_LDS (C64_NST * C64_STAGE + C64_SLAB)

struct c64_args {
  const void* x;          // bf16 NHWC [F][H][W][64]
  const void* w;          // bf16 [64][576], k = (kh*3 + kw)*64 + ci
  const float* scale;     // [64]
  const float* shift;
  const void* resid;      // bf16 [F*H*W][64] or null (added before the ReLU)
  void* out;              // bf16 [F*H*W][64]
  int M, H, W;            // M = F*H*W
  int tiles, tpw;         // 32-row tiles, tiles per wave
  int relu;
};

#define C64_WAIT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

template <bool RESID>
__global__ __launch_bounds__(256, 1) void conv3x3_c64_bf16_kernel(c64_args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* wl = smem;                                         // [64][C64_WPITCH]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  {
    const char* src = reinterpret_cast<const char*>(a.w);
    for (int i = tid; i < 64 * 72; i += 256) {             // 72 x 16-B chunks per 1152-B row
      const int n = i / 72, c = i - n * 72;
      *reinterpret_cast<f32x4*>(wl + n * C64_WPITCH + c * 16) = *reinterpret_cast<const f32x4*>(src + (size_t)i * 16);
    }
  }
  __syncthreads();
  const int gw = blockIdx.x * 4 + wave;
  const int t_begin = gw * a.tpw, t_end = min(a.tiles, t_begin + a.tpw);
  if (t_begin >= t_end) return;                            // (no workgroup barrier below this line)
  char* ring = smem + 64 * C64_WPITCH + wave * C64_WAVE_LDS;
  float* cs = reinterpret_cast<float*>(ring + C64_NST * C64_STAGE);
  constexpr unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)OOB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, a.M * 128, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void*)(RESID ? a.resid : a.x), 0, a.M * 128, 0x00020000);
  // One LDS-DMA piece: 64 lanes x 16 B -> 1 KiB at a wave-uniform LDS address; offsets >= num_records return zeros
  // (halo taps, the M tail, and the dummy pieces that keep the vmcnt arithmetic uniform past the last tile).
  auto dma16 = [&](unsigned voff, char* dst) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (__attribute__((address_space(3))) void*)dst, 16, (int)voff, 0, 0, 0);
  };

  // This lane inside DMA piece j stages row r = 8*j + (lane >> 3); LDS chunk (lane & 7) of that row holds the
  // row's logical chunk (lane & 7) ^ ((r >> 1) & 7)  (swizzle on the SOURCE address, LDS image lane-linear).
  const int prow = lane >> 3, pch = lane & 7;
  const float inv_w = 1.0f / (float)a.W, inv_h = 1.0f / (float)a.H;
  int ph[4], pw[4];                                        // (h, w) of the staged rows of the tile being issued
  unsigned aoff[4], amask[4];
  auto set_tile = [&](int t) {                             // masks + offsets from (ph, pw); dense NHWC: pixel index == m
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = 8 * j + prow, m = t * 32 + r;
      aoff[j] = (unsigned)((m - a.W - 1) * 128 + ((pch ^ ((r >> 1) & 7)) << 4));      // tap (0,0); wraps when masked
      unsigned mask = 0;
      if (m < a.M) {
        unsigned colm = 0;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
          if ((unsigned)(pw[j] - 1 + kw) < (unsigned)a.W) colm |= 1u << kw;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
          if ((unsigned)(ph[j] - 1 + kh) < (unsigned)a.H) mask |= colm << (3 * kh);
      }
      amask[j] = mask;
    }
  };
  {
    const int HW = a.H * a.W;
#pragma unroll
    for (int j = 0; j < 4; ++j) {                          // one real division per wave; afterwards carries only
      const int m = t_begin * 32 + 8 * j + prow;
      const int rem = m % HW;
      ph[j] = rem / a.W;
      pw[j] = rem - ph[j] * a.W;
    }
  }
  auto advance_tile = [&]() {                              // + 32 positions; floor(x / d) exact as (int)((x + .5f) / d) for x < 2^16
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int x = pw[j] + 32;
      const int q1 = (int)(((float)x + 0.5f) * inv_w);
      pw[j] = x - q1 * a.W;
      const int y = ph[j] + q1;
      const int q2 = (int)(((float)y + 0.5f) * inv_h);
      ph[j] = y - q2 * a.H;
    }
  };
  auto issue = [&](int tap, int stage, bool live) {        // 4 pieces = one 32 x 128 B stage
    const unsigned delta = (unsigned)(((tap / 3) * a.W + (tap % 3)) * 128);
    const unsigned bit = 1u << tap;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      dma16((live && (amask[j] & bit)) ? aoff[j] + delta : OOB, ring + stage * C64_STAGE + j * 1024);
  };

  float sc[2], sh[2];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) { sc[cb] = a.scale[32 * cb + l31]; sh[cb] = a.shift[32 * cb + l31]; }
  const char* a_rd = ring + l31 * 128;                     // fragment row; chunk (2s+lh) ^ ((l31>>1)&7)
  const unsigned a_sw = (unsigned)((l31 >> 1) & 7);
  const char* b_rd = wl + l31 * C64_WPITCH + lh * 16;

  // ---- flat (tile, tap) stream: step s = 9*(t - t_begin) + tap lives in stage s & 3 and is issued 3 steps ahead
  set_tile(t_begin);
  issue(0, 0, true); issue(1, 1, true); issue(2, 2, true);
  for (int t = t_begin; t < t_end; ++t) {
    const int sbase = (9 * (t - t_begin)) & (C64_NST - 1);
    f32x16 acc[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
    u32x4 rv[2][2];
    if constexpr (RESID) {                                 // issued at the top: long landed when the epilogue needs them
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int pos = t * 32 + 16 * i + (lane >> 2);
          rv[cb][i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsR, pos * 128 + cb * 64 + (lane & 3) * 16, 0, 0));
        }
    }
    const bool first = t == t_begin;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      // issue step s+3 = (t, tap+3) or (t+1, tap-6)
      if (tap == 6) { advance_tile(); set_tile(t + 1); }
      issue((tap + 3) % 9, (sbase + tap + 3) & (C64_NST - 1), tap < 6 || t + 1 < t_end);
      // Wait for step s: in-order completion, so "all but the N youngest" with N = the 12 pieces of s+1..s+3
      // + what was issued behind the pieces of s by the epilogue of the previous tile (4 stores) and the top of this
      // one (4 residual loads) — only the first three taps of a tile see those.
      if (tap < 3) {
        if (first) { if constexpr (RESID) C64_WAIT(16); else C64_WAIT(12); }
        else { if constexpr (RESID) C64_WAIT(20); else C64_WAIT(16); }
      } else {
        C64_WAIT(12);
      }
      const char* st = a_rd + ((sbase + tap) & (C64_NST - 1)) * C64_STAGE;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bf16x8 af = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(st + (((2 * s + lh) ^ a_sw) << 4)));
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          const bf16x8 bf = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(b_rd + cb * 32 * C64_WPITCH + tap * 128 + s * 32));
          acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[cb], 0, 0, 0);
        }
      }
    }
    // ---- epilogue: per 32-channel half through the wave's slab (fp32, pitch 36 floats); 16 bytes per lane to HBM
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) cs[((r & 3) + 8 * (r >> 2) + 4 * lh) * 36 + l31] = acc[cb][r] * sc[cb] + sh[cb];
#pragma unroll
      for (int i = 0; i < 2; ++i) {                        // lane -> (position 16i + lane/4, channels 8*(lane&3) ..+7)
        const int row = 16 * i + (lane >> 2), c8 = (lane & 3) * 8;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(cs + row * 36 + c8);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(cs + row * 36 + c8 + 4);
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        if constexpr (RESID) {
          const bf16x8 rr = __builtin_bit_cast(bf16x8, rv[cb][i]);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += (float)rr[e];
        }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (__bf16)(a.relu ? fmaxf(v[e], 0.f) : v[e]);
        const int pos = t * 32 + row;                      // rows >= M fall outside the descriptor: dropped
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rsC, pos * 128 + cb * 64 + c8 * 2, 0, 0);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// v2: every input pixel is brought into LDS ONCE.  The workgroup (4 waves x 32 positions = one 128-position tile per
// step) streams the activation through a 512-slot ring over the flattened position axis (one slot = one pixel =
// 128 B): tap (kh, kw) of output position p is pixel p + (kh-1)*W + (kw-1), i.e. ring index
// (p - P0 + kh*W + kw) & 511 with the ring anchored at L0 = P0 - W - 1 — nine taps = nine address offsets into the
// same resident pixels, the taps that fall outside the frame (row / column halo) are zeroed on the fragment with the
// per-position 9-bit mask.  Per tile the workgroup issues ONE block of 128 new pixels (16 LDS-DMA pieces, 4 per wave)
// instead of nine 32-pixel stages per wave: 9x fewer bytes through the L2 -> LDS path, 9x fewer DMA instructions, and
// the block being loaded is not needed before the next tile: a whole tile of lead time.  One workgroup barrier per tile
// (publishes the block every wave waited for with its own vmcnt, and fences the overwrite of the oldest block).
#define C64_R 512                  // ring slots (pixels)
#define C64_V2_LDS (64 * C64_WPITCH + C64_R * 128 + 4 * C64_SLAB)

template <bool RESID>
__global__ __launch_bounds__(256, 1) void conv3x3_c64_bf16_v2_kernel(c64_args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* wl = smem;                                         // [64][C64_WPITCH]
  char* ring = smem + 64 * C64_WPITCH;                     // [512][128 B], chunk c of slot q at ((c ^ ((q>>1)&7)) << 4)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  float* cs = reinterpret_cast<float*>(ring + C64_R * 128 + wave * C64_SLAB);
  {
    const char* src = reinterpret_cast<const char*>(a.w);
    for (int i = tid; i < 64 * 72; i += 256) {
      const int n = i / 72, c = i - n * 72;
      *reinterpret_cast<f32x4*>(wl + n * C64_WPITCH + c * 16) = *reinterpret_cast<const f32x4*>(src + (size_t)i * 16);
    }
  }
  const int t_begin = blockIdx.x * a.tpw, t_end = min(a.tiles, t_begin + a.tpw);      // 128-position tiles of this workgroup
  const int nt = t_end - t_begin;                          // uniform over the workgroup (barriers below)
  const int P0 = t_begin * 128, L0 = P0 - a.W - 1;
  const int nh = (2 * a.W + 1) >> 7;                       // extra halo blocks a tile reaches into (0 or 1: W <= 127)
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.M * 128, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, a.M * 128, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void*)(RESID ? a.resid : a.x), 0, a.M * 128, 0x00020000);
  // block jb = pixels [L0 + 128 jb, +128): 16 pieces of 8 pixels, this wave issues pieces 4w .. 4w+3.  Pixels before
  // the tensor (negative -> huge unsigned offset) or past it fall outside the descriptor and arrive as zeros.
  auto issue_block = [&](int jb) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int q = 128 * jb + 8 * (4 * wave + k);         // ring-relative index of the piece's first pixel
      const int qi = q + (lane >> 3);
      const unsigned voff = (unsigned)((L0 + qi) * 128 + (((lane & 7) ^ ((qi >> 1) & 7)) << 4));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (__attribute__((address_space(3))) void*)(ring + (q & (C64_R - 1)) * 128), 16,
                                               (int)voff, 0, 0, 0);
    }
  };
  // (h, w) of this lane's output position inside the tile (fragment row l31 of wave `wave`)
  const float inv_w = 1.0f / (float)a.W, inv_h = 1.0f / (float)a.H;
  int ph, pw;
  {
    const int m = P0 + 32 * wave + l31, HW = a.H * a.W;
    const int rem = m % HW;
    ph = rem / a.W;
    pw = rem - ph * a.W;
  }
  float sc[2], sh[2];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) { sc[cb] = a.scale[32 * cb + l31]; sh[cb] = a.shift[32 * cb + l31]; }
  const char* b_rd = wl + l31 * C64_WPITCH + lh * 16;

  for (int jb = 0; jb <= 1 + nh; ++jb) issue_block(jb);    // what tile 0 needs
  for (int T = 0; T < nt; ++T) {
    const int t = t_begin + T;
    // own pieces of block T+1+nh have landed (behind them in the queue: only the 4 stores of the previous epilogue)
    if (T == 0) C64_WAIT(0); else C64_WAIT(4);
    // raw barrier (a __syncthreads() fence would drain vmcnt to 0, i.e. also wait for the previous tile's stores):
    // block T+1+nh complete and visible; every wave is done reading for tile T-1
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue_block(T + 2 + nh);                               // overwrites block T-2+nh (4 blocks in the ring): no reader left
    u32x4 rv[2][2];
    if constexpr (RESID) {
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int pos = t * 128 + 32 * wave + 16 * i + (lane >> 2);
          rv[cb][i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsR, pos * 128 + cb * 64 + (lane & 3) * 16, 0, 0));
        }
    }
    unsigned mask = 0;
    {
      const int m = t * 128 + 32 * wave + l31;
      if (m < a.M) {
        unsigned colm = 0;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
          if ((unsigned)(pw - 1 + kw) < (unsigned)a.W) colm |= 1u << kw;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
          if ((unsigned)(ph - 1 + kh) < (unsigned)a.H) mask |= colm << (3 * kh);
      }
    }
    f32x16 acc[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
    const int qb = 128 * T + 32 * wave + l31;              // ring-relative index of tap (0,0) of this lane's position
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int qi = (qb + (tap / 3) * a.W + (tap % 3)) & (C64_R - 1);
      const char* arow = ring + qi * 128;
      const unsigned sw = (unsigned)((qi >> 1) & 7);
      const bool on = (mask >> tap) & 1u;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        f32x4 av = *reinterpret_cast<const f32x4*>(arow + (((2 * s + lh) ^ sw) << 4));
        if (!on) av = f32x4{0.f, 0.f, 0.f, 0.f};             // halo tap of this position: zero padding
        const bf16x8 af = __builtin_bit_cast(bf16x8, av);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          const bf16x8 bf = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(b_rd + cb * 32 * C64_WPITCH + tap * 128 + s * 32));
          acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[cb], 0, 0, 0);
        }
      }
    }
    // ---- epilogue (as v1): per 32-channel half through the wave's slab, 16 bytes per lane to HBM
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) cs[((r & 3) + 8 * (r >> 2) + 4 * lh) * 36 + l31] = acc[cb][r] * sc[cb] + sh[cb];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = 16 * i + (lane >> 2), c8 = (lane & 3) * 8;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(cs + row * 36 + c8);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(cs + row * 36 + c8 + 4);
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        if constexpr (RESID) {
          const bf16x8 rr = __builtin_bit_cast(bf16x8, rv[cb][i]);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += (float)rr[e];
        }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (__bf16)(a.relu ? fmaxf(v[e], 0.f) : v[e]);
        const int pos = t * 128 + 32 * wave + row;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rsC, pos * 128 + cb * 64 + c8 * 2, 0, 0);
      }
    }
    {                                                      // next tile: + 128 positions (exact float-reciprocal floor, x < 2^16)
      const int x = pw + 128;
      const int q1 = (int)(((float)x + 0.5f) * inv_w);
      pw = x - q1 * a.W;
      const int y = ph + q1;
      const int q2 = (int)(((float)y + 0.5f) * inv_h);
      ph = y - q2 * a.H;
    }
  }
}

// 0 = auto (v2 where it applies), 1 = force v1 (per-wave im2col stages), for A/B runs: CADRE_C64_VARIANT
static int g_c64_variant = [] { const char* e = getenv("CADRE_C64_VARIANT"); return e ? atoi(e) : 0; }();

extern "C" int cadre_conv3x3_c64_bf16(const void* x, const void* w, const float* scale, const float* shift,
                                      const void* resid, void* out, int32_t F, int32_t H, int32_t W, int32_t relu,
                                      void* stream) {
  if (!x || !w || !scale || !shift || !out || F < 1 || H < 1 || W < 1) return cadre_fail("cadre_conv3x3_c64_bf16: bad argument");
  const long long M = (long long)F * H * W;
  if (M * 128 >= (1ll << 31)) return cadre_fail("cadre_conv3x3_c64_bf16: activation spans >= 2 GiB: chunk the batch");
  if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)out | (uintptr_t)resid) & 15) return cadre_fail("cadre_conv3x3_c64_bf16: operands must be 16-byte aligned");
  c64_args a;
  a.x = x; a.w = w; a.scale = scale; a.shift = shift; a.resid = resid; a.out = out;
  a.M = (int)M; a.H = H; a.W = W; a.relu = relu;
  hipStream_t st = (hipStream_t)stream;
  if (W <= 127 && W >= 2 && g_c64_variant != 1) {          // v2: shared position ring, each pixel loaded once
    a.tiles = (int)((M + 127) / 128);
    int wgs = 256;
    if (a.tiles < wgs * 2) wgs = (a.tiles + 1) / 2 > 0 ? (a.tiles + 1) / 2 : 1;
    a.tpw = (a.tiles + wgs - 1) / wgs;
    const dim3 grid((a.tiles + a.tpw - 1) / a.tpw), block(256);
    if (resid) {
      (void)hipFuncSetAttribute((const void*)conv3x3_c64_bf16_v2_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipLaunchKernelGGL(conv3x3_c64_bf16_v2_kernel<true>, grid, block, C64_V2_LDS, st, a);
    } else {
      (void)hipFuncSetAttribute((const void*)conv3x3_c64_bf16_v2_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipLaunchKernelGGL(conv3x3_c64_bf16_v2_kernel<false>, grid, block, C64_V2_LDS, st, a);
    }
    return (int)hipGetLastError();
  }
  a.tiles = (int)((M + 31) / 32);
  int waves = 1024;                                        // 256 CUs x 4 autonomous waves
  if (a.tiles < waves * 4) waves = (a.tiles + 3) / 4 > 0 ? (a.tiles + 3) / 4 : 1;
  a.tpw = (a.tiles + waves - 1) / waves;
  const int nwave = (a.tiles + a.tpw - 1) / a.tpw;
  const dim3 grid((nwave + 3) / 4), block(256);
  const size_t lds = 64 * C64_WPITCH + 4 * C64_WAVE_LDS;
  if (resid) {
    (void)hipFuncSetAttribute((const void*)conv3x3_c64_bf16_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(conv3x3_c64_bf16_kernel<true>, grid, block, lds, st, a);
  } else {
    (void)hipFuncSetAttribute((const void*)conv3x3_c64_bf16_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(conv3x3_c64_bf16_kernel<false>, grid, block, lds, st, a);
  }
  return (int)hipGetLastError();
}
