// winograd_mats.h — Cook-Toom transform matrices of Winograd F(m x m, 3x3), m = 2, 3, 4, shared by winograd.hip (stand-alone
// transforms) and winograd_fused.hip (batched plane GEMM with the inverse transform in its epilogue).  The host's G (cadre_amd/
// encoder.py _WINO_G) matches these B^T / A^T.  DESIGN.md 3.7 has the point search and the measured rounding errors.
#pragma once

// F(2x2): points 0, 1, -1, infinity; F(3x3): 0, 3/4, -3/4, 2, infinity; F(4x4): 0, +-3/4, +-3/2, infinity; F(6x6): 0, +-1/2, +-1, +-2, infinity.
// F(2x2, 3x3): 4x4 input tiles, 16 planes, 2.25x fewer multiplies than direct; F(3x3, 3x3): 5x5 tiles, 25 planes, 3.24x fewer and
// only 2.78x (not 4x) the input in transform-domain traffic; F(4x4, 3x3): 6x6 tiles, 36 planes, 4x fewer, 2.25x the input.
template <int M> struct wino_mat;
template <> struct wino_mat<2> {
  static constexpr int N = 4;
  static constexpr float BT[4][4] = {{1, 0, -1, 0}, {0, 1, 1, 0}, {0, -1, 1, 0}, {0, 1, 0, -1}};
  static constexpr float AT[2][4] = {{1, 1, 1, 0}, {0, 1, -1, -1}};
};
template <> struct wino_mat<3> {
  static constexpr int N = 5;
  // points 0, 3/4, -3/4, 2, infinity: 30 % less rounding error than 0, 1, -1, 2 (rms 2.2e-7 against 3.2e-7 of the tensor's max
  // at K = 512; the direct conv: 1.6e-7) — every coefficient is a dyadic rational, exact in fp32
  static constexpr float BT[5][5] = {{1.125f, -0.5625f, -2.f, 1.f, 0.f}, {0.f, -1.5f, -1.25f, 1.f, 0.f}, {0.f, 1.5f, -2.75f, 1.f, 0.f},
                                     {0.f, -0.5625f, 0.f, 1.f, 0.f}, {0.f, 1.125f, -0.5625f, -2.f, 1.f}};
  static constexpr float AT[3][5] = {{1.f, 1.f, 1.f, 1.f, 0.f}, {0.f, 0.75f, -0.75f, 2.f, 0.f}, {0.f, 0.5625f, 0.5625f, 4.f, 1.f}};
};
template <> struct wino_mat<4> {
  static constexpr int N = 6;
  // F(4x4, 3x3): 6x6 tiles, 36 planes.  Points 0, 3/4, -3/4, 3/2, -3/2, infinity — searched like the F(3x3) set (float32
  // emulation: rms error 1.6x the F(3x3) set's, against 4.2x for the textbook 0, +-1, +-2); rows of B^T scaled by powers of two,
  // every coefficient dyadic
  static constexpr float BT[6][6] = {{1.265625f, 0.f, -2.8125f, 0.f, 1.f, 0.f},      {0.f, 1.6875f, 2.25f, -0.75f, -1.f, 0.f},
                                     {0.f, -1.6875f, 2.25f, 0.75f, -1.f, 0.f},       {0.f, -0.84375f, -0.5625f, 1.5f, 1.f, 0.f},
                                     {0.f, 0.84375f, -0.5625f, -1.5f, 1.f, 0.f},     {0.f, 1.265625f, 0.f, -2.8125f, 0.f, 1.f}};
  static constexpr float AT[4][6] = {{1.f, 1.f, 1.f, 1.f, 1.f, 0.f},                 {0.f, 0.75f, -0.75f, 1.5f, -1.5f, 0.f},
                                     {0.f, 0.5625f, 0.5625f, 2.25f, 2.25f, 0.f},     {0.f, 0.421875f, -0.421875f, 3.375f, -3.375f, 1.f}};
};
template <> struct wino_mat<6> {
  static constexpr int N = 8;
  // F(6x6, 3x3) (round 6): 8x8 tiles, 64 planes, 5.06x fewer multiplies than direct (F(3x3): 3.24x) and 1.78x the input in
  // transform-domain traffic (F(3x3): 2.78x).  Points 0, +-1/2, +-1, +-2, infinity — the best of the 56 sets of three magnitudes out of
  // {1/2, 3/4, 1, 5/4, 3/2, 2, 5/2, 3} in a float32 emulation of one 256-channel conv (tools/dbg/wino_points.py: rms 1.25e-6 / max
  // 1.5e-5 of the tensor's max against 3.0e-7 / 2.5e-6 for the F(4x4) set in the same emulation); every coefficient dyadic.
  // Taken only where 6 x 6 tiles cover the map exactly and the three-launch form runs (the 18 x 18 maps of layer3).
  static constexpr float BT[8][8] = {{-0.5f, 0.f, 2.625f, 0.f, -2.625f, 0.f, 0.5f, 0.f},   {0.f, 1.f, 2.f, -1.25f, -2.5f, 0.25f, 0.5f, 0.f},
                                     {0.f, -1.f, 2.f, 1.25f, -2.5f, -0.25f, 0.5f, 0.f},    {0.f, 0.5f, 0.5f, -2.125f, -2.125f, 0.5f, 0.5f, 0.f},
                                     {0.f, -0.5f, 0.5f, 2.125f, -2.125f, -0.5f, 0.5f, 0.f}, {0.f, 0.5f, 0.25f, -2.5f, -1.25f, 2.f, 1.f, 0.f},
                                     {0.f, -0.5f, 0.25f, 2.5f, -1.25f, -2.f, 1.f, 0.f},    {0.f, -0.5f, 0.f, 2.625f, 0.f, -2.625f, 0.f, 0.5f}};
  static constexpr float AT[6][8] = {{1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 0.f},             {0.f, 0.5f, -0.5f, 1.f, -1.f, 2.f, -2.f, 0.f},
                                     {0.f, 0.25f, 0.25f, 1.f, 1.f, 4.f, 4.f, 0.f},         {0.f, 0.125f, -0.125f, 1.f, -1.f, 8.f, -8.f, 0.f},
                                     {0.f, 0.0625f, 0.0625f, 1.f, 1.f, 16.f, 16.f, 0.f},   {0.f, 0.03125f, -0.03125f, 1.f, -1.f, 32.f, -32.f, 1.f}};
};
constexpr float wino_mat<2>::BT[4][4];
constexpr float wino_mat<2>::AT[2][4];
constexpr float wino_mat<3>::BT[5][5];
constexpr float wino_mat<3>::AT[3][5];
constexpr float wino_mat<4>::BT[6][6];
constexpr float wino_mat<4>::AT[4][6];
constexpr float wino_mat<6>::BT[8][8];
constexpr float wino_mat<6>::AT[6][8];

// sum_k c[k] * v[k * stride] over the non-zero constants (unrolled at compile time; +-1 become adds).  T: float or a float vector
template <int N, typename T>
__device__ __forceinline__ T wino_dot(const float (&c)[N], const T* v, int stride) {
  T acc = T{};
  bool first = true;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    if (c[k] == 0.f) continue;
    const T x = v[k * stride];
    if (first) { acc = c[k] == 1.f ? x : (c[k] == -1.f ? -x : x * c[k]); first = false; }
    else if (c[k] == 1.f) acc += x;
    else if (c[k] == -1.f) acc -= x;
    else acc += x * c[k];
  }
  return acc;
}
