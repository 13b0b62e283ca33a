// Shared device helpers for the MI355X (gfx950) TensorNet energy+force kernels.
//
// Data layout conventions:
//   * per-atom Cartesian tensors: 9 irreducible components per channel, row-major [N][9][F], F fastest
//       c=0      I            X = I*1 + A + S                    (reference tensornet.py:18-39)
//       c=1..3   v0,v1,v2     A = skew(v): A01=-v2 A02=+v1 A12=-v0 (reference tensornet.py:84-103)
//       c=4..8   s0..s4       S = [[s0,s1,s2],[s1,s3,s4],[s2,s4,-(s0+s3)]]
//   * undirected pair list (i > j), one extra "self pair" (d = 0) at index P
//   * symmetric CSR adjacency sorted by neighbour index, self edge included
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define TN_WAVE 64

namespace tn {

struct M3 {  // 3x3 matrix in registers
  float a[3][3];
};

// Canonical pair geometry: delta = pos[hi] - pos[lo] (+ triclinic minimum image, z -> y -> x; reference neighbors_brute.py:112-135,
// models/utils.py:206-229), returns |delta|^2.  ONE definition for every kernel that decides "is this a pair" - the count and the
// fill pass of a row, the two rows of a pair, brute force and cell list - with contraction off and the fused multiply-adds
// spelled out: left to the compiler, a*b+c was contracted differently in the count and the fill instantiation, and a pair within one
// ulp of the cutoff (about one per 3e5 atoms of water) was counted by one pass and not the other - the row then held one entry
// too many and a real neighbour fell off its end (round 4: wrong forces on ~50 atoms of a 273 375-atom box).
__device__ __forceinline__ float pair_geometry(const float* __restrict__ pos, int hi, int lo, const float* __restrict__ box, float& dx,
                                               float& dy, float& dz) {
#pragma clang fp contract(off)
  dx = pos[hi * 3 + 0] - pos[lo * 3 + 0];
  dy = pos[hi * 3 + 1] - pos[lo * 3 + 1];
  dz = pos[hi * 3 + 2] - pos[lo * 3 + 2];
  if (box) {
    const float s3 = -roundf(dz / box[8]);
    dx = __builtin_fmaf(s3, box[6], dx);
    dy = __builtin_fmaf(s3, box[7], dy);
    dz = __builtin_fmaf(s3, box[8], dz);
    const float s2 = -roundf(dy / box[4]);
    dx = __builtin_fmaf(s2, box[3], dx);
    dy = __builtin_fmaf(s2, box[4], dy);
    const float s1 = -roundf(dx / box[0]);
    dx = __builtin_fmaf(s1, box[0], dx);
  }
  const float xx = dx * dx;
  return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, xx));
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float silu(float x) { return x * sigmoidf_(x); }
__device__ __forceinline__ float silu_grad(float x) {
  float s = sigmoidf_(x);
  return s * (1.0f + x * (1.0f - s));
}

// compose: 9 irreducible components -> full 3x3
__device__ __forceinline__ M3 compose(const float u[9]) {
  M3 m;
  m.a[0][0] = u[0] + u[4];
  m.a[0][1] = u[5] - u[3];
  m.a[0][2] = u[6] + u[2];
  m.a[1][0] = u[5] + u[3];
  m.a[1][1] = u[0] + u[7];
  m.a[1][2] = u[8] - u[1];
  m.a[2][0] = u[6] - u[2];
  m.a[2][1] = u[8] + u[1];
  m.a[2][2] = u[0] - u[4] - u[7];
  return m;
}

// dec: full 3x3 -> 9 irreducible components (I = tr/3, A = asym, S = sym - I*1)
__device__ __forceinline__ void decompose(const M3& m, float u[9]) {
  float I = (m.a[0][0] + m.a[1][1] + m.a[2][2]) * (1.0f / 3.0f);
  u[0] = I;
  u[1] = 0.5f * (m.a[2][1] - m.a[1][2]);
  u[2] = 0.5f * (m.a[0][2] - m.a[2][0]);
  u[3] = 0.5f * (m.a[1][0] - m.a[0][1]);
  u[4] = m.a[0][0] - I;
  u[5] = 0.5f * (m.a[0][1] + m.a[1][0]);
  u[6] = 0.5f * (m.a[0][2] + m.a[2][0]);
  u[7] = m.a[1][1] - I;
  u[8] = 0.5f * (m.a[1][2] + m.a[2][1]);
}

// adjoint of compose: gradient wrt the 9 stored components given the gradient wrt the 3x3
__device__ __forceinline__ void compose_T(const M3& g, float u[9]) {
  u[0] = g.a[0][0] + g.a[1][1] + g.a[2][2];
  u[1] = g.a[2][1] - g.a[1][2];
  u[2] = g.a[0][2] - g.a[2][0];
  u[3] = g.a[1][0] - g.a[0][1];
  u[4] = g.a[0][0] - g.a[2][2];
  u[5] = g.a[0][1] + g.a[1][0];
  u[6] = g.a[0][2] + g.a[2][0];
  u[7] = g.a[1][1] - g.a[2][2];
  u[8] = g.a[1][2] + g.a[2][1];
}

// adjoint of dec: gradient wrt the 3x3 given the gradient wrt its 9 irreducible components
__device__ __forceinline__ M3 decompose_T(const float g[9]) {
  M3 m;
  float t = (g[0] - g[4] - g[7]) * (1.0f / 3.0f);
  m.a[0][0] = t + g[4];
  m.a[0][1] = 0.5f * (g[5] - g[3]);
  m.a[0][2] = 0.5f * (g[6] + g[2]);
  m.a[1][0] = 0.5f * (g[5] + g[3]);
  m.a[1][1] = t + g[7];
  m.a[1][2] = 0.5f * (g[8] - g[1]);
  m.a[2][0] = 0.5f * (g[6] - g[2]);
  m.a[2][1] = 0.5f * (g[8] + g[1]);
  m.a[2][2] = t;
  return m;
}

// Frobenius norm^2 of compose(u) (reference tensornet.py:144-146) and its gradient wrt u
__device__ __forceinline__ float quad(const float u[9]) {
  float t = u[4] + u[7];
  return 3.0f * u[0] * u[0] + 2.0f * (u[1] * u[1] + u[2] * u[2] + u[3] * u[3]) + u[4] * u[4] + u[7] * u[7] + t * t +
         2.0f * (u[5] * u[5] + u[6] * u[6] + u[8] * u[8]);
}
__device__ __forceinline__ void dquad(const float u[9], float g[9]) {
  g[0] = 6.0f * u[0];
  g[1] = 4.0f * u[1];
  g[2] = 4.0f * u[2];
  g[3] = 4.0f * u[3];
  g[4] = 4.0f * u[4] + 2.0f * u[7];
  g[5] = 4.0f * u[5];
  g[6] = 4.0f * u[6];
  g[7] = 4.0f * u[7] + 2.0f * u[4];
  g[8] = 4.0f * u[8];
}

__device__ __forceinline__ M3 matmul(const M3& x, const M3& y) {
  M3 r;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) r.a[i][j] = x.a[i][0] * y.a[0][j] + x.a[i][1] * y.a[1][j] + x.a[i][2] * y.a[2][j];
  return r;
}
__device__ __forceinline__ M3 transpose(const M3& x) {
  M3 r;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) r.a[i][j] = x.a[j][i];
  return r;
}
__device__ __forceinline__ M3 add(const M3& x, const M3& y) {
  M3 r;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) r.a[i][j] = x.a[i][j] + y.a[i][j];
  return r;
}
__device__ __forceinline__ M3 scale(const M3& x, float s) {
  M3 r;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) r.a[i][j] = x.a[i][j] * s;
  return r;
}
__device__ __forceinline__ float frob2(const M3& x) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) s += x.a[i][j] * x.a[i][j];
  return s;
}

__device__ __forceinline__ int type_of(int c) { return c == 0 ? 0 : (c < 4 ? 1 : 2); }

// wave-level sum over 64 lanes (all lanes get the result)
// Workgroups are dealt to the 8 XCDs round-robin by linear id and each XCD has a private 4 MB L2.  For kernels
// whose block b works on item b (an atom row, a tile of pairs), neighbouring items share gathered rows (same
// molecule / cell), so give every XCD one CONTIGUOUS chunk of the items: bijective map block id -> item id.
__device__ __forceinline__ int xcd_chunk(int b, int n) {
  const int q = n >> 3, r = n & 7, x = b & 7, k = b >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

// Capacity-sized grids (static shapes: the grid covers the pair CAPACITY, only the first n_act items exist): deal the
// existing items to the XCDs in contiguous chunks, -1 for a block that has none.  With xcd_chunk over the capacity the
// existing items would all land on the first few XCDs.  Needs gridDim >= n_act.
__device__ __forceinline__ int xcd_chunk_act(int b, int n_act) {
  const int q = n_act >> 3, r = n_act & 7, x = b & 7, k = b >> 3;
  if (k >= q + (x < r ? 1 : 0)) return -1;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

// all-reduce over aligned groups of 2 / 4 / 8 / 16 lanes with DPP moves (VALU, no LDS round trip): quad_perm xor 1,
// quad_perm xor 2, row_half_mirror (lane i <-> 7 - i, pairs the two quads of 8), row_mirror (i <-> 15 - i)
__device__ __forceinline__ float row_sum(float v, int n) {  // n in {1, 2, 4, 8, 16}
  if (n >= 2) v = v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
  if (n >= 4) v = v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
  if (n >= 8) v = v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));
  if (n >= 16) v = v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));
  return v;
}
// sum over the 64 lanes of a wave, result in every lane: 4 DPP steps inside the rows of 16, 2 cross-row exchanges
__device__ __forceinline__ float wave_sum(float v) {
  v = row_sum(v, 16);
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

// four wave-wide sums for the price of seven exchanges: halve the value set at xor 32 and xor 16, then reduce one value
// over 16 lanes.  Totals land in lane 0 (a), 16 (b), 32 (c), 48 (d).
__device__ __forceinline__ float wave_sum4(float a, float b, float c, float d, int lane) {
  const bool hi = lane & 32;
  float k0 = hi ? c : a, k1 = hi ? d : b;          // kept pair
  const float s0 = hi ? a : c, s1 = hi ? b : d;    // pair handed to the partner lane
  k0 += __shfl_xor(s0, 32, 64);
  k1 += __shfl_xor(s1, 32, 64);
  const bool q = lane & 16;
  float v = q ? k1 : k0;
  v += __shfl_xor(q ? k0 : k1, 16, 64);
  return row_sum(v, 16);
}

// CosineCutoff and derivative (reference models/utils.py:506-528)
__device__ __forceinline__ void cosine_cutoff(float d, float lo, float up, float& c, float& dc) {
  const float PI = 3.14159265358979323846f;
  if (lo > 0.f) {
    float k = 2.0f * PI / (up - lo);
    float arg = PI * (2.0f * (d - lo) / (up - lo) + 1.0f);
    bool in = (d < up) && (d > lo);
    c = in ? 0.5f * (cosf(arg) + 1.0f) : 0.f;
    dc = in ? -0.5f * sinf(arg) * k : 0.f;
  } else {
    float k = PI / up;
    bool in = d < up;
    c = in ? 0.5f * (cosf(d * k) + 1.0f) : 0.f;
    dc = in ? -0.5f * sinf(d * k) * k : 0.f;
  }
}

}  // namespace tn
