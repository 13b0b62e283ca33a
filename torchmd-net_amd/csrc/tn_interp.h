// Cubic Hermite interpolation of the radial tables (tn_edge_table.hip) - ONE definition of the arithmetic, shared by the kernel
// that writes the per-pair rows and by the sweeps that evaluate them on the fly ("recompute_pair_rows"): the products and sums
// are spelled as fmaf chains so that every kernel rounds identically (recompute == stored, bit for bit).
#pragma once
#include <hip/hip_runtime.h>

namespace tn {

struct InterpCoef {
  float aD, a0, a1;  // value: f0 + (aD D + a0 s0 + a1 s1)
  float bD, b0, b1;  // slope: bD D + b0 s0 + b1 s1
  int k;             // grid interval
};
// table row k = [value | slope | divided difference] x R; row k + 1's slope sits 4 R further on; row T + 1 = the self pair (d = 0)
__device__ __forceinline__ InterpCoef interp_coef(float d, float lo, float h, float inv_h, int T) {
#pragma clang fp contract(off)
  InterpCoef c;
  // contraction off (the pragma above covers the operators written HERE - HIP's __fmul_rn / __fadd_rn are inline functions of a
  // header compiled with contraction on, and x - k was fused with the product before it) and every fused multiply-add spelled
  // out: left to itself the compiler contracts a*b+c differently from one kernel to the next, and the row a sweep interpolates
  // then differs from the stored one in its last bit
  const float x = (d - lo) * inv_h;
  int k = (int)x;
  k = k < 0 ? 0 : (k > T - 1 ? T - 1 : k);
  c.k = k;
  const float t = x - (float)k, t2 = t * t, t3 = t2 * t;
  const float m3 = 3.f * t2, m6 = 6.f * t, m2 = -2.f * t;
  c.aD = __builtin_fmaf(-2.f, t3, m3) * h;                       // (3 t^2 - 2 t^3) h
  c.a0 = (__builtin_fmaf(-2.f, t2, t3) + t) * h;                 // (t^3 - 2 t^2 + t) h
  c.a1 = (t3 - t2) * h;                                          // (t^3 - t^2) h
  c.bD = __builtin_fmaf(-6.f, t2, m6);                           // 6 t - 6 t^2
  c.b0 = __builtin_fmaf(3.f, t2, __builtin_fmaf(-4.f, t, 1.f));  // 3 t^2 - 4 t + 1
  c.b1 = __builtin_fmaf(3.f, t2, m2);                            // 3 t^2 - 2 t
  return c;
}
__device__ __forceinline__ float interp_value(float f0, float s0, float D, float s1, const InterpCoef& c) {
#pragma clang fp contract(off)
  const float p0 = c.aD * D;
  return f0 + __builtin_fmaf(c.a1, s1, __builtin_fmaf(c.a0, s0, p0));
}
__device__ __forceinline__ float interp_slope(float s0, float D, float s1, const InterpCoef& c) {
#pragma clang fp contract(off)
  const float p0 = c.bD * D;
  return __builtin_fmaf(c.b1, s1, __builtin_fmaf(c.b0, s0, p0));
}

// what a sweep needs to evaluate a layer's per-pair row itself
struct PairRowTable {
  const float* tab;  // [T + 2][3][R] or null (rows stored per pair)
  const float* pd;   // [P + 1] pair distances
  const int* counts; // counts[0] = P = index of the self pair (device side: static shapes have no host copy)
  float lo, h, inv_h;
  int T;
};
// row[type t](f) of pair p and, with DW, its distance derivative: 4 table loads per type instead of 1 (2) stored-row loads
template <bool DW>
__device__ __forceinline__ void pair_row_eval(const PairRowTable& q, int p, int R, int F, int f, float w[3], float dw[3]) {
  if (p == q.counts[0]) {
    const float* row = q.tab + (int64_t)(q.T + 1) * 3 * R + f;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      w[t] = row[t * F];
      if (DW) dw[t] = row[R + t * F];
    }
    return;
  }
  const InterpCoef c = interp_coef(q.pd[p], q.lo, q.h, q.inv_h, q.T);
  const float* row = q.tab + (int64_t)c.k * 3 * R + f;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const float f0 = row[t * F], s0 = row[R + t * F], D = row[2 * R + t * F], s1 = row[4 * R + t * F];
    w[t] = interp_value(f0, s0, D, s1, c);
    if (DW) dw[t] = interp_slope(s0, D, s1, c);
  }
}

}  // namespace tn
