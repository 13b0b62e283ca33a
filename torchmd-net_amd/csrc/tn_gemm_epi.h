// Epilogue kinds shared by the fp32 MFMA GEMM kernels (tn_gemm.hip: throughput tiles, tn_gemm_skinny.hip:
// latency-oriented split-K tiles).  The kind is a template parameter: no per-element branching.
#pragma once
#include <hip/hip_runtime.h>

#include "tn_gemm.h"

namespace tn {

enum EpiKind : int {
  EPI_PLAIN = 0,          // C = acc (+bias)
  EPI_SILU_PRE,           // pre = acc+bias ; C = silu(pre)
  EPI_SILU_PRE_ROWSCALE,  // pre = acc+bias ; C = silu(pre) * rowscale[m]
  EPI_MULAUX_PRE,         // pre = acc ; C = pre * aux[m,n]
  EPI_MULDSILU,           // C = acc * silu'(aux[m,n])
  EPI_ACCUM,              // C += acc
  EPI_GENERIC             // run-time flags (any other combination)
};

__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float fast_silu(float x) { return x * fast_sigmoid(x); }
__device__ __forceinline__ float fast_silu_grad(float x) {
  float s = fast_sigmoid(x);
  return s * (1.0f + x * (1.0f - s));
}

static inline int epi_kind(const GemmArgs& a) {
  const int f = a.flags;
  if (f == 0 && !a.pre) return EPI_PLAIN;
  if (f == GEMM_ACT_SILU && a.pre) return EPI_SILU_PRE;
  if (f == (GEMM_ACT_SILU | GEMM_ROWSCALE) && a.pre) return EPI_SILU_PRE_ROWSCALE;
  if (f == GEMM_MUL_AUX && a.pre) return EPI_MULAUX_PRE;
  if (f == GEMM_MUL_DSILU_AUX && !a.pre) return EPI_MULDSILU;
  if (f == GEMM_ACCUM && !a.pre) return EPI_ACCUM;
  return EPI_GENERIC;
}

// one output element: v = acc + bias already applied by the caller
template <int EPI>
__device__ __forceinline__ void epilogue_store(const GemmArgs& a, float* __restrict__ C, float* __restrict__ pre,
                                               const float* __restrict__ aux, int row, int col, float v) {
  float* cp = C + (int64_t)row * a.ldc + col;
  if (EPI == EPI_PLAIN) {
    *cp = v;
  } else if (EPI == EPI_SILU_PRE) {
    pre[(int64_t)row * a.ldpre + col] = v;
    *cp = fast_silu(v);
  } else if (EPI == EPI_SILU_PRE_ROWSCALE) {
    pre[(int64_t)row * a.ldpre + col] = v;
    *cp = fast_silu(v) * a.rowscale[row];
  } else if (EPI == EPI_MULAUX_PRE) {
    pre[(int64_t)row * a.ldpre + col] = v;
    *cp = v * aux[(int64_t)row * a.ldaux + col];
  } else if (EPI == EPI_MULDSILU) {
    *cp = v * fast_silu_grad(aux[(int64_t)row * a.ldaux + col]);
  } else if (EPI == EPI_ACCUM) {
    *cp = v + *cp;
  } else {
    const int flags = a.flags;
    if (pre) pre[(int64_t)row * a.ldpre + col] = v;
    if (flags & GEMM_ACT_SILU) v = fast_silu(v);
    if (flags & GEMM_ROWSCALE) v *= a.rowscale[row];
    if (flags & GEMM_MUL_AUX) v *= aux[(int64_t)row * a.ldaux + col];
    if (flags & GEMM_MUL_DSILU_AUX) v *= fast_silu_grad(aux[(int64_t)row * a.ldaux + col]);
    if (flags & GEMM_ACCUM) v += *cp;
    *cp = v;
  }
}

// latency-oriented variant for small row counts (tn_gemm_skinny.hip)
int launch_gemm_skinny(const GemmArgs& a, hipStream_t stream);

}  // namespace tn
