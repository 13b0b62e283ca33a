// 16-byte-per-lane variants of the gather / scatter kernels of the TensorNet path (gfx950).
//
// These kernels are bound by L2 / HBM gathers (per directed edge 3F edge weights + 9F source
// features).  Each thread owns 4 consecutive channels so every access is a global_load_dwordx4
// (1 KiB per wave-instruction); an atom (or pair) is served by TPA = F/4 lanes, a wave by 64/TPA
// atoms, and the per-pair reductions over channels are xor-shuffles inside the TPA-lane group.
// Used when F % 4 == 0, F/4 is a power of two and F/4 <= 64; otherwise the scalar kernels in
// tn_kernels.hip run.  Same arithmetic, same summation order over edges (deterministic).
#include <cstdlib>

#include "tn_common.h"
#include "tn_kernels.h"

namespace tn {

static inline int cdiv_(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

bool gather_v4_ok(int F) {
  static const bool off = getenv("TMDNET_NO_V4") != nullptr;  // developer switch: force the scalar kernels
  if (off || (F & 3)) return false;
  int f4 = F >> 2;
  return f4 >= 1 && f4 <= 64 && (f4 & (f4 - 1)) == 0;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void fma4(float4& a, const float4& w, const float4& s) {
  a.x += w.x * s.x;
  a.y += w.y * s.y;
  a.z += w.z * s.z;
  a.w += w.w * s.w;
}
__device__ __forceinline__ float comp(const float4& v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : (k == 2 ? v.z : v.w)); }
__device__ __forceinline__ void setc(float4& v, int k, float x) {
  if (k == 0) v.x = x; else if (k == 1) v.y = x; else if (k == 2) v.z = x; else v.w = x;
}
__device__ __forceinline__ float group_sum(float v, int tpa) {
  for (int off = tpa >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ void csr_gather4(const Graph& g, int i, int F, int f, const float* __restrict__ w,
                                            const float* __restrict__ src, float4 acc[9]) {
  const int e0 = g.rowptr[i], e1 = g.rowptr[i + 1];
  const int F3 = 3 * F, F9 = 9 * F;
#pragma unroll
  for (int c = 0; c < 9; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int e = e0; e < e1; ++e) {
    const int j = g.col[e], p = g.epair[e];
    const float* wp = w + (int64_t)p * F3 + f;
    const float* sp = src + (int64_t)j * F9 + f;
    const float4 w0 = ld4(wp), w1 = ld4(wp + F), w2 = ld4(wp + 2 * F);
    float4 s[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) s[c] = ld4(sp + c * F);
    fma4(acc[0], w0, s[0]);
    fma4(acc[1], w1, s[1]);
    fma4(acc[2], w1, s[2]);
    fma4(acc[3], w1, s[3]);
#pragma unroll
    for (int c = 4; c < 9; ++c) fma4(acc[c], w2, s[c]);
  }
}

// message passing + group product + normalisation (reference tensornet.py:757-806); ADJOINT: gPn[i] += sum_e w gMi[j]
template <bool ADJOINT>
__global__ __launch_bounds__(256) void k_message_v4(Graph g, int N, int F, const float* __restrict__ w,
                                                    const float* __restrict__ src, const float* __restrict__ q,
                                                    const int64_t* __restrict__ batch, int o3, float* __restrict__ out0,
                                                    float* __restrict__ out1) {
  const int tpa = F >> 2, apb = 256 / tpa;
  const int i = xcd_chunk(blockIdx.x, gridDim.x) * apb + threadIdx.x / tpa;
  if (i >= N || g.counts[2]) return;
  const int f = (threadIdx.x % tpa) << 2;
  float4 m[9];
  csr_gather4(g, i, F, f, w, src, m);
  const int64_t base = (int64_t)i * 9 * F + f;
  if (ADJOINT) {
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      float4 o = ld4(out0 + base + c * F);
      o.x += m[c].x; o.y += m[c].y; o.z += m[c].z; o.w += m[c].w;
      st4(out0 + base + c * F, o);
    }
    return;
  }
  const float kap = q ? (batch ? 1.0f + 0.1f * q[batch[i]] : q[i]) : 1.0f;
  float4 y[9], ch[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) {
    y[c] = ld4(src + base + c * F);
    st4(out0 + base + c * F, m[c]);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float yu[9], mu[9], uc[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      yu[c] = comp(y[c], k);
      mu[c] = comp(m[c], k);
    }
    const M3 Y = compose(yu), M = compose(mu);
    const M3 Cm = o3 ? scale(add(matmul(Y, M), matmul(M, Y)), kap) : scale(matmul(Y, M), 2.0f);
    decompose(Cm, uc);
    const float inv = 1.0f / (frob2(Cm) + 1.0f);
#pragma unroll
    for (int c = 0; c < 9; ++c) setc(ch[c], k, uc[c] * inv);
  }
#pragma unroll
  for (int c = 0; c < 9; ++c) st4(out1 + base + c * F, ch[c]);
}

void launch_message_v4(const Graph& g, int N, int F, const float* w, const float* src, const float* q, const int64_t* batch,
                       int o3, float* Mi, float* Ch, hipStream_t s) {
  const int apb = 256 / (F >> 2);
  hipLaunchKernelGGL(k_message_v4<false>, dim3(cdiv_(N, apb)), dim3(256), 0, s, g, N, F, w, src, q, batch, o3, Mi, Ch);
}
void launch_message_adjoint_v4(const Graph& g, int N, int F, const float* w, const float* gMi, float* gPn, hipStream_t s) {
  const int apb = 256 / (F >> 2);
  hipLaunchKernelGGL(k_message_v4<true>, dim3(cdiv_(N, apb)), dim3(256), 0, s, g, N, F, w, gMi, nullptr, nullptr, 0, gPn,
                     nullptr);
}


// embedding scatter (see k_embed_scatter in tn_kernels.hip)
__global__ __launch_bounds__(256) void k_embed_scatter_v4(Graph g, int N, int F, const int64_t* __restrict__ z,
                                                          const float* __restrict__ Utab, const float* __restrict__ Vtab,
                                                          const float* __restrict__ Q, const float* __restrict__ C,
                                                          float* __restrict__ u0, float* __restrict__ s0n) {
  const int tpa = F >> 2, apb = 256 / tpa;
  const int i = xcd_chunk(blockIdx.x, gridDim.x) * apb + threadIdx.x / tpa;
  if (i >= N || g.counts[2]) return;
  const int f = (threadIdx.x % tpa) << 2;
  const int e0 = g.rowptr[i], e1 = g.rowptr[i + 1];
  const int F3 = 3 * F;
  const float4 Ui = ld4(Utab + z[i] * F + f);
  float4 acc[10];
#pragma unroll
  for (int c = 0; c < 10; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int e = e0; e < e1; ++e) {
    const int j = g.col[e], p = g.epair[e];
    const float sg = g.esign[e];
    float rx = 0.f, ry = 0.f, rz = 0.f;
    if (sg != 0.f) {
      rx = sg * g.prhat[p * 3];
      ry = sg * g.prhat[p * 3 + 1];
      rz = sg * g.prhat[p * 3 + 2];
    }
    const float cp = C[p];
    const float4 Vj = ld4(Vtab + z[j] * F + f);
    const float* qp = Q + (int64_t)p * F3 + f;
    const float4 q0 = ld4(qp), q1 = ld4(qp + F), q2 = ld4(qp + 2 * F);
    const float rr[6] = {rx * rx, rx * ry, rx * rz, ry * ry, ry * rz, rz * rz};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float cz = cp * (comp(Ui, k) + comp(Vj, k));
      const float W0 = cz * comp(q0, k), W1 = cz * comp(q1, k), W2 = cz * comp(q2, k);
      setc(acc[0], k, comp(acc[0], k) + W0);
      setc(acc[1], k, comp(acc[1], k) + W1 * rx);
      setc(acc[2], k, comp(acc[2], k) + W1 * ry);
      setc(acc[3], k, comp(acc[3], k) + W1 * rz);
#pragma unroll
      for (int c = 0; c < 6; ++c) setc(acc[4 + c], k, comp(acc[4 + c], k) + W2 * rr[c]);
    }
  }
  // acc[4..9] = T00,T01,T02,T11,T12,T22
  float4 out[9], sn;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float t00 = comp(acc[4], k), t11 = comp(acc[7], k), t22 = comp(acc[9], k);
    const float tr3 = (t00 + t11 + t22) * (1.0f / 3.0f);
    float u[9] = {comp(acc[0], k), comp(acc[1], k), comp(acc[2], k), comp(acc[3], k), t00 - tr3,
                  comp(acc[5], k), comp(acc[6], k), t11 - tr3, comp(acc[8], k)};
#pragma unroll
    for (int c = 0; c < 9; ++c) setc(out[c], k, u[c]);
    setc(sn, k, quad(u));
  }
  float* o = u0 + (int64_t)i * 9 * F + f;
#pragma unroll
  for (int c = 0; c < 9; ++c) st4(o + c * F, out[c]);
  st4(s0n + (int64_t)i * F + f, sn);
}
void launch_embed_scatter_v4(const Graph& g, int N, int F, const int64_t* z, const float* Utab, const float* Vtab, const float* Q,
                             const float* C, float* u0, float* s0n, hipStream_t s) {
  const int apb = 256 / (F >> 2);
  hipLaunchKernelGGL(k_embed_scatter_v4, dim3(cdiv_(N, apb)), dim3(256), 0, s, g, N, F, z, Utab, Vtab, Q, C, u0, s0n);
}


}  // namespace tn
