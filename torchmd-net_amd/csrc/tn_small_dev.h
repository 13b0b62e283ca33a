// Device helpers shared by the per-atom phase kernels of small systems (tn_small.hip: one atom per block) and of mid-size
// systems (tn_mid.hip: four atoms per block).  Nothing here touches a block's LDS layout.
#pragma once
#include <hip/hip_runtime.h>

#include "tn_common.h"

namespace tn {
namespace {

constexpr int SM_G = 8;        // thread groups per block
constexpr int SM_FMAX = 128;   // channels
struct Blk {
  int F, g, f, tid, T;
};


__device__ __forceinline__ void ld9(const float* __restrict__ p, int F, float u[9]) {
#pragma unroll
  for (int c = 0; c < 9; ++c) u[c] = p[c * F];
}
__device__ __forceinline__ void st9(float* __restrict__ p, int F, const float u[9]) {
#pragma unroll
  for (int c = 0; c < 9; ++c) p[c * F] = u[c];
}

// ---- nine-component linear: every thread its share KQ = F / 8 of the input channels; the sums of the groups meet in part.
// Weights come from L2 and a block streams all 3 F^2 of them through its CU's miss path (~4 us per product, in-kernel
// timestamps): they are REQUESTED early (tlin_issue: 3 KQ registers) - before the sweep, or before group 0's 3x3 algebra - and
// multiplied once the input vector is in xs (tlin_finish).
template <int KQ>
struct TlinW {
  float w0[KQ], w1[KQ], w2[KQ];
};
template <int KQ>
__device__ __forceinline__ void tlin_issue(const float* const W[3], const Blk& b, TlinW<KQ>& t) {
  const int F = b.F, k0 = b.g * KQ;
#pragma unroll
  for (int k = 0; k < KQ; ++k) {
    t.w0[k] = W[0][(k0 + k) * F + b.f];
    t.w1[k] = W[1][(k0 + k) * F + b.f];
    t.w2[k] = W[2][(k0 + k) * F + b.f];
  }
}

// acc[c] = sum_k w_type(c)[k] * x[c][k0 + k]: the input vector sits in LDS as [9][F]; every lane of a wave reads the same address
// (k0 depends on the group only), so these reads are bound by instruction issue, not by LDS bandwidth: as 16-byte reads they are
// 9 KQ / 4 instructions instead of 9 KQ (the 64-atom step went 0.137 -> 0.130 ms with this alone, profiles/r04_notes.md section 7)
template <int KQ>
__device__ __forceinline__ void tlin_fma(const TlinW<KQ>& t, const float* x /* LDS, 16-byte aligned */, int F, int k0, float acc[9]) {
  typedef float v4 __attribute__((ext_vector_type(4)));
  static_assert(KQ % 4 == 0, "16-byte pieces");
#pragma unroll
  for (int c = 0; c < 9; ++c) {
    float s = 0.f;
#pragma unroll
    for (int k4 = 0; k4 < KQ / 4; ++k4) {
      const v4 xv = *reinterpret_cast<const v4*>(x + c * F + k0 + 4 * k4);
      const float* wt = c == 0 ? t.w0 : (c < 4 ? t.w1 : t.w2);
      s += wt[4 * k4] * xv.x;
      s += wt[4 * k4 + 1] * xv.y;
      s += wt[4 * k4 + 2] * xv.z;
      s += wt[4 * k4 + 3] * xv.w;
    }
    acc[c] = s;
  }
}

constexpr int SM_MV_KG = 16;
typedef float f4 __attribute__((ext_vector_type(4)));
struct MvW {
  f4 w[8];
};
// the slice of the input channels thread tid multiplies: [k0, k1), both multiples of 4 (K is: F, 2F, 3F or the head width), so
// that the input vector is read from LDS in 16-byte pieces; returns the number of slices in use
__device__ __forceinline__ int mv_slice(int K, int Nout, const Blk& b, int& c4, int& q, int& k0, int& k1) {
  const int n4 = Nout >> 2;
  const int kg = min(b.T / n4, SM_MV_KG);
  q = b.tid / n4;
  c4 = b.tid - q * n4;
  const int kq = (((K + kg - 1) / kg) + 3) & ~3;
  k0 = q * kq;
  k1 = min(K, k0 + kq);
  return (K + kq - 1) / kq;  // slices q < this one hold channels
}
// request the first eight weight rows of the thread's slice ahead of time (while the input vector is still being made)
__device__ __forceinline__ void mv_issue(const float* __restrict__ WT, int K, int Nout, const Blk& b, MvW& t) {
  int c4, q, k0, k1;
  const int kg = mv_slice(K, Nout, b, c4, q, k0, k1);
  if (q >= kg) return;
#pragma unroll
  for (int u = 0; u < 8; ++u)
    if (k0 + u < k1) t.w[u] = *reinterpret_cast<const f4*>(WT + (int64_t)(k0 + u) * Nout + 4 * c4);
}
// acc[a] += W[k .. k + 4) x[a][k .. k + 4) for NA input vectors (stride xs), the four weight rows given
template <int NA>
__device__ __forceinline__ void mv_fma4(const f4 w[4], const float* x, int xs, int k, f4 acc[NA]) {
#pragma unroll
  for (int a = 0; a < NA; ++a) {
    const f4 xv = *reinterpret_cast<const f4*>(x + a * xs + k);
    acc[a] += w[0] * xv.x;
    acc[a] += w[1] * xv.y;
    acc[a] += w[2] * xv.z;
    acc[a] += w[3] * xv.w;
  }
}

// LayerNorm statistics of v[0..R) (LDS), computed by every wave for itself: no barrier
__device__ __forceinline__ void row_stats(const float* v, int R, float& mean, float& rs) {
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int k = lane; k < R; k += 64) s += v[k];
  mean = wave_sum(s) / R;
  float var = 0.f;
  for (int k = lane; k < R; k += 64) {
    const float d = v[k] - mean;
    var += d * d;
  }
  rs = 1.0f / sqrtf(wave_sum(var) / R + 1e-5f);
}
// LayerNorm adjoint sums of g (LDS) against the stored normalised row xh (global) and the weight w: s1 = mean(g w), s2 = mean(g w xh)
__device__ __forceinline__ void lnbwd_stats(const float* g, const float* __restrict__ xh, const float* __restrict__ w, int R, float& s1,
                                            float& s2) {
  const int lane = threadIdx.x & 63;
  float a = 0.f, c = 0.f;
  for (int k = lane; k < R; k += 64) {
    const float gw = g[k] * w[k];
    a += gw;
    c += gw * xh[k];
  }
  s1 = wave_sum(a) / R;
  s2 = wave_sum(c) / R;
}

// group product + normalisation (reference tensornet.py:800-806): C_hat from Y = Pn[i], M = message
__device__ __forceinline__ void group_product(const float y[9], const float m[9], int o3, float kap, float ch[9]) {
  const M3 Y = compose(y), M = compose(m);
  const M3 Cm = o3 ? scale(add(matmul(Y, M), matmul(M, Y)), kap) : scale(matmul(Y, M), 2.0f);
  decompose(Cm, ch);
  const float inv = 1.0f / (frob2(Cm) + 1.0f);
#pragma unroll
  for (int c = 0; c < 9; ++c) ch[c] *= inv;
}
// its adjoint (k_message_bwd_node)
__device__ __forceinline__ void group_product_bwd(const float gc[9], const float y[9], const float m[9], int o3, float kap, float gm[9],
                                                  float gy[9]) {
  const M3 Y = compose(y), M = compose(m);
  const M3 Cm = o3 ? scale(add(matmul(Y, M), matmul(M, Y)), kap) : scale(matmul(Y, M), 2.0f);
  float uc[9];
  decompose(Cm, uc);
  const float inv = 1.0f / (frob2(Cm) + 1.0f);
  float dot = 0.f, guc[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) {
    dot += gc[c] * uc[c];
    guc[c] = gc[c] * inv;
  }
  const float g_t = -dot * inv * inv;
  const M3 gCm = add(decompose_T(guc), scale(Cm, 2.0f * g_t));
  const M3 Yt = transpose(Y), Mt = transpose(M);
  M3 gY, gM;
  if (o3) {
    gY = scale(add(matmul(gCm, Mt), matmul(Mt, gCm)), kap);
    gM = scale(add(matmul(Yt, gCm), matmul(gCm, Yt)), kap);
  } else {
    gY = scale(matmul(gCm, Mt), 2.0f);
    gM = scale(matmul(Yt, gCm), 2.0f);
  }
  compose_T(gM, gm);
  compose_T(gY, gy);
}
// g_D = compose^T(Gf + kappa (Gf dX^T + dX^T Gf)), Gf = dec^T(G)   (k_update_bwd)
__device__ __forceinline__ void update_bwd(const float gg[9], const float d[9], float kap, float o[9]) {
  const M3 Gf = decompose_T(gg);
  const M3 dXt = transpose(compose(d));
  const M3 gdx = add(Gf, scale(add(matmul(Gf, dXt), matmul(dXt, Gf)), kap));
  compose_T(gdx, o);
}


}  // namespace
}  // namespace tn
