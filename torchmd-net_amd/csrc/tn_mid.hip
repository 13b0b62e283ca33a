// Per-atom phase kernels with FOUR atoms per block: systems of 513 .. 1 024 atoms (MD stepping of a larger molecule; one round of
// blocks on the 256 CUs - beyond that the general schedule is as fast).  Same cut of the step as tn_small.hip - embedding, one kernel per layer, one per
// reverse layer, nothing but the neighbour sweeps between them - and the same arithmetic; what changes is who shares what:
//
// * a block of tn_small.hip streams every weight matrix of its phase through its own CU for ONE atom (3.5 MB per block and step,
//   ~64 of the 133 us of a 64-atom step): fine while the blocks are few, hopeless at 2 048 atoms (tn_small.hip's kernels lose to
//   the general schedule from 1 024 atoms on).  Here a thread keeps its slice of the weights in registers and applies it to the
//   input vectors of the block's A = 4 atoms: a quarter of the weight traffic per atom;
// * the 8 thread groups of a block are dealt to the atoms: a sweep gives every atom 8 / A = 2 groups (each takes every second
//   edge of the row), the per-channel 3x3 algebra of atom a runs on group a ("owner"), products use all groups for all atoms.
//
// Deterministic (fixed summation orders; they differ from tn_small.hip's and from the general schedule's by rounding only).
#include <cstdlib>

#include "tn_common.h"
#include "tn_small.h"
#include "tn_small_dev.h"

namespace tn {

namespace {

constexpr int MD_PART = 18432;  // floats: sweep partials [8][10][F], tensor-linear partials [8][9][F], MLP partials [kg <= 16][A][Nout]
constexpr int MD_V = 3 * SM_FMAX;  // stride of an atom's vectors (longest: 3F)
template <int A>
struct __attribute__((aligned(16))) MidLds {
  float part[MD_PART];
  float xs[A * 9 * SM_FMAX];    // tensor-linear inputs of the block's atoms [atom][component][channel]
  float red[A * 10 * SM_FMAX];  // reduced partial sums [atom][component][channel]
  float va[A * MD_V], vb[A * MD_V], vc[A * MD_V];
};

template <int A>
struct Own {       // what a thread group owns
  int a, i;        // atom slot of the block, atom index (group g < A owns slot g)
  bool is, valid;  // this group owns a slot; the slot holds a real atom
};

// tensor linear of the block's A atoms: owners hand in u, get out (3 * KQ weights per thread, requested after the first barrier)
template <int KQ, int A>
__device__ __forceinline__ void tlin_multi(const float* const W[3], const float u[9], const Own<A>& o, MidLds<A>& L, const Blk& b,
                                           float out[9]) {
  const int F = b.F, k0 = b.g * KQ;
  if (o.is) {
#pragma unroll
    for (int c = 0; c < 9; ++c) L.xs[(o.a * 9 + c) * F + b.f] = o.valid ? u[c] : 0.f;
  }
  __syncthreads();
  TlinW<KQ> t;
  tlin_issue<KQ>(W, b, t);
  // two atoms per pass (their partials fill the 72 KB of part): 2 barriers per pair of atoms
  static_assert(A % 2 == 0, "atoms are processed in pairs");
#pragma unroll 1
  for (int a = 0; a < A; a += 2) {
    float acc[2][9];
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      tlin_fma<KQ>(t, L.xs + (a + v) * 9 * F, F, k0, acc[v]);
#pragma unroll
      for (int c = 0; c < 9; ++c) L.part[((v * SM_G + b.g) * 9 + c) * F + b.f] = acc[v][c];
    }
    __syncthreads();
    for (int vc = b.g; vc < 18; vc += SM_G) {  // 18 (atom of the pair, component) sums over the 8 groups
      const int v = vc / 9, c = vc - v * 9;
      float s = 0.f;
#pragma unroll
      for (int gg = 0; gg < SM_G; ++gg) s += L.part[((v * SM_G + gg) * 9 + c) * F + b.f];
      L.red[((a + v) * 10 + c) * F + b.f] = s;
    }
    __syncthreads();
  }
  if (o.is) {
#pragma unroll
    for (int c = 0; c < 9; ++c) out[c] = L.red[(o.a * 10 + c) * F + b.f];
  }
}

// y[a][n] = bias[n] + sum_k WT[k, n] x[a][k] for the block's A atoms (x, y in LDS with stride MD_V per atom); a thread's weight
// rows are loaded once and multiplied with all A input vectors
template <int A>
__device__ __forceinline__ void matvec_multi(const float* __restrict__ WT, const float* __restrict__ bias, const float* x, int K, int Nout,
                                             float* part, float* y, const Blk& b) {
  int c4, q, k0, k1;
  const int kg = mv_slice(K, Nout, b, c4, q, k0, k1);  // slices in use
  if (q < kg) {
    f4 acc[A];
#pragma unroll
    for (int a = 0; a < A; ++a) acc[a] = (f4)(0.f);
    int k = k0;
#pragma unroll 1
    for (; k + 8 <= k1; k += 8) {
      f4 wv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) wv[u] = *reinterpret_cast<const f4*>(WT + (int64_t)(k + u) * Nout + 4 * c4);
      mv_fma4<A>(wv, x, MD_V, k, acc);
      mv_fma4<A>(wv + 4, x, MD_V, k + 4, acc);
    }
    if (k < k1) {
      f4 wv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) wv[u] = *reinterpret_cast<const f4*>(WT + (int64_t)(k + u) * Nout + 4 * c4);
      mv_fma4<A>(wv, x, MD_V, k, acc);
    }
#pragma unroll
    for (int a = 0; a < A; ++a) *reinterpret_cast<f4*>(part + (q * A + a) * Nout + 4 * c4) = acc[a];
  }
  __syncthreads();
  for (int idx = b.tid; idx < A * Nout; idx += b.T) {
    const int a = idx / Nout, n = idx - a * Nout;
    float s = bias ? bias[n] : 0.f;
    for (int qq = 0; qq < kg; ++qq) s += part[(qq * A + a) * Nout + n];
    y[a * MD_V + n] = s;
  }
  __syncthreads();
}

// message sweep of the block's A rows: the 8 / A groups of atom a take every (8 / A)-th edge of its row, two edges per trip with
// the next trip's records requested ahead (tn_small.hip sweep); partial sums -> part[group][9][F]
template <int MODE, int A>
__device__ __forceinline__ void sweep_multi(const Graph& g, int i0, int N, const float* __restrict__ w, const float* __restrict__ dw,
                                            const float* __restrict__ src, const float* __restrict__ Y, float* __restrict__ slots,
                                            int64_t slot_stride, float* part, const Blk& b) {
  constexpr int GPA = SM_G / A;
  const int F = b.F, F3 = 3 * F, F9 = 9 * F, f = b.f;
  const int a = b.g / GPA, sub = b.g - a * GPA, i = i0 + a;
  float acc[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) acc[c] = 0.f;
  if (i < N) {
    const int e0 = g.rowptr[i], e1 = g.rowptr[i + 1];
    float y[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) y[c] = MODE == 2 ? Y[(int64_t)i * F9 + c * F + f] : 0.f;
    auto rec = [&](int e, int& j, int& p, float& sg) __attribute__((always_inline)) {
      const bool ok = e < e1;
      j = ok ? g.col[e] : 0;
      p = ok ? g.epair[e] : 0;
      sg = ok && MODE == 2 ? g.esign[e] : 0.f;
    };
    int jA, pA, jB, pB;
    float sA, sB;
    int e = e0 + sub;
    rec(e, jA, pA, sA);
    rec(e + GPA, jB, pB, sB);
    for (; e < e1; e += 2 * GPA) {
      const bool two = e + GPA < e1;
      const int j0 = jA, p0 = pA, j1 = jB, p1 = pB;
      const float sg0 = sA, sg1 = sB;
      float wr[2][3], dr[2][3], s9[2][9];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int jj = u ? j1 : j0, pp = u ? p1 : p0;
        const float* wp = w + (int64_t)pp * F3 + f;
        const float* sp = src + (int64_t)jj * F9 + f;
        if (u == 0 || two) {
#pragma unroll
          for (int k = 0; k < 3; ++k) wr[u][k] = wp[k * F];
          if (MODE == 2) {
            const float* dp = dw + (int64_t)pp * F3 + f;
#pragma unroll
            for (int k = 0; k < 3; ++k) dr[u][k] = dp[k * F];
          }
#pragma unroll
          for (int c = 0; c < 9; ++c) s9[u][c] = sp[c * F];
        } else {
#pragma unroll
          for (int k = 0; k < 3; ++k) wr[u][k] = dr[u][k] = 0.f;
#pragma unroll
          for (int c = 0; c < 9; ++c) s9[u][c] = 0.f;
        }
      }
      rec(e + 2 * GPA, jA, pA, sA);
      rec(e + 3 * GPA, jB, pB, sB);
      float h[2] = {0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        acc[0] += wr[u][0] * s9[u][0];
        acc[1] += wr[u][1] * s9[u][1];
        acc[2] += wr[u][1] * s9[u][2];
        acc[3] += wr[u][1] * s9[u][3];
        acc[4] += wr[u][2] * s9[u][4];
        acc[5] += wr[u][2] * s9[u][5];
        acc[6] += wr[u][2] * s9[u][6];
        acc[7] += wr[u][2] * s9[u][7];
        acc[8] += wr[u][2] * s9[u][8];
        if (MODE == 2)
          h[u] = dr[u][0] * (s9[u][0] * y[0]) + dr[u][1] * (s9[u][1] * y[1] + s9[u][2] * y[2] + s9[u][3] * y[3]) +
                 dr[u][2] * (s9[u][4] * y[4] + s9[u][5] * y[5] + s9[u][6] * y[6] + s9[u][7] * y[7] + s9[u][8] * y[8]);
      }
      if (MODE == 2) {
        const float h0 = wave_sum(h[0]), h1 = wave_sum(h[1]);
        if ((f & 63) == 0) {
          if (sg0 != 0.f) slots[(int64_t)(f >> 6) * slot_stride + 2 * p0 + (sg0 > 0.f ? 0 : 1)] = h0;
          if (two && sg1 != 0.f) slots[(int64_t)(f >> 6) * slot_stride + 2 * p1 + (sg1 > 0.f ? 0 : 1)] = h1;
        }
      }
    }
  }
#pragma unroll
  for (int c = 0; c < 9; ++c) part[(b.g * 9 + c) * F + f] = acc[c];
}
// owner of slot a: sum of its groups' partials (after the barrier that follows the sweep)
template <int NC, int A>
__device__ __forceinline__ void own_sum(const float* part, const Own<A>& o, const Blk& b, float out[NC]) {
  constexpr int GPA = SM_G / A;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    float s = 0.f;
#pragma unroll
    for (int sub = 0; sub < GPA; ++sub) s += part[((o.a * GPA + sub) * NC + c) * b.F + b.f];
    out[c] = s;
  }
}

}  // namespace

// =====================================================================================================================
template <int F, int A>
__global__ __launch_bounds__(SM_G * F) void k_mid_embed(SmallEmbedArgs a) {
  constexpr int KQ = F / SM_G, GPA = SM_G / A;
  __shared__ MidLds<A> L;
  const Graph& g = a.g;
  if (g.counts[2]) return;  // pair overflow: the adjacency was not filled (the host reports the error)
  const int i0 = blockIdx.x * A, F3 = 3 * F;
  Blk b{F, (int)threadIdx.x / F, (int)threadIdx.x % F, (int)threadIdx.x, SM_G * F};
  const int f = b.f;
  const Own<A> o{b.g, i0 + b.g, b.g < A, b.g < A && i0 + b.g < a.N};

  // ---- embedding scatter: I0 = sum W0 ; v = sum W1 r ; T = sum W2 r r^T   (k_embed_scatter_split, 8 / A groups per atom)
  {
    const int sl = b.g / GPA, sub = b.g - sl * GPA, i = i0 + sl;
    float s[10];
#pragma unroll
    for (int c = 0; c < 10; ++c) s[c] = 0.f;
    if (i < a.N) {
      const int e0 = g.rowptr[i], e1 = g.rowptr[i + 1];
      const int64_t zi = a.z[i];
      const float Ui = a.Utab[zi * F + f];
      // three edges per trip: their operand chains (record -> geometry, species -> table row, Q row) are independent
      constexpr int EU = 3;
      for (int e = e0 + sub; e < e1; e += EU * GPA) {
        float rx[EU], ry[EU], rz[EU], cz[EU], q0[EU], q1[EU], q2[EU];
#pragma unroll
        for (int u = 0; u < EU; ++u) {
          const int ee = e + u * GPA;
          const bool ok = ee < e1;
          const int j = ok ? g.col[ee] : 0, p = ok ? g.epair[ee] : 0;
          const float sg = ok ? g.esign[ee] : 0.f;
          rx[u] = sg * g.prhat[p * 3];
          ry[u] = sg * g.prhat[p * 3 + 1];
          rz[u] = sg * g.prhat[p * 3 + 2];
          if (sg == 0.f) rx[u] = ry[u] = rz[u] = 0.f;  // self edge: the self pair's geometry slot is not written
          cz[u] = ok ? a.C[p] * (Ui + a.Vtab[a.z[j] * F + f]) : 0.f;
          const float* qq = a.Q + (int64_t)p * F3 + f;
          q0[u] = qq[0];
          q1[u] = qq[F];
          q2[u] = qq[2 * F];
        }
#pragma unroll
        for (int u = 0; u < EU; ++u) {
          const float W0 = cz[u] * q0[u], W1 = cz[u] * q1[u], W2 = cz[u] * q2[u];
          s[0] += W0;
          s[1] += W1 * rx[u]; s[2] += W1 * ry[u]; s[3] += W1 * rz[u];
          s[4] += W2 * rx[u] * rx[u]; s[5] += W2 * rx[u] * ry[u]; s[6] += W2 * rx[u] * rz[u];
          s[7] += W2 * ry[u] * ry[u]; s[8] += W2 * ry[u] * rz[u]; s[9] += W2 * rz[u] * rz[u];
        }
      }
    }
#pragma unroll
    for (int c = 0; c < 10; ++c) L.part[(b.g * 10 + c) * F + f] = s[c];
  }
  __syncthreads();
  float u[9];
  if (o.is) {
    float s[10];
    own_sum<10, A>(L.part, o, b, s);
    const float tr3 = (s[4] + s[7] + s[9]) * (1.0f / 3.0f);
    u[0] = s[0]; u[1] = s[1]; u[2] = s[2]; u[3] = s[3];
    u[4] = s[4] - tr3; u[5] = s[5]; u[6] = s[6]; u[7] = s[7] - tr3; u[8] = s[8];
    if (o.valid) st9(a.u0 + (int64_t)o.i * 9 * F + f, F, u);
    // ---- LayerNorm of the invariant (reference tensornet.py:595-611): the owner's two waves have the whole row between them
    L.va[o.a * MD_V + f] = o.valid ? quad(u) : 0.f;
  }
  __syncthreads();
  if (o.is) {
    float mean, rs;
    row_stats(L.va + o.a * MD_V, F, mean, rs);
    const float xh = (L.va[o.a * MD_V + f] - mean) * rs;
    L.vb[o.a * MD_V + f] = xh * a.ln0_w[f] + a.ln0_b[f];
    if (o.valid) {
      a.xh0[(int64_t)o.i * F + f] = xh;
      if (f == 0) a.rstd0[o.i] = rs;
    }
  }
  __syncthreads();
  matvec_multi<A>(a.L1T, a.bL1, L.vb, F, 2 * F, L.part, L.va, b);
  for (int idx = b.tid; idx < A * 2 * F; idx += b.T) {
    const int sl = idx / (2 * F), n = idx - sl * 2 * F;
    const float v = L.va[sl * MD_V + n];
    if (i0 + sl < a.N) a.a1[(int64_t)(i0 + sl) * 2 * F + n] = v;
    L.va[sl * MD_V + n] = silu(v);
  }
  __syncthreads();
  matvec_multi<A>(a.L2T, a.bL2, L.va, 2 * F, 3 * F, L.part, L.vb, b);
  for (int idx = b.tid; idx < A * 3 * F; idx += b.T) {
    const int sl = idx / (3 * F), n = idx - sl * 3 * F;
    const float v = L.vb[sl * MD_V + n], gt = silu(v);
    if (i0 + sl < a.N) {
      a.a2[(int64_t)(i0 + sl) * 3 * F + n] = v;
      a.gates[(int64_t)(i0 + sl) * 3 * F + n] = gt;
    }
    L.vb[sl * MD_V + n] = gt;
  }
  // ---- tensor linear, gates -> X0 (the first barrier inside tlin_multi publishes vb as well)
  float ux[9], x0[9];
  tlin_multi<KQ, A>(a.UeT, u, o, L, b, ux);
  if (o.is) {
#pragma unroll
    for (int c = 0; c < 9; ++c) x0[c] = ux[c] * L.vb[o.a * MD_V + type_of(c) * F + f];
    if (o.valid) {
      st9(a.UX + (int64_t)o.i * 9 * F + f, F, ux);
      st9(a.X0 + (int64_t)o.i * 9 * F + f, F, x0);
    }
    const float inv = 1.0f / (quad(x0) + 1.0f);  // layer 0: X_hat = X / (||X||^2 + 1)
#pragma unroll
    for (int c = 0; c < 9; ++c) x0[c] *= inv;
  }
  if (a.L == 0) return;
  float pn[9];
  tlin_multi<KQ, A>(a.V0T, x0, o, L, b, pn);
  if (o.valid) st9(a.Pn0 + (int64_t)o.i * 9 * F + f, F, pn);
}

// =====================================================================================================================
template <int F, int A, int LAST>
__global__ __launch_bounds__(SM_G * F) void k_mid_layer(SmallLayerArgs a) {
  constexpr int KQ = F / SM_G;
  __shared__ MidLds<A> L;
  const Graph& g = a.g;
  if (g.counts[2]) return;
  const int i0 = blockIdx.x * A;
  Blk b{F, (int)threadIdx.x / F, (int)threadIdx.x % F, (int)threadIdx.x, SM_G * F};
  const int f = b.f;
  const Own<A> o{b.g, i0 + b.g, b.g < A, b.g < A && i0 + b.g < a.N};
  const int64_t row9 = (int64_t)o.i * 9 * F + f;
  const float kap = (a.kap && o.valid) ? a.kap[o.i] : 1.0f;

  sweep_multi<0, A>(g, i0, a.N, a.w, nullptr, a.Pn, nullptr, nullptr, 0, L.part, b);
  __syncthreads();
  float ch[9];
  if (o.is) {
    float m[9], y[9];
    own_sum<9, A>(L.part, o, b, m);
    if (o.valid) {
      st9(a.Mi + row9, F, m);
      ld9(a.Pn + row9, F, y);
      group_product(y, m, a.o3, kap, ch);
    }
  }
  __syncthreads();  // part is free again
  float d[9], xn[9];
  tlin_multi<KQ, A>(a.VbT, ch, o, L, b, d);
  if (o.valid) {
    st9(a.D + row9, F, d);
    float xh[9];
    ld9(a.X + row9, F, xh);
    const float inv = 1.0f / (quad(xh) + 1.0f);
#pragma unroll
    for (int c = 0; c < 9; ++c) xh[c] *= inv;
    const M3 dX = compose(d);
    const M3 Xf = add(add(compose(xh), dX), scale(matmul(dX, dX), kap));
    decompose(Xf, xn);
    st9(a.Xn + row9, F, xn);
  }
  if (!LAST) {
    float xh[9], pn[9];
    if (o.valid) {
      const float inv = 1.0f / (quad(xn) + 1.0f);
#pragma unroll
      for (int c = 0; c < 9; ++c) xh[c] = xn[c] * inv;
    }
    tlin_multi<KQ, A>(a.VnT, xh, o, L, b, pn);
    if (o.valid) st9(a.Pn_next + row9, F, pn);
    return;
  }
  // ---- readout: invariants, LayerNorm, MLP, head (reference tensornet.py:384-398, output_modules.py:43-73)
  if (o.is) {
    float ft[3] = {0.f, 0.f, 0.f};
    if (o.valid) {
      const float t = xn[4] + xn[7];
      ft[0] = 3.0f * xn[0] * xn[0];
      ft[1] = 2.0f * (xn[1] * xn[1] + xn[2] * xn[2] + xn[3] * xn[3]);
      ft[2] = xn[4] * xn[4] + xn[7] * xn[7] + t * t + 2.0f * (xn[5] * xn[5] + xn[6] * xn[6] + xn[8] * xn[8]);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) L.va[o.a * MD_V + k * F + f] = ft[k];
  }
  __syncthreads();
  float rs = 0.f;
  if (o.is) {
    float mean;
    row_stats(L.va + o.a * MD_V, 3 * F, mean, rs);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int n = k * F + f;
      const float xh = (L.va[o.a * MD_V + n] - mean) * rs;
      if (o.valid) a.xhr[(int64_t)o.i * 3 * F + n] = xh;
      L.vb[o.a * MD_V + n] = xh * a.lnr_w[n] + a.lnr_b[n];
    }
    if (o.valid && f == 0) a.rstdr[o.i] = rs;
  }
  __syncthreads();
  matvec_multi<A>(a.LinT, a.bLin, L.vb, 3 * F, F, L.part, L.vc, b);  // vc[.][0..F) = al
  for (int idx = b.tid; idx < A * F; idx += b.T) {
    const int sl = idx / F, n = idx - sl * F;
    const float v = L.vc[sl * MD_V + n], xv = silu(v);
    if (i0 + sl < a.N) {
      a.al[(int64_t)(i0 + sl) * F + n] = v;
      a.x[(int64_t)(i0 + sl) * F + n] = xv;
    }
    L.va[sl * MD_V + n] = xv;
  }
  __syncthreads();
  matvec_multi<A>(a.O1T, a.bO1, L.va, F, a.H, L.part, L.vb, b);  // vb[.][0..H) = ao
  const float wgt = (a.atom_w && o.valid) ? a.atom_w[a.perm ? a.perm[o.i] : o.i] : 1.0f;  // weight of the owner's atom in the energy sum
  if (o.valid && f < 64) {
    float s = 0.f;
    for (int k = f; k < a.H; k += 64) s += silu(L.vb[o.a * MD_V + k]) * a.O2[k];
    s = wave_sum(s);
    if (f == 0) {
      float e = (s + a.bO2[0]) * a.std_;
      if (a.atomref) e += a.atomref[a.z[o.i]];
      a.ea[o.i] = wgt * e;
    }
  }
  if (!a.want_forces) return;
  // ---- reverse: head, MLP, LayerNorm, invariants -> G = d E / d X[L]
  if (o.is) L.vc[o.a * MD_V + 2 * F + (f & 63)] = wgt;  // the owners' weights, read by every thread below (vc[.][2F..] is free)
  __syncthreads();
  for (int idx = b.tid; idx < A * a.H; idx += b.T) {
    const int sl = idx / a.H, k = idx - sl * a.H;
    L.va[sl * MD_V + k] = L.vc[sl * MD_V + 2 * F] * a.std_ * a.O2[k] * silu_grad(L.vb[sl * MD_V + k]);  // g_ao
  }
  __syncthreads();
  matvec_multi<A>(a.O1, nullptr, L.va, a.H, F, L.part, L.vb, b);  // vb[.][0..F) = g_ao O1
  for (int idx = b.tid; idx < A * F; idx += b.T) {
    const int sl = idx / F, n = idx - sl * F;
    L.vb[sl * MD_V + n] *= silu_grad(L.vc[sl * MD_V + n]);  // g_al
  }
  __syncthreads();
  matvec_multi<A>(a.Lin, nullptr, L.vb, F, 3 * F, L.part, L.va, b);  // va[.][0..3F) = g_ln
  float gD[9];
  if (o.valid) {
    const float* xh = a.xhr + (int64_t)o.i * 3 * F;
    const float* gl = L.va + o.a * MD_V;
    float s1, s2;
    lnbwd_stats(gl, xh, a.lnr_w, 3 * F, s1, s2);
    const float gI = (gl[f] * a.lnr_w[f] - s1 - xh[f] * s2) * rs;
    const float gA = (gl[F + f] * a.lnr_w[F + f] - s1 - xh[F + f] * s2) * rs;
    const float gS = (gl[2 * F + f] * a.lnr_w[2 * F + f] - s1 - xh[2 * F + f] * s2) * rs;
    float dq[9];
    dquad(xn, dq);
    float G[9] = {dq[0] * gI, dq[1] * gA, dq[2] * gA, dq[3] * gA, dq[4] * gS, dq[5] * gS, dq[6] * gS, dq[7] * gS, dq[8] * gS};
    st9(a.G + row9, F, G);
    update_bwd(G, d, kap, gD);
  }
  float gch[9];
  tlin_multi<KQ, A>(a.Vb, gD, o, L, b, gch);
  if (o.valid) {
    float y[9], m[9], gm[9], gy[9];
    ld9(a.Pn + row9, F, y);
    ld9(a.Mi + row9, F, m);
    group_product_bwd(gch, y, m, a.o3, kap, gm, gy);
    st9(a.gMi + row9, F, gm);
    st9(a.gPn + row9, F, gy);
  }
}

// =====================================================================================================================
template <int F, int A>
__global__ __launch_bounds__(SM_G * F) void k_mid_rev(SmallRevArgs a) {
  constexpr int KQ = F / SM_G;
  __shared__ MidLds<A> L;
  const Graph& g = a.g;
  if (g.counts[2]) return;
  const int i0 = blockIdx.x * A;
  Blk b{F, (int)threadIdx.x / F, (int)threadIdx.x % F, (int)threadIdx.x, SM_G * F};
  const int f = b.f;
  const Own<A> o{b.g, i0 + b.g, b.g < A, b.g < A && i0 + b.g < a.N};
  const int64_t row9 = (int64_t)o.i * 9 * F + f;
  const float kap = (a.kap && o.valid) ? a.kap[o.i] : 1.0f;

  // per-molecule energies (reference output_modules.py:43-73): fixed order, one wave per molecule
  if (a.energy && b.tid < 64) {
    for (int m = blockIdx.x; m < a.B; m += gridDim.x) {
      float s = 0.f;
      if (!g.counts[3]) {
        for (int n = g.mstart[m] + b.tid; n < g.mend[m]; n += 64) s += a.ea[n];
      } else {
        for (int n = b.tid; n < a.N; n += 64)
          if (a.batch[n] == m) s += a.ea[n];
      }
      s = wave_sum(s);
      if (b.tid == 0) a.energy[m] = s + a.mean;
    }
  }

  sweep_multi<2, A>(g, i0, a.N, a.w, a.dw, a.gMi_in, a.Pn, a.slots, a.slot_stride, L.part, b);
  __syncthreads();
  float gp[9];
  if (o.is) {
    float s[9];
    own_sum<9, A>(L.part, o, b, s);
    if (o.valid) {
      ld9(a.gPn + row9, F, gp);
#pragma unroll
      for (int c = 0; c < 9; ++c) gp[c] += s[c];
    }
  }
  __syncthreads();  // part is free again
  float gxl[9], gx[9];
  tlin_multi<KQ, A>(a.Va, gp, o, L, b, gxl);
  // normalisation adjoint with the residual stream's G (k_norm_bwd)
  if (o.valid) {
    float u[9], dq[9];
    ld9(a.X + row9, F, u);
    ld9(a.G + row9, F, gx);
    dquad(u, dq);
    const float inv = 1.0f / (quad(u) + 1.0f);
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      gx[c] += gxl[c];
      dot += gx[c] * u[c];
    }
    const float g_s = -dot * inv * inv;
#pragma unroll
    for (int c = 0; c < 9; ++c) gx[c] = gx[c] * inv + dq[c] * g_s;
    st9(a.G + row9, F, gx);
  }
  if (!a.first) {
    // the layer below: update adjoint, second linear^T, group-product adjoint
    float gD[9], gch[9];
    if (o.valid) {
      float d[9];
      ld9(a.D_prev + row9, F, d);
      update_bwd(gx, d, kap, gD);
    }
    tlin_multi<KQ, A>(a.Vb_prev, gD, o, L, b, gch);
    if (o.valid) {
      float yp[9], mp[9], gm[9], gy[9];
      ld9(a.Pn_prev + row9, F, yp);
      ld9(a.Mi_prev + row9, F, mp);
      group_product_bwd(gch, yp, mp, a.o3, kap, gm, gy);
      st9(a.gMi_out + row9, F, gm);
      st9(a.gPn + row9, F, gy);
    }
    return;
  }
  // ---- layer 0: embedding gate adjoint (k_embed_gate_bwd), gate MLP^T, LayerNorm adjoint, tensor linear^T, embedding adjoint
  float gux[9];
  if (o.is) {
    float acc[3] = {0.f, 0.f, 0.f};
    if (o.valid) {
      float ux[9];
      ld9(a.UX + row9, F, ux);
      const float* gt = a.gates + (int64_t)o.i * 3 * F + f;
      const float g3[3] = {gt[0], gt[F], gt[2 * F]};
#pragma unroll
      for (int c = 0; c < 9; ++c) {
        gux[c] = gx[c] * g3[type_of(c)];
        acc[type_of(c)] += gx[c] * ux[c];
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) acc[k] *= silu_grad(a.a2[(int64_t)o.i * 3 * F + k * F + f]);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) L.va[o.a * MD_V + k * F + f] = acc[k];  // g_a2
  }
  __syncthreads();
  matvec_multi<A>(a.L2, nullptr, L.va, 3 * F, 2 * F, L.part, L.vb, b);
  for (int idx = b.tid; idx < A * 2 * F; idx += b.T) {
    const int sl = idx / (2 * F), n = idx - sl * 2 * F;
    if (i0 + sl < a.N) L.vb[sl * MD_V + n] *= silu_grad(a.a1[(int64_t)(i0 + sl) * 2 * F + n]);  // g_a1
  }
  __syncthreads();
  matvec_multi<A>(a.L1, nullptr, L.vb, 2 * F, F, L.part, L.va, b);  // va[.][0..F) = g_ln0
  float gs0 = 0.f;
  if (o.valid) {
    float s1, s2;
    lnbwd_stats(L.va + o.a * MD_V, a.xh0 + (int64_t)o.i * F, a.ln0_w, F, s1, s2);
    gs0 = (L.va[o.a * MD_V + f] * a.ln0_w[f] - s1 - a.xh0[(int64_t)o.i * F + f] * s2) * a.rstd0[o.i];
  }
  float gl[9];
  tlin_multi<KQ, A>(a.Ue, gux, o, L, b, gl);
  if (o.valid) {
    float u[9], dq[9];
    ld9(a.u0 + row9, F, u);
    dquad(u, dq);
#pragma unroll
    for (int c = 0; c < 9; ++c) gl[c] += dq[c] * gs0;
    const float third = (gl[4] + gl[7]) * (1.0f / 3.0f);
    float* op = a.gA + (int64_t)o.i * 10 * F + f;
    op[0] = gl[0];
    op[F] = gl[1];
    op[2 * F] = gl[2];
    op[3 * F] = gl[3];
    op[4 * F] = gl[4] - third;  // T00
    op[5 * F] = gl[5];          // T01
    op[6 * F] = gl[6];          // T02
    op[7 * F] = gl[7] - third;  // T11
    op[8 * F] = gl[8];          // T12
    op[9 * F] = -third;         // T22
  }
}

// ---------------------------------------------------------------------------------------------------------------------
constexpr int MID_A = 4;
bool mid_fused_ok(int N, int F, int H, int L) {
  static const int max_atoms = [] {
    // one round of blocks on 256 CUs; beyond it the general schedule is as fast (measured, replayed step, on / off: 768 atoms
    // 0.316 / 0.411 ms, 1 024: 0.357 / 0.483, 1 280: 0.544 / 0.529, 2 048: 0.630 / 0.649)
    const char* e = getenv("TMDNET_MID_FUSED_MAX");  // developer switch: 0 turns the four-atoms-per-block kernels off
    return e ? atoi(e) : 1024;
  }();
  return N > 0 && N <= max_atoms && L >= 1 && (F == 64 || F == 128) && H >= 4 && H % 4 == 0 && H <= 3 * F;
}
void launch_mid_embed(const SmallEmbedArgs& a, hipStream_t s) {
  const dim3 grid((a.N + MID_A - 1) / MID_A);
  if (a.F == 128) hipLaunchKernelGGL((k_mid_embed<128, MID_A>), grid, dim3(SM_G * 128), 0, s, a);
  else hipLaunchKernelGGL((k_mid_embed<64, MID_A>), grid, dim3(SM_G * 64), 0, s, a);
}
void launch_mid_layer(const SmallLayerArgs& a, bool last, hipStream_t s) {
  const dim3 grid((a.N + MID_A - 1) / MID_A);
  if (a.F == 128) {
    if (last) hipLaunchKernelGGL((k_mid_layer<128, MID_A, 1>), grid, dim3(SM_G * 128), 0, s, a);
    else hipLaunchKernelGGL((k_mid_layer<128, MID_A, 0>), grid, dim3(SM_G * 128), 0, s, a);
  } else {
    if (last) hipLaunchKernelGGL((k_mid_layer<64, MID_A, 1>), grid, dim3(SM_G * 64), 0, s, a);
    else hipLaunchKernelGGL((k_mid_layer<64, MID_A, 0>), grid, dim3(SM_G * 64), 0, s, a);
  }
}
void launch_mid_rev(const SmallRevArgs& a, hipStream_t s) {
  const dim3 grid((a.N + MID_A - 1) / MID_A);
  if (a.F == 128) hipLaunchKernelGGL((k_mid_rev<128, MID_A>), grid, dim3(SM_G * 128), 0, s, a);
  else hipLaunchKernelGGL((k_mid_rev<64, MID_A>), grid, dim3(SM_G * 64), 0, s, a);
}

}  // namespace tn
