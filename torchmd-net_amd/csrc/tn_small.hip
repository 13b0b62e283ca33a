// Fused per-atom phase kernels for small systems (reference tensornet.py:405-445 embedding, 729-814 interaction, 384-398 readout,
// output_modules.py:43-117 head, and their adjoints), gfx950.
//
// A 64-atom molecule in MD stepping - and a rank's shard under strong scaling - is bound by launch-to-launch latency: the step
// was ~45 dependent launches of a few microseconds each (0.233 ms replayed as one HIP graph).  Between two neighbour sweeps
// every operation of the model is PER ATOM (LayerNorm, the gate MLP, the nine-component tensor linears, the 3x3 algebra, the
// readout MLP and its adjoint), so the node side of the step is cut at the sweeps only:
//
//   k_small_embed   embedding scatter, LayerNorm, gate MLP, tensor linear x gates -> X0, normalise, linear -> Pn[0]
//   k_small_layer   message sweep, group product + normalisation, linear, update -> X[l+1]; then the next layer's normalise +
//                   linear, or (last layer) readout LayerNorm + MLP + head -> atom energy AND the reverse pass back down to
//                   the inputs of this layer's adjoint sweep (head, MLP, LayerNorm, invariants, update, linear^T, group product)
//   k_small_rev     adjoint sweep with the distance-gradient slots, linear^T, normalisation adjoint; then the layer below's
//                   update adjoint + linear^T + group-product adjoint, or (layer 0) gate adjoint, gate MLP^T, LayerNorm adjoint,
//                   linear^T, embedding adjoint -> gA
//
// 1 + L + L launches for the node side of an energy + force step with L layers (35 before), 10 for the whole step with the
// graph kernel, the table interpolation, the embedding's pair kernel and the force gather.
//
// One block per atom, G = 8 thread groups of F threads (thread = (group g, channel f)).  Sweeps: group g takes every 8th edge
// of the row (k_message_split).  Products: the reduction index is split over the groups - a thread multiplies its output
// channel by F/8 (tensor linear) or K/kg (MLP) input channels, weights read in [in][out] order (coalesced over f) from L2, the
// partial sums meet in LDS and are added in a fixed order.  fp32 FMA throughout: at this size the matrix cores have nothing to
// amortise their fill over (a tensor linear of one atom is 9 x 128 x 128 multiply-adds).  Per-channel 3x3 algebra and the
// LayerNorm statistics run on group 0 / redundantly per wave between the block barriers.  Deterministic.
#include <cstdlib>

#include "tn_common.h"
#include "tn_small.h"
#include "tn_small_dev.h"

namespace tn {

namespace {

struct __attribute__((aligned(16))) SmallLds {
  float part[SM_G * 10 * SM_FMAX];  // partial sums [group][component][channel] (sweeps: 9 or 10 components) / [k-group][output]
  float xs[10 * SM_FMAX];           // input of a tensor linear [component][channel]; reduced partial sums
  float va[3 * SM_FMAX], vb[3 * SM_FMAX], vc[3 * SM_FMAX];  // vectors of the MLP chains
};

// sum of the groups' partial sums, every thread of the block calls (one barrier inside): group g adds up component g (groups 0
// and 1 also components 8 and 9) into xs, group 0 then picks up its NC values.  Group 0 alone reading all 8 NC partials kept 72
// LDS results in registers next to its 3x3 algebra (30 registers spilled at the 128-register limit of a 1024-thread block).
template <int NC>
__device__ __forceinline__ void part_reduce(SmallLds& L, const Blk& b, float out[NC]) {
  const int F = b.F;
  for (int c = b.g; c < NC; c += SM_G) {
    float s = 0.f;
#pragma unroll
    for (int gg = 0; gg < SM_G; ++gg) s += L.part[(gg * NC + c) * F + b.f];
    L.xs[c * F + b.f] = s;
  }
  __syncthreads();
  if (b.g == 0) {
#pragma unroll
    for (int c = 0; c < NC; ++c) out[c] = L.xs[c * F + b.f];
  }
}
// xs <- u (group 0), barrier, partial products, barrier, group 0: out = sum.  Leaves xs / part free after the caller's next barrier.
template <int KQ>
__device__ __forceinline__ void tlin_finish(const TlinW<KQ>& t, const float u[9], SmallLds& L, const Blk& b, float out[9]) {
  const int F = b.F, k0 = b.g * KQ;
  if (b.g == 0) {
#pragma unroll
    for (int c = 0; c < 9; ++c) L.xs[c * F + b.f] = u[c];
  }
  __syncthreads();
  float acc[9];
  tlin_fma<KQ>(t, L.xs, F, k0, acc);
#pragma unroll
  for (int c = 0; c < 9; ++c) L.part[(b.g * 9 + c) * F + b.f] = acc[c];
  __syncthreads();
  part_reduce<9>(L, b, out);
}

// the same with the weights requested only now (after the barrier: the compiler must not move 3 KQ loads into the caller's algebra)
template <int KQ>
__device__ __forceinline__ void tlin_now(const float* const W[3], const float u[9], SmallLds& L, const Blk& b, float out[9]) {
  const int F = b.F, k0 = b.g * KQ;
  if (b.g == 0) {
#pragma unroll
    for (int c = 0; c < 9; ++c) L.xs[c * F + b.f] = u[c];
  }
  __syncthreads();
  TlinW<KQ> t;
  tlin_issue<KQ>(W, b, t);
  float acc[9];
  tlin_fma<KQ>(t, L.xs, F, k0, acc);
#pragma unroll
  for (int c = 0; c < 9; ++c) L.part[(b.g * 9 + c) * F + b.f] = acc[c];
  __syncthreads();
  part_reduce<9>(L, b, out);
}

// ---- y[n] = bias[n] + sum_k WT[k, n] x[k]  (x, y in LDS; every thread of the block calls; y is visible on return).  Nout % 4 == 0.
// A thread owns four consecutive outputs (one 16-byte weight load per input channel) and a slice of the input channels; up to
// sixteen slices, their sums added in a fixed order.  Eight loads in flight per thread.
template <bool PRE>
__device__ __forceinline__ void matvec_(const float* __restrict__ WT, const float* __restrict__ bias, const float* x, int K, int Nout,
                                        float* part, float* y, const Blk& b, const MvW* pre) {
  int c4, q, k0, k1;
  const int kg = mv_slice(K, Nout, b, c4, q, k0, k1);  // slices in use
  if (q < kg) {
    f4 acc[1] = {(f4)(0.f)};
    int k = k0;
    if (PRE) {  // the first eight rows were requested ahead
      mv_fma4<1>(pre->w, x, 0, k0, acc);
      if (k0 + 4 < k1) mv_fma4<1>(pre->w + 4, x, 0, k0 + 4, acc);
      k = min(k1, k0 + 8);
    }
#pragma unroll 1
    for (; k + 8 <= k1; k += 8) {  // one chunk of eight requests at a time (unrolled, the chunks' loads pile up in registers)
      f4 wv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) wv[u] = *reinterpret_cast<const f4*>(WT + (int64_t)(k + u) * Nout + 4 * c4);
      mv_fma4<1>(wv, x, 0, k, acc);
      mv_fma4<1>(wv + 4, x, 0, k + 4, acc);
    }
    if (k < k1) {  // one more piece of four
      f4 wv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) wv[u] = *reinterpret_cast<const f4*>(WT + (int64_t)(k + u) * Nout + 4 * c4);
      mv_fma4<1>(wv, x, 0, k, acc);
    }
    *reinterpret_cast<f4*>(part + q * Nout + 4 * c4) = acc[0];
  }
  __syncthreads();
  if (b.tid < Nout) {
    float s = bias ? bias[b.tid] : 0.f;
    for (int qq = 0; qq < kg; ++qq) s += part[qq * Nout + b.tid];
    y[b.tid] = s;
  }
  __syncthreads();
}
__device__ __forceinline__ void matvec(const float* __restrict__ WT, const float* __restrict__ bias, const float* x, int K, int Nout,
                                       float* part, float* y, const Blk& b) {
  matvec_<false>(WT, bias, x, K, Nout, part, y, b, nullptr);
}
__device__ __forceinline__ void matvec(const float* __restrict__ WT, const float* __restrict__ bias, const float* x, int K, int Nout,
                                       float* part, float* y, const Blk& b, const MvW& pre) {
  matvec_<true>(WT, bias, x, K, Nout, part, y, b, &pre);
}

// message sweep of row i by the block's groups (k_message_split): acc = sum_e w[pair, type(c)] * src[col, c]; MODE 2 also the
// distance-gradient halves h(i <- j) = sum_f dw * src[j] . y per edge -> slots.  The partial sums go to part; group 0 adds them.
template <int MODE>
__device__ __forceinline__ void sweep(const Graph& g, int i, const float* __restrict__ w, const float* __restrict__ dw,
                                      const float* __restrict__ src, const float y[9], float* __restrict__ slots, int64_t slot_stride,
                                      float* part, const Blk& b) {
  const int F = b.F, F3 = 3 * F, F9 = 9 * F, f = b.f;
  const int e0 = g.rowptr[i], e1 = g.rowptr[i + 1];
  float acc[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) acc[c] = 0.f;
  // two edges of the group per trip (independent loads); the (column, pair, sign) records of the NEXT trip are requested before
  // this trip's rows are multiplied, so the index fetch is off the chain
  auto rec = [&](int e, int& j, int& p, float& sg) __attribute__((always_inline)) {
    const bool ok = e < e1;
    j = ok ? g.col[e] : 0;
    p = ok ? g.epair[e] : 0;
    sg = ok && MODE == 2 ? g.esign[e] : 0.f;
  };
  int jA, pA, jB, pB;
  float sA, sB;
  int e = e0 + b.g;
  rec(e, jA, pA, sA);
  rec(e + SM_G, jB, pB, sB);
  for (; e < e1; e += 2 * SM_G) {
    const bool two = e + SM_G < e1;
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
    rec(e + 2 * SM_G, jA, pA, sA);
    rec(e + 3 * SM_G, jB, pB, sB);
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
#pragma unroll
  for (int c = 0; c < 9; ++c) part[(b.g * 9 + c) * F + f] = acc[c];
}

}  // namespace

// =====================================================================================================================
template <int F>
__global__ __launch_bounds__(SM_G * F) void k_small_embed(SmallEmbedArgs a) {
  constexpr int KQ = F / SM_G;
  __shared__ SmallLds L;
  const Graph& g = a.g;
  if (g.counts[2]) return;  // pair overflow: the adjacency was not filled (the host reports the error)
  const int i = blockIdx.x, F3 = 3 * F;
  Blk b{F, (int)threadIdx.x / F, (int)threadIdx.x % F, (int)threadIdx.x, SM_G * F};
  const int f = b.f;
  MvW mw;
  mv_issue(a.L1T, F, 2 * F, b, mw);  // the gate MLP's first weights travel while the row is swept

  // ---- embedding scatter (k_embed_scatter_split): I0 = sum W0 ; v = sum W1 r ; T = sum W2 r r^T
  {
    const int e0 = g.rowptr[i], e1 = g.rowptr[i + 1];
    const int64_t zi = a.z[i];
    const float Ui = a.Utab[zi * F + f];
    float s[10];
#pragma unroll
    for (int c = 0; c < 10; ++c) s[c] = 0.f;
    for (int e = e0 + b.g; e < e1; e += SM_G) {
      const int j = g.col[e], p = g.epair[e];
      const float sg = g.esign[e];
      float rx = 0.f, ry = 0.f, rz = 0.f;
      if (sg != 0.f) {
        rx = sg * g.prhat[p * 3];
        ry = sg * g.prhat[p * 3 + 1];
        rz = sg * g.prhat[p * 3 + 2];
      }
      const float cz = a.C[p] * (Ui + a.Vtab[a.z[j] * F + f]);
      const float* qq = a.Q + (int64_t)p * F3 + f;
      const float W0 = cz * qq[0], W1 = cz * qq[F], W2 = cz * qq[2 * F];
      s[0] += W0;
      s[1] += W1 * rx; s[2] += W1 * ry; s[3] += W1 * rz;
      s[4] += W2 * rx * rx; s[5] += W2 * rx * ry; s[6] += W2 * rx * rz;
      s[7] += W2 * ry * ry; s[8] += W2 * ry * rz; s[9] += W2 * rz * rz;
    }
#pragma unroll
    for (int c = 0; c < 10; ++c) L.part[(b.g * 10 + c) * F + f] = s[c];
  }
  __syncthreads();
  float u[9];
  float s10[10];
  part_reduce<10>(L, b, s10);
  if (b.g == 0) {
    const float* s = s10;
    const float tr3 = (s[4] + s[7] + s[9]) * (1.0f / 3.0f);
    u[0] = s[0]; u[1] = s[1]; u[2] = s[2]; u[3] = s[3];
    u[4] = s[4] - tr3; u[5] = s[5]; u[6] = s[6]; u[7] = s[7] - tr3; u[8] = s[8];
    st9(a.u0 + (int64_t)i * 9 * F + f, F, u);
    L.va[f] = quad(u);
  }
  __syncthreads();
  // ---- LayerNorm of the invariant, gate MLP (reference tensornet.py:595-611)
  {
    float mean, rs;
    row_stats(L.va, F, mean, rs);
    if (b.g == 0) {
      const float xh = (L.va[f] - mean) * rs;
      a.xh0[(int64_t)i * F + f] = xh;
      L.vb[f] = xh * a.ln0_w[f] + a.ln0_b[f];
      if (f == 0) a.rstd0[i] = rs;
    }
  }
  __syncthreads();
  matvec(a.L1T, a.bL1, L.vb, F, 2 * F, L.part, L.va, b, mw);
  if (b.tid < 2 * F) {
    const float v = L.va[b.tid];
    a.a1[(int64_t)i * 2 * F + b.tid] = v;
    L.va[b.tid] = silu(v);
  }
  __syncthreads();
  matvec(a.L2T, a.bL2, L.va, 2 * F, 3 * F, L.part, L.vb, b);
  if (b.tid < 3 * F) {
    const float v = L.vb[b.tid];
    a.a2[(int64_t)i * 3 * F + b.tid] = v;
    const float gt = silu(v);
    a.gates[(int64_t)i * 3 * F + b.tid] = gt;
    L.vb[b.tid] = gt;
  }
  // ---- tensor linear, gates -> X0 (the barrier inside tlin publishes vb as well)
  float ux[9];
  TlinW<KQ> tw;
  tlin_now<KQ>(a.UeT, u, L, b, ux);
  if (a.L > 0) tlin_issue<KQ>(a.V0T, b, tw);  // requested before group 0 turns to its 3x3 algebra
  float x0[9];
  if (b.g == 0) {
    st9(a.UX + (int64_t)i * 9 * F + f, F, ux);
#pragma unroll
    for (int c = 0; c < 9; ++c) x0[c] = ux[c] * L.vb[type_of(c) * F + f];
    st9(a.X0 + (int64_t)i * 9 * F + f, F, x0);
  }
  if (a.L == 0) return;
  // ---- layer 0: X_hat = X / (||X||^2 + 1), first linear -> Pn[0]
  if (b.g == 0) {
    const float inv = 1.0f / (quad(x0) + 1.0f);
#pragma unroll
    for (int c = 0; c < 9; ++c) x0[c] *= inv;
  }
  __syncthreads();  // part was read by group 0 just now
  float pn[9];
  tlin_finish<KQ>(tw, x0, L, b, pn);
  if (b.g == 0) st9(a.Pn0 + (int64_t)i * 9 * F + f, F, pn);
}

// =====================================================================================================================
template <int F, int LAST>
__global__ __launch_bounds__(SM_G * F) void k_small_layer(SmallLayerArgs a) {
  constexpr int KQ = F / SM_G;
  __shared__ SmallLds L;
  const Graph& g = a.g;
  if (g.counts[2]) return;
  const int i = blockIdx.x;
  Blk b{F, (int)threadIdx.x / F, (int)threadIdx.x % F, (int)threadIdx.x, SM_G * F};
  const int f = b.f;
  const int64_t row9 = (int64_t)i * 9 * F + f;
  const float kap = a.kap ? a.kap[i] : 1.0f;
  const float none[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  TlinW<KQ> tw;
  if (!LAST) tlin_issue<KQ>(a.VbT, b, tw);  // the second linear's weights travel while the row is swept (last layer: the kernel is
                                            // at its register limit with the readout chain; requested where they are used)

  sweep<0>(g, i, a.w, nullptr, a.Pn, none, nullptr, 0, L.part, b);
  __syncthreads();
  float ch[9];
  float m[9];
  part_reduce<9>(L, b, m);
  if (b.g == 0) {
    float y[9];
    st9(a.Mi + row9, F, m);
    ld9(a.Pn + row9, F, y);
    group_product(y, m, a.o3, kap, ch);
  }
  __syncthreads();
  float d[9];
  if (LAST) tlin_now<KQ>(a.VbT, ch, L, b, d);
  else tlin_finish<KQ>(tw, ch, L, b, d);
  if (!LAST) tlin_issue<KQ>(a.VnT, b, tw);
  float xn[9];
  if (b.g == 0) {
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
    float xh[9];
    if (b.g == 0) {
      const float inv = 1.0f / (quad(xn) + 1.0f);
#pragma unroll
      for (int c = 0; c < 9; ++c) xh[c] = xn[c] * inv;
    }
    __syncthreads();
    float pn[9];
    tlin_finish<KQ>(tw, xh, L, b, pn);
    if (b.g == 0) st9(a.Pn_next + row9, F, pn);
    return;
  }
  // ---- readout: invariants, LayerNorm, MLP, head (reference tensornet.py:384-398, output_modules.py:43-73)
  if (b.g == 0) {
    const float t = xn[4] + xn[7];
    L.va[f] = 3.0f * xn[0] * xn[0];
    L.va[F + f] = 2.0f * (xn[1] * xn[1] + xn[2] * xn[2] + xn[3] * xn[3]);
    L.va[2 * F + f] = xn[4] * xn[4] + xn[7] * xn[7] + t * t + 2.0f * (xn[5] * xn[5] + xn[6] * xn[6] + xn[8] * xn[8]);
  }
  __syncthreads();
  {
    float mean, rs;
    row_stats(L.va, 3 * F, mean, rs);
    if (b.tid < 3 * F) {
      const float xh = (L.va[b.tid] - mean) * rs;
      a.xhr[(int64_t)i * 3 * F + b.tid] = xh;
      L.vb[b.tid] = xh * a.lnr_w[b.tid] + a.lnr_b[b.tid];
    }
    if (b.tid == 0) a.rstdr[i] = rs;
    __syncthreads();
    matvec(a.LinT, a.bLin, L.vb, 3 * F, F, L.part, L.vc, b);  // vc[0..F) = al
    if (b.tid < F) {
      const float v = L.vc[b.tid];
      a.al[(int64_t)i * F + b.tid] = v;
      const float xv = silu(v);
      a.x[(int64_t)i * F + b.tid] = xv;
      L.va[b.tid] = xv;
    }
    __syncthreads();
    matvec(a.O1T, a.bO1, L.va, F, a.H, L.part, L.vb, b);  // vb[0..H) = ao
    const float wgt = a.atom_w ? a.atom_w[a.perm ? a.perm[i] : i] : 1.0f;  // weight of this atom in the energy sum
    if (b.tid < 64) {
      float s = 0.f;
      for (int k = b.tid; k < a.H; k += 64) s += silu(L.vb[k]) * a.O2[k];
      s = wave_sum(s);
      if (b.tid == 0) {
        float e = (s + a.bO2[0]) * a.std_;
        if (a.atomref) e += a.atomref[a.z[i]];
        a.ea[i] = wgt * e;
      }
    }
    if (!a.want_forces) return;
    // ---- reverse: head, MLP, LayerNorm, invariants -> G = d E / d X[L]
    __syncthreads();
    if (b.tid < a.H) L.va[b.tid] = wgt * a.std_ * a.O2[b.tid] * silu_grad(L.vb[b.tid]);  // g_ao
    __syncthreads();
    matvec(a.O1, nullptr, L.va, a.H, F, L.part, L.vb, b);  // vb[0..F) = g_ao O1
    if (b.tid < F) L.vb[b.tid] *= silu_grad(L.vc[b.tid]);  // g_al
    __syncthreads();
    matvec(a.Lin, nullptr, L.vb, F, 3 * F, L.part, L.va, b);  // va[0..3F) = g_ln
    float s1, s2;
    lnbwd_stats(L.va, a.xhr + (int64_t)i * 3 * F, a.lnr_w, 3 * F, s1, s2);
    float gD[9];
    if (b.g == 0) {
      const float* xh = a.xhr + (int64_t)i * 3 * F;
      const float gI = (L.va[f] * a.lnr_w[f] - s1 - xh[f] * s2) * rs;
      const float gA = (L.va[F + f] * a.lnr_w[F + f] - s1 - xh[F + f] * s2) * rs;
      const float gS = (L.va[2 * F + f] * a.lnr_w[2 * F + f] - s1 - xh[2 * F + f] * s2) * rs;
      float dq[9];
      dquad(xn, dq);
      float G[9] = {dq[0] * gI, dq[1] * gA, dq[2] * gA, dq[3] * gA, dq[4] * gS, dq[5] * gS, dq[6] * gS, dq[7] * gS, dq[8] * gS};
      st9(a.G + row9, F, G);
      update_bwd(G, d, kap, gD);
    }
    __syncthreads();
    float gch[9];
    tlin_now<KQ>(a.Vb, gD, L, b, gch);
    if (b.g == 0) {
      float y[9], m[9], gm[9], gy[9];  // reloaded (this thread stored m above): keeping them live across the readout spilled
      ld9(a.Pn + row9, F, y);
      ld9(a.Mi + row9, F, m);
      group_product_bwd(gch, y, m, a.o3, kap, gm, gy);
      st9(a.gMi + row9, F, gm);
      st9(a.gPn + row9, F, gy);
    }
  }
}

// =====================================================================================================================
template <int F>
__global__ __launch_bounds__(SM_G * F) void k_small_rev(SmallRevArgs a) {
  constexpr int KQ = F / SM_G;
  __shared__ SmallLds L;
  const Graph& g = a.g;
  if (g.counts[2]) return;
  const int i = blockIdx.x;
  Blk b{F, (int)threadIdx.x / F, (int)threadIdx.x % F, (int)threadIdx.x, SM_G * F};
  const int f = b.f;
  const int64_t row9 = (int64_t)i * 9 * F + f;
  const float kap = a.kap ? a.kap[i] : 1.0f;

  // per-molecule energies (reference output_modules.py:43-73): fixed order, one wave per molecule
  if (a.energy && i < a.B && b.tid < 64) {
    float s = 0.f;
    if (!g.counts[3]) {
      for (int n = g.mstart[i] + b.tid; n < g.mend[i]; n += 64) s += a.ea[n];
    } else {
      for (int n = b.tid; n < a.N; n += 64)
        if (a.batch[n] == i) s += a.ea[n];
    }
    s = wave_sum(s);
    if (b.tid == 0) a.energy[i] = s + a.mean;
  }

  float y[9];
  ld9(a.Pn + row9, F, y);
  sweep<2>(g, i, a.w, a.dw, a.gMi_in, y, a.slots, a.slot_stride, L.part, b);
  __syncthreads();
  float gp[9];
  float s[9];
  part_reduce<9>(L, b, s);
  if (b.g == 0) {
    ld9(a.gPn + row9, F, gp);
#pragma unroll
    for (int c = 0; c < 9; ++c) gp[c] += s[c];
  }
  __syncthreads();
  float gxl[9];
  tlin_now<KQ>(a.Va, gp, L, b, gxl);
  // normalisation adjoint with the residual stream's G (k_norm_bwd)
  float gx[9];
  if (b.g == 0) {
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
  }
  if (b.g == 0) st9(a.G + row9, F, gx);
  if (!a.first) {
    // the layer below: update adjoint, second linear^T, group-product adjoint
    float gD[9], d[9];
    if (b.g == 0) {
      ld9(a.D_prev + row9, F, d);
      update_bwd(gx, d, kap, gD);
    }
    __syncthreads();
    float gch[9];
    tlin_now<KQ>(a.Vb_prev, gD, L, b, gch);
    if (b.g == 0) {
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
  if (b.g == 0) {
    float ux[9];
    ld9(a.UX + row9, F, ux);
    const float* gt = a.gates + (int64_t)i * 3 * F + f;
    const float g3[3] = {gt[0], gt[F], gt[2 * F]};
    float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      gux[c] = gx[c] * g3[type_of(c)];
      acc[type_of(c)] += gx[c] * ux[c];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) L.va[k * F + f] = acc[k] * silu_grad(a.a2[(int64_t)i * 3 * F + k * F + f]);  // g_a2
  }
  __syncthreads();
  matvec(a.L2, nullptr, L.va, 3 * F, 2 * F, L.part, L.vb, b);
  if (b.tid < 2 * F) L.vb[b.tid] *= silu_grad(a.a1[(int64_t)i * 2 * F + b.tid]);  // g_a1
  __syncthreads();
  matvec(a.L1, nullptr, L.vb, 2 * F, F, L.part, L.va, b);  // va[0..F) = g_ln0
  float s1, s2;
  lnbwd_stats(L.va, a.xh0 + (int64_t)i * F, a.ln0_w, F, s1, s2);
  float gs0 = 0.f;
  if (b.g == 0) gs0 = (L.va[f] * a.ln0_w[f] - s1 - a.xh0[(int64_t)i * F + f] * s2) * a.rstd0[i];
  __syncthreads();
  float gl[9];
  tlin_now<KQ>(a.Ue, gux, L, b, gl);
  if (b.g == 0) {
    float u[9], dq[9];
    ld9(a.u0 + row9, F, u);
    dquad(u, dq);
#pragma unroll
    for (int c = 0; c < 9; ++c) gl[c] += dq[c] * gs0;
    const float third = (gl[4] + gl[7]) * (1.0f / 3.0f);
    float* o = a.gA + (int64_t)i * 10 * F + f;
    o[0] = gl[0];
    o[F] = gl[1];
    o[2 * F] = gl[2];
    o[3 * F] = gl[3];
    o[4 * F] = gl[4] - third;  // T00
    o[5 * F] = gl[5];          // T01
    o[6 * F] = gl[6];          // T02
    o[7 * F] = gl[7] - third;  // T11
    o[8 * F] = gl[8];          // T12
    o[9 * F] = -third;         // T22
  }
}

// ---------------------------------------------------------------------------------------------------------------------
bool small_fused_ok(int N, int F, int H, int L) {
  static const int max_atoms = [] {
    const char* e = getenv("TMDNET_SMALL_FUSED_MAX");  // developer switch: 0 turns the fused small-system kernels off
    return e ? atoi(e) : 512;
  }();
  return N > 0 && N <= max_atoms && L >= 1 && (F == 64 || F == 128) && H >= 4 && H % 4 == 0 && H <= 3 * F;
}
void launch_small_embed(const SmallEmbedArgs& a, hipStream_t s) {
  if (a.F == 128) hipLaunchKernelGGL((k_small_embed<128>), dim3(a.N), dim3(SM_G * 128), 0, s, a);
  else hipLaunchKernelGGL((k_small_embed<64>), dim3(a.N), dim3(SM_G * 64), 0, s, a);
}
void launch_small_layer(const SmallLayerArgs& a, bool last, hipStream_t s) {
  if (a.F == 128) {
    if (last) hipLaunchKernelGGL((k_small_layer<128, 1>), dim3(a.N), dim3(SM_G * 128), 0, s, a);
    else hipLaunchKernelGGL((k_small_layer<128, 0>), dim3(a.N), dim3(SM_G * 128), 0, s, a);
  } else {
    if (last) hipLaunchKernelGGL((k_small_layer<64, 1>), dim3(a.N), dim3(SM_G * 64), 0, s, a);
    else hipLaunchKernelGGL((k_small_layer<64, 0>), dim3(a.N), dim3(SM_G * 64), 0, s, a);
  }
}
void launch_small_rev(const SmallRevArgs& a, hipStream_t s) {
  if (a.F == 128) hipLaunchKernelGGL((k_small_rev<128>), dim3(a.N), dim3(SM_G * 128), 0, s, a);
  else hipLaunchKernelGGL((k_small_rev<64>), dim3(a.N), dim3(SM_G * 64), 0, s, a);
}

}  // namespace tn
