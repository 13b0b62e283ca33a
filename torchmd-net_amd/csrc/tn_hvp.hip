// Kernels of the analytic second-order pass of force-matching training on TensorNet (gfx950):
//     d/d theta of  s = v . d(sum_m E_m)/d pos          (reference: autograd twice, model.py:618-628; warp_ops/*_bwd_bwd)
// The arithmetic of every kernel is one function of tn_hvp_math.h (one logical thread = one (atom, channel), pair-row element or
// LayerNorm row), specified by oracle/tensornet_second_order.py and checked against it on the host (tests/test_hvp_host.py);
// here is only the index arithmetic: consecutive lanes take consecutive channels (coalesced rows of the [N][9][F] tensors,
// wave-uniform neighbour loops for F >= 64).  This pass belongs to the training path, not to the inference step: it is written to
// be exact and simple - one launch per statement group of the specification - not tuned.
#include "tn_hvp.h"

#include "tn_common.h"
#include "tn_hvp_math.h"

namespace tn {
namespace hvp {

namespace {
constexpr int TB = 256;
inline dim3 grid_for(int64_t n) { return dim3((unsigned)((n + TB - 1) / TB)); }
struct RedWave {  // the row's lanes = one wave (all 64 active: the row index is wave-uniform)
  __device__ float operator()(float v) const { return wave_sum(v); }
};
#define NF_INDEX                                                           \
  const int64_t idx_ = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     \
  if (idx_ >= (int64_t)N * F) return;                                      \
  const int n = (int)(idx_ / F), f = (int)(idx_ - (int64_t)n * F);

__global__ __launch_bounds__(TB) void k_pair_tangent(Graph g, int P, int K, const float* __restrict__ v, const float* __restrict__ dphi,
                               const float* __restrict__ dC, float* __restrict__ d_t, float* __restrict__ rhat_t, float* __restrict__ phi_t,
                               float* __restrict__ C_t) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p > P) return;
  pair_tangent(p, P, K, g.pair_i, g.pair_j, g.prhat, g.pd, v, dphi, dC, d_t, rhat_t, phi_t, C_t);
}
__global__ __launch_bounds__(TB) void k_embed_scatter_dual(Graph g, int N, int F, int P, const int64_t* __restrict__ z, const float* __restrict__ Utab,
                                     const float* __restrict__ Vtab, const float* __restrict__ Q, const float* __restrict__ Q_t,
                                     const float* __restrict__ C, const float* __restrict__ C_t, const float* __restrict__ rhat_t,
                                     float* __restrict__ u0, float* __restrict__ u0_t, float* __restrict__ s0n, float* __restrict__ s0n_t) {
  NF_INDEX
  embed_scatter_dual(n, f, F, P, g.rowptr, g.col, g.epair, g.esign, z, Utab, Vtab, Q, Q_t, C, C_t, g.prhat, rhat_t, u0, u0_t, s0n, s0n_t);
}
__global__ __launch_bounds__(TB) void k_ln_dual(int R, int W, const float* __restrict__ x, const float* __restrict__ x_t, const float* __restrict__ w,
                          const float* __restrict__ b, float* __restrict__ y, float* __restrict__ xh, float* __restrict__ rstd,
                          float* __restrict__ y_t, float* __restrict__ xh_t, float* __restrict__ rstd_t) {
  const int r = blockIdx.x * (TB / 64) + (threadIdx.x >> 6);  // a wave per row
  if (r < R) ln_dual_lanes(r, W, (int)(threadIdx.x & 63), 64, RedWave{}, x, x_t, w, b, y, xh, rstd, y_t, xh_t, rstd_t);
}
__global__ __launch_bounds__(TB) void k_lnbwd_dual(int R, int W, const float* __restrict__ g, const float* __restrict__ g_t, const float* __restrict__ xh,
                             const float* __restrict__ xh_t, const float* __restrict__ rstd, const float* __restrict__ rstd_t,
                             const float* __restrict__ w, float* __restrict__ o, float* __restrict__ o_t) {
  const int r = blockIdx.x * (TB / 64) + (threadIdx.x >> 6);  // a wave per row
  if (r < R) lnbwd_dual_lanes(r, W, (int)(threadIdx.x & 63), 64, RedWave{}, g, g_t, xh, xh_t, rstd, rstd_t, w, o, o_t);
}
__global__ __launch_bounds__(TB) void k_silu_tangent(int64_t n, const float* __restrict__ a, const float* __restrict__ a_t, float* __restrict__ h_t) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) silu_tangent(i, a, a_t, h_t);
}
__global__ __launch_bounds__(TB) void k_dsilu_dual(int64_t n, const float* __restrict__ g, const float* __restrict__ g_t, const float* __restrict__ a,
                             const float* __restrict__ a_t, float* __restrict__ o, float* __restrict__ o_t) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dsilu_dual(i, g, g_t, a, a_t, o, o_t);
}
__global__ __launch_bounds__(TB) void k_gate_mul_dual(int N, int F, const float* __restrict__ UX, const float* __restrict__ UX_t, const float* __restrict__ gates,
                                const float* __restrict__ gates_t, float* __restrict__ X, float* __restrict__ X_t) {
  NF_INDEX
  gate_mul_dual(n, f, F, UX, UX_t, gates, gates_t, X, X_t);
}
__global__ __launch_bounds__(TB) void k_w_dual(int64_t total, int F3, const float* __restrict__ e3, const float* __restrict__ e3_t, const float* __restrict__ C,
                         const float* __restrict__ C_t, float* __restrict__ w, float* __restrict__ w_t) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) w_dual(i, F3, e3, e3_t, C, C_t, w, w_t);
}
__global__ __launch_bounds__(TB) void k_norm_dual(int N, int F, const float* __restrict__ X, const float* __restrict__ X_t, float* __restrict__ Xh,
                            float* __restrict__ Xh_t) {
  NF_INDEX
  norm_dual(n, f, F, X, X_t, Xh, Xh_t);
}
__global__ __launch_bounds__(TB) void k_group_dual(int N, int F, const float* __restrict__ Pn, const float* __restrict__ Pn_t, const float* __restrict__ Mi,
                             const float* __restrict__ Mi_t, const float* __restrict__ kap, int o3, float* __restrict__ Ch,
                             float* __restrict__ Ch_t) {
  NF_INDEX
  group_dual(n, f, F, Pn, Pn_t, Mi, Mi_t, kap, o3, Ch, Ch_t);
}
__global__ __launch_bounds__(TB) void k_update_dual(int N, int F, const float* __restrict__ Xh, const float* __restrict__ Xh_t, const float* __restrict__ D,
                              const float* __restrict__ D_t, const float* __restrict__ kap, float* __restrict__ Xn,
                              float* __restrict__ Xn_t) {
  NF_INDEX
  update_dual(n, f, F, Xh, Xh_t, D, D_t, kap, Xn, Xn_t);
}
__global__ __launch_bounds__(TB) void k_feat_dual(int N, int F, const float* __restrict__ X, const float* __restrict__ X_t, float* __restrict__ feat,
                            float* __restrict__ feat_t) {
  NF_INDEX
  feat_dual(n, f, F, X, X_t, feat, feat_t);
}
__global__ __launch_bounds__(TB) void k_head_dual(int64_t total, int H, const float* __restrict__ ao, const float* __restrict__ ao_t, const float* __restrict__ O2,
                            float std_, const float* __restrict__ ge, const int64_t* __restrict__ batch, float* __restrict__ g_ao,
                            float* __restrict__ g_ao_t, float* __restrict__ headv) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) head_dual(i, H, ao, ao_t, O2, std_, ge, batch, g_ao, g_ao_t, headv);
}
// out[0] = - std sum_n ge[molecule(n)]: ONE block, each thread a strided partial sum, the 256 partials added in order by thread 0
__global__ __launch_bounds__(TB) void k_head_bias_seed(int N, float std_, const float* __restrict__ ge, const int64_t* __restrict__ batch,
                                                       float* __restrict__ out) {
  __shared__ float part[TB];
  float a = 0.f;
  for (int n = threadIdx.x; n < N; n += TB) a += head_bias_seed_term(n, ge, batch);
  part[threadIdx.x] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int k = 0; k < TB; ++k) t += part[k];
    out[0] = -std_ * t;
  }
}
__global__ __launch_bounds__(TB) void k_row_seed(int64_t total, int W, const float* __restrict__ ge, const int64_t* __restrict__ batch,
                                                 const float* __restrict__ x, float* __restrict__ x_t) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) row_seed(i, W, ge, batch, x, x_t);
}
__global__ __launch_bounds__(TB) void k_readout_bwd_dual(int N, int F, const float* __restrict__ X, const float* __restrict__ X_t,
                                   const float* __restrict__ g_feat, const float* __restrict__ g_feat_t, float* __restrict__ G,
                                   float* __restrict__ G_t) {
  NF_INDEX
  readout_bwd_dual(n, f, F, X, X_t, g_feat, g_feat_t, G, G_t);
}
__global__ __launch_bounds__(TB) void k_update_bwd_dual(int N, int F, const float* __restrict__ G, const float* __restrict__ G_t, const float* __restrict__ D,
                                  const float* __restrict__ D_t, const float* __restrict__ kap, float* __restrict__ g_D,
                                  float* __restrict__ g_D_t) {
  NF_INDEX
  update_bwd_dual(n, f, F, G, G_t, D, D_t, kap, g_D, g_D_t);
}
__global__ __launch_bounds__(TB) void k_group_bwd_dual(int N, int F, const float* __restrict__ g_Ch, const float* __restrict__ g_Ch_t,
                                 const float* __restrict__ Pn, const float* __restrict__ Pn_t, const float* __restrict__ Mi,
                                 const float* __restrict__ Mi_t, const float* __restrict__ kap, int o3, float* __restrict__ g_Mi,
                                 float* __restrict__ g_Mi_t, float* __restrict__ g_PnY, float* __restrict__ g_PnY_t) {
  NF_INDEX
  group_bwd_dual(n, f, F, g_Ch, g_Ch_t, Pn, Pn_t, Mi, Mi_t, kap, o3, g_Mi, g_Mi_t, g_PnY, g_PnY_t);
}
__global__ __launch_bounds__(TB) void k_pair_gw_dual(Graph g, int P, int F, const float* __restrict__ g_Mi, const float* __restrict__ g_Mi_t,
                               const float* __restrict__ Pn, const float* __restrict__ Pn_t, const float* __restrict__ self_gw,
                               const float* __restrict__ self_gw_t, const float* __restrict__ e3, const float* __restrict__ e3_t,
                               const float* __restrict__ C, const float* __restrict__ C_t, float* __restrict__ g_e3,
                               float* __restrict__ g_e3_t, float* __restrict__ gcp, float* __restrict__ gcp_t) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (int64_t)(P + 1) * F)
    pair_gw_dual(i, P, F, g.pair_i, g.pair_j, g_Mi, g_Mi_t, Pn, Pn_t, self_gw, self_gw_t, e3, e3_t, C, C_t, g_e3, g_e3_t, gcp, gcp_t);
}
__global__ __launch_bounds__(TB) void k_norm_bwd_dual(int N, int F, const float* __restrict__ X, const float* __restrict__ X_t, const float* __restrict__ G,
                                const float* __restrict__ G_t, const float* __restrict__ gL, const float* __restrict__ gL_t,
                                float* __restrict__ Gn, float* __restrict__ Gn_t) {
  NF_INDEX
  norm_bwd_dual(n, f, F, X, X_t, G, G_t, gL, gL_t, Gn, Gn_t);
}
__global__ __launch_bounds__(TB) void k_gate_bwd_dual(int N, int F, const float* __restrict__ G, const float* __restrict__ G_t, const float* __restrict__ UX,
                                const float* __restrict__ UX_t, const float* __restrict__ gates, const float* __restrict__ gates_t,
                                const float* __restrict__ a2, const float* __restrict__ a2_t, float* __restrict__ g_UX,
                                float* __restrict__ g_UX_t, float* __restrict__ g_a2, float* __restrict__ g_a2_t) {
  NF_INDEX
  gate_bwd_dual(n, f, F, G, G_t, UX, UX_t, gates, gates_t, a2, a2_t, g_UX, g_UX_t, g_a2, g_a2_t);
}
__global__ __launch_bounds__(TB) void k_embed_bwd_atom_dual(int N, int F, const float* __restrict__ gL, const float* __restrict__ gL_t,
                                      const float* __restrict__ u0, const float* __restrict__ u0_t, const float* __restrict__ g_s0n,
                                      const float* __restrict__ g_s0n_t, float* __restrict__ gA, float* __restrict__ gA_t) {
  NF_INDEX
  embed_bwd_atom_dual(n, f, F, gL, gL_t, u0, u0_t, g_s0n, g_s0n_t, gA, gA_t);
}
__global__ __launch_bounds__(TB) void k_embed_edge_dual(Graph g, int N, int F, int P, const int64_t* __restrict__ z, const float* __restrict__ Utab,
                                  const float* __restrict__ Vtab, const float* __restrict__ Q, const float* __restrict__ Q_t,
                                  const float* __restrict__ C, const float* __restrict__ C_t, const float* __restrict__ rhat_t,
                                  const float* __restrict__ gA, const float* __restrict__ gA_t, float* __restrict__ gq,
                                  float* __restrict__ gq_t, int64_t dir_stride, float* __restrict__ selfq, float* __restrict__ selfq_t,
                                  float* __restrict__ gZu_t, float* __restrict__ gZv_t) {
  NF_INDEX
  embed_edge_dual(n, f, F, P, g.rowptr, g.col, g.epair, g.esign, z, Utab, Vtab, Q, Q_t, C, C_t, g.prhat, rhat_t, gA, gA_t, gq, gq_t,
                  dir_stride, selfq, selfq_t, gZu_t, gZv_t);
}
__global__ __launch_bounds__(TB) void k_radial2(Graph g, int P, int K, const float* __restrict__ means, const float* __restrict__ betas,
                                                float lo, float up, float* __restrict__ d2phi, float* __restrict__ d2C) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (int64_t)(P + 1) * K) radial2(i, P, K, g.pd, means, betas, lo, up, d2phi, d2C);
}
// the two row kernels of the position gradient spread the channels of a row over a WAVE (coalesced rows, one wave_sum per result)
// where the bodies of tn_hvp_math.h walk them in one thread - same summands, another (fixed) order
__global__ __launch_bounds__(TB) void k_pair_rowdot(int rows, int W, const float* __restrict__ x, const float* __restrict__ x_t,
                                                    const float* __restrict__ y, const float* __restrict__ y2,
                                                    const float* __restrict__ d_t, int accumulate, float* __restrict__ out,
                                                    float* __restrict__ out_t) {
  const int p = blockIdx.x * (TB / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (p >= rows) return;  // wave-uniform
  const int64_t b = (int64_t)p * W;
  float s0 = 0.f, st = 0.f, s2 = 0.f;
  for (int j = lane; j < W; j += 64) {
    const float yy = y ? y[b + j] : 1.0f;
    s0 += x[b + j] * yy;
    st += x_t[b + j] * yy;
    if (y2) s2 += x[b + j] * y2[b + j];
  }
  s0 = wave_sum(s0);
  st = wave_sum(st);
  s2 = wave_sum(s2);
  if (lane == 0) {
    if (y2) st += s2 * d_t[p];
    out[p] = (accumulate ? out[p] : 0.f) + s0;
    out_t[p] = (accumulate ? out_t[p] : 0.f) + st;
  }
}
__global__ __launch_bounds__(TB) void k_edge_geom_dual(Graph g, int E, int N, int F, int P, const int64_t* __restrict__ z,
                                                       const float* __restrict__ Utab, const float* __restrict__ Vtab,
                                                       const float* __restrict__ Q, const float* __restrict__ Q_t,
                                                       const float* __restrict__ C, const float* __restrict__ C_t,
                                                       const float* __restrict__ rhat_t, const float* __restrict__ gA,
                                                       const float* __restrict__ gA_t, float* __restrict__ ec, float* __restrict__ ec_t,
                                                       int64_t dir_stride) {
  const int e = blockIdx.x * (TB / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (e >= E) return;  // wave-uniform, as everything up to the channel loop
  const float sg = g.esign[e];
  const int p = g.epair[e];
  if (sg == 0.f || p >= P) return;
  const int i = edge_geom_row(e, N, g.rowptr), j = g.col[e];
  float r[3], rt[3], acc[4] = {0.f, 0.f, 0.f, 0.f}, acc_t[4] = {0.f, 0.f, 0.f, 0.f};
  for (int a = 0; a < 3; ++a) {
    r[a] = sg * g.prhat[p * 3 + a];
    rt[a] = sg * rhat_t[p * 3 + a];
  }
  const float c = C[p], ct = C_t[p];
  for (int f = lane; f < F; f += 64) edge_geom_term(i, j, p, f, F, r, rt, c, ct, z, Utab, Vtab, Q, Q_t, gA, gA_t, acc, acc_t);
  for (int x = 0; x < 4; ++x) {
    acc[x] = wave_sum(acc[x]);
    acc_t[x] = wave_sum(acc_t[x]);
  }
  if (lane == 0) {
    const int64_t o = (sg > 0.f ? 0 : dir_stride) + (int64_t)p * 4;
    for (int x = 0; x < 4; ++x) {
      ec[o + x] = acc[x];
      ec_t[o + x] = acc_t[x];
    }
  }
}
// the same per (edge, channel) arithmetic with a WAVE PER ROW ATOM: the atom's gA / gA_t (20 values per channel) stay in registers while
// the wave walks the row's edges - the wave-per-edge kernel above re-read them for every edge (13 KB per edge instead of 3: 0.94 ms
// at the bench batch).  The next edge's records and rows are requested before the current edge is reduced.  F <= 64 NCH.
template <int NCH>
__global__ __launch_bounds__(TB) void k_edge_geom_rows(Graph g, int N, int F, int P, const int64_t* __restrict__ z,
                                                       const float* __restrict__ Utab, const float* __restrict__ Vtab,
                                                       const float* __restrict__ Q, const float* __restrict__ Q_t,
                                                       const float* __restrict__ C, const float* __restrict__ C_t,
                                                       const float* __restrict__ rhat_t, const float* __restrict__ gA,
                                                       const float* __restrict__ gA_t, float* __restrict__ ec, float* __restrict__ ec_t,
                                                       int64_t dir_stride) {
  const int i = blockIdx.x * (TB / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= N) return;  // wave-uniform
  const int F3 = 3 * F, F10 = 10 * F;
  const int e0 = g.rowptr[i], e1 = g.rowptr[i + 1];
  float a[NCH][10], at[NCH][10], ui[NCH];
  const int64_t zi = z[i];
#pragma unroll
  for (int u = 0; u < NCH; ++u) {
    const int f = min(lane + 64 * u, F - 1);  // lanes past F: a valid channel, masked below
    ui[u] = Utab[zi * F + f];
#pragma unroll
    for (int cc = 0; cc < 10; ++cc) {
      a[u][cc] = gA[(int64_t)i * F10 + cc * F + f];
      at[u][cc] = gA_t[(int64_t)i * F10 + cc * F + f];
    }
  }
  struct Rec {
    float sg, c, ct, r[3], rt[3], q[NCH][3], qt[NCH][3], vj[NCH];
    int p;
  };
  auto load = [&](int e, Rec& o) {
    e = min(e, e1 - 1);
    o.sg = g.esign[e];
    o.p = g.epair[e];
    const int pc = min(o.p, P);  // (row P exists: the self pair's)
    const int64_t zj = z[g.col[e]];
    o.c = C[pc];
    o.ct = C_t[pc];
#pragma unroll
    for (int x = 0; x < 3; ++x) {
      o.r[x] = g.prhat[pc * 3 + x];
      o.rt[x] = rhat_t[pc * 3 + x];
    }
#pragma unroll
    for (int u = 0; u < NCH; ++u) {
      const int f = min(lane + 64 * u, F - 1);
      o.vj[u] = Vtab[zj * F + f];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        o.q[u][k] = Q[(int64_t)pc * F3 + k * F + f];
        o.qt[u][k] = Q_t[(int64_t)pc * F3 + k * F + f];
      }
    }
  };
  if (e0 >= e1) return;
  Rec cur, nxt;
  load(e0, cur);
  for (int e = e0; e < e1; ++e) {
    load(e + 1, nxt);
    if (cur.sg != 0.f && cur.p < P) {  // wave-uniform
      float r[3], rt[3], acc[4] = {0.f, 0.f, 0.f, 0.f}, acc_t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int x = 0; x < 3; ++x) {
        r[x] = cur.sg * cur.r[x];
        rt[x] = cur.sg * cur.rt[x];
      }
#pragma unroll
      for (int u = 0; u < NCH; ++u)
        if (lane + 64 * u < F) edge_geom_core(a[u], at[u], ui[u] + cur.vj[u], cur.q[u], cur.qt[u], r, rt, cur.c, cur.ct, acc, acc_t);
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        acc[x] = wave_sum(acc[x]);
        acc_t[x] = wave_sum(acc_t[x]);
      }
      if (lane == 0) {
        const int64_t o = (cur.sg > 0.f ? 0 : dir_stride) + (int64_t)cur.p * 4;
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          ec[o + x] = acc[x];
          ec_t[o + x] = acc_t[x];
        }
      }
    }
    cur = nxt;
  }
}
__global__ __launch_bounds__(TB) void k_geom_dual(Graph g, int P, const float* __restrict__ d_t, const float* __restrict__ rhat_t,
                                                  const float* __restrict__ dC, const float* __restrict__ d2C, const float* __restrict__ gC,
                                                  const float* __restrict__ gC_t, const float* __restrict__ gphid,
                                                  const float* __restrict__ gphid_t, const float* __restrict__ ec,
                                                  const float* __restrict__ ec_t, int64_t dir_stride, float* __restrict__ gdel,
                                                  float* __restrict__ gdel_t) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < P) geom_dual(p, P, g.pd, g.prhat, d_t, rhat_t, dC, d2C, gC, gC_t, gphid, gphid_t, ec, ec_t, dir_stride, gdel, gdel_t);
}
__global__ __launch_bounds__(TB) void k_pair_to_atom(Graph g, int N, int P, const float* __restrict__ gp, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) pair_to_atom(i, P, g.rowptr, g.epair, g.esign, gp, out);
}
}  // namespace

#define LAUNCH(kernel, count, ...)                                                            \
  do {                                                                                        \
    if ((count) > 0) hipLaunchKernelGGL(kernel, grid_for(count), dim3(TB), 0, s, __VA_ARGS__); \
  } while (0)

void launch_pair_tangent(const Graph& g, int P, int K, const float* v, const float* dphi, const float* dC, float* d_t, float* rhat_t,
                         float* phi_t, float* C_t, hipStream_t s) {
  LAUNCH(k_pair_tangent, (int64_t)P + 1, g, P, K, v, dphi, dC, d_t, rhat_t, phi_t, C_t);
}
void launch_embed_scatter_dual(const Graph& g, int N, int F, int P, const int64_t* z, const float* Utab, const float* Vtab, const float* Q,
                               const float* Q_t, const float* C, const float* C_t, const float* rhat_t, float* u0, float* u0_t, float* s0n,
                               float* s0n_t, hipStream_t s) {
  LAUNCH(k_embed_scatter_dual, (int64_t)N * F, g, N, F, P, z, Utab, Vtab, Q, Q_t, C, C_t, rhat_t, u0, u0_t, s0n, s0n_t);
}
void launch_ln_dual(int R, int W, const float* x, const float* x_t, const float* w, const float* b, float* y, float* xh, float* rstd,
                    float* y_t, float* xh_t, float* rstd_t, hipStream_t s) {
  LAUNCH(k_ln_dual, (int64_t)R * 64, R, W, x, x_t, w, b, y, xh, rstd, y_t, xh_t, rstd_t);
}
void launch_lnbwd_dual(int R, int W, const float* g, const float* g_t, const float* xh, const float* xh_t, const float* rstd,
                       const float* rstd_t, const float* w, float* o, float* o_t, hipStream_t s) {
  LAUNCH(k_lnbwd_dual, (int64_t)R * 64, R, W, g, g_t, xh, xh_t, rstd, rstd_t, w, o, o_t);
}
void launch_silu_tangent(int64_t n, const float* a, const float* a_t, float* h_t, hipStream_t s) { LAUNCH(k_silu_tangent, n, n, a, a_t, h_t); }
void launch_dsilu_dual(int64_t n, const float* g, const float* g_t, const float* a, const float* a_t, float* o, float* o_t, hipStream_t s) {
  LAUNCH(k_dsilu_dual, n, n, g, g_t, a, a_t, o, o_t);
}
void launch_gate_mul_dual(int N, int F, const float* UX, const float* UX_t, const float* gates, const float* gates_t, float* X, float* X_t,
                          hipStream_t s) {
  LAUNCH(k_gate_mul_dual, (int64_t)N * F, N, F, UX, UX_t, gates, gates_t, X, X_t);
}
void launch_w_dual(int64_t rows, int F3, const float* e3, const float* e3_t, const float* C, const float* C_t, float* w, float* w_t,
                   hipStream_t s) {
  LAUNCH(k_w_dual, rows * F3, rows * F3, F3, e3, e3_t, C, C_t, w, w_t);
}
void launch_norm_dual(int N, int F, const float* X, const float* X_t, float* Xh, float* Xh_t, hipStream_t s) {
  LAUNCH(k_norm_dual, (int64_t)N * F, N, F, X, X_t, Xh, Xh_t);
}
void launch_group_dual(int N, int F, const float* Pn, const float* Pn_t, const float* Mi, const float* Mi_t, const float* kap, int o3,
                       float* Ch, float* Ch_t, hipStream_t s) {
  LAUNCH(k_group_dual, (int64_t)N * F, N, F, Pn, Pn_t, Mi, Mi_t, kap, o3, Ch, Ch_t);
}
void launch_update_dual(int N, int F, const float* Xh, const float* Xh_t, const float* D, const float* D_t, const float* kap, float* Xn,
                        float* Xn_t, hipStream_t s) {
  LAUNCH(k_update_dual, (int64_t)N * F, N, F, Xh, Xh_t, D, D_t, kap, Xn, Xn_t);
}
void launch_feat_dual(int N, int F, const float* X, const float* X_t, float* feat, float* feat_t, hipStream_t s) {
  LAUNCH(k_feat_dual, (int64_t)N * F, N, F, X, X_t, feat, feat_t);
}
void launch_head_dual(int N, int H, const float* ao, const float* ao_t, const float* O2, float std_, const float* ge, const int64_t* batch,
                      float* g_ao, float* g_ao_t, float* headv, hipStream_t s) {
  LAUNCH(k_head_dual, (int64_t)N * H, (int64_t)N * H, H, ao, ao_t, O2, std_, ge, batch, g_ao, g_ao_t, headv);
}
void launch_head_bias_seed(int N, float std_, const float* ge, const int64_t* batch, float* out, hipStream_t s) {
  hipLaunchKernelGGL(k_head_bias_seed, dim3(1), dim3(TB), 0, s, N, std_, ge, batch, out);
}
void launch_row_seed(int N, int W, const float* ge, const int64_t* batch, const float* x, float* x_t, hipStream_t s) {
  LAUNCH(k_row_seed, (int64_t)N * W, (int64_t)N * W, W, ge, batch, x, x_t);
}
void launch_readout_bwd_dual(int N, int F, const float* X, const float* X_t, const float* g_feat, const float* g_feat_t, float* G,
                             float* G_t, hipStream_t s) {
  LAUNCH(k_readout_bwd_dual, (int64_t)N * F, N, F, X, X_t, g_feat, g_feat_t, G, G_t);
}
void launch_update_bwd_dual(int N, int F, const float* G, const float* G_t, const float* D, const float* D_t, const float* kap, float* g_D,
                            float* g_D_t, hipStream_t s) {
  LAUNCH(k_update_bwd_dual, (int64_t)N * F, N, F, G, G_t, D, D_t, kap, g_D, g_D_t);
}
void launch_group_bwd_dual(int N, int F, const float* g_Ch, const float* g_Ch_t, const float* Pn, const float* Pn_t, const float* Mi,
                           const float* Mi_t, const float* kap, int o3, float* g_Mi, float* g_Mi_t, float* g_PnY, float* g_PnY_t,
                           hipStream_t s) {
  LAUNCH(k_group_bwd_dual, (int64_t)N * F, N, F, g_Ch, g_Ch_t, Pn, Pn_t, Mi, Mi_t, kap, o3, g_Mi, g_Mi_t, g_PnY, g_PnY_t);
}
void launch_pair_gw_dual(const Graph& g, int P, int F, const float* g_Mi, const float* g_Mi_t, const float* Pn, const float* Pn_t,
                         const float* self_gw, const float* self_gw_t, const float* e3, const float* e3_t, const float* C, const float* C_t,
                         float* g_e3, float* g_e3_t, float* gcp, float* gcp_t, hipStream_t s) {
  LAUNCH(k_pair_gw_dual, (int64_t)(P + 1) * F, g, P, F, g_Mi, g_Mi_t, Pn, Pn_t, self_gw, self_gw_t, e3, e3_t, C, C_t, g_e3, g_e3_t, gcp,
         gcp_t);
}
void launch_norm_bwd_dual(int N, int F, const float* X, const float* X_t, const float* G, const float* G_t, const float* gL,
                          const float* gL_t, float* Gn, float* Gn_t, hipStream_t s) {
  LAUNCH(k_norm_bwd_dual, (int64_t)N * F, N, F, X, X_t, G, G_t, gL, gL_t, Gn, Gn_t);
}
void launch_gate_bwd_dual(int N, int F, const float* G, const float* G_t, const float* UX, const float* UX_t, const float* gates,
                          const float* gates_t, const float* a2, const float* a2_t, float* g_UX, float* g_UX_t, float* g_a2, float* g_a2_t,
                          hipStream_t s) {
  LAUNCH(k_gate_bwd_dual, (int64_t)N * F, N, F, G, G_t, UX, UX_t, gates, gates_t, a2, a2_t, g_UX, g_UX_t, g_a2, g_a2_t);
}
void launch_embed_bwd_atom_dual(int N, int F, const float* gL, const float* gL_t, const float* u0, const float* u0_t, const float* g_s0n,
                                const float* g_s0n_t, float* gA, float* gA_t, hipStream_t s) {
  LAUNCH(k_embed_bwd_atom_dual, (int64_t)N * F, N, F, gL, gL_t, u0, u0_t, g_s0n, g_s0n_t, gA, gA_t);
}
void launch_embed_edge_dual(const Graph& g, int N, int F, int P, const int64_t* z, const float* Utab, const float* Vtab, const float* Q,
                            const float* Q_t, const float* C, const float* C_t, const float* rhat_t, const float* gA, const float* gA_t,
                            float* gq, float* gq_t, int64_t dir_stride, float* selfq, float* selfq_t, float* gZu_t, float* gZv_t,
                            hipStream_t s) {
  LAUNCH(k_embed_edge_dual, (int64_t)N * F, g, N, F, P, z, Utab, Vtab, Q, Q_t, C, C_t, rhat_t, gA, gA_t, gq, gq_t, dir_stride, selfq,
         selfq_t, gZu_t, gZv_t);
}

void launch_radial2(const Graph& g, int P, int K, const float* means, const float* betas, float lo, float up, float* d2phi, float* d2C,
                    hipStream_t s) {
  LAUNCH(k_radial2, (int64_t)(P + 1) * K, g, P, K, means, betas, lo, up, d2phi, d2C);
}
void launch_pair_rowdot(int rows, int W, const float* x, const float* x_t, const float* y, const float* y2, const float* d_t, bool accumulate,
                        float* out, float* out_t, hipStream_t s) {
  LAUNCH(k_pair_rowdot, (int64_t)rows * 64, rows, W, x, x_t, y, y2, d_t, accumulate ? 1 : 0, out, out_t);  // a wave per row
}
void launch_edge_geom_dual(const Graph& g, int E, int N, int F, int P, const int64_t* z, const float* Utab, const float* Vtab, const float* Q,
                           const float* Q_t, const float* C, const float* C_t, const float* rhat_t, const float* gA, const float* gA_t,
                           float* ec, float* ec_t, int64_t dir_stride, hipStream_t s) {
  if (F <= 64) LAUNCH((k_edge_geom_rows<1>), (int64_t)N * 64, g, N, F, P, z, Utab, Vtab, Q, Q_t, C, C_t, rhat_t, gA, gA_t, ec, ec_t, dir_stride);  // a wave per row atom
  else if (F <= 128) LAUNCH((k_edge_geom_rows<2>), (int64_t)N * 64, g, N, F, P, z, Utab, Vtab, Q, Q_t, C, C_t, rhat_t, gA, gA_t, ec, ec_t, dir_stride);
  else LAUNCH(k_edge_geom_dual, (int64_t)E * 64, g, E, N, F, P, z, Utab, Vtab, Q, Q_t, C, C_t, rhat_t, gA, gA_t, ec, ec_t, dir_stride);  // a wave per edge
}
void launch_geom_dual(const Graph& g, int P, const float* d_t, const float* rhat_t, const float* dC, const float* d2C, const float* gC,
                      const float* gC_t, const float* gphid, const float* gphid_t, const float* ec, const float* ec_t, int64_t dir_stride,
                      float* gdel, float* gdel_t, hipStream_t s) {
  LAUNCH(k_geom_dual, (int64_t)P, g, P, d_t, rhat_t, dC, d2C, gC, gC_t, gphid, gphid_t, ec, ec_t, dir_stride, gdel, gdel_t);
}
void launch_pair_to_atom(const Graph& g, int N, int P, const float* gp, float* out, hipStream_t s) {
  LAUNCH(k_pair_to_atom, (int64_t)N, g, N, P, gp, out);
}

}  // namespace hvp
}  // namespace tn
