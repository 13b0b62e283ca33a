// Kernels of the analytic second-order pass that only TensorNet2 + ScalarPlusWeightedCoulomb needs (gfx950): index arithmetic around
// the bodies of tn_tn2_hvp_math.h, as tn_hvp.hip is around tn_hvp_math.h.  Specification oracle/tn2_second_order.py; the same bodies
// run on the host in tests/tn2_hvp_host_mirror.py.  Training path, exact and simple, not tuned.
#include "tn_hvp.h"

#include "tn_common.h"
#include "tn_tn2_hvp_math.h"

namespace tn {
namespace hvp {

namespace {
constexpr int TB = 256;
inline dim3 grid_for(int64_t n) { return dim3((unsigned)((n + TB - 1) / TB)); }
#define IDX2(ROWS, W)                                                   \
  const int64_t idx_ = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  \
  if (idx_ >= (int64_t)(ROWS) * (W)) return;                            \
  const int n = (int)(idx_ / (W)), f = (int)(idx_ - (int64_t)n * (W));
#define IDX1(TOTAL)                                                  \
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  \
  if (i >= (int64_t)(TOTAL)) return;

__global__ __launch_bounds__(TB) void k_cp_feat_dual(int N, int F, const float* __restrict__ X, const float* __restrict__ X_t,
                                                     float* __restrict__ feat, float* __restrict__ feat_t) {
  IDX2(N, F)
  cp_feat_dual(n, f, F, X, X_t, feat, feat_t);
}
__global__ __launch_bounds__(TB) void k_cp_feat_bwd_dual(int N, int F, const float* __restrict__ X, const float* __restrict__ X_t,
                                                         const float* __restrict__ g_feat, const float* __restrict__ g_feat_t,
                                                         float* __restrict__ G, float* __restrict__ G_t) {
  IDX2(N, F)
  cp_feat_bwd_dual(n, f, F, X, X_t, g_feat, g_feat_t, G, G_t);
}
__global__ __launch_bounds__(TB) void k_cp_mol_sums(Graph g, int B, int qd, const float* __restrict__ out, const float* __restrict__ out_t,
                                                    float* __restrict__ sums) {
  IDX2(B, qd)
  cp_mol_sums(n, f, qd, g.mstart, g.mend, out, out_t, sums);
}
__global__ __launch_bounds__(TB) void k_cp_qeq_dual(int N, int qd, const int64_t* __restrict__ batch, const float* __restrict__ Qmol,
                                                    const float* __restrict__ out, const float* __restrict__ out_t,
                                                    const float* __restrict__ sums, float* __restrict__ ch, float* __restrict__ ch_t, int ldc,
                                                    int off) {
  IDX2(N, qd)
  cp_qeq_dual(n, f, qd, batch, Qmol, out, out_t, sums, ch, ch_t, ldc, off);
}
__global__ __launch_bounds__(TB) void k_cp_mol_sums_bwd(Graph g, int B, int qd, const int64_t* __restrict__ batch,
                                                        const float* __restrict__ Qmol, const float* __restrict__ out,
                                                        const float* __restrict__ out_t, const float* __restrict__ sums,
                                                        const float* __restrict__ g_ch, const float* __restrict__ g_ch_t, int ldg, int off,
                                                        float* __restrict__ bs) {
  IDX2(B, qd)
  cp_mol_sums_bwd(n, f, qd, g.mstart, g.mend, batch, Qmol, out, out_t, sums, g_ch, g_ch_t, ldg, off, bs);
}
__global__ __launch_bounds__(TB) void k_cp_qeq_bwd_dual(int N, int qd, const int64_t* __restrict__ batch, const float* __restrict__ Qmol,
                                                        const float* __restrict__ out, const float* __restrict__ out_t,
                                                        const float* __restrict__ sums, const float* __restrict__ bs,
                                                        const float* __restrict__ g_ch, const float* __restrict__ g_ch_t, int ldg, int off,
                                                        float* __restrict__ g_out, float* __restrict__ g_out_t) {
  IDX2(N, qd)
  cp_qeq_bwd_dual(n, f, qd, batch, Qmol, out, out_t, sums, bs, g_ch, g_ch_t, ldg, off, g_out, g_out_t);
}
__global__ __launch_bounds__(TB) void k_tn2_edge_pre1_dual(Graph g, int E, int N, int F, const float* __restrict__ Ap,
                                                           const float* __restrict__ Ap_t, const float* __restrict__ Bt,
                                                           const float* __restrict__ Bt_t, const float* __restrict__ Cs,
                                                           const float* __restrict__ Cs_t, float* __restrict__ pre1, float* __restrict__ e1_t,
                                                           float* __restrict__ he1, float* __restrict__ he1_t) {
  IDX1((int64_t)E * F)
  tn2_edge_pre1_dual(i, N, F, g.rowptr, g.col, g.epair, Ap, Ap_t, Bt, Bt_t, Cs, Cs_t, pre1, e1_t, he1, he1_t);
}
__global__ __launch_bounds__(TB) void k_tn2_w_dual(Graph g, int64_t total, int F3, const float* __restrict__ e3,
                                                   const float* __restrict__ e3_t, const float* __restrict__ C, const float* __restrict__ C_t,
                                                   float* __restrict__ w, float* __restrict__ w_t) {
  IDX1(total)
  tn2_w_dual(i, F3, g.epair, e3, e3_t, C, C_t, w, w_t);
}
__global__ __launch_bounds__(TB) void k_edge_sweep2(Graph g, int N, int F, const int* __restrict__ emap, const float* __restrict__ wA,
                                                    const float* __restrict__ srcA, const float* __restrict__ wB,
                                                    const float* __restrict__ srcB, const float* __restrict__ init, float* __restrict__ out) {
  IDX2(N, F)
  edge_sweep2(n, f, F, g.rowptr, g.col, emap, wA, srcA, wB, srcB, init, out);
}
__global__ __launch_bounds__(TB) void k_tn2_edge_gw_dual(Graph g, int E, int N, int F, const float* __restrict__ g_Mi,
                                                         const float* __restrict__ g_Mi_t, const float* __restrict__ Pn,
                                                         const float* __restrict__ Pn_t, const float* __restrict__ e3,
                                                         const float* __restrict__ e3_t, const float* __restrict__ C,
                                                         const float* __restrict__ C_t, float* __restrict__ g_e3, float* __restrict__ g_e3_t,
                                                         float* __restrict__ gcp, float* __restrict__ gcp_t) {
  IDX1((int64_t)E * F)
  tn2_edge_gw_dual(i, N, F, g.rowptr, g.col, g.epair, g_Mi, g_Mi_t, Pn, Pn_t, e3, e3_t, C, C_t, g_e3, g_e3_t, gcp, gcp_t);
}
__global__ __launch_bounds__(TB) void k_tn2_edge_reduce_dual(Graph g, int N, int F, const int* __restrict__ erev, const float* __restrict__ g1,
                                                             const float* __restrict__ g1_t, float* __restrict__ gB, float* __restrict__ gB_t,
                                                             float* __restrict__ gCs, float* __restrict__ gCs_t, float* __restrict__ gself,
                                                             float* __restrict__ gself_t) {
  IDX2(N, F)
  tn2_edge_reduce_dual(n, f, F, g.rowptr, g.col, erev, g1, g1_t, gB, gB_t, gCs, gCs_t, gself, gself_t);
}
__global__ __launch_bounds__(TB) void k_tn2_pair_reduce_dual(int64_t total, int F, const int* __restrict__ pair_edge,
                                                             const int* __restrict__ erev, const float* __restrict__ g1,
                                                             const float* __restrict__ g1_t, float* __restrict__ gAp, float* __restrict__ gAp_t) {
  IDX1(total)
  tn2_pair_reduce_dual(i, F, pair_edge, erev, g1, g1_t, gAp, gAp_t);
}
__global__ __launch_bounds__(TB) void k_edge_rowdot(Graph g, int E, int W, int ldx, const float* __restrict__ x, const float* __restrict__ x_t,
                                                    const float* __restrict__ y, const float* __restrict__ y2, const float* __restrict__ d_t,
                                                    int accumulate, float* __restrict__ val, float* __restrict__ val_t) {
  // a wave per directed edge (edge_rowdot's sums, lanes strided over the row, as k_pair_rowdot in tn_hvp.hip)
  const int e = blockIdx.x * (TB / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (e >= E) return;  // wave-uniform
  const int p = g.epair[e];
  float s0 = 0.f, st = 0.f, s2 = 0.f;
  for (int j = lane; j < W; j += 64) {
    const float yy = y ? y[(int64_t)p * W + j] : 1.0f;
    s0 += x[(int64_t)e * ldx + j] * yy;
    st += x_t[(int64_t)e * ldx + j] * yy;
    if (y2) s2 += x[(int64_t)e * ldx + j] * y2[(int64_t)p * W + j];
  }
  s0 = wave_sum(s0);
  st = wave_sum(st);
  s2 = wave_sum(s2);
  if (lane == 0) {
    if (y2) st += s2 * d_t[p];
    val[e] = (accumulate ? val[e] : 0.f) + s0;
    val_t[e] = (accumulate ? val_t[e] : 0.f) + st;
  }
}
__global__ __launch_bounds__(TB) void k_pair_from_edges(int P, const int* __restrict__ pair_edge, const int* __restrict__ erev,
                                                        const float* __restrict__ val, const float* __restrict__ val_t, float* __restrict__ out,
                                                        float* __restrict__ out_t) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < P) pair_from_edges(p, pair_edge, erev, val, val_t, out, out_t);
}
__global__ __launch_bounds__(TB) void k_coulomb_atom_dual(Graph g, int N, int QC, const int64_t* __restrict__ batch,
                                                          const float* __restrict__ pos, const float* __restrict__ v,
                                                          const float* __restrict__ box, int box_per_mol, const float* __restrict__ ch,
                                                          const float* __restrict__ ch_t, const float* __restrict__ wq,
                                                          const float* __restrict__ wsum, float cut, float eps, float scale,
                                                          float* __restrict__ e_atom, float* __restrict__ e_atom_t, float* __restrict__ g_q,
                                                          float* __restrict__ g_q_t, float* __restrict__ g_pos, float* __restrict__ hv) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N)
    coulomb_atom_dual(i, QC, g.mstart, g.mend, batch, pos, v, box, box_per_mol, ch, ch_t, wq, wsum[0], cut, eps, scale, e_atom, e_atom_t, g_q,
                      g_q_t, g_pos, hv);
}
__global__ __launch_bounds__(TB) void k_vec_sum(int n, const float* __restrict__ x, float* __restrict__ out) {  // out[0] = sum x (n <= 64)
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    float s = 0.f;
    for (int k = 0; k < n; ++k) s += x[k];
    out[0] = s;
  }
}
__global__ __launch_bounds__(TB) void k_add_cols(int64_t total, int qd, const float* __restrict__ src, float* __restrict__ dst, int ld,
                                                 int off) {
  IDX1(total)
  add_cols(i, qd, src, dst, ld, off);
}
__global__ __launch_bounds__(TB) void k_axpy1(int64_t total, float a, const float* __restrict__ x, float* __restrict__ y) {
  IDX1(total)
  axpy1(i, a, x, y);
}
__global__ __launch_bounds__(TB) void k_scale1(int64_t total, float a, const float* __restrict__ x, float* __restrict__ y) {
  IDX1(total)
  scale1(i, a, x, y);
}
}  // namespace

#define LAUNCH(kernel, count, ...)                                                            \
  do {                                                                                        \
    if ((count) > 0) hipLaunchKernelGGL(kernel, grid_for(count), dim3(TB), 0, s, __VA_ARGS__); \
  } while (0)

void launch_cp_feat_dual(int N, int F, const float* X, const float* X_t, float* feat, float* feat_t, hipStream_t s) {
  LAUNCH(k_cp_feat_dual, (int64_t)N * F, N, F, X, X_t, feat, feat_t);
}
void launch_cp_feat_bwd_dual(int N, int F, const float* X, const float* X_t, const float* g_feat, const float* g_feat_t, float* G, float* G_t,
                             hipStream_t s) {
  LAUNCH(k_cp_feat_bwd_dual, (int64_t)N * F, N, F, X, X_t, g_feat, g_feat_t, G, G_t);
}
void launch_cp_qeq_dual(const Graph& g, int N, int B, int qd, const int64_t* batch, const float* Qmol, const float* out, const float* out_t,
                        float* sums, float* ch, float* ch_t, int ldc, int off, hipStream_t s) {
  LAUNCH(k_cp_mol_sums, (int64_t)B * qd, g, B, qd, out, out_t, sums);
  LAUNCH(k_cp_qeq_dual, (int64_t)N * qd, N, qd, batch, Qmol, out, out_t, sums, ch, ch_t, ldc, off);
}
void launch_cp_qeq_bwd_dual(const Graph& g, int N, int B, int qd, const int64_t* batch, const float* Qmol, const float* out,
                            const float* out_t, const float* sums, const float* g_ch, const float* g_ch_t, int ldg, int off, float* bs,
                            float* g_out, float* g_out_t, hipStream_t s) {
  LAUNCH(k_cp_mol_sums_bwd, (int64_t)B * qd, g, B, qd, batch, Qmol, out, out_t, sums, g_ch, g_ch_t, ldg, off, bs);
  LAUNCH(k_cp_qeq_bwd_dual, (int64_t)N * qd, N, qd, batch, Qmol, out, out_t, sums, bs, g_ch, g_ch_t, ldg, off, g_out, g_out_t);
}
void launch_tn2_edge_pre1_dual(const Graph& g, int E, int N, int F, const float* Ap, const float* Ap_t, const float* Bt, const float* Bt_t,
                               const float* Cs, const float* Cs_t, float* pre1, float* e1_t, float* he1, float* he1_t, hipStream_t s) {
  LAUNCH(k_tn2_edge_pre1_dual, (int64_t)E * F, g, E, N, F, Ap, Ap_t, Bt, Bt_t, Cs, Cs_t, pre1, e1_t, he1, he1_t);
}
void launch_tn2_w_dual(const Graph& g, int E, int F3, const float* e3, const float* e3_t, const float* C, const float* C_t, float* w,
                       float* w_t, hipStream_t s) {
  LAUNCH(k_tn2_w_dual, (int64_t)E * F3, g, (int64_t)E * F3, F3, e3, e3_t, C, C_t, w, w_t);
}
void launch_edge_sweep2(const Graph& g, int N, int F, const int* emap, const float* wA, const float* srcA, const float* wB, const float* srcB,
                        const float* init, float* out, hipStream_t s) {
  LAUNCH(k_edge_sweep2, (int64_t)N * F, g, N, F, emap, wA, srcA, wB, srcB, init, out);
}
void launch_tn2_edge_gw_dual(const Graph& g, int E, int N, int F, const float* g_Mi, const float* g_Mi_t, const float* Pn, const float* Pn_t,
                             const float* e3, const float* e3_t, const float* C, const float* C_t, float* g_e3, float* g_e3_t, float* gcp,
                             float* gcp_t, hipStream_t s) {
  LAUNCH(k_tn2_edge_gw_dual, (int64_t)E * F, g, E, N, F, g_Mi, g_Mi_t, Pn, Pn_t, e3, e3_t, C, C_t, g_e3, g_e3_t, gcp, gcp_t);
}
void launch_tn2_edge_reduce_dual(const Graph& g, int N, int P, int F, const int* erev, const int* pair_edge, const float* g1,
                                 const float* g1_t, float* gB, float* gB_t, float* gCs, float* gCs_t, float* gself, float* gself_t,
                                 float* gAp, float* gAp_t, hipStream_t s) {
  LAUNCH(k_tn2_edge_reduce_dual, (int64_t)N * F, g, N, F, erev, g1, g1_t, gB, gB_t, gCs, gCs_t, gself, gself_t);
  LAUNCH(k_tn2_pair_reduce_dual, (int64_t)P * F, (int64_t)P * F, F, pair_edge, erev, g1, g1_t, gAp, gAp_t);
}
void launch_edge_rowdot(const Graph& g, int E, int W, int ldx, const float* x, const float* x_t, const float* y, const float* y2,
                        const float* d_t, bool accumulate, float* val, float* val_t, hipStream_t s) {
  LAUNCH(k_edge_rowdot, (int64_t)E * 64, g, E, W, ldx, x, x_t, y, y2, d_t, accumulate ? 1 : 0, val, val_t);  // a wave per edge
}
void launch_pair_from_edges(int P, const int* pair_edge, const int* erev, const float* val, const float* val_t, float* out, float* out_t,
                            hipStream_t s) {
  LAUNCH(k_pair_from_edges, (int64_t)P, P, pair_edge, erev, val, val_t, out, out_t);
}
void launch_coulomb_atom_dual(const Graph& g, int N, int QC, const int64_t* batch, const float* pos, const float* v, const float* box,
                              int box_per_mol, const float* ch, const float* ch_t, const float* wq, float* wsum_scratch, float cut, float eps,
                              float scale, float* e_atom, float* e_atom_t, float* g_q, float* g_q_t, float* g_pos, float* hv, hipStream_t s) {
  hipLaunchKernelGGL(k_vec_sum, dim3(1), dim3(TB), 0, s, QC, wq, wsum_scratch);
  LAUNCH(k_coulomb_atom_dual, (int64_t)N, g, N, QC, batch, pos, v, box, box_per_mol, ch, ch_t, wq, wsum_scratch, cut, eps, scale, e_atom,
         e_atom_t, g_q, g_q_t, g_pos, hv);
}
void launch_add_cols(int N, int qd, const float* src, float* dst, int ld, int off, hipStream_t s) {
  LAUNCH(k_add_cols, (int64_t)N * qd, (int64_t)N * qd, qd, src, dst, ld, off);
}
void launch_axpy1(int64_t n, float a, const float* x, float* y, hipStream_t s) { LAUNCH(k_axpy1, n, n, a, x, y); }
void launch_scale1(int64_t n, float a, const float* x, float* y, hipStream_t s) { LAUNCH(k_scale1, n, n, a, x, y); }

}  // namespace hvp
}  // namespace tn
