// TEST INFRASTRUCTURE ONLY.  The per-element bodies of the analytic second-order pass (torchmd-net_amd/csrc/tn_hvp_math.h) in plain
// host loops, compiled host-only (no device code, no HIP runtime calls), so that tests/test_hvp_host.py can check the arithmetic
// the kernels of tn_hvp.hip run against oracle/tensornet_second_order.py on a machine without a GPU.  Nothing under
// torchmd-net_amd/ links or loads this file.
#include "../torchmd-net_amd/csrc/tn_hvp_math.h"

using namespace tn::hvp;

extern "C" {

void hh_pair_tangent(int P, int K, const int* pi, const int* pj, const float* prhat, const float* pd, const float* v, const float* dphi,
                     const float* dC, float* d_t, float* rhat_t, float* phi_t, float* C_t) {
  for (int p = 0; p <= P; ++p) pair_tangent(p, P, K, pi, pj, prhat, pd, v, dphi, dC, d_t, rhat_t, phi_t, C_t);
}
void hh_embed_scatter_dual(int N, int F, int P, const int* rowptr, const int* col, const int* epair, const float* esign, const int64_t* z,
                           const float* Utab, const float* Vtab, const float* Q, const float* Q_t, const float* C, const float* C_t,
                           const float* prhat, const float* rhat_t, float* u0, float* u0_t, float* s0n, float* s0n_t) {
  for (int i = 0; i < N; ++i)
    for (int f = 0; f < F; ++f)
      embed_scatter_dual(i, f, F, P, rowptr, col, epair, esign, z, Utab, Vtab, Q, Q_t, C, C_t, prhat, rhat_t, u0, u0_t, s0n, s0n_t);
}
void hh_ln_dual(int R, int W, const float* x, const float* x_t, const float* w, const float* b, float* y, float* xh, float* rstd,
                float* y_t, float* xh_t, float* rstd_t) {
  for (int r = 0; r < R; ++r) ln_dual(r, W, x, x_t, w, b, y, xh, rstd, y_t, xh_t, rstd_t);
}
void hh_lnbwd_dual(int R, int W, const float* g, const float* g_t, const float* xh, const float* xh_t, const float* rstd,
                   const float* rstd_t, const float* w, float* o, float* o_t) {
  for (int r = 0; r < R; ++r) lnbwd_dual(r, W, g, g_t, xh, xh_t, rstd, rstd_t, w, o, o_t);
}
void hh_silu_tangent(int64_t n, const float* a, const float* a_t, float* h_t) {
  for (int64_t i = 0; i < n; ++i) silu_tangent(i, a, a_t, h_t);
}
void hh_dsilu_dual(int64_t n, const float* g, const float* g_t, const float* a, const float* a_t, float* o, float* o_t) {
  for (int64_t i = 0; i < n; ++i) dsilu_dual(i, g, g_t, a, a_t, o, o_t);
}
void hh_gate_mul_dual(int N, int F, const float* UX, const float* UX_t, const float* gates, const float* gates_t, float* X, float* X_t) {
  for (int n = 0; n < N; ++n)
    for (int f = 0; f < F; ++f) gate_mul_dual(n, f, F, UX, UX_t, gates, gates_t, X, X_t);
}
void hh_w_dual(int64_t rows, int F3, const float* e3, const float* e3_t, const float* C, const float* C_t, float* w, float* w_t) {
  for (int64_t i = 0; i < rows * F3; ++i) w_dual(i, F3, e3, e3_t, C, C_t, w, w_t);
}
void hh_norm_dual(int N, int F, const float* X, const float* X_t, float* Xh, float* Xh_t) {
  for (int n = 0; n < N; ++n)
    for (int f = 0; f < F; ++f) norm_dual(n, f, F, X, X_t, Xh, Xh_t);
}
void hh_group_dual(int N, int F, const float* Pn, const float* Pn_t, const float* Mi, const float* Mi_t, const float* kap, int o3,
                   float* Ch, float* Ch_t) {
  for (int n = 0; n < N; ++n)
    for (int f = 0; f < F; ++f) group_dual(n, f, F, Pn, Pn_t, Mi, Mi_t, kap, o3, Ch, Ch_t);
}
void hh_update_dual(int N, int F, const float* Xh, const float* Xh_t, const float* D, const float* D_t, const float* kap, float* Xn,
                    float* Xn_t) {
  for (int n = 0; n < N; ++n)
    for (int f = 0; f < F; ++f) update_dual(n, f, F, Xh, Xh_t, D, D_t, kap, Xn, Xn_t);
}
void hh_feat_dual(int N, int F, const float* X, const float* X_t, float* feat, float* feat_t) {
  for (int n = 0; n < N; ++n)
    for (int f = 0; f < F; ++f) feat_dual(n, f, F, X, X_t, feat, feat_t);
}
void hh_head_dual(int64_t n, int H, const float* ao, const float* ao_t, const float* O2, float std_, float* g_ao, float* g_ao_t,
                  float* headv) {
  for (int64_t i = 0; i < n; ++i) head_dual(i, H, ao, ao_t, O2, std_, g_ao, g_ao_t, headv);
}
void hh_readout_bwd_dual(int N, int F, const float* X, const float* X_t, const float* g_feat, const float* g_feat_t, float* G, float* G_t) {
  for (int n = 0; n < N; ++n)
    for (int f = 0; f < F; ++f) readout_bwd_dual(n, f, F, X, X_t, g_feat, g_feat_t, G, G_t);
}
void hh_update_bwd_dual(int N, int F, const float* G, const float* G_t, const float* D, const float* D_t, const float* kap, float* g_D,
                        float* g_D_t) {
  for (int n = 0; n < N; ++n)
    for (int f = 0; f < F; ++f) update_bwd_dual(n, f, F, G, G_t, D, D_t, kap, g_D, g_D_t);
}
void hh_group_bwd_dual(int N, int F, const float* g_Ch, const float* g_Ch_t, const float* Pn, const float* Pn_t, const float* Mi,
                       const float* Mi_t, const float* kap, int o3, float* g_Mi, float* g_Mi_t, float* g_PnY, float* g_PnY_t) {
  for (int n = 0; n < N; ++n)
    for (int f = 0; f < F; ++f) group_bwd_dual(n, f, F, g_Ch, g_Ch_t, Pn, Pn_t, Mi, Mi_t, kap, o3, g_Mi, g_Mi_t, g_PnY, g_PnY_t);
}
void hh_pair_gw_dual(int P, int F, const int* pi, const int* pj, const float* g_Mi, const float* g_Mi_t, const float* Pn,
                     const float* Pn_t, const float* self_gw, const float* self_gw_t, const float* e3, const float* e3_t, const float* C,
                     const float* C_t, float* g_e3, float* g_e3_t, float* gcp, float* gcp_t) {
  for (int64_t i = 0; i < (int64_t)(P + 1) * F; ++i)
    pair_gw_dual(i, P, F, pi, pj, g_Mi, g_Mi_t, Pn, Pn_t, self_gw, self_gw_t, e3, e3_t, C, C_t, g_e3, g_e3_t, gcp, gcp_t);
}
void hh_norm_bwd_dual(int N, int F, const float* X, const float* X_t, const float* G, const float* G_t, const float* gL, const float* gL_t,
                      float* Gn, float* Gn_t) {
  for (int n = 0; n < N; ++n)
    for (int f = 0; f < F; ++f) norm_bwd_dual(n, f, F, X, X_t, G, G_t, gL, gL_t, Gn, Gn_t);
}
void hh_gate_bwd_dual(int N, int F, const float* G, const float* G_t, const float* UX, const float* UX_t, const float* gates,
                      const float* gates_t, const float* a2, const float* a2_t, float* g_UX, float* g_UX_t, float* g_a2, float* g_a2_t) {
  for (int n = 0; n < N; ++n)
    for (int f = 0; f < F; ++f) gate_bwd_dual(n, f, F, G, G_t, UX, UX_t, gates, gates_t, a2, a2_t, g_UX, g_UX_t, g_a2, g_a2_t);
}
void hh_embed_bwd_atom_dual(int N, int F, const float* gL, const float* gL_t, const float* u0, const float* u0_t, const float* g_s0n,
                            const float* g_s0n_t, float* gA, float* gA_t) {
  for (int n = 0; n < N; ++n)
    for (int f = 0; f < F; ++f) embed_bwd_atom_dual(n, f, F, gL, gL_t, u0, u0_t, g_s0n, g_s0n_t, gA, gA_t);
}
void hh_embed_edge_dual(int N, int F, int P, const int* rowptr, const int* col, const int* epair, const float* esign, const int64_t* z,
                        const float* Utab, const float* Vtab, const float* Q, const float* Q_t, const float* C, const float* C_t,
                        const float* prhat, const float* rhat_t, const float* gA, const float* gA_t, float* gq, float* gq_t,
                        int64_t dir_stride, float* selfq, float* selfq_t, float* gZu_t, float* gZv_t) {
  for (int i = 0; i < N; ++i)
    for (int f = 0; f < F; ++f)
      embed_edge_dual(i, f, F, P, rowptr, col, epair, esign, z, Utab, Vtab, Q, Q_t, C, C_t, prhat, rhat_t, gA, gA_t, gq, gq_t, dir_stride,
                      selfq, selfq_t, gZu_t, gZv_t);
}
void hh_radial2(int P, int K, const float* pd, const float* means, const float* betas, float lo, float up, float* d2phi, float* d2C) {
  for (int64_t i = 0; i < (int64_t)(P + 1) * K; ++i) radial2(i, P, K, pd, means, betas, lo, up, d2phi, d2C);
}
void hh_pair_rowdot(int rows, int W, const float* x, const float* x_t, const float* y, const float* y2, const float* d_t, int accumulate,
                    float* out, float* out_t) {
  for (int p = 0; p < rows; ++p) pair_rowdot(p, W, x, x_t, y, y2, d_t, accumulate, out, out_t);
}
void hh_edge_geom_dual(int E, int N, int F, int P, const int* rowptr, const int* col, const int* epair, const float* esign, const int64_t* z,
                       const float* Utab, const float* Vtab, const float* Q, const float* Q_t, const float* C, const float* C_t,
                       const float* prhat, const float* rhat_t, const float* gA, const float* gA_t, float* ec, float* ec_t,
                       int64_t dir_stride) {
  for (int e = 0; e < E; ++e)
    edge_geom_dual(e, N, F, P, rowptr, col, epair, esign, z, Utab, Vtab, Q, Q_t, C, C_t, prhat, rhat_t, gA, gA_t, ec, ec_t, dir_stride);
}
void hh_geom_dual(int P, const float* pd, const float* prhat, const float* d_t, const float* rhat_t, const float* dC, const float* d2C,
                  const float* gC, const float* gC_t, const float* gphid, const float* gphid_t, const float* ec, const float* ec_t,
                  int64_t dir_stride, float* gdel, float* gdel_t) {
  for (int p = 0; p < P; ++p) geom_dual(p, P, pd, prhat, d_t, rhat_t, dC, d2C, gC, gC_t, gphid, gphid_t, ec, ec_t, dir_stride, gdel, gdel_t);
}
void hh_pair_to_atom(int N, int P, const int* rowptr, const int* epair, const float* esign, const float* g, float* out) {
  for (int i = 0; i < N; ++i) pair_to_atom(i, P, rowptr, epair, esign, g, out);
}

}  // extern "C"
