// TEST INFRASTRUCTURE ONLY.  The per-element bodies of the analytic second-order pass (torchmd-net_amd/csrc/tn_hvp_math.h) in plain
// host loops, compiled host-only (no device code, no HIP runtime calls), so that tests/test_hvp_host.py can check the arithmetic
// the kernels of tn_hvp.hip run against oracle/tensornet_second_order.py on a machine without a GPU.  Nothing under
// torchmd-net_amd/ links or loads this file.
#include "../torchmd-net_amd/csrc/tn_hvp_math.h"
#include "../torchmd-net_amd/csrc/tn_et_hvp_math.h"
#include "../torchmd-net_amd/csrc/tn_tn2_hvp_math.h"

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
void hh_head_dual(int64_t n, int H, const float* ao, const float* ao_t, const float* O2, float std_, const float* ge, const int64_t* batch,
                  float* g_ao, float* g_ao_t, float* headv) {
  for (int64_t i = 0; i < n; ++i) head_dual(i, H, ao, ao_t, O2, std_, ge, batch, g_ao, g_ao_t, headv);
}
void hh_head_bias_seed(int N, float std_, const float* ge, const int64_t* batch, float* out) {  // k_head_bias_seed's order of summation
  float part[256];
  for (int t = 0; t < 256; ++t) {
    float a = 0.f;
    for (int n = t; n < N; n += 256) a += head_bias_seed_term(n, ge, batch);
    part[t] = a;
  }
  float tot = 0.f;
  for (int k = 0; k < 256; ++k) tot += part[k];
  out[0] = -std_ * tot;
}
void hh_row_seed(int N, int W, const float* ge, const int64_t* batch, const float* x, float* x_t) {
  for (int64_t i = 0; i < (int64_t)N * W; ++i) row_seed(i, W, ge, batch, x, x_t);
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

// ---------------------------------------------------------------- Equivariant Transformer (tn_et_hvp_math.h)
void he_rowscale_dual(int64_t rows, int W, const float* e, const float* e_t, const float* C, const float* C_t, float* o, float* o_t) {
  for (int64_t i = 0; i < rows * W; ++i) rowscale_dual(i, W, e, e_t, C, C_t, o, o_t);
}
void he_nbr_embed_dual(int N, int F, int P, const int* rowptr, const int* col, const int* epair, const float* esign, const int64_t* z,
                       const float* emb, const float* embN, const float* Wn, const float* Wn_t, float* xcat, float* xcat_t) {
  for (int i = 0; i < N; ++i)
    for (int f = 0; f < F; ++f) et_nbr_embed_dual(i, f, F, P, rowptr, col, epair, esign, z, emb, embN, Wn, Wn_t, xcat, xcat_t);
}
void he_embed_dual(int N, int F, const int64_t* z, const float* emb, float* x, float* x_t) {
  for (int i = 0; i < N; ++i)
    for (int f = 0; f < F; ++f) et_embed_dual(i, f, F, z, emb, x, x_t);
}
static EtAttn mk_attn(const float* qkv, const float* qkv_t, const float* vec, const float* vec_t, const float* dkv, const float* dkv_t,
                      const float* C, const float* C_t, const float* prhat, const float* rhat_t, int F, int hd, int Wd, int ok, int ov, int vc,
                      int P) {
  return EtAttn{qkv, qkv_t, vec, vec_t, dkv, dkv_t, C, C_t, prhat, rhat_t, F, hd, Wd, ok, ov, vc, P};
}
void he_attn_fwd_dual(int N, int H, const float* qkv, const float* qkv_t, const float* vec, const float* vec_t, const float* dkv,
                      const float* dkv_t, const float* C, const float* C_t, const float* prhat, const float* rhat_t, int F, int hd, int Wd, int ok,
                      int ov, int vc, int P, const int* rowptr, const int* col, const int* epair, const float* esign, float* xagg,
                      float* xagg_t, float* vagg, float* vagg_t) {
  const EtAttn A_ = mk_attn(qkv, qkv_t, vec, vec_t, dkv, dkv_t, C, C_t, prhat, rhat_t, F, hd, Wd, ok, ov, vc, P);
  for (int t = 0; t < N; ++t)
    for (int h = 0; h < H; ++h) et_attn_fwd_dual(t, h, A_, rowptr, col, epair, esign, xagg, xagg_t, vagg, vagg_t);
}
void he_update_dual(int N, int F, const float* x, const float* x_t, const float* vec, const float* vec_t, const float* vp, const float* vp_t,
                    const float* o, const float* o_t, const float* vagg, const float* vagg_t, float* xn, float* xn_t, float* vecn,
                    float* vecn_t, float* vdot, float* vdot_t) {
  for (int n = 0; n < N; ++n)
    for (int f = 0; f < F; ++f) et_update_dual(n, f, F, x, x_t, vec, vec_t, vp, vp_t, o, o_t, vagg, vagg_t, xn, xn_t, vecn, vecn_t, vdot, vdot_t);
}
void he_cat_norm_dual(int N, int W, const float* x, const float* x_t, int Fx, const float* u, const float* u_t, int ldu, int Fn, int ldo,
                      float* out, float* out_t) {
  for (int n = 0; n < N; ++n)
    for (int f = 0; f < W; ++f) et_cat_norm_dual(n, f, x, x_t, Fx, u, u_t, ldu, Fn, ldo, out, out_t);
}
void he_head_mid_dual(int N, int F2, const float* y, const float* y_t, const float* u2, const float* u2_t, int ldu, float* hcat2,
                      float* hcat2_t, float* vq, float* vq_t) {
  for (int n = 0; n < N; ++n)
    for (int f = 0; f < F2; ++f) et_head_mid_dual(n, f, F2, y, y_t, u2, u2_t, ldu, hcat2, hcat2_t, vq, vq_t);
}
void he_norm_bwd_dual(int N, int W, const float* g_n, const float* g_n_t, int ldg, const float* u, const float* u_t, int ldu, float* g_u,
                      float* g_u_t, int ldgu) {
  for (int n = 0; n < N; ++n)
    for (int f = 0; f < W; ++f) et_norm_bwd_dual(n, f, g_n, g_n_t, ldg, u, u_t, ldu, g_u, g_u_t, ldgu);
}
void he_head_mid_bwd_dual(int N, int F2, const float* y, const float* y_t, const float* u2, const float* u2_t, int ldu, const float* g_h2,
                          const float* g_h2_t, const float* g_vq, const float* g_vq_t, float* g_y, float* g_y_t, float* g_u2, float* g_u2_t,
                          int ldgu) {
  for (int n = 0; n < N; ++n)
    for (int f = 0; f < F2; ++f) et_head_mid_bwd_dual(n, f, F2, y, y_t, u2, u2_t, ldu, g_h2, g_h2_t, g_vq, g_vq_t, g_y, g_y_t, g_u2, g_u2_t, ldgu);
}
void he_update_bwd_dual(int N, int F, const float* g_x, const float* g_x_t, const float* g_vec, const float* g_vec_t, const float* vp,
                        const float* vp_t, const float* o, const float* o_t, const float* vdot, const float* vdot_t, float* g_o, float* g_o_t,
                        float* g_vp, float* g_vp_t) {
  for (int n = 0; n < N; ++n)
    for (int f = 0; f < F; ++f) et_update_bwd_dual(n, f, F, g_x, g_x_t, g_vec, g_vec_t, vp, vp_t, o, o_t, vdot, vdot_t, g_o, g_o_t, g_vp, g_vp_t);
}
void he_attn_bwd_dual(int N, int H, const float* qkv, const float* qkv_t, const float* vec, const float* vec_t, const float* dkv,
                      const float* dkv_t, const float* C, const float* C_t, const float* prhat, const float* rhat_t, int F, int hd, int Wd, int ok,
                      int ov, int vc, int P, const int* rowptr, const int* col, const int* epair, const float* esign, const float* g_xagg,
                      const float* g_xagg_t, const float* g_vagg, const float* g_vagg_t, float* g_qkv, float* g_qkv_t, float* g_vec_in,
                      float* g_vec_in_t, float* gq, float* gq_t, int64_t dir_stride, float* selfq, float* selfq_t, float* slots, float* slots_t,
                      int64_t slot_dir_stride) {
  const EtAttn A_ = mk_attn(qkv, qkv_t, vec, vec_t, dkv, dkv_t, C, C_t, prhat, rhat_t, F, hd, Wd, ok, ov, vc, P);
  for (int t = 0; t < N; ++t)
    for (int h = 0; h < H; ++h) {
      et_attn_bwd_tgt_dual(t, h, H, A_, rowptr, col, epair, esign, g_xagg, g_xagg_t, g_vagg, g_vagg_t, g_qkv, g_qkv_t, gq, gq_t, dir_stride,
                           selfq, selfq_t, slots, slots_t, slot_dir_stride);
      et_attn_bwd_src_dual(t, h, A_, rowptr, col, epair, esign, g_xagg, g_xagg_t, g_vagg, g_vagg_t, g_qkv, g_qkv_t, g_vec_in, g_vec_in_t);
    }
}
void he_filter_gpre_dual(int P, int Wd, const float* gq, const float* gq_t, int64_t dir_stride, const float* self_g, const float* self_g_t,
                         const float* ekv, const float* ekv_t, float* g_e, float* g_e_t) {
  for (int64_t i = 0; i < (int64_t)(P + 1) * Wd; ++i) et_filter_gpre_dual(i, P, Wd, gq, gq_t, dir_stride, self_g, self_g_t, ekv, ekv_t, g_e, g_e_t);
}
void he_nbr_bwd_dual(int N, int F, int P, const int* rowptr, const int* col, const int* epair, const float* esign, const int64_t* z,
                     const float* embN, const float* Wn, const float* Wn_t, const float* g_xcat, const float* g_xcat_t, float* gq, float* gq_t,
                     int64_t dir_stride, float* gZ_t) {
  for (int i = 0; i < N; ++i)
    for (int f = 0; f < F; ++f) et_nbr_bwd_dual(i, f, F, P, rowptr, col, epair, esign, z, embN, Wn, Wn_t, g_xcat, g_xcat_t, gq, gq_t, dir_stride, gZ_t);
}
void he_nbr_pair_dual(int P, int F, const float* gq, const float* gq_t, int64_t dir_stride, const float* C, const float* C_t, float* g_Wn,
                      float* g_Wn_t, float* g_en, float* g_en_t) {
  for (int64_t i = 0; i < (int64_t)P * F; ++i) et_nbr_pair_dual(i, F, gq, gq_t, dir_stride, C, C_t, g_Wn, g_Wn_t, g_en, g_en_t);
}
void he_pair_rowdot2(int rows, int W, const float* x, const float* x_t, const float* y, const float* y_t, int accumulate, float* out,
                     float* out_t) {
  for (int p = 0; p < rows; ++p) pair_rowdot2(p, W, x, x_t, y, y_t, accumulate, out, out_t);
}
void he_pair_slots_dual(int P, int H, int nsets, const float* slots, const float* slots_t, int64_t set_stride, int64_t slot_dir_stride,
                        float* g_cut, float* g_cut_t, float* g_rh, float* g_rh_t) {
  for (int p = 0; p < P; ++p) et_pair_slots_dual(p, H, nsets, slots, slots_t, set_stride, slot_dir_stride, g_cut, g_cut_t, g_rh, g_rh_t);
}
void he_geom_dual(int P, const float* pd, const float* prhat, const float* d_t, const float* rhat_t, const float* dC, const float* d2C,
                  const float* g_cut, const float* g_cut_t, const float* g_dphi, const float* g_dphi_t, const float* g_rh, const float* g_rh_t,
                  float* gdel, float* gdel_t) {
  for (int p = 0; p < P; ++p) et_geom_dual(p, pd, prhat, d_t, rhat_t, dC, d2C, g_cut, g_cut_t, g_dphi, g_dphi_t, g_rh, g_rh_t, gdel, gdel_t);
}

// ---------------------------------------------------------------- TensorNet2 (tn_tn2_hvp_math.h)
void h2_cp_feat_dual(int N, int F, const float* X, const float* X_t, float* feat, float* feat_t) {
  for (int n = 0; n < N; ++n)
    for (int f = 0; f < F; ++f) cp_feat_dual(n, f, F, X, X_t, feat, feat_t);
}
void h2_cp_feat_bwd_dual(int N, int F, const float* X, const float* X_t, const float* g_feat, const float* g_feat_t, float* G, float* G_t) {
  for (int n = 0; n < N; ++n)
    for (int f = 0; f < F; ++f) cp_feat_bwd_dual(n, f, F, X, X_t, g_feat, g_feat_t, G, G_t);
}
void h2_cp_mol_sums(int B, int qd, const int* mstart, const int* mend, const float* out, const float* out_t, float* sums) {
  for (int m = 0; m < B; ++m)
    for (int q = 0; q < qd; ++q) cp_mol_sums(m, q, qd, mstart, mend, out, out_t, sums);
}
void h2_cp_qeq_dual(int N, int qd, const int64_t* batch, const float* Qmol, const float* out, const float* out_t, const float* sums, float* ch,
                    float* ch_t, int ldc, int off) {
  for (int n = 0; n < N; ++n)
    for (int q = 0; q < qd; ++q) cp_qeq_dual(n, q, qd, batch, Qmol, out, out_t, sums, ch, ch_t, ldc, off);
}
void h2_cp_qeq_bwd_dual(int N, int B, int qd, const int* mstart, const int* mend, const int64_t* batch, const float* Qmol, const float* out,
                        const float* out_t, const float* sums, const float* g_ch, const float* g_ch_t, int ldg, int off, float* bs,
                        float* g_out, float* g_out_t) {
  for (int m = 0; m < B; ++m)
    for (int q = 0; q < qd; ++q) cp_mol_sums_bwd(m, q, qd, mstart, mend, batch, Qmol, out, out_t, sums, g_ch, g_ch_t, ldg, off, bs);
  for (int n = 0; n < N; ++n)
    for (int q = 0; q < qd; ++q) cp_qeq_bwd_dual(n, q, qd, batch, Qmol, out, out_t, sums, bs, g_ch, g_ch_t, ldg, off, g_out, g_out_t);
}
void h2_edge_pre1_dual(int E, int N, int F, const int* rowptr, const int* col, const int* epair, const float* Ap, const float* Ap_t,
                       const float* Bt, const float* Bt_t, const float* Cs, const float* Cs_t, float* pre1, float* e1_t, float* he1,
                       float* he1_t) {
  for (int64_t i = 0; i < (int64_t)E * F; ++i) tn2_edge_pre1_dual(i, N, F, rowptr, col, epair, Ap, Ap_t, Bt, Bt_t, Cs, Cs_t, pre1, e1_t, he1, he1_t);
}
void h2_edge_reduce_dual(int N, int F, const int* rowptr, const int* col, const int* erev, const float* g1, const float* g1_t, float* gB,
                         float* gB_t, float* gCs, float* gCs_t, float* gself, float* gself_t) {
  for (int i = 0; i < N; ++i)
    for (int f = 0; f < F; ++f) tn2_edge_reduce_dual(i, f, F, rowptr, col, erev, g1, g1_t, gB, gB_t, gCs, gCs_t, gself, gself_t);
}
void h2_pair_reduce_dual(int P, int F, const int* pair_edge, const int* erev, const float* g1, const float* g1_t, float* gAp, float* gAp_t) {
  for (int64_t i = 0; i < (int64_t)P * F; ++i) tn2_pair_reduce_dual(i, F, pair_edge, erev, g1, g1_t, gAp, gAp_t);
}
void h2_w_dual(int E, int F3, const int* epair, const float* e3, const float* e3_t, const float* C, const float* C_t, float* w, float* w_t) {
  for (int64_t i = 0; i < (int64_t)E * F3; ++i) tn2_w_dual(i, F3, epair, e3, e3_t, C, C_t, w, w_t);
}
void h2_edge_sweep2(int N, int F, const int* rowptr, const int* col, const int* emap, const float* wA, const float* srcA, const float* wB,
                    const float* srcB, const float* init, float* out) {
  for (int n = 0; n < N; ++n)
    for (int f = 0; f < F; ++f) edge_sweep2(n, f, F, rowptr, col, emap, wA, srcA, wB, srcB, init, out);
}
void h2_edge_gw_dual(int E, int N, int F, const int* rowptr, const int* col, const int* epair, const float* g_Mi, const float* g_Mi_t,
                     const float* Pn, const float* Pn_t, const float* e3, const float* e3_t, const float* C, const float* C_t, float* g_e3,
                     float* g_e3_t, float* gcp, float* gcp_t) {
  for (int64_t i = 0; i < (int64_t)E * F; ++i)
    tn2_edge_gw_dual(i, N, F, rowptr, col, epair, g_Mi, g_Mi_t, Pn, Pn_t, e3, e3_t, C, C_t, g_e3, g_e3_t, gcp, gcp_t);
}
void h2_edge_rowdot(int E, int W, int ldx, const int* epair, const float* x, const float* x_t, const float* y, const float* y2,
                    const float* d_t, int accumulate, float* val, float* val_t) {
  for (int e = 0; e < E; ++e) edge_rowdot(e, W, ldx, epair, x, x_t, y, y2, d_t, accumulate, val, val_t);
}
void h2_pair_from_edges(int P, const int* pair_edge, const int* erev, const float* val, const float* val_t, float* out, float* out_t) {
  for (int p = 0; p < P; ++p) pair_from_edges(p, pair_edge, erev, val, val_t, out, out_t);
}
void h2_coulomb_atom_dual(int N, int QC, const int* mstart, const int* mend, const int64_t* batch, const float* pos, const float* v,
                          const float* box, int box_per_mol, const float* ch, const float* ch_t, const float* wq, float wsum, float cut, float eps,
                          float scale, float* e_atom, float* e_atom_t, float* g_q, float* g_q_t, float* g_pos, float* hv) {
  for (int i = 0; i < N; ++i)
    coulomb_atom_dual(i, QC, mstart, mend, batch, pos, v, box, box_per_mol, ch, ch_t, wq, wsum, cut, eps, scale, e_atom, e_atom_t, g_q, g_q_t,
                      g_pos, hv);
}

}  // extern "C"
