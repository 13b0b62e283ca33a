// Launchers of the analytic second-order pass (tn_hvp.hip; arithmetic in tn_hvp_math.h, schedule in tn_hvp_api.hip).
// "_t" = tangent along the direction v in the positions; node tensors [N][9][F], pair rows [P + 1][...], self pair at row P.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tn_kernels.h"

namespace tn {
namespace hvp {

void launch_pair_tangent(const Graph& g, int P, int K, const float* v, const float* dphi, const float* dC, float* d_t, float* rhat_t,
                         float* phi_t, float* C_t, hipStream_t s);
void launch_embed_scatter_dual(const Graph& g, int N, int F, int P, const int64_t* z, const float* Utab, const float* Vtab, const float* Q,
                               const float* Q_t, const float* C, const float* C_t, const float* rhat_t, float* u0, float* u0_t, float* s0n,
                               float* s0n_t, hipStream_t s);
void launch_ln_dual(int R, int W, const float* x, const float* x_t, const float* w, const float* b, float* y, float* xh, float* rstd,
                    float* y_t, float* xh_t, float* rstd_t, hipStream_t s);
void launch_lnbwd_dual(int R, int W, const float* g, const float* g_t, const float* xh, const float* xh_t, const float* rstd,
                       const float* rstd_t, const float* w, float* o, float* o_t, hipStream_t s);
void launch_silu_tangent(int64_t n, const float* a, const float* a_t, float* h_t, hipStream_t s);
void launch_dsilu_dual(int64_t n, const float* g, const float* g_t, const float* a, const float* a_t, float* o, float* o_t, hipStream_t s);
void launch_gate_mul_dual(int N, int F, const float* UX, const float* UX_t, const float* gates, const float* gates_t, float* X, float* X_t,
                          hipStream_t s);
void launch_w_dual(int64_t rows, int F3, const float* e3, const float* e3_t, const float* C, const float* C_t, float* w, float* w_t,
                   hipStream_t s);
void launch_norm_dual(int N, int F, const float* X, const float* X_t, float* Xh, float* Xh_t, hipStream_t s);
void launch_group_dual(int N, int F, const float* Pn, const float* Pn_t, const float* Mi, const float* Mi_t, const float* kap, int o3,
                       float* Ch, float* Ch_t, hipStream_t s);
void launch_update_dual(int N, int F, const float* Xh, const float* Xh_t, const float* D, const float* D_t, const float* kap, float* Xn,
                        float* Xn_t, hipStream_t s);
void launch_feat_dual(int N, int F, const float* X, const float* X_t, float* feat, float* feat_t, hipStream_t s);
// ge != null: the energy seed of one-pass training (tn_hvp_math.h head_dual); launch_head_bias_seed: out[0] = - std sum_n ge[molecule(n)];
// launch_row_seed: x_t[n, :] -= ge[molecule(n)] x[n, :]
void launch_head_bias_seed(int N, float std_, const float* ge, const int64_t* batch, float* out, hipStream_t s);
void launch_row_seed(int N, int W, const float* ge, const int64_t* batch, const float* x, float* x_t, hipStream_t s);
void launch_head_dual(int N, int H, const float* ao, const float* ao_t, const float* O2, float std_, const float* ge, const int64_t* batch,
                      float* g_ao, float* g_ao_t,
                      float* headv, hipStream_t s);
void launch_readout_bwd_dual(int N, int F, const float* X, const float* X_t, const float* g_feat, const float* g_feat_t, float* G,
                             float* G_t, hipStream_t s);
void launch_update_bwd_dual(int N, int F, const float* G, const float* G_t, const float* D, const float* D_t, const float* kap, float* g_D,
                            float* g_D_t, hipStream_t s);
void launch_group_bwd_dual(int N, int F, const float* g_Ch, const float* g_Ch_t, const float* Pn, const float* Pn_t, const float* Mi,
                           const float* Mi_t, const float* kap, int o3, float* g_Mi, float* g_Mi_t, float* g_PnY, float* g_PnY_t,
                           hipStream_t s);
void launch_pair_gw_dual(const Graph& g, int P, int F, const float* g_Mi, const float* g_Mi_t, const float* Pn, const float* Pn_t,
                         const float* self_gw, const float* self_gw_t, const float* e3, const float* e3_t, const float* C, const float* C_t,
                         float* g_e3, float* g_e3_t, float* gcp, float* gcp_t, hipStream_t s);  // gcp: per-channel parts of g_C (or null)
void launch_norm_bwd_dual(int N, int F, const float* X, const float* X_t, const float* G, const float* G_t, const float* gL,
                          const float* gL_t, float* Gn, float* Gn_t, hipStream_t s);
void launch_gate_bwd_dual(int N, int F, const float* G, const float* G_t, const float* UX, const float* UX_t, const float* gates,
                          const float* gates_t, const float* a2, const float* a2_t, float* g_UX, float* g_UX_t, float* g_a2, float* g_a2_t,
                          hipStream_t s);
void launch_embed_bwd_atom_dual(int N, int F, const float* gL, const float* gL_t, const float* u0, const float* u0_t, const float* g_s0n,
                                const float* g_s0n_t, float* gA, float* gA_t, hipStream_t s);
void launch_embed_edge_dual(const Graph& g, int N, int F, int P, const int64_t* z, const float* Utab, const float* Vtab, const float* Q,
                            const float* Q_t, const float* C, const float* C_t, const float* rhat_t, const float* gA, const float* gA_t,
                            float* gq, float* gq_t, int64_t dir_stride, float* selfq, float* selfq_t, float* gZu_t, float* gZv_t,
                            hipStream_t s);

// ---- H v in the positions
void launch_radial2(const Graph& g, int P, int K, const float* means, const float* betas, float lo, float up, float* d2phi, float* d2C,
                    hipStream_t s);
void launch_pair_rowdot(int rows, int W, const float* x, const float* x_t, const float* y, const float* y2, const float* d_t, bool accumulate,
                        float* out, float* out_t, hipStream_t s);
void launch_edge_geom_dual(const Graph& g, int E, int N, int F, int P, const int64_t* z, const float* Utab, const float* Vtab, const float* Q,
                           const float* Q_t, const float* C, const float* C_t, const float* rhat_t, const float* gA, const float* gA_t,
                           float* ec, float* ec_t, int64_t dir_stride, hipStream_t s);
void launch_geom_dual(const Graph& g, int P, const float* d_t, const float* rhat_t, const float* dC, const float* d2C, const float* gC,
                      const float* gC_t, const float* gphid, const float* gphid_t, const float* ec, const float* ec_t, int64_t dir_stride,
                      float* gdel, float* gdel_t, hipStream_t s);
void launch_pair_to_atom(const Graph& g, int N, int P, const float* gp, float* out, hipStream_t s);

// ---- Equivariant Transformer (tn_et_hvp.hip; bodies and the EtAttn operand pack in tn_et_hvp_math.h)
struct EtAttn;
void launch_rowscale_dual(int64_t rows, int W, const float* e, const float* e_t, const float* C, const float* C_t, float* o, float* o_t,
                          hipStream_t s);
void launch_et_nbr_embed_dual(const Graph& g, int N, int F, int P, const int64_t* z, const float* emb, const float* embN, const float* Wn,
                              const float* Wn_t, float* xcat, float* xcat_t, hipStream_t s);
void launch_et_embed_dual(int N, int F, const int64_t* z, const float* emb, float* x, float* x_t, hipStream_t s);
void launch_et_attn_fwd_dual(const Graph& g, int N, int H, const EtAttn& A_, float* xagg, float* xagg_t, float* vagg, float* vagg_t,
                             hipStream_t s);
void launch_et_update_dual(int N, int F, const float* x, const float* x_t, const float* vec, const float* vec_t, const float* vp,
                           const float* vp_t, const float* o, const float* o_t, const float* vagg, const float* vagg_t, float* xn,
                           float* xn_t, float* vecn, float* vecn_t, float* vdot, float* vdot_t, hipStream_t s);
void launch_et_cat_norm_dual(int N, int W, const float* x, const float* x_t, int Fx, const float* u, const float* u_t, int ldu, int Fn,
                             int ldo, float* out, float* out_t, hipStream_t s);
void launch_et_head_mid_dual(int N, int F2, const float* y, const float* y_t, const float* u2, const float* u2_t, int ldu, float* hcat2,
                             float* hcat2_t, float* vq, float* vq_t, hipStream_t s);
void launch_et_norm_bwd_dual(int N, int W, const float* g_n, const float* g_n_t, int ldg, const float* u, const float* u_t, int ldu,
                             float* g_u, float* g_u_t, int ldgu, hipStream_t s);
void launch_et_head_mid_bwd_dual(int N, int F2, const float* y, const float* y_t, const float* u2, const float* u2_t, int ldu,
                                 const float* g_h2, const float* g_h2_t, const float* g_vq, const float* g_vq_t, float* g_y, float* g_y_t,
                                 float* g_u2, float* g_u2_t, int ldgu, hipStream_t s);
void launch_et_update_bwd_dual(int N, int F, const float* g_x, const float* g_x_t, const float* g_vec, const float* g_vec_t, const float* vp,
                               const float* vp_t, const float* o, const float* o_t, const float* vdot, const float* vdot_t, float* g_o,
                               float* g_o_t, float* g_vp, float* g_vp_t, hipStream_t s);
void launch_et_attn_bwd_dual(const Graph& g, int N, int H, const EtAttn& A_, const float* g_xagg, const float* g_xagg_t, const float* g_vagg,
                             const float* g_vagg_t, float* g_qkv, float* g_qkv_t, float* g_vec_in, float* g_vec_in_t, float* gq, float* gq_t,
                             int64_t dir_stride, float* selfq, float* selfq_t, float* slots, float* slots_t, int64_t slot_dir_stride,
                             hipStream_t s);
void launch_et_filter_gpre_dual(int P, int Wd, const float* gq, const float* gq_t, int64_t dir_stride, const float* self_g,
                                const float* self_g_t, const float* ekv, const float* ekv_t, float* g_e, float* g_e_t, hipStream_t s);
void launch_et_nbr_bwd_dual(const Graph& g, int N, int F, int P, const int64_t* z, const float* embN, const float* Wn, const float* Wn_t,
                            const float* g_xcat, const float* g_xcat_t, float* gq, float* gq_t, int64_t dir_stride, float* gZ_t,
                            hipStream_t s);
void launch_et_nbr_pair_dual(int P, int F, const float* gq, const float* gq_t, int64_t dir_stride, const float* C, const float* C_t,
                             float* g_Wn, float* g_Wn_t, float* g_en, float* g_en_t, hipStream_t s);
void launch_pair_rowdot2(int rows, int W, const float* x, const float* x_t, const float* y, const float* y_t, bool accumulate, float* out,
                         float* out_t, hipStream_t s);
void launch_et_pair_slots_dual(int P, int H, int nsets, const float* slots, const float* slots_t, int64_t set_stride,
                               int64_t slot_dir_stride, float* g_cut, float* g_cut_t, float* g_rh, float* g_rh_t, hipStream_t s);
void launch_et_geom_dual(const Graph& g, int P, const float* d_t, const float* rhat_t, const float* dC, const float* d2C, const float* g_cut,
                         const float* g_cut_t, const float* g_dphi, const float* g_dphi_t, const float* g_rh, const float* g_rh_t, float* gdel,
                         float* gdel_t, hipStream_t s);
void launch_add2(int64_t n, const float* a, float* o, hipStream_t s);  // o += a

// ---- TensorNet2 + Coulomb head (tn_tn2_hvp.hip; bodies in tn_tn2_hvp_math.h)
void launch_cp_feat_dual(int N, int F, const float* X, const float* X_t, float* feat, float* feat_t, hipStream_t s);
void launch_cp_feat_bwd_dual(int N, int F, const float* X, const float* X_t, const float* g_feat, const float* g_feat_t, float* G, float* G_t,
                             hipStream_t s);
void launch_cp_qeq_dual(const Graph& g, int N, int B, int qd, const int64_t* batch, const float* Qmol, const float* out, const float* out_t,
                        float* sums, float* ch, float* ch_t, int ldc, int off, hipStream_t s);
void launch_cp_qeq_bwd_dual(const Graph& g, int N, int B, int qd, const int64_t* batch, const float* Qmol, const float* out,
                            const float* out_t, const float* sums, const float* g_ch, const float* g_ch_t, int ldg, int off, float* bs,
                            float* g_out, float* g_out_t, hipStream_t s);
void launch_tn2_edge_pre1_dual(const Graph& g, int E, int N, int F, const float* Ap, const float* Ap_t, const float* Bt, const float* Bt_t,
                               const float* Cs, const float* Cs_t, float* pre1, float* e1_t, float* he1, float* he1_t, hipStream_t s);
void launch_tn2_w_dual(const Graph& g, int E, int F3, const float* e3, const float* e3_t, const float* C, const float* C_t, float* w,
                       float* w_t, hipStream_t s);
void launch_edge_sweep2(const Graph& g, int N, int F, const int* emap, const float* wA, const float* srcA, const float* wB, const float* srcB,
                        const float* init, float* out, hipStream_t s);
void launch_tn2_edge_gw_dual(const Graph& g, int E, int N, int F, const float* g_Mi, const float* g_Mi_t, const float* Pn, const float* Pn_t,
                             const float* e3, const float* e3_t, const float* C, const float* C_t, float* g_e3, float* g_e3_t, float* gcp,
                             float* gcp_t, hipStream_t s);
void launch_tn2_edge_reduce_dual(const Graph& g, int N, int P, int F, const int* erev, const int* pair_edge, const float* g1,
                                 const float* g1_t, float* gB, float* gB_t, float* gCs, float* gCs_t, float* gself, float* gself_t,
                                 float* gAp, float* gAp_t, hipStream_t s);
void launch_edge_rowdot(const Graph& g, int E, int W, int ldx, const float* x, const float* x_t, const float* y, const float* y2,
                        const float* d_t, bool accumulate, float* val, float* val_t, hipStream_t s);
void launch_pair_from_edges(int P, const int* pair_edge, const int* erev, const float* val, const float* val_t, float* out, float* out_t,
                            hipStream_t s);
void launch_coulomb_atom_dual(const Graph& g, int N, int QC, const int64_t* batch, const float* pos, const float* v, const float* box,
                              int box_per_mol, const float* ch, const float* ch_t, const float* wq, float* wsum_scratch, float cut, float eps,
                              float scale, float* e_atom, float* e_atom_t, float* g_q, float* g_q_t, float* g_pos, float* hv, hipStream_t s);
void launch_add_cols(int N, int qd, const float* src, float* dst, int ld, int off, hipStream_t s);
void launch_axpy1(int64_t n, float a, const float* x, float* y, hipStream_t s);   // y += a x
void launch_scale1(int64_t n, float a, const float* x, float* y, hipStream_t s);  // y = a x

}  // namespace hvp
}  // namespace tn
