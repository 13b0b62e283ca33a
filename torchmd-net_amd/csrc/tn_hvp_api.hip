// Analytic second-order pass of force-matching training, TensorNet + Scalar head (C ABI: tmdnet_force_param_grads).
//
//     s(theta) = v . d(sum_m E_m)/d pos ,  v = d loss / d F      ->      d loss / d theta |_forces = - d s / d theta
//
// The reference differentiates its autograd graph twice (torchmdnet/models/model.py:618-628 with create_graph = self.training,
// and the *_bwd_bwd kernels of warp_ops/).  Here d s / d theta is the forward-mode tangent, along v, of the hand-written E + F
// program: one self-contained pass that carries (value, tangent) through the forward and (adjoint, tangent of the adjoint)
// through the reverse sweep, and takes every weight gradient as  g_y_t^T x + g_y^T x_t  where the first-order pass takes
// g_y^T x.  Statement by statement it is oracle/tensornet_second_order.py (pinned to autograd-of-autograd in fp64);
// launch by launch it is tests/hvp_host_mirror.py, which runs the same kernel bodies on the host.  Dense products go through the
// engine's GEMM launchers (gemm / tensor_linear / launch_tn_gemm / launch_colsum), the neighbour sums through its plain CSR
// sweep (launch_message_adjoint), everything else is a kernel of tn_hvp.hip.
// The gradient buffer has the layout of tmdnet_energy_param_grads (tmdnet_param_grad_entry).
#include <algorithm>
#include <string>

#include "tn_hvp.h"
#include "tn_model.h"
#include "tn_tn2.h"

using namespace tn;

namespace {

struct HvpLayer {
  float *pre1, *he1, *e1_t, *he1_t, *pre2, *he2, *e2_t, *he2_t, *pre3, *e3_t, *w, *w_t;
  float *Xh, *Xh_t, *Pn, *Pn_t, *Mi, *Mi_t, *Ch, *Ch_t, *D, *D_t;
};
struct HvpCp {  // one ChargePredict head of TensorNet2 (kept for the reverse sweep)
  float *cf, *cf_t, *ln, *xh, *rstd, *ln_t, *xh_t, *rstd_t, *a1, *h1, *a1_t, *h1_t, *a2, *h2, *a2_t, *h2_t, *out, *out_t, *sums;
};
struct HvpTn2 {  // what TensorNet2 + Coulomb head adds
  int *erev, *eid, *pair_edge;
  float *charges, *charges_t, *g_charges, *g_charges_t, *Ap, *Ap_t, *Bt, *Bt_t, *Cs, *Cs_t, *gB, *gB_t, *gCs, *gCs_t, *gself, *gself_t, *gAp,
      *gAp_t, *g_cl, *g_cl_t, *gCe, *gCe_t, *e_c, *e_c_t, *g_q, *g_q_t, *gpos_c, *hv_c, *wsum;
  float *bs, *g_out, *g_out_t, *g_h2, *g_h2_t, *g_a2, *g_a2_t, *g_h1, *g_h1_t, *g_a1, *g_a1_t, *g_ln, *g_ln_t, *g_cf, *g_cf_t;
  std::vector<HvpCp> cp;                // L + 1
  std::vector<float*> chl, chl_t;       // per layer: the charge block its edge MLP reads, contiguous [N][qd]
};
struct HvpBuffers {
  HvpTn2 t2;
  // forward (kept for the reverse sweep)
  float *phi, *dphi, *C, *dC, *d_t, *rhat_t, *phi_t, *C_t, *kap;
  float *Q, *Q_t, *u0, *u0_t, *s0n, *s0n_t, *ln0, *xh0, *rstd0, *ln0_t, *xh0_t, *rstd0_t;
  float *a1, *h1, *a1_t, *h1_t, *a2, *gates, *a2_t, *gates_t, *UX, *UX_t;
  std::vector<float*> X, X_t;  // L + 1
  std::vector<HvpLayer> lay;
  float *feat, *feat_t, *lnr, *xhr, *rstdr, *lnr_t, *xhr_t, *rstdr_t, *al, *x, *al_t, *x_t, *ao, *ao_t, *g_ao, *g_ao_t, *headv;
  // reverse sweep (reused layer after layer)
  float *g_x, *g_x_t, *g_al, *g_al_t, *g_ln, *g_ln_t, *g_feat, *g_feat_t, *G, *G_t, *Gn, *Gn_t;
  float *g_D, *g_D_t, *g_Ch, *g_Ch_t, *g_Mi, *g_Mi_t, *g_Pn, *g_Pn_t, *gXl, *gXl_t;
  float *self_gw, *self_gw_t, *g3, *g3_t, *gh2, *gh2_t, *g2, *g2_t, *gh1, *gh1_t, *g1, *g1_t;
  float *g_UX, *g_UX_t, *g_a2, *g_a2_t, *g_h1, *g_h1_t, *g_a1, *g_a1_t, *g_ln0, *g_ln0_t, *g_s0n, *g_s0n_t, *g_u0l, *g_u0l_t;
  float *gA, *gA_t, *gq, *gq_t, *selfq, *selfq_t, *gZu_t, *gZv_t, *onehot, *part;
  // H v in the positions
  float *d2phi, *d2C, *gC, *gC_t, *gphid, *gphid_t, *dQ, *d2Q, *de1, *d2e1, *gcp, *gcp_t, *ec, *ec_t, *gdel, *gdel_t, *gpos;
};

HvpBuffers carve_hvp(void* ws, const tmdnet_hparams& hp, int64_t N, int64_t P, size_t* total, int64_t qd = 0, int64_t B = 0) {
  const int64_t F = hp.hidden_channels, K = hp.num_rbf, L = hp.num_layers, H = hp.head_hidden, Z = hp.max_z, P1 = P + 1;
  const int64_t N9 = N * 9 * F, NF = N * F;
  const int64_t E = 2 * P + N, RW = qd > 0 ? E : P1;  // rows of the edge-MLP buffers: pairs, or directed edges for TensorNet2
  Carver c(ws);
  HvpBuffers b;
  auto f = [&](int64_t n) { return c.take<float>(n); };
  b.phi = f(P1 * K); b.dphi = f(P1 * K); b.C = f(P1); b.dC = f(P1); b.d_t = f(P1); b.rhat_t = f(P1 * 3); b.phi_t = f(P1 * K); b.C_t = f(P1);
  b.kap = f(N);
  b.Q = f(P1 * 3 * F); b.Q_t = f(P1 * 3 * F); b.u0 = f(N9); b.u0_t = f(N9); b.s0n = f(NF); b.s0n_t = f(NF);
  b.ln0 = f(NF); b.xh0 = f(NF); b.rstd0 = f(N); b.ln0_t = f(NF); b.xh0_t = f(NF); b.rstd0_t = f(N);
  b.a1 = f(2 * NF); b.h1 = f(2 * NF); b.a1_t = f(2 * NF); b.h1_t = f(2 * NF);
  b.a2 = f(3 * NF); b.gates = f(3 * NF); b.a2_t = f(3 * NF); b.gates_t = f(3 * NF);
  b.UX = f(N9); b.UX_t = f(N9);
  for (int l = 0; l <= L; ++l) {
    b.X.push_back(f(N9));
    b.X_t.push_back(f(N9));
  }
  for (int l = 0; l < L; ++l) {
    HvpLayer y;
    y.pre1 = f(RW * F); y.he1 = f(RW * F); y.e1_t = f(RW * F); y.he1_t = f(RW * F);
    y.pre2 = f(RW * 2 * F); y.he2 = f(RW * 2 * F); y.e2_t = f(RW * 2 * F); y.he2_t = f(RW * 2 * F);
    y.pre3 = f(RW * 3 * F); y.e3_t = f(RW * 3 * F); y.w = f(RW * 3 * F); y.w_t = f(RW * 3 * F);
    y.Xh = f(N9); y.Xh_t = f(N9); y.Pn = f(N9); y.Pn_t = f(N9); y.Mi = f(N9); y.Mi_t = f(N9); y.Ch = f(N9); y.Ch_t = f(N9);
    y.D = f(N9); y.D_t = f(N9);
    b.lay.push_back(y);
  }
  b.feat = f(3 * NF); b.feat_t = f(3 * NF); b.lnr = f(3 * NF); b.xhr = f(3 * NF); b.rstdr = f(N); b.lnr_t = f(3 * NF); b.xhr_t = f(3 * NF);
  b.rstdr_t = f(N); b.al = f(NF); b.x = f(NF); b.al_t = f(NF); b.x_t = f(NF);
  b.ao = f(N * H); b.ao_t = f(N * H); b.g_ao = f(N * H); b.g_ao_t = f(N * H); b.headv = f(N * H);
  b.g_x = f(NF); b.g_x_t = f(NF); b.g_al = f(NF); b.g_al_t = f(NF);
  b.g_ln = f(3 * NF); b.g_ln_t = f(3 * NF); b.g_feat = f(3 * NF); b.g_feat_t = f(3 * NF);
  b.G = f(N9); b.G_t = f(N9); b.Gn = f(N9); b.Gn_t = f(N9);
  b.g_D = f(N9); b.g_D_t = f(N9); b.g_Ch = f(N9); b.g_Ch_t = f(N9); b.g_Mi = f(N9); b.g_Mi_t = f(N9);
  b.g_Pn = f(N9); b.g_Pn_t = f(N9); b.gXl = f(N9); b.gXl_t = f(N9);
  b.self_gw = f(3 * F); b.self_gw_t = f(3 * F);
  b.g3 = f(RW * 3 * F); b.g3_t = f(RW * 3 * F); b.gh2 = f(RW * 2 * F); b.gh2_t = f(RW * 2 * F); b.g2 = f(RW * 2 * F); b.g2_t = f(RW * 2 * F);
  b.gh1 = f(RW * F); b.gh1_t = f(RW * F); b.g1 = f(RW * F); b.g1_t = f(RW * F);
  b.g_UX = f(N9); b.g_UX_t = f(N9); b.g_a2 = f(3 * NF); b.g_a2_t = f(3 * NF); b.g_h1 = f(2 * NF); b.g_h1_t = f(2 * NF);
  b.g_a1 = f(2 * NF); b.g_a1_t = f(2 * NF); b.g_ln0 = f(NF); b.g_ln0_t = f(NF); b.g_s0n = f(NF); b.g_s0n_t = f(NF);
  b.g_u0l = f(N9); b.g_u0l_t = f(N9);
  b.gA = f(N * 10 * F); b.gA_t = f(N * 10 * F); b.gq = f(2 * P1 * 3 * F); b.gq_t = f(2 * P1 * 3 * F);
  b.selfq = f(NF); b.selfq_t = f(NF); b.gZu_t = f(NF); b.gZv_t = f(NF); b.onehot = f(N * Z);
  const int64_t big = std::max<int64_t>({6 * F * F, 3 * F * K, Z * F, H * F, 3 * F * F});
  b.part = f((int64_t)train_part_floats((int)std::max<int64_t>(RW, 5 * N), big));
  b.d2phi = f(P1 * K); b.d2C = f(P1); b.gC = f(P1); b.gC_t = f(P1); b.gphid = f(P1); b.gphid_t = f(P1); b.dQ = f(P1 * 3 * F); b.d2Q = f(P1 * 3 * F);
  b.de1 = f(P1 * F); b.d2e1 = f(P1 * F); b.gcp = f(RW * F); b.gcp_t = f(RW * F); b.ec = f(2 * P1 * 4); b.ec_t = f(2 * P1 * 4);
  b.gdel = f(P1 * 3); b.gdel_t = f(P1 * 3); b.gpos = f(N * 3);
  if (qd > 0) {  // TensorNet2: charge heads, the three-block first edge layer, Coulomb head
    HvpTn2& t = b.t2;
    const int64_t QC = (L + 1) * qd;
    t.erev = c.take<int>(E); t.eid = c.take<int>(E); t.pair_edge = c.take<int>(P1);
    t.charges = f(N * QC); t.charges_t = f(N * QC); t.g_charges = f(N * QC); t.g_charges_t = f(N * QC);
    t.Ap = f(P1 * F); t.Ap_t = f(P1 * F); t.Bt = f(NF); t.Bt_t = f(NF); t.Cs = f(NF); t.Cs_t = f(NF);
    t.gB = f(NF); t.gB_t = f(NF); t.gCs = f(NF); t.gCs_t = f(NF); t.gself = f(NF); t.gself_t = f(NF); t.gAp = f(P1 * F); t.gAp_t = f(P1 * F);
    t.g_cl = f(N * qd); t.g_cl_t = f(N * qd); t.gCe = f(E); t.gCe_t = f(E);
    t.e_c = f(N); t.e_c_t = f(N); t.g_q = f(N * QC); t.g_q_t = f(N * QC); t.gpos_c = f(N * 3); t.hv_c = f(N * 3); t.wsum = f(1);
    t.bs = f(B * 4 * qd); t.g_out = f(N * 2 * qd); t.g_out_t = f(N * 2 * qd); t.g_h2 = f(NF); t.g_h2_t = f(NF); t.g_a2 = f(NF); t.g_a2_t = f(NF);
    t.g_h1 = f(NF); t.g_h1_t = f(NF); t.g_a1 = f(NF); t.g_a1_t = f(NF); t.g_ln = f(3 * NF); t.g_ln_t = f(3 * NF); t.g_cf = f(3 * NF);
    t.g_cf_t = f(3 * NF);
    for (int h = 0; h <= L; ++h) {
      HvpCp k;
      k.cf = f(3 * NF); k.cf_t = f(3 * NF); k.ln = f(3 * NF); k.xh = f(3 * NF); k.rstd = f(N); k.ln_t = f(3 * NF); k.xh_t = f(3 * NF);
      k.rstd_t = f(N); k.a1 = f(NF); k.h1 = f(NF); k.a1_t = f(NF); k.h1_t = f(NF); k.a2 = f(NF); k.h2 = f(NF); k.a2_t = f(NF); k.h2_t = f(NF);
      k.out = f(N * 2 * qd); k.out_t = f(N * 2 * qd); k.sums = f(B * 4 * qd);
      t.cp.push_back(k);
    }
    for (int l = 0; l < L; ++l) {
      t.chl.push_back(f(N * qd));
      t.chl_t.push_back(f(N * qd));
    }
  }
  if (total) *total = c.off;
  return b;
}

// last call's buffers (developer / test hook tmdnet_hvp_debug_tensor: every intermediate by the name it has in this file)
thread_local HvpBuffers g_last;
thread_local int64_t g_lastN = -1, g_lastP = 0;
thread_local const tmdnet_model* g_last_model = nullptr;

}  // namespace

int tmdnet_hvp_debug_tensor(tmdnet_model* m, void* stream, const char* name, float* out, int64_t numel) {
  if (!m || !name) return TMDNET_ERR_INVALID;
  if (m->et) return et_hvp_debug_tensor(m, reinterpret_cast<hipStream_t>(stream), name, out, numel);
  if (g_last_model != m || g_lastN < 0) return fail(m, TMDNET_ERR_STATE, "no second-order pass has run on this handle (this thread)");
  const tmdnet_hparams& hp = m->hp;
  const int64_t F = hp.hidden_channels, K = hp.num_rbf, L = hp.num_layers, H = hp.head_hidden, N = g_lastN, P1 = g_lastP + 1;
  const int64_t N9 = N * 9 * F, NF = N * F;
  const HvpBuffers& b = g_last;
  std::map<std::string, std::pair<const float*, int64_t>> t;
#define T_(field, n) t[#field] = {b.field, (n)}
  T_(phi, P1 * K); T_(dphi, P1 * K); T_(C, P1); T_(dC, P1); T_(d_t, P1); T_(rhat_t, P1 * 3); T_(phi_t, P1 * K); T_(C_t, P1);
  T_(Q, P1 * 3 * F); T_(Q_t, P1 * 3 * F); T_(u0, N9); T_(u0_t, N9); T_(s0n, NF); T_(s0n_t, NF); T_(ln0, NF); T_(xh0, NF); T_(rstd0, N);
  T_(ln0_t, NF); T_(xh0_t, NF); T_(rstd0_t, N); T_(a1, 2 * NF); T_(h1, 2 * NF); T_(a1_t, 2 * NF); T_(h1_t, 2 * NF); T_(a2, 3 * NF);
  T_(gates, 3 * NF); T_(a2_t, 3 * NF); T_(gates_t, 3 * NF); T_(UX, N9); T_(UX_t, N9);
  T_(feat, 3 * NF); T_(feat_t, 3 * NF); T_(lnr, 3 * NF); T_(xhr, 3 * NF); T_(rstdr, N); T_(lnr_t, 3 * NF); T_(xhr_t, 3 * NF); T_(rstdr_t, N);
  T_(al, NF); T_(x, NF); T_(al_t, NF); T_(x_t, NF); T_(ao, N * H); T_(ao_t, N * H); T_(g_ao, N * H); T_(g_ao_t, N * H); T_(headv, N * H);
  T_(g_x, NF); T_(g_x_t, NF); T_(g_al, NF); T_(g_al_t, NF); T_(g_ln, 3 * NF); T_(g_ln_t, 3 * NF); T_(g_feat, 3 * NF); T_(g_feat_t, 3 * NF);
  T_(g_D, N9); T_(g_D_t, N9); T_(g_Ch, N9); T_(g_Ch_t, N9); T_(g_Mi, N9); T_(g_Mi_t, N9); T_(g_Pn, N9);
  T_(g_Pn_t, N9); T_(gXl, N9); T_(gXl_t, N9); T_(self_gw, 3 * F); T_(self_gw_t, 3 * F); T_(g3, P1 * 3 * F); T_(g3_t, P1 * 3 * F);
  T_(gh2, P1 * 2 * F); T_(gh2_t, P1 * 2 * F); T_(g2, P1 * 2 * F); T_(g2_t, P1 * 2 * F); T_(gh1, P1 * F); T_(gh1_t, P1 * F); T_(g1, P1 * F);
  T_(g1_t, P1 * F); T_(g_UX, N9); T_(g_UX_t, N9); T_(g_a2, 3 * NF); T_(g_a2_t, 3 * NF); T_(g_h1, 2 * NF); T_(g_h1_t, 2 * NF);
  T_(g_a1, 2 * NF); T_(g_a1_t, 2 * NF); T_(g_ln0, NF); T_(g_ln0_t, NF); T_(g_s0n, NF); T_(g_s0n_t, NF); T_(g_u0l, N9); T_(g_u0l_t, N9);
  T_(gA, N * 10 * F); T_(gA_t, N * 10 * F); T_(gq, 2 * P1 * 3 * F); T_(gq_t, 2 * P1 * 3 * F); T_(selfq, NF); T_(selfq_t, NF);
  T_(gZu_t, NF); T_(gZv_t, NF);
  T_(d2phi, P1 * K); T_(d2C, P1); T_(gC, P1); T_(gC_t, P1); T_(gphid, P1); T_(gphid_t, P1); T_(ec, 2 * P1 * 4); T_(ec_t, 2 * P1 * 4);
  T_(gdel, (P1 - 1) * 3); T_(gdel_t, (P1 - 1) * 3);
#undef T_
  for (int l = 0; l <= L; ++l) {
    t["X" + std::to_string(l)] = {b.X[l], N9};
    t["X_t" + std::to_string(l)] = {b.X_t[l], N9};
  }
  // the residual stream's adjoint ping-pongs between (G, Gn): after the sweep the one that reached the embedding is "G_emb"
  t["G_emb"] = {(L & 1) ? b.Gn : b.G, N9};
  t["G_emb_t"] = {(L & 1) ? b.Gn_t : b.G_t, N9};
  for (int l = 0; l < L; ++l) {
    const HvpLayer& y = b.lay[l];
    const std::string q = "l" + std::to_string(l) + ".";
#define L_(field, n) t[q + #field] = {y.field, (n)}
    L_(pre1, P1 * F); L_(he1, P1 * F); L_(e1_t, P1 * F); L_(he1_t, P1 * F); L_(pre2, P1 * 2 * F); L_(he2, P1 * 2 * F); L_(e2_t, P1 * 2 * F);
    L_(he2_t, P1 * 2 * F); L_(pre3, P1 * 3 * F); L_(e3_t, P1 * 3 * F); L_(w, P1 * 3 * F); L_(w_t, P1 * 3 * F); L_(Xh, N9); L_(Xh_t, N9);
    L_(Pn, N9); L_(Pn_t, N9); L_(Mi, N9); L_(Mi_t, N9); L_(Ch, N9); L_(Ch_t, N9); L_(D, N9); L_(D_t, N9);
#undef L_
  }
  if (m->tn2) {  // TensorNet2: the charge heads, the blocks of the first edge layer (last layer processed), the Coulomb head; edge-MLP
                 // rows of the "l*." tensors are the E = 2 P + N directed edges
    const int64_t qd = m->tn2->hp.q_dim, QC = (L + 1) * qd, E = 2 * (P1 - 1) + N;
    const HvpTn2& u = b.t2;
#define U_(field, n) t[#field] = {u.field, (n)}
    U_(charges, N * QC); U_(charges_t, N * QC); U_(g_charges, N * QC); U_(g_charges_t, N * QC); U_(g_q, N * QC); U_(g_q_t, N * QC);
    U_(e_c, N); U_(e_c_t, N); U_(gpos_c, N * 3); U_(hv_c, N * 3); U_(gCe, E); U_(gCe_t, E); U_(gAp, P1 * F); U_(gAp_t, P1 * F); U_(gB, NF);
    U_(gB_t, NF); U_(gCs, NF); U_(gCs_t, NF); U_(gself, NF); U_(gself_t, NF); U_(Ap, P1 * F); U_(Ap_t, P1 * F); U_(Bt, NF); U_(Cs, NF);
#undef U_
    for (int h = 0; h <= L; ++h) {
      const HvpCp& k = u.cp[h];
      const std::string q = "cp" + std::to_string(h) + ".";
#define C_(field, n) t[q + #field] = {k.field, (n)}
      C_(cf, 3 * NF); C_(cf_t, 3 * NF); C_(ln, 3 * NF); C_(ln_t, 3 * NF); C_(xh, 3 * NF); C_(xh_t, 3 * NF); C_(rstd, N); C_(rstd_t, N);
      C_(a1, NF); C_(a1_t, NF); C_(h1, NF); C_(h1_t, NF); C_(a2, NF); C_(a2_t, NF); C_(h2, NF); C_(h2_t, NF); C_(out, N * 2 * qd);
      C_(out_t, N * 2 * qd);
#undef C_
    }
    for (int l = 0; l < L; ++l) {
      const HvpLayer& y = b.lay[l];
      const std::string q = "l" + std::to_string(l) + ".";
      for (auto& kv : std::vector<std::pair<std::string, std::pair<const float*, int64_t>>>{
               {"pre1", {y.pre1, E * F}}, {"he1", {y.he1, E * F}}, {"e1_t", {y.e1_t, E * F}}, {"he1_t", {y.he1_t, E * F}},
               {"pre2", {y.pre2, E * 2 * F}}, {"he2", {y.he2, E * 2 * F}}, {"e2_t", {y.e2_t, E * 2 * F}}, {"he2_t", {y.he2_t, E * 2 * F}},
               {"pre3", {y.pre3, E * 3 * F}}, {"e3_t", {y.e3_t, E * 3 * F}}, {"w", {y.w, E * 3 * F}}, {"w_t", {y.w_t, E * 3 * F}}})
        t[q + kv.first] = kv.second;
    }
    for (const char* nm : {"g3", "g3_t"}) t[nm].second = E * 3 * F;
    for (const char* nm : {"gh2", "gh2_t", "g2", "g2_t"}) t[nm].second = E * 2 * F;
    for (const char* nm : {"gh1", "gh1_t", "g1", "g1_t"}) t[nm].second = E * F;
  }
  auto it = t.find(name);
  if (it == t.end()) return fail(m, TMDNET_ERR_INVALID, std::string("unknown second-order tensor: ") + name);
  if (!out) return (int)std::min<int64_t>(it->second.second, 0x7fffffff);  // size query
  if (numel != it->second.second) return fail(m, TMDNET_ERR_INVALID, "tensor " + std::string(name) + " has " + std::to_string(it->second.second) + " elements");
  HIP_TRY(m, hipMemcpyAsync(out, it->second.first, (size_t)numel * sizeof(float), hipMemcpyDeviceToDevice, reinterpret_cast<hipStream_t>(stream)));
  return TMDNET_OK;
}

int tmdnet_force_param_workspace_bytes(tmdnet_model* m, int64_t n_atoms, int64_t n_mol, int64_t n_pairs, size_t* bytes) {
  (void)n_mol;
  if (!m || !bytes || n_atoms < 0 || n_pairs < 0) return TMDNET_ERR_INVALID;
  if (m->et) return et_force_param_workspace_bytes(m, n_atoms, n_pairs, bytes);
  carve_hvp(nullptr, m->hp, n_atoms, n_pairs, bytes, m->tn2 ? m->tn2->hp.q_dim : 0, n_mol);
  return TMDNET_OK;
}

int tmdnet_force_param_grads(tmdnet_model* m, void* stream, void* graph_ws, void* ws, size_t ws_bytes, int64_t n_atoms, int64_t n_mol,
                             int64_t n_pairs, const int64_t* z, const int64_t* batch, const float* q, const float* v, float* grads, float* hv) {
  return tmdnet_loss_param_grads(m, stream, graph_ws, ws, ws_bytes, n_atoms, n_mol, n_pairs, z, batch, q, v, nullptr, grads, hv);
}

// ge == null: gradient of s = v . d(sum E)/d pos.  ge [n_mol] = d loss / d E: gradient of S = s - sum_m ge_m E_m, i.e. minus the whole
// gradient of loss(E, F) in ONE pass: the tangent adjoint minus the adjoint of sum_m ge_m E_m obeys the tangent adjoint's recursion,
// so only its seed at the head changes (head_dual; the head's last bias; the Coulomb head's adjoints for TensorNet2).
int tmdnet_loss_param_grads(tmdnet_model* m, void* stream, void* graph_ws, void* ws, size_t ws_bytes, int64_t n_atoms, int64_t n_mol,
                            int64_t n_pairs, const int64_t* z, const int64_t* batch, const float* q, const float* v, const float* ge,
                            float* grads, float* hv) {
  if (!m || !graph_ws || !ws || !v || !grads) return TMDNET_ERR_INVALID;
  if (ge && n_mol > 1 && !batch) return fail(m, TMDNET_ERR_INVALID, "an energy seed for several molecules needs batch");
  if (!m->finalized) return fail(m, TMDNET_ERR_STATE, "parameters not finalised");
  if (n_pairs < 0) return fail(m, TMDNET_ERR_INVALID, "the second-order pass needs the exact pair count (dynamic shapes)");
  recall_graph(m, graph_ws);
  if (m->graph_is_cell) return fail(m, TMDNET_ERR_STATE, "second-order pass: build the graph without the cell list");
  if (m->atom_w) return fail(m, TMDNET_ERR_INVALID, "atom weights (tmdnet_set_atom_weights) are implemented for TensorNet inference only");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (m->et) {
    if (q) return fail(m, TMDNET_ERR_INVALID, "the Equivariant Transformer takes no total charge (reference torchmd_et.py:188-196)");
    Graph ge_ = carve_graph(graph_ws, n_atoms, n_mol, (int64_t)m->hp.max_num_neighbors * n_atoms, nullptr);
    if (n_pairs > ge_.pcap) return fail(m, TMDNET_ERR_INVALID, "n_pairs out of range");
    if (m->graph_has_z) z = ge_.z_c;
    if (!z) return fail(m, TMDNET_ERR_INVALID, "z is required (here or in tmdnet_build_graph)");
    CurScope cur_e(m);
    return et_force_param_grads(m, s, ge_, ws, ws_bytes, n_atoms, n_mol, n_pairs, z, batch, v, ge, grads, hv);
  }
  const tmdnet_hparams& hp = m->hp;
  const int F = hp.hidden_channels, K = hp.num_rbf, L = hp.num_layers, Z = hp.max_z, H = hp.head_hidden, o3 = hp.group_o3;
  const int N = (int)n_atoms, B = (int)n_mol, P = (int)n_pairs, P1 = P + 1;
  const int64_t N9 = (int64_t)N * 9 * F;
  Graph g = carve_graph(graph_ws, n_atoms, n_mol, (int64_t)hp.max_num_neighbors * n_atoms, nullptr);
  if (n_pairs > g.pcap) return fail(m, TMDNET_ERR_INVALID, "n_pairs out of range");
  size_t need = 0;
  const Tn2Model* const T2 = m->tn2;  // TensorNet2 + Coulomb head: extra statements marked T2 below
  const int qd = T2 ? T2->hp.q_dim : 0, QC = (L + 1) * qd, E = 2 * P + N, RW = T2 ? E : P1;
  if (T2 && !batch) return fail(m, TMDNET_ERR_INVALID, "TensorNet2: batch is required");
  if (T2 && !m->g_pos) return fail(m, TMDNET_ERR_STATE, "TensorNet2: build the graph first (it keeps the positions for the Coulomb sum)");
  HvpBuffers b = carve_hvp(ws, hp, n_atoms, n_pairs, &need, qd, n_mol);
  if (need > ws_bytes) return fail(m, TMDNET_ERR_WORKSPACE, "second-order workspace too small: need " + std::to_string(need));
  if (m->graph_has_z) z = g.z_c;  // validated (clamped) by the graph phase
  if (!z) return fail(m, TMDNET_ERR_INVALID, "z is required (here or in tmdnet_build_graph)");
  const DevParams& W = m->P;
  CurScope cur_(m);
  g_gemm_cat = CAT_GEMM_NODE;  // exact row counts everywhere: no device-side row count
  g_mdev = nullptr;
  g_madd = 0;

  // gradient buffer: layout of tmdnet_param_grad_entry
  std::map<std::string, int64_t> off;
  int64_t total = 0;
  for (const auto& kv : (T2 ? tn2_train_layout(m) : param_grad_layout(m))) {
    off[kv.first] = total;
    total += (kv.second + 63) & ~int64_t(63);
  }
  launch_fill(grads, 0.f, total, s);
  auto at = [&](const std::string& k) { return grads + off.at(k); };
  const RowMap rH = rows_plain(H), rF = rows_plain(F), r2F = rows_plain(2 * F), r3F = rows_plain(3 * F), rK = rows_plain(K);
  const RowMap rc_[3] = {rows_comp(F, 1), rows_comp(F, 3), rows_comp(F, 5)};
  const int c0_[3] = {0, 1, 4}, nc_[3] = {1, 3, 5};
  // dW = g_y_t^T x + g_y^T x_t  and  db = colsum(g_y_t)  of a dense layer y = x W^T + b
  auto dense_grad = [&](const float* gy, const float* gy_t, RowMap mg, const float* x, const float* x_t, RowMap mx, int R, int Nout, int Kin,
                        const std::string& wkey, const std::string& bkey) {
    launch_tn_gemm_pair(s, gy_t, x, gy, x_t, mg, mx, nullptr, R, Nout, Kin, at(wkey), false, b.part);
    if (!bkey.empty()) launch_colsum(s, gy_t, mg, nullptr, mg, nullptr, nullptr, R, Nout, at(bkey), false, b.part);
  };
  // the three weight sets of a 9-component tensor linear out_c = in_c W_type(c)^T
  auto tensor_linear_grad = [&](const float* gO, const float* gO_t, const float* In, const float* In_t, const std::string& key) {
    for (int t = 0; t < 3; ++t) {
      const int64_t o = (int64_t)c0_[t] * F;
      launch_tn_gemm_pair(s, gO_t + o, In + o, gO + o, In_t + o, rc_[t], rc_[t], nullptr, N * nc_[t], F, F, at(key + std::to_string(t)), false,
                          b.part);
    }
  };
  // scale-vector gradients of a LayerNorm: d w = colsum(g_t xh + g xh_t), d b = colsum(g_t)
  auto ln_grad = [&](const float* gy, const float* gy_t, const float* xh, const float* xh_t, RowMap mr, int R, int W_, const std::string& wkey,
                     const std::string& bkey) {
    launch_colsum(s, gy_t, mr, xh, mr, nullptr, nullptr, R, W_, at(wkey), false, b.part);
    launch_colsum(s, gy, mr, xh_t, mr, nullptr, nullptr, R, W_, at(wkey), true, b.part);
    launch_colsum(s, gy_t, mr, nullptr, mr, nullptr, nullptr, R, W_, at(bkey), false, b.part);
  };
  const float* kap = nullptr;
  if (q && !T2) {  // TensorNet: charge factor 1 + 0.1 q of the group product / update; TensorNet2: q is the total charge of the equilibration
    launch_kappa(q, batch, N, B, b.kap, s);
    kap = b.kap;
  }
  HvpTn2& t2 = b.t2;
  // one ChargePredict head (T2): invariants of X -> LayerNorm -> MLP -> (c | f) -> equilibration -> column block h of `charges`
  auto charge_head = [&](int h, const float* X, const float* X_t) {
    const CpParams& cw = T2->cp[h];
    HvpCp& k = t2.cp[h];
    hvp::launch_cp_feat_dual(N, F, X, X_t, k.cf, k.cf_t, s);
    hvp::launch_ln_dual(N, 3 * F, k.cf, k.cf_t, cw.ln_w, cw.ln_b, k.ln, k.xh, k.rstd, k.ln_t, k.xh_t, k.rstd_t, s);
    gemm(s, k.ln, 3 * F, cw.W1, 3 * F, cw.b1, k.h1, F, N, F, 3 * F, GEMM_ACT_SILU, k.a1, F);
    gemm(s, k.ln_t, 3 * F, cw.W1, 3 * F, nullptr, k.a1_t, F, N, F, 3 * F);
    hvp::launch_silu_tangent((int64_t)N * F, k.a1, k.a1_t, k.h1_t, s);
    gemm(s, k.h1, F, cw.W2, F, cw.b2, k.h2, F, N, F, F, GEMM_ACT_SILU, k.a2, F);
    gemm(s, k.h1_t, F, cw.W2, F, nullptr, k.a2_t, F, N, F, F);
    hvp::launch_silu_tangent((int64_t)N * F, k.a2, k.a2_t, k.h2_t, s);
    gemm(s, k.h2, F, cw.W3, F, cw.b3, k.out, 2 * qd, N, 2 * qd, F);
    gemm(s, k.h2_t, F, cw.W3, F, nullptr, k.out_t, 2 * qd, N, 2 * qd, F);
    hvp::launch_cp_qeq_dual(g, N, B, qd, batch, q, k.out, k.out_t, k.sums, t2.charges, t2.charges_t, QC, h * qd, s);
  };
  // its adjoint, added into the residual stream's adjoint (G, G_t); g_ch = column block h of g_charges
  auto charge_head_bwd = [&](int h, const float* X, const float* X_t, float* Gx, float* Gx_t) {
    const CpParams& cw = T2->cp[h];
    HvpCp& k = t2.cp[h];
    const std::string c_ = "cp" + std::to_string(h) + ".";
    hvp::launch_cp_qeq_bwd_dual(g, N, B, qd, batch, q, k.out, k.out_t, k.sums, t2.g_charges, t2.g_charges_t, QC, h * qd, t2.bs, t2.g_out,
                                t2.g_out_t, s);
    dense_grad(t2.g_out, t2.g_out_t, rows_plain(2 * qd), k.h2, k.h2_t, rF, N, 2 * qd, F, c_ + "W3", c_ + "b3");
    gemm(s, t2.g_out, 2 * qd, cw.W3T, 2 * qd, nullptr, t2.g_h2, F, N, F, 2 * qd);
    gemm(s, t2.g_out_t, 2 * qd, cw.W3T, 2 * qd, nullptr, t2.g_h2_t, F, N, F, 2 * qd);
    hvp::launch_dsilu_dual((int64_t)N * F, t2.g_h2, t2.g_h2_t, k.a2, k.a2_t, t2.g_a2, t2.g_a2_t, s);
    dense_grad(t2.g_a2, t2.g_a2_t, rF, k.h1, k.h1_t, rF, N, F, F, c_ + "W2", c_ + "b2");
    gemm(s, t2.g_a2, F, cw.W2T, F, nullptr, t2.g_h1, F, N, F, F);
    gemm(s, t2.g_a2_t, F, cw.W2T, F, nullptr, t2.g_h1_t, F, N, F, F);
    hvp::launch_dsilu_dual((int64_t)N * F, t2.g_h1, t2.g_h1_t, k.a1, k.a1_t, t2.g_a1, t2.g_a1_t, s);
    dense_grad(t2.g_a1, t2.g_a1_t, rF, k.ln, k.ln_t, r3F, N, F, 3 * F, c_ + "W1", c_ + "b1");
    gemm(s, t2.g_a1, F, cw.W1T, F, nullptr, t2.g_ln, 3 * F, N, 3 * F, F);
    gemm(s, t2.g_a1_t, F, cw.W1T, F, nullptr, t2.g_ln_t, 3 * F, N, 3 * F, F);
    ln_grad(t2.g_ln, t2.g_ln_t, k.xh, k.xh_t, r3F, N, 3 * F, c_ + "ln_w", c_ + "ln_b");
    hvp::launch_lnbwd_dual(N, 3 * F, t2.g_ln, t2.g_ln_t, k.xh, k.xh_t, k.rstd, k.rstd_t, cw.ln_w, t2.g_cf, t2.g_cf_t, s);
    hvp::launch_cp_feat_bwd_dual(N, F, X, X_t, t2.g_cf, t2.g_cf_t, Gx, Gx_t, s);
  };
  if (T2) launch_edge_reverse(g, N, t2.erev, t2.eid, t2.pair_edge, s);

  // ================= geometry: radial functions per pair, tangent of the geometry along v
  launch_radial(g, P, RadialParams{W.means, W.betas, K, hp.cutoff_lower, hp.cutoff_upper}, b.phi, b.dphi, b.C, b.dC, s);
  hvp::launch_pair_tangent(g, P, K, v, b.dphi, b.dC, b.d_t, b.rhat_t, b.phi_t, b.C_t, s);
  if (hv) {  // position gradient: second derivatives of the radial functions, per-pair accumulators of the distance gradient
    hvp::launch_radial2(g, P, K, W.means, W.betas, hp.cutoff_lower, hp.cutoff_upper, b.d2phi, b.d2C, s);
    launch_fill(b.gC, 0.f, P1, s);
    launch_fill(b.gC_t, 0.f, P1, s);
    launch_fill(b.gphid, 0.f, P1, s);
    launch_fill(b.gphid_t, 0.f, P1, s);
  }

  // ================= embedding, forward (reference tensornet.py:543-619)
  gemm(s, b.phi, K, W.Wdp, K, W.bdp, b.Q, 3 * F, P1, 3 * F, K);
  gemm(s, b.phi_t, K, W.Wdp, K, nullptr, b.Q_t, 3 * F, P1, 3 * F, K);
  hvp::launch_embed_scatter_dual(g, N, F, P, z, W.Utab, W.Vtab, b.Q, b.Q_t, b.C, b.C_t, b.rhat_t, b.u0, b.u0_t, b.s0n, b.s0n_t, s);
  hvp::launch_ln_dual(N, F, b.s0n, b.s0n_t, W.ln0_w, W.ln0_b, b.ln0, b.xh0, b.rstd0, b.ln0_t, b.xh0_t, b.rstd0_t, s);
  gemm(s, b.ln0, F, W.L1, F, W.bL1, b.h1, 2 * F, N, 2 * F, F, GEMM_ACT_SILU, b.a1, 2 * F);
  gemm(s, b.ln0_t, F, W.L1, F, nullptr, b.a1_t, 2 * F, N, 2 * F, F);
  hvp::launch_silu_tangent((int64_t)N * 2 * F, b.a1, b.a1_t, b.h1_t, s);
  gemm(s, b.h1, 2 * F, W.L2, 2 * F, W.bL2, b.gates, 3 * F, N, 3 * F, 2 * F, GEMM_ACT_SILU, b.a2, 3 * F);
  gemm(s, b.h1_t, 2 * F, W.L2, 2 * F, nullptr, b.a2_t, 3 * F, N, 3 * F, 2 * F);
  hvp::launch_silu_tangent((int64_t)N * 3 * F, b.a2, b.a2_t, b.gates_t, s);
  tensor_linear(s, b.u0, W.Ue, b.UX, N, F);
  tensor_linear(s, b.u0_t, W.Ue, b.UX_t, N, F);
  hvp::launch_gate_mul_dual(N, F, b.UX, b.UX_t, b.gates, b.gates_t, b.X[0], b.X_t[0], s);
  if (T2) charge_head(0, b.X[0], b.X_t[0]);

  // ================= interaction layers, forward (reference tensornet.py:729-814)
  for (int l = 0; l < L; ++l) {
    const LayerP& q_ = W.layer[l];
    HvpLayer& y = b.lay[l];
    if (T2) {  // first layer per DIRECTED edge in three blocks: pre1[e] = (M1a phi + b1)[pair] + (M1b c)[target] + (M1c c)[source]
      const Tn2LayerP& q2 = T2->layer[l];
      launch_slice_cols(t2.charges, QC, l * qd, N, qd, t2.chl[l], s);
      launch_slice_cols(t2.charges_t, QC, l * qd, N, qd, t2.chl_t[l], s);
      gemm(s, b.phi, K, q2.M1a, K, q_.b1, t2.Ap, F, P1, F, K);
      gemm(s, b.phi_t, K, q2.M1a, K, nullptr, t2.Ap_t, F, P1, F, K);
      gemm(s, t2.chl[l], qd, q2.M1b, qd, nullptr, t2.Bt, F, N, F, qd);
      gemm(s, t2.chl_t[l], qd, q2.M1b, qd, nullptr, t2.Bt_t, F, N, F, qd);
      gemm(s, t2.chl[l], qd, q2.M1c, qd, nullptr, t2.Cs, F, N, F, qd);
      gemm(s, t2.chl_t[l], qd, q2.M1c, qd, nullptr, t2.Cs_t, F, N, F, qd);
      hvp::launch_tn2_edge_pre1_dual(g, E, N, F, t2.Ap, t2.Ap_t, t2.Bt, t2.Bt_t, t2.Cs, t2.Cs_t, y.pre1, y.e1_t, y.he1, y.he1_t, s);
    } else {
      gemm(s, b.phi, K, q_.M1, K, q_.b1, y.he1, F, P1, F, K, GEMM_ACT_SILU, y.pre1, F);
      gemm(s, b.phi_t, K, q_.M1, K, nullptr, y.e1_t, F, P1, F, K);
      hvp::launch_silu_tangent((int64_t)P1 * F, y.pre1, y.e1_t, y.he1_t, s);
    }
    gemm(s, y.he1, F, q_.M2, F, q_.b2, y.he2, 2 * F, RW, 2 * F, F, GEMM_ACT_SILU, y.pre2, 2 * F);
    gemm(s, y.he1_t, F, q_.M2, F, nullptr, y.e2_t, 2 * F, RW, 2 * F, F);
    hvp::launch_silu_tangent((int64_t)RW * 2 * F, y.pre2, y.e2_t, y.he2_t, s);
    gemm(s, y.he2, 2 * F, q_.M3, 2 * F, q_.b3, y.pre3, 3 * F, RW, 3 * F, 2 * F);
    gemm(s, y.he2_t, 2 * F, q_.M3, 2 * F, nullptr, y.e3_t, 3 * F, RW, 3 * F, 2 * F);
    if (T2) hvp::launch_tn2_w_dual(g, E, 3 * F, y.pre3, y.e3_t, b.C, b.C_t, y.w, y.w_t, s);
    else hvp::launch_w_dual(P1, 3 * F, y.pre3, y.e3_t, b.C, b.C_t, y.w, y.w_t, s);
    hvp::launch_norm_dual(N, F, b.X[l], b.X_t[l], y.Xh, y.Xh_t, s);
    tensor_linear(s, y.Xh, q_.V, y.Pn, N, F);
    tensor_linear(s, y.Xh_t, q_.V, y.Pn_t, N, F);
    // neighbour sums through the engine's plain sweep (out += sum_e w[pair(e)] src[col(e)]): Mi_t = sweep(w, Pn_t) + sweep(w_t, Pn)
    if (T2) {  // weights per directed edge: own sweep
      hvp::launch_edge_sweep2(g, N, F, nullptr, y.w, y.Pn, nullptr, nullptr, nullptr, y.Mi, s);
      hvp::launch_edge_sweep2(g, N, F, nullptr, y.w, y.Pn_t, y.w_t, y.Pn, nullptr, y.Mi_t, s);
    } else {
      launch_message_dual(g, N, F, y.w, y.w_t, y.Pn, y.Pn_t, y.Mi, y.Mi_t, false, s);  // one sweep: every pair row read once per row atom
    }
    hvp::launch_group_dual(N, F, y.Pn, y.Pn_t, y.Mi, y.Mi_t, kap, o3, y.Ch, y.Ch_t, s);
    tensor_linear(s, y.Ch, q_.V + 3, y.D, N, F);
    tensor_linear(s, y.Ch_t, q_.V + 3, y.D_t, N, F);
    hvp::launch_update_dual(N, F, y.Xh, y.Xh_t, y.D, y.D_t, kap, b.X[l + 1], b.X_t[l + 1], s);
    if (T2) charge_head(l + 1, b.X[l + 1], b.X_t[l + 1]);
  }

  // ================= readout and head, forward (reference tensornet.py:384-398, output_modules.py:108-117)
  hvp::launch_feat_dual(N, F, b.X[L], b.X_t[L], b.feat, b.feat_t, s);
  hvp::launch_ln_dual(N, 3 * F, b.feat, b.feat_t, W.lnr_w, W.lnr_b, b.lnr, b.xhr, b.rstdr, b.lnr_t, b.xhr_t, b.rstdr_t, s);
  gemm(s, b.lnr, 3 * F, W.Lin, 3 * F, W.bLin, b.x, F, N, F, 3 * F, GEMM_ACT_SILU, b.al, F);
  gemm(s, b.lnr_t, 3 * F, W.Lin, 3 * F, nullptr, b.al_t, F, N, F, 3 * F);
  hvp::launch_silu_tangent((int64_t)N * F, b.al, b.al_t, b.x_t, s);
  gemm(s, b.x, F, W.O1, F, W.bO1, b.ao, H, N, H, F);
  gemm(s, b.x_t, F, W.O1, F, nullptr, b.ao_t, H, N, H, F);
  hvp::launch_head_dual(N, H, b.ao, b.ao_t, W.O2, W.std, ge, batch, b.g_ao, b.g_ao_t, b.headv, s);

  // ================= reverse sweep with tangents: head and readout
  launch_colsum(s, b.headv, rH, nullptr, rH, nullptr, nullptr, N, H, at("O2"), false, b.part);  // d s / d bO2 = 0 (filled above)
  if (ge) hvp::launch_head_bias_seed(N, W.std, ge, batch, at("bO2"), s);                         // d S / d bO2 = - std sum_n ge[molecule(n)]
  dense_grad(b.g_ao, b.g_ao_t, rH, b.x, b.x_t, rF, N, H, F, "O1", "bO1");
  gemm(s, b.g_ao, H, W.O1T, H, nullptr, b.g_x, F, N, F, H);
  gemm(s, b.g_ao_t, H, W.O1T, H, nullptr, b.g_x_t, F, N, F, H);
  hvp::launch_dsilu_dual((int64_t)N * F, b.g_x, b.g_x_t, b.al, b.al_t, b.g_al, b.g_al_t, s);
  dense_grad(b.g_al, b.g_al_t, rF, b.lnr, b.lnr_t, r3F, N, F, 3 * F, "Lin", "bLin");
  gemm(s, b.g_al, F, W.LinT, F, nullptr, b.g_ln, 3 * F, N, 3 * F, F);
  gemm(s, b.g_al_t, F, W.LinT, F, nullptr, b.g_ln_t, 3 * F, N, 3 * F, F);
  ln_grad(b.g_ln, b.g_ln_t, b.xhr, b.xhr_t, r3F, N, 3 * F, "lnr_w", "lnr_b");
  hvp::launch_lnbwd_dual(N, 3 * F, b.g_ln, b.g_ln_t, b.xhr, b.xhr_t, b.rstdr, b.rstdr_t, W.lnr_w, b.g_feat, b.g_feat_t, s);
  hvp::launch_readout_bwd_dual(N, F, b.X[L], b.X_t[L], b.g_feat, b.g_feat_t, b.G, b.G_t, s);
  if (T2) {  // Coulomb head: adjoints of all charge channels (and its own pair geometry for H v), then the last charge head
    hvp::launch_coulomb_atom_dual(g, N, QC, batch, m->g_pos, v, m->g_box_mode ? m->g_box : nullptr, m->g_box_mode == 2 ? 1 : 0, t2.charges, t2.charges_t, T2->qweights,
                                  t2.wsum, T2->hp.coulomb_cutoff, T2->hp.coulomb_epsilon_solvent, 0.5f * 27.211386024367243f * 0.5291772105638411f,
                                  t2.e_c, t2.e_c_t, t2.g_q, t2.g_q_t, t2.gpos_c, t2.hv_c, s);
    if (ge) {  // the Coulomb pairs lie inside a molecule: its adjoints take their atom's factor
      hvp::launch_row_seed(N, QC, ge, batch, t2.g_q, t2.g_q_t, s);
      hvp::launch_row_seed(N, 3, ge, batch, t2.gpos_c, t2.hv_c, s);
    }
    hvp::launch_scale1((int64_t)N * QC, W.std, t2.g_q, t2.g_charges, s);
    hvp::launch_scale1((int64_t)N * QC, W.std, t2.g_q_t, t2.g_charges_t, s);
    charge_head_bwd(L, b.X[L], b.X_t[L], b.G, b.G_t);
    launch_fill(t2.gCe, 0.f, E, s);
    launch_fill(t2.gCe_t, 0.f, E, s);
  }

  // ================= reverse sweep with tangents: interaction layers
  float *G = b.G, *G_t = b.G_t, *Gn = b.Gn, *Gn_t = b.Gn_t;
  for (int l = L - 1; l >= 0; --l) {
    const LayerP& q_ = W.layer[l];
    HvpLayer& y = b.lay[l];
    const std::string t_ = "l" + std::to_string(l) + ".";
    hvp::launch_update_bwd_dual(N, F, G, G_t, y.D, y.D_t, kap, b.g_D, b.g_D_t, s);
    tensor_linear_grad(b.g_D, b.g_D_t, y.Ch, y.Ch_t, t_ + "Vb");
    tensor_linear(s, b.g_D, q_.VT + 3, b.g_Ch, N, F);
    tensor_linear(s, b.g_D_t, q_.VT + 3, b.g_Ch_t, N, F);
    // g_Pn = compose_T(g_Y) (written by the group-product adjoint) + sweep(w, g_Mi); the tangent adds sweep(w_t, g_Mi)
    if (T2) {  // adjoint wrt the sources with the weights of the REVERSE edge (col <- row); g_w per directed edge
      hvp::launch_group_bwd_dual(N, F, b.g_Ch, b.g_Ch_t, y.Pn, y.Pn_t, y.Mi, y.Mi_t, kap, o3, b.g_Mi, b.g_Mi_t, b.gXl, b.gXl_t, s);
      hvp::launch_edge_sweep2(g, N, F, t2.erev, y.w, b.g_Mi, nullptr, nullptr, b.gXl, b.g_Pn, s);
      hvp::launch_edge_sweep2(g, N, F, t2.erev, y.w, b.g_Mi_t, y.w_t, b.g_Mi, b.gXl_t, b.g_Pn_t, s);
      hvp::launch_tn2_edge_gw_dual(g, E, N, F, b.g_Mi, b.g_Mi_t, y.Pn, y.Pn_t, y.pre3, y.e3_t, b.C, b.C_t, b.g3, b.g3_t, hv ? b.gcp : nullptr,
                                   hv ? b.gcp_t : nullptr, s);
      if (hv) hvp::launch_edge_rowdot(g, E, F, F, b.gcp, b.gcp_t, nullptr, nullptr, b.d_t, true, t2.gCe, t2.gCe_t, s);
    } else {
    hvp::launch_group_bwd_dual(N, F, b.g_Ch, b.g_Ch_t, y.Pn, y.Pn_t, y.Mi, y.Mi_t, kap, o3, b.g_Mi, b.g_Mi_t, b.g_Pn, b.g_Pn_t, s);
    launch_message_dual(g, N, F, y.w, y.w_t, b.g_Mi, b.g_Mi_t, b.g_Pn, b.g_Pn_t, true, s);
    // edge MLP: g_w per pair (self pair: summed over the atoms, per irreducible type), back through silu(.) C, M3, M2, M1
    for (int k = 0; k < 3; ++k) {
      const int64_t o = (int64_t)c0_[k] * F;
      launch_colsum(s, b.g_Mi + o, rc_[k], y.Pn + o, rc_[k], nullptr, nullptr, N * nc_[k], F, b.self_gw + (int64_t)k * F, false, b.part);
      launch_colsum(s, b.g_Mi_t + o, rc_[k], y.Pn + o, rc_[k], nullptr, nullptr, N * nc_[k], F, b.self_gw_t + (int64_t)k * F, false, b.part);
      launch_colsum(s, b.g_Mi + o, rc_[k], y.Pn_t + o, rc_[k], nullptr, nullptr, N * nc_[k], F, b.self_gw_t + (int64_t)k * F, true, b.part);
    }
    hvp::launch_pair_gw_dual(g, P, F, b.g_Mi, b.g_Mi_t, y.Pn, y.Pn_t, b.self_gw, b.self_gw_t, y.pre3, y.e3_t, b.C, b.C_t, b.g3, b.g3_t,
                             hv ? b.gcp : nullptr, hv ? b.gcp_t : nullptr, s);
    if (hv) hvp::launch_pair_rowdot(P1, F, b.gcp, b.gcp_t, nullptr, nullptr, b.d_t, true, b.gC, b.gC_t, s);  // g_C[p] += sum g_w silu(e3)
    }  // !T2
    const float* const M3T_ = T2 ? T2->layer[l].M3T : q_.M3T;
    const float* const M2T_ = T2 ? T2->layer[l].M2T : q_.M2T;
    dense_grad(b.g3, b.g3_t, r3F, y.he2, y.he2_t, r2F, RW, 3 * F, 2 * F, t_ + "M2", t_ + "b2");
    gemm(s, b.g3, 3 * F, M3T_, 3 * F, nullptr, b.gh2, 2 * F, RW, 2 * F, 3 * F);
    gemm(s, b.g3_t, 3 * F, M3T_, 3 * F, nullptr, b.gh2_t, 2 * F, RW, 2 * F, 3 * F);
    hvp::launch_dsilu_dual((int64_t)RW * 2 * F, b.gh2, b.gh2_t, y.pre2, y.e2_t, b.g2, b.g2_t, s);
    dense_grad(b.g2, b.g2_t, r2F, y.he1, y.he1_t, rF, RW, 2 * F, F, t_ + "M1", t_ + "b1");
    gemm(s, b.g2, 2 * F, M2T_, 2 * F, nullptr, b.gh1, F, RW, F, 2 * F);
    gemm(s, b.g2_t, 2 * F, M2T_, 2 * F, nullptr, b.gh1_t, F, RW, F, 2 * F);
    hvp::launch_dsilu_dual((int64_t)RW * F, b.gh1, b.gh1_t, y.pre1, y.e1_t, b.g1, b.g1_t, s);
    if (T2) {  // the three blocks of the first layer: pair rows (self pair: summed over the atoms' self edges), target and source charges
      const Tn2LayerP& q2 = T2->layer[l];
      const RowMap rq = rows_plain(qd);
      hvp::launch_tn2_edge_reduce_dual(g, N, P, F, t2.erev, t2.pair_edge, b.g1, b.g1_t, t2.gB, t2.gB_t, t2.gCs, t2.gCs_t, t2.gself, t2.gself_t,
                                       t2.gAp, t2.gAp_t, s);
      launch_colsum(s, t2.gself, rF, nullptr, rF, nullptr, nullptr, N, F, t2.gAp + (int64_t)P * F, false, b.part);
      launch_colsum(s, t2.gself_t, rF, nullptr, rF, nullptr, nullptr, N, F, t2.gAp_t + (int64_t)P * F, false, b.part);
      dense_grad(t2.gAp, t2.gAp_t, rF, b.phi, b.phi_t, rK, P1, F, K, t_ + "M0", t_ + "b0");
      dense_grad(t2.gB, t2.gB_t, rF, t2.chl[l], t2.chl_t[l], rq, N, F, qd, t_ + "M0b", "");
      dense_grad(t2.gCs, t2.gCs_t, rF, t2.chl[l], t2.chl_t[l], rq, N, F, qd, t_ + "M0c", "");
      // charges of head l entered this layer's edge MLP: g_c += gB M1b + gCs M1c
      gemm(s, t2.gB, F, q2.M1bT, F, nullptr, t2.g_cl, qd, N, qd, F);
      gemm(s, t2.gCs, F, q2.M1cT, F, nullptr, t2.g_cl, qd, N, qd, F, GEMM_ACCUM);
      gemm(s, t2.gB_t, F, q2.M1bT, F, nullptr, t2.g_cl_t, qd, N, qd, F);
      gemm(s, t2.gCs_t, F, q2.M1cT, F, nullptr, t2.g_cl_t, qd, N, qd, F, GEMM_ACCUM);
      hvp::launch_add_cols(N, qd, t2.g_cl, t2.g_charges, QC, l * qd, s);
      hvp::launch_add_cols(N, qd, t2.g_cl_t, t2.g_charges_t, QC, l * qd, s);
      if (hv) {  // (g_Ap M1a) . phi' = g_Ap . (phi' M1a^T)
        gemm(s, b.dphi, K, q2.M1a, K, nullptr, b.de1, F, P1, F, K);
        gemm(s, b.d2phi, K, q2.M1a, K, nullptr, b.d2e1, F, P1, F, K);
        hvp::launch_pair_rowdot(P, F, t2.gAp, t2.gAp_t, b.de1, b.d2e1, b.d_t, true, b.gphid, b.gphid_t, s);
      }
    } else {
    dense_grad(b.g1, b.g1_t, rF, b.phi, b.phi_t, rK, P1, F, K, t_ + "M0", t_ + "b0");
    if (hv) {  // (g_e1 M1) . phi' = g_e1 . (phi' M1^T): the distance tangents of e1 instead of a K-wide adjoint of phi
      gemm(s, b.dphi, K, q_.M1, K, nullptr, b.de1, F, P1, F, K);
      gemm(s, b.d2phi, K, q_.M1, K, nullptr, b.d2e1, F, P1, F, K);
      hvp::launch_pair_rowdot(P1, F, b.g1, b.g1_t, b.de1, b.d2e1, b.d_t, true, b.gphid, b.gphid_t, s);
    }
    }
    tensor_linear_grad(b.g_Pn, b.g_Pn_t, y.Xh, y.Xh_t, t_ + "Va");
    tensor_linear(s, b.g_Pn, q_.VT, b.gXl, N, F);
    tensor_linear(s, b.g_Pn_t, q_.VT, b.gXl_t, N, F);
    hvp::launch_norm_bwd_dual(N, F, b.X[l], b.X_t[l], G, G_t, b.gXl, b.gXl_t, Gn, Gn_t, s);
    std::swap(G, Gn);
    std::swap(G_t, Gn_t);
    if (T2) charge_head_bwd(l, b.X[l], b.X_t[l], G, G_t);  // its channels went into the Coulomb term and into this layer's edge MLP
  }

  // ================= reverse sweep with tangents: embedding
  hvp::launch_gate_bwd_dual(N, F, G, G_t, b.UX, b.UX_t, b.gates, b.gates_t, b.a2, b.a2_t, b.g_UX, b.g_UX_t, b.g_a2, b.g_a2_t, s);
  dense_grad(b.g_a2, b.g_a2_t, r3F, b.h1, b.h1_t, r2F, N, 3 * F, 2 * F, "L2", "bL2");
  gemm(s, b.g_a2, 3 * F, W.L2T, 3 * F, nullptr, b.g_h1, 2 * F, N, 2 * F, 3 * F);
  gemm(s, b.g_a2_t, 3 * F, W.L2T, 3 * F, nullptr, b.g_h1_t, 2 * F, N, 2 * F, 3 * F);
  hvp::launch_dsilu_dual((int64_t)N * 2 * F, b.g_h1, b.g_h1_t, b.a1, b.a1_t, b.g_a1, b.g_a1_t, s);
  dense_grad(b.g_a1, b.g_a1_t, r2F, b.ln0, b.ln0_t, rF, N, 2 * F, F, "L1", "bL1");
  gemm(s, b.g_a1, 2 * F, W.L1T, 2 * F, nullptr, b.g_ln0, F, N, F, 2 * F);
  gemm(s, b.g_a1_t, 2 * F, W.L1T, 2 * F, nullptr, b.g_ln0_t, F, N, F, 2 * F);
  ln_grad(b.g_ln0, b.g_ln0_t, b.xh0, b.xh0_t, rF, N, F, "ln0_w", "ln0_b");
  hvp::launch_lnbwd_dual(N, F, b.g_ln0, b.g_ln0_t, b.xh0, b.xh0_t, b.rstd0, b.rstd0_t, W.ln0_w, b.g_s0n, b.g_s0n_t, s);
  tensor_linear_grad(b.g_UX, b.g_UX_t, b.u0, b.u0_t, "Ue");
  tensor_linear(s, b.g_UX, W.UeT, b.g_u0l, N, F);
  tensor_linear(s, b.g_UX_t, W.UeT, b.g_u0l_t, N, F);
  hvp::launch_embed_bwd_atom_dual(N, F, b.g_u0l, b.g_u0l_t, b.u0, b.u0_t, b.g_s0n, b.g_s0n_t, b.gA, b.gA_t, s);
  // edge weights W_k = C (U[z_i] + V[z_j]) (Wdp phi + bdp)_k: per direction of every pair, the self edges apart (I block only)
  const int64_t dir = (int64_t)P1 * 3 * F;
  hvp::launch_embed_edge_dual(g, N, F, P, z, W.Utab, W.Vtab, b.Q, b.Q_t, b.C, b.C_t, b.rhat_t, b.gA, b.gA_t, b.gq, b.gq_t, dir, b.selfq,
                              b.selfq_t, b.gZu_t, b.gZv_t, s);
  if (hv) {
    gemm(s, b.dphi, K, W.Wdp, K, nullptr, b.dQ, 3 * F, P1, 3 * F, K);
    gemm(s, b.d2phi, K, W.Wdp, K, nullptr, b.d2Q, 3 * F, P1, 3 * F, K);
    hvp::launch_pair_rowdot(P, 3 * F, b.gq, b.gq_t, b.dQ, b.d2Q, b.d_t, true, b.gphid, b.gphid_t, s);  // both directions of every pair
    hvp::launch_pair_rowdot(P, 3 * F, b.gq + dir, b.gq_t + dir, b.dQ, b.d2Q, b.d_t, true, b.gphid, b.gphid_t, s);
    const int64_t edir = (int64_t)P1 * 4;
    launch_fill(b.ec, 0.f, 2 * edir, s);
    launch_fill(b.ec_t, 0.f, 2 * edir, s);
    const int E = 2 * P + N;  // symmetric adjacency: both directions of every pair and one self edge per atom
    hvp::launch_edge_geom_dual(g, E, N, F, P, z, W.Utab, W.Vtab, b.Q, b.Q_t, b.C, b.C_t, b.rhat_t, b.gA, b.gA_t, b.ec, b.ec_t, edir, s);
    if (T2) hvp::launch_pair_from_edges(P, t2.pair_edge, t2.erev, t2.gCe, t2.gCe_t, b.gC, b.gC_t, s);  // both directions of a pair
    hvp::launch_geom_dual(g, P, b.d_t, b.rhat_t, b.dC, b.d2C, b.gC, b.gC_t, b.gphid, b.gphid_t, b.ec, b.ec_t, edir, b.gdel, b.gdel_t, s);
    hvp::launch_pair_to_atom(g, N, P, b.gdel_t, hv, s);
    if (T2) hvp::launch_axpy1((int64_t)N * 3, W.std, t2.hv_c, hv, s);  // the Coulomb term's own pair geometry
  }
  float* dWdp = at("Wdp");
  float* dbdp = at("bdp");
  launch_tn_gemm_pair(s, b.gq_t, b.phi, b.gq, b.phi_t, r3F, rK, nullptr, P, 3 * F, K, dWdp, false, b.part);              // direction i <- j
  launch_tn_gemm_pair(s, b.gq_t + dir, b.phi, b.gq + dir, b.phi_t, r3F, rK, nullptr, P, 3 * F, K, dWdp, true, b.part);   // direction j <- i
  launch_tn_gemm(s, b.selfq_t, rF, b.phi + (int64_t)P * K, rows_plain(0), nullptr, nullptr, N, F, K, dWdp, true, b.part);  // self pair: phi_t = 0
  launch_colsum(s, b.gq_t, r3F, nullptr, r3F, nullptr, nullptr, P, 3 * F, dbdp, false, b.part);
  launch_colsum(s, b.gq_t + dir, r3F, nullptr, r3F, nullptr, nullptr, P, 3 * F, dbdp, true, b.part);
  launch_colsum(s, b.selfq_t, rF, nullptr, rF, nullptr, nullptr, N, F, dbdp, true, b.part);
  launch_onehot(z, N, Z, b.onehot, s);
  launch_tn_gemm(s, b.onehot, rows_plain(Z), b.gZu_t, rF, nullptr, nullptr, N, Z, F, at("Utab"), false, b.part);
  launch_tn_gemm(s, b.onehot, rows_plain(Z), b.gZv_t, rF, nullptr, nullptr, N, Z, F, at("Vtab"), false, b.part);
  HIP_TRY(m, hipGetLastError());
  g_last = b;
  g_lastN = N;
  g_lastP = P;
  g_last_model = m;
  return TMDNET_OK;
}
