// Kernel schedule of the TensorNet2 + ScalarPlusWeightedCoulomb energy+force path (AceFF-2.0).
//
// Reference call stack: TorchMD_Net.forward (model.py:530-631) -> TensorNet2.forward (tensornet2.py:372-463) ->
// TensorEmbedding (shared with TensorNet) / ChargePredict (:49-157) / Interaction x L (:466-626) -> readout ->
// ScalarPlusWeightedCoulomb.pre_reduce (output_modules.py:440-606) -> reduce; forces by the hand-written reverse pass below
// (specification: oracle/tn2_torch.py, pinned to the unmodified reference; every adjoint checked against its autograd).
//
// What is TensorNet's and reused as is: graph, radial tables / radial kernel, embedding and its adjoint, tensor linears,
// normalisation / update / group-product kernels and adjoints, the CSR message sweeps (their weight rows go through
// Graph::epair: identity here, the reverse-edge map in the adjoint), readout, force gather.  What is new (tn_tn2.hip): the
// charge heads with per-molecule equilibration, the per-directed-edge MLP with the charge channels of both endpoints
// (forward as GEMMs on E rows after a gather-add first layer, reverse-mode through it: the weights are no longer a function
// of the distance alone), the Coulomb pair sum.  Distance gradients still use forward tangents where the dependence is on d
// only: the pair block M1a phi(d) of the first layer (table or value+tangent GEMM) and the cutoff C(d).
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "tn_model.h"
#include "tn_tn2.h"
#include "tn_tlin9.h"

using namespace tn;

namespace {

struct Tn2Buffers {
  FwdBuffers b;  // the tensors TensorNet's schedule has as well (same names)
  int *erev, *eid, *pair_edge;
  std::vector<float*> featq, lnq, xhq, rstdq, a1q, h1q, a2q, h2q, outq, chg, FuQ;  // ChargePredict heads, L + 1
  std::vector<float*> Ap, dAp, Bt, Cs, pre1, he1, pre2, he2, pre3, we, Ce;         // per layer
  float *chg_all, *ec, *eatom;
  float *g_q, *fcoul, *g_c, *g_ctmp, *g_outq, *g_h2q, *g_h1q, *g_lnq, *g_featq;
  float *g_pre3, *g_pre2, *g_pre1, *gCe_slots, *gB, *gCs, *gAp;
};

Tn2Buffers tn2_carve(void* ws, const tmdnet_model* m, int64_t N, int64_t B, int64_t P, int64_t E, bool bwd, size_t* total, int ntp) {
  Carver c(ws);
  Tn2Buffers t{};
  FwdBuffers& b = t.b;
  const tmdnet_hparams& hp = m->hp;
  const int64_t F = hp.hidden_channels, K = hp.num_rbf, L = hp.num_layers, H = hp.head_hidden, qd = m->tn2->hp.q_dim;
  const int64_t P1 = P + 1, N9 = N * 9 * F, QC = (L + 1) * qd;
  b.phi = c.take<float>(P1 * K);
  b.dphi = c.take<float>(P1 * K);
  b.C = c.take<float>(P1);
  b.dC = c.take<float>(P1);
  b.Q = c.take<float>(P1 * 3 * F);
  b.dQ = c.take<float>(bwd ? P1 * 3 * F : 0);
  b.u0 = c.take<float>(N9);
  b.s0n = c.take<float>(N * F);
  b.ln0 = c.take<float>(N * F);
  b.xh0 = c.take<float>(N * F);
  b.rstd0 = c.take<float>(N);
  b.a1 = c.take<float>(N * 2 * F);
  b.h1 = c.take<float>(N * 2 * F);
  b.a2 = c.take<float>(N * 3 * F);
  b.gates = c.take<float>(N * 3 * F);
  b.UX = c.take<float>(N9);
  for (int l = 0; l <= L; ++l) b.X.push_back(c.take<float>(N9));
  for (int l = 0; l < L; ++l) {
    b.Pn.push_back(c.take<float>(N9));
    b.Mi.push_back(c.take<float>(N9));
    b.D.push_back(c.take<float>(N9));
    t.Ap.push_back(c.take<float>(P1 * F));
    t.dAp.push_back(c.take<float>(bwd ? P1 * F : 0));
    t.Bt.push_back(c.take<float>(N * F));
    t.Cs.push_back(c.take<float>(N * F));
    t.pre1.push_back(c.take<float>(E * F));
    t.he1.push_back(c.take<float>(E * F));
    t.pre2.push_back(c.take<float>(E * 2 * F));
    t.he2.push_back(c.take<float>(E * 2 * F));
    t.pre3.push_back(c.take<float>(E * 3 * F));
    t.we.push_back(c.take<float>(E * 3 * F));
    t.Ce.push_back(c.take<float>(E));
  }
  for (int l = 0; l <= L; ++l) {
    t.featq.push_back(c.take<float>(N * 3 * F));
    t.lnq.push_back(c.take<float>(N * 3 * F));
    t.xhq.push_back(c.take<float>(N * 3 * F));
    t.rstdq.push_back(c.take<float>(N));
    t.a1q.push_back(c.take<float>(N * F));
    t.h1q.push_back(c.take<float>(N * F));
    t.a2q.push_back(c.take<float>(N * F));
    t.h2q.push_back(c.take<float>(N * F));
    t.outq.push_back(c.take<float>(N * 2 * qd));
    t.chg.push_back(c.take<float>(N * qd));
    t.FuQ.push_back(c.take<float>(B * 2 * qd));
  }
  b.Xh = c.take<float>(N9);
  b.Ch = c.take<float>(N9);
  b.feat = c.take<float>(N * 3 * F);
  b.lnr = c.take<float>(N * 3 * F);
  b.xhr = c.take<float>(N * 3 * F);
  b.rstdr = c.take<float>(N);
  b.al = c.take<float>(N * F);
  b.x = c.take<float>(N * F);
  b.ao = c.take<float>(N * H);
  b.ea = c.take<float>(N);
  b.skeys_s = c.take<unsigned>(P1);
  b.svals_s = c.take<int>(P1);
  b.shist = c.take<int>(65536 + 2);
  t.erev = c.take<int>(E);
  t.eid = c.take<int>(E);
  t.pair_edge = c.take<int>(P1);
  t.chg_all = c.take<float>(N * QC);
  t.ec = c.take<float>(N);
  t.eatom = c.take<float>(N);
  if (bwd) {
    b.g_ao = c.take<float>(N * H);
    b.g_al = c.take<float>(N * F);
    b.g_ln = c.take<float>(N * 3 * F);
    b.g_feat = c.take<float>(N * 3 * F);
    b.G = c.take<float>(N9);
    b.gD = c.take<float>(N9);
    b.gCh = c.take<float>(N9);
    b.gMi = c.take<float>(N9);
    b.gPn = c.take<float>(N9);
    b.gXl = c.take<float>(N9);
    b.gd = c.take<float>(P1);
    b.gUX = c.take<float>(N9);
    b.g_a2 = c.take<float>(N * 3 * F);
    b.g_a1 = c.take<float>(N * 2 * F);
    b.g_ln0 = c.take<float>(N * F);
    b.g_s0n = c.take<float>(N * F);
    b.g_u0l = c.take<float>(N9);
    b.gA = c.take<float>(N * 10 * F);
    b.g_rhat = c.take<float>(P1 * 3);
    b.g_delta = c.take<float>(P1 * 3);
    t.g_q = c.take<float>(N * QC);
    t.fcoul = c.take<float>(N * 3);
    t.g_c = c.take<float>(N * qd);
    t.g_ctmp = c.take<float>(N * qd);
    t.g_outq = c.take<float>(N * 2 * qd);
    t.g_h2q = c.take<float>(N * F);
    t.g_h1q = c.take<float>(N * F);
    t.g_lnq = c.take<float>(N * 3 * F);
    t.g_featq = c.take<float>(N * 3 * F);
    t.g_pre3 = c.take<float>(E * 3 * F);
    t.g_pre2 = c.take<float>(E * 2 * F);
    t.g_pre1 = c.take<float>(E * F);
    t.gCe_slots = c.take<float>(E * tn2_gw_slots((int)F));
    t.gB = c.take<float>(N * F);
    t.gCs = c.take<float>(N * F);
    t.gAp = c.take<float>(P1 * F);
  }
  b.mom = c.take<float>(ntp ? embed_rb_moment_elems(N, ntp, (int)K) : 0);
  b.gmom = c.take<float>(ntp && bwd ? embed_rb_gmoment_elems(N, ntp, (int)K) : 0);
  b.ps = c.take<float>(ntp ? P1 * 8 : 0);
  if (total) *total = c.off;
  return t;
}

}  // namespace

// parameter-gradient pass: gradient buffer layout and the extra buffers of a TensorNet2 handle
std::vector<std::pair<std::string, int64_t>> tn2_train_layout(const tmdnet_model* m) {
  const int64_t F = m->hp.hidden_channels, K = m->hp.num_rbf, L = m->hp.num_layers, H = m->hp.head_hidden, Z = m->hp.max_z,
                qd = m->tn2->hp.q_dim;
  std::vector<std::pair<std::string, int64_t>> e = {{"Wdp", 3 * F * K}, {"bdp", 3 * F}, {"Utab", Z * F}, {"Vtab", Z * F}, {"Ue0", F * F},
                                                    {"Ue1", F * F}, {"Ue2", F * F}, {"L1", 2 * F * F}, {"bL1", 2 * F}, {"L2", 6 * F * F},
                                                    {"bL2", 3 * F}, {"ln0_w", F}, {"ln0_b", F}};
  for (int l = 0; l < L; ++l) {
    const std::string t = "l" + std::to_string(l) + ".";
    for (auto kv : std::vector<std::pair<std::string, int64_t>>{{"M0", F * K}, {"M0b", F * qd}, {"M0c", F * qd}, {"b0", F}, {"M1", 2 * F * F},
                                                                 {"b1", 2 * F}, {"M2", 6 * F * F}, {"b2", 3 * F}})
      e.push_back({t + kv.first, kv.second});
    for (const char* ab : {"Va", "Vb"})
      for (int k = 0; k < 3; ++k) e.push_back({t + ab + std::to_string(k), F * F});
  }
  for (int h = 0; h <= L; ++h) {
    const std::string c = "cp" + std::to_string(h) + ".";
    for (auto kv : std::vector<std::pair<std::string, int64_t>>{{"ln_w", 3 * F}, {"ln_b", 3 * F}, {"W1", 3 * F * F}, {"b1", F}, {"W2", F * F},
                                                                 {"b2", F}, {"W3", 2 * qd * F}, {"b3", 2 * qd}})
      e.push_back({c + kv.first, kv.second});
  }
  for (auto kv : std::vector<std::pair<std::string, int64_t>>{{"lnr_w", 3 * F}, {"lnr_b", 3 * F}, {"Lin", 3 * F * F}, {"bLin", F}, {"O1", H * F},
                                                               {"bO1", H}, {"O2", H}, {"bO2", 1}})
    e.push_back(kv);
  return e;
}
void tn2_carve_train(void* ws, const tmdnet_model* m, int64_t N, int64_t P, int64_t E, TrainCtx* tc, size_t* total) {
  const int64_t F = m->hp.hidden_channels, K = m->hp.num_rbf, L = m->hp.num_layers, H = m->hp.head_hidden, Z = m->hp.max_z, P1 = P + 1,
                N9 = N * 9 * F;
  Carver c(ws);
  TrainCtx t;
  for (int l = 0; l < L; ++l) {
    t.Ch.push_back(c.take<float>(N9));
    t.Xh.push_back(c.take<float>(N9));
  }
  t.g1 = c.take<float>(N * F);  // self-edge rows of g_pre1
  t.gq = c.take<float>(2 * P1 * 3 * F);
  t.selfq = c.take<float>(N * F);
  t.gZu = c.take<float>(N * F);
  t.gZv = c.take<float>(N * F);
  t.onehot = c.take<float>(N * Z);
  t.head = c.take<float>(N * (H + 1));
  t.forces = c.take<float>(N * 3);
  const int64_t big = std::max<int64_t>({6 * F * F, 3 * F * K, Z * F, H * F, 3 * F * F});
  t.part = c.take<float>((int64_t)train_part_floats((int)std::max<int64_t>({P1, 5 * N, E}), big));
  if (tc) {
    const float* ge = tc->gE;
    float* gr = tc->grads;
    const int ph = tc->phase;
    *tc = t;
    tc->gE = ge;
    tc->grads = gr;
    tc->phase = ph;
  }
  if (total) *total = c.off;
}

int tn2_forward_workspace_bytes(const tmdnet_model* m, int64_t n_atoms, int64_t n_mol, int64_t n_pairs, int64_t n_edges, int32_t want_forces,
                                size_t* bytes) {
  if (n_edges < 0) n_edges = (int64_t)m->hp.max_num_neighbors * n_atoms;  // static shapes: the edge capacity
  const int ntp = rb_ntp(m, n_atoms, n_pairs);
  if (n_pairs < 0) n_pairs = ((int64_t)m->hp.max_num_neighbors * n_atoms) / 2 + 1;
  tn2_carve(nullptr, m, n_atoms, n_mol, n_pairs, n_edges, want_forces != 0, bytes, ntp);
  return TMDNET_OK;
}

int tn2_energy_forces(tmdnet_model* m, hipStream_t s, const Graph& g0, void* ws, size_t ws_bytes, int64_t n_atoms, int64_t n_mol,
                      int64_t n_pairs, const int64_t* z, const int64_t* batch, const float* q, int32_t want_forces, float* energy,
                      float* forces) {
  const tmdnet_hparams& hp = m->hp;
  const Tn2Model& T2 = *m->tn2;
  const int F = hp.hidden_channels, K = hp.num_rbf, L = hp.num_layers, H = hp.head_hidden, qd = T2.hp.q_dim, QC = (L + 1) * qd;
  const int N = (int)n_atoms, B = (int)n_mol;
  Graph g = g0;
  const int P = n_pairs >= 0 ? (int)n_pairs : (int)g.pcap, P1 = P + 1;
  const bool static_mode = n_pairs < 0;
  const int64_t E = static_mode ? g.ecap : m->lastE;  // row capacity of the per-edge arrays
  size_t need = 0;
  const int ntp = rb_ntp(m, n_atoms, n_pairs);  // > 0: embedding in the radial basis (tn_embed_rb.hip), no Q / dQ rows
  Tn2Buffers t = tn2_carve(ws, m, n_atoms, n_mol, P, E, want_forces != 0, &need, ntp);
  if (need > ws_bytes) return fail(m, TMDNET_ERR_WORKSPACE, "forward workspace too small: need " + std::to_string(need));
  FwdBuffers& b = t.b;
  const DevParams& W = m->P;
  const int o3 = hp.group_o3;
  const int* perm = m->graph_is_cell ? g.perm : nullptr;
  const bool bwd = want_forces != 0;
  if (T2.hp.coulomb_cutoff <= 0.f && m->g_box_mode != 0)
    return fail(m, TMDNET_ERR_INVALID, "PBC is not supported with coulomb_cutoff = None");  // output_modules.py:451-453
  const double Nd = N, Pd = P, Fd = F, Ed = (double)E, nodeB = Nd * 9 * Fd * 4;
  auto PAIR = [&]() { g_gemm_cat = CAT_GEMM_EDGE; g_mdev = static_mode ? g.counts : nullptr; g_madd = 1; };
  auto EDGE = [&]() { g_gemm_cat = CAT_GEMM_EDGE; g_mdev = static_mode ? g.counts + 1 : nullptr; g_madd = 0; };
  auto NODE = [&]() { g_gemm_cat = CAT_GEMM_NODE; g_mdev = nullptr; g_madd = 0; };
  const int Erows = (int)E;

  // ---- graph extras: reverse-edge map, identity edge ids, lower edge of every pair
  KR(CAT_GRAPH, Ed * 16, launch_edge_reverse(g, N, t.erev, t.eid, t.pair_edge, s));
  Graph gE = g, gR = g;  // the sweeps index their weight rows through epair: per directed edge here
  gE.epair = t.eid;
  gR.epair = t.erev;

  // ---- per-pair functions of the distance: Q (embedding) and the pair block of every layer's first edge-MLP layer
  TrainCtx* const tc = m->train;  // parameter gradients: direct radial functions, per-layer X_hat / group product kept
  const bool use_tab = !tc && m->tabs.ok && (int64_t)P1 >= m->tab_min_pairs && (int)m->tabs.tab.size() == 1 + L;
  if (use_tab) {
    const float* t0[1] = {m->tabs.tab[0]};
    float* o0[1] = {b.Q};
    float* d0[1] = {bwd ? b.dQ : nullptr};
    std::vector<const float*> tl;
    std::vector<float*> ol, dl;
    for (int l = 0; l < L; ++l) {
      tl.push_back(m->tabs.tab[1 + l]);
      ol.push_back(t.Ap[l]);
      dl.push_back(bwd ? t.dAp[l] : nullptr);
    }
    KR(CAT_EDGE_TABLE, (Pd + 1) * (12 * Fd + 4 * Fd * L) * (bwd ? 2 : 1),
       (launch_pair_buckets(g, P, hp.cutoff_lower, hp.cutoff_upper, m->tabs.T, b.C, b.dC, b.shist, b.skeys_s, b.svals_s, s),
        ntp ? (void)0
            : launch_edge_interp(g, P, hp.cutoff_lower, hp.cutoff_upper, m->tabs.T, 3 * F, 1, t0, o0, d0, b.skeys_s, b.svals_s, s, b.C, b.dC),
        L ? launch_edge_interp(g, P, hp.cutoff_lower, hp.cutoff_upper, m->tabs.T, F, L, tl.data(), ol.data(), dl.data(), b.skeys_s,
                               b.svals_s, s, ntp ? b.C : nullptr, ntp ? b.dC : nullptr)
          : (void)0));
  } else {
    RadialParams rp{W.means, W.betas, K, hp.cutoff_lower, hp.cutoff_upper};
    KR(CAT_ELEMENTWISE, Pd * (2 * K + 3) * 4, launch_radial(g, P, rp, b.phi, b.dphi, b.C, b.dC, s));
    PAIR();
    if (ntp) {
    } else if (bwd) gemm_dual(s, 0, b.phi, b.dphi, K, W.Wdp, W.bdp, b.Q, b.dQ, 3 * F, P1, 3 * F, K, nullptr, nullptr, W.Wdp_sb);
    else gemm(s, b.phi, K, W.Wdp, K, W.bdp, b.Q, 3 * F, P1, 3 * F, K);
    for (int l = 0; l < L; ++l) {
      if (bwd) gemm_dual(s, 0, b.phi, b.dphi, K, T2.layer[l].M1a, W.layer[l].b1, t.Ap[l], t.dAp[l], F, P1, F, K, nullptr, nullptr, nullptr);
      else gemm(s, b.phi, K, T2.layer[l].M1a, K, W.layer[l].b1, t.Ap[l], F, P1, F, K);
    }
  }

  // ---- embedding (TensorNet's: tensornet.py:543-619)
  const RadialParams rbp{W.means, W.betas, K, hp.cutoff_lower, hp.cutoff_upper};
  const double momB = (double)embed_rb_moment_elems(N, ntp ? ntp : 4, K) * 4;
  if (ntp) {
    KR(CAT_SCATTER, Ed * 12 + Pd * 48 + momB,
       (launch_pair_scalars(g, P, hp.cutoff_lower, hp.cutoff_upper, b.ps, s), launch_embed_moments(g, N, rbp, ntp, b.ps, b.mom, s)));
    KR(CAT_SCATTER, momB + Nd * 10 * Fd * 4, launch_embed_combine(g, N, F, K, ntp, z, W.Utab, W.Vtab, m->rb_fwd, b.mom, b.u0, b.s0n, s));
  } else {
    KR(CAT_SCATTER, Pd * 12 * Fd + Ed * 12 + Nd * 10 * Fd * 4, launch_embed_scatter(g, N, F, z, W.Utab, W.Vtab, b.Q, b.C, b.u0, b.s0n, s));
  }
  KR(CAT_ELEMENTWISE, Nd * Fd * 12, launch_layernorm_fwd(b.s0n, W.ln0_w, W.ln0_b, N, F, b.ln0, b.xh0, b.rstd0, s));
  NODE();
  gemm(s, b.ln0, F, W.L1, F, W.bL1, b.h1, 2 * F, N, 2 * F, F, GEMM_ACT_SILU, b.a1, 2 * F);
  gemm(s, b.h1, 2 * F, W.L2, 2 * F, W.bL2, b.gates, 3 * F, N, 3 * F, 2 * F, GEMM_ACT_SILU, b.a2, 3 * F);
  // fused 9-component tensor linears at batch scale (tn_tlin9.hip, as in tn_api.hip); the parameter-gradient pass keeps the
  // unfused schedule (its weight-gradient products read the intermediates)
  const bool t9 = !tc && tlin9_ok(N, F) && tlin9_images(W.Ue) && tlin9_images(W.UeT) && (L == 0 || (tlin9_images(W.layer[0].V) && tlin9_images(W.layer[0].VT)));
  if (t9) {
    Tl9Args ta{};
    ta.A = b.u0; ta.C = b.X[0]; ta.o1 = b.UX; ta.e3 = b.gates; ta.N = N; ta.F = F;
    tlin9(s, TL9_PRO_PLAIN, TL9_EPI_MULGATE, W.Ue, ta, 3.0 + 1.0 / 3.0, "gate");
  } else
    tensor_linear(s, b.u0, W.Ue, b.X[0], N, F, GEMM_MUL_AUX, b.UX, b.gates);

  // ---- ChargePredict head l on X[l] (tensornet2.py:139-157)
  auto charge_predict = [&](int l) {
    const CpParams& cp = T2.cp[l];
    KR(CAT_ELEMENTWISE, nodeB + Nd * 3 * Fd * 4, launch_cp_feat(b.X[l], N, F, t.featq[l], s));
    KR(CAT_ELEMENTWISE, Nd * 3 * Fd * 12, launch_layernorm_fwd(t.featq[l], cp.ln_w, cp.ln_b, N, 3 * F, t.lnq[l], t.xhq[l], t.rstdq[l], s));
    NODE();
    gemm(s, t.lnq[l], 3 * F, cp.W1, 3 * F, cp.b1, t.h1q[l], F, N, F, 3 * F, GEMM_ACT_SILU, t.a1q[l], F);
    gemm(s, t.h1q[l], F, cp.W2, F, cp.b2, t.h2q[l], F, N, F, F, GEMM_ACT_SILU, t.a2q[l], F);
    gemm(s, t.h2q[l], F, cp.W3, F, cp.b3, t.outq[l], 2 * qd, N, 2 * qd, F);
    KR(CAT_ELEMENTWISE, Nd * qd * 12, launch_qeq_fwd(g, t.outq[l], q, batch, N, B, qd, t.chg[l], t.FuQ[l], s));
    KR(CAT_ELEMENTWISE, Nd * qd * 8, launch_put_cols(t.chg[l], N, qd, t.chg_all, QC, l * qd, s));
  };
  charge_predict(0);

  // ---- interaction layers (tensornet2.py:536-626)
  for (int l = 0; l < L; ++l) {
    const LayerP& q_ = W.layer[l];
    const Tn2LayerP& q2 = T2.layer[l];
    NODE();
    gemm(s, t.chg[l], qd, q2.M1b, qd, nullptr, t.Bt[l], F, N, F, qd);   // target-atom block of the first layer
    gemm(s, t.chg[l], qd, q2.M1c, qd, nullptr, t.Cs[l], F, N, F, qd);   // source-atom block
    KR(CAT_ELEMENTWISE, Ed * Fd * 8 + Pd * Fd * 4, launch_tn2_edge_pre1(g, N, F, t.Ap[l], t.Bt[l], t.Cs[l], b.C, t.pre1[l], t.he1[l], t.Ce[l], s));
    EDGE();
    gemm(s, t.he1[l], F, q_.M2, F, q_.b2, t.he2[l], 2 * F, Erows, 2 * F, F, GEMM_ACT_SILU, t.pre2[l], 2 * F);
    gemm(s, t.he2[l], 2 * F, q_.M3, 2 * F, q_.b3, t.we[l], 3 * F, Erows, 3 * F, 2 * F, GEMM_ACT_SILU | GEMM_ROWSCALE, t.pre3[l], 3 * F,
         nullptr, 0, t.Ce[l]);
    float* const Xh_l = tc ? tc->Xh[l] : b.Xh;
    float* const Ch_l = tc ? tc->Ch[l] : b.Ch;
    if (t9) {
      Tl9Args ta{};
      ta.A = b.X[l]; ta.C = b.Pn[l]; ta.N = N; ta.F = F;
      tlin9(s, TL9_PRO_NORM, TL9_EPI_PLAIN, q_.V, ta, 2.0, "norm");
      KR(CAT_MESSAGE, Ed * (12 * Fd + 12) + 3 * nodeB, launch_message(gE, N, F, t.we[l], b.Pn[l], nullptr, nullptr, o3, b.Mi[l], Ch_l, s));
      Tl9Args tb{};
      tb.A = Ch_l; tb.C = b.D[l]; tb.e0 = b.X[l]; tb.o1 = b.X[l + 1]; tb.o2 = b.feat; tb.want_feat = l + 1 == L; tb.N = N; tb.F = F;
      tlin9(s, TL9_PRO_PLAIN, TL9_EPI_UPDATE, q_.V + 3, tb, 4.0 + (l + 1 == L ? 1.0 / 3.0 : 0.0), "update");
    } else {
    KR(CAT_ELEMENTWISE, 2 * nodeB, launch_norm_x(b.X[l], Xh_l, N, F, s));
    tensor_linear(s, Xh_l, q_.V, b.Pn[l], N, F);
    KR(CAT_MESSAGE, Ed * (12 * Fd + 12) + 3 * nodeB, launch_message(gE, N, F, t.we[l], b.Pn[l], nullptr, nullptr, o3, b.Mi[l], Ch_l, s));
    tensor_linear(s, Ch_l, q_.V + 3, b.D[l], N, F);
    // X_new = X_hat + dX + dX dX (no charge factor, tensornet2.py:624); the readout invariants come with the last layer
    KR(CAT_ELEMENTWISE, 4 * nodeB, launch_layer_update(Xh_l, b.D[l], nullptr, nullptr, N, F, b.X[l + 1], l + 1 < L ? 0 : 2, b.feat, s));
    }
    charge_predict(l + 1);
  }

  // ---- readout + head + Coulomb + per-molecule sum
  if (L == 0) KR(CAT_ELEMENTWISE, nodeB + Nd * 3 * Fd * 4, launch_readout_feat(b.X[L], N, F, b.feat, s));
  KR(CAT_ELEMENTWISE, Nd * 3 * Fd * 12, launch_layernorm_fwd(b.feat, W.lnr_w, W.lnr_b, N, 3 * F, b.lnr, b.xhr, b.rstdr, s));
  NODE();
  gemm(s, b.lnr, 3 * F, W.Lin, 3 * F, W.bLin, b.x, F, N, F, 3 * F, GEMM_ACT_SILU, b.al, F);
  gemm(s, b.x, F, W.O1, F, W.bO1, b.ao, H, N, H, F);
  const float* cpos = m->graph_is_cell ? g.pos_s : m->g_pos;
  const float* cbox = m->graph_is_cell ? g.boxd : m->g_box;
  const int64_t* cbatch = (m->graph_is_cell && !m->graph_cell_multi) ? nullptr : batch;  // several molecules in cell order: their internal batch
  if (!cpos) return fail(m, TMDNET_ERR_STATE, "TensorNet2 needs the positions of tmdnet_build_graph for its Coulomb head");
  {
    ProfScope ps_(s, CAT_PAIR, 0.0, Nd * (12 + 8.0 * QC), "launch_coulomb");
    if (launch_coulomb(g, cpos, cbatch, cbox, m->g_box_mode, N, B, QC, t.chg_all, T2.qweights, T2.hp.coulomb_cutoff,
                       T2.hp.coulomb_epsilon_solvent, W.std, t.ec, bwd ? t.g_q : nullptr, bwd ? t.fcoul : nullptr, s))
      return fail(m, TMDNET_ERR_INVALID, "too many charge channels for the Coulomb kernel");
  }
  KR(CAT_ELEMENTWISE, Nd * H * 4, launch_tn2_head_energy(b.ao, W.O2, W.bO2, t.ec, N, H, W.std, W.atomref, z, t.eatom, s));
  KR(CAT_ELEMENTWISE, Nd * 4, launch_mol_sum(g, t.eatom, batch, N, B, W.mean, energy, s));

  if (bwd) {
    // ---- head + readout adjoints (TensorNet's)
    KR(CAT_ELEMENTWISE, Nd * H * 8, launch_head_bwd(b.ao, W.O2, N, H, W.std, b.g_ao, s));
    NODE();
    // parameter gradients: dW (+)= g_out^T in, db = colsum(g_out) wherever an adjoint meets its layer's input (tn_train.hip)
    auto RP = [](int64_t ld) { return rows_plain(ld); };
    auto dW = [&](const std::string& key, const float* gOut, int64_t ldg, const float* In, int64_t ldi, int R, int Nout, int Kin,
                  bool acc = false) { launch_tn_gemm(s, gOut, RP(ldg), In, RP(ldi), nullptr, nullptr, R, Nout, Kin, tc->at(key), acc, tc->part); };
    auto dB = [&](const std::string& key, const float* gOut, int64_t ldg, int R, int ncol, const float* mul = nullptr, int64_t ldm = 0,
                  bool acc = false) { launch_colsum(s, gOut, RP(ldg), mul, RP(ldm), nullptr, nullptr, R, ncol, tc->at(key), acc, tc->part); };
    const RowMap rc_[3] = {rows_comp(F, 1), rows_comp(F, 3), rows_comp(F, 5)};
    const int c0_[3] = {0, 1, 4}, nc_[3] = {1, 3, 5};
    auto tensor_linear_grad = [&](const float* gOut, const float* In, const std::string& key) {
      for (int k = 0; k < 3; ++k)
        launch_tn_gemm(s, gOut + (int64_t)c0_[k] * F, rc_[k], In + (int64_t)c0_[k] * F, rc_[k], nullptr, nullptr, N * nc_[k], F, F,
                       tc->at(key + std::to_string(k)), false, tc->part);
    };
    if (tc) {
      // seeds d loss / d E_mol(i): rows of g_ao and of the Coulomb charge gradient; everything below is linear in the two
      launch_train_seed(b.ao, tc->gE, batch, N, H, W.std, b.g_ao, tc->head, s);
      launch_slice_cols(tc->head, H + 1, H, N, 1, tc->selfq, s);                     // s_i = seed * std
      launch_scale_rows(t.g_q, tc->selfq, 1.0f / W.std, N, QC, s);                   // g_q was computed with seed 1
      dB("O2", tc->head, H + 1, N, H);
      dB("bO2", tc->head + H, H + 1, N, 1);
      dW("O1", b.g_ao, H, b.x, F, N, H, F);
      dB("bO1", b.g_ao, H, N, H);
    }
    gemm(s, b.g_ao, H, W.O1T, H, nullptr, b.g_al, F, N, F, H, GEMM_MUL_DSILU_AUX, nullptr, 0, b.al, F);
    gemm(s, b.g_al, F, W.LinT, F, nullptr, b.g_ln, 3 * F, N, 3 * F, F);
    if (tc) {
      dW("Lin", b.g_al, F, b.lnr, 3 * F, N, F, 3 * F);
      dB("bLin", b.g_al, F, N, F);
      dB("lnr_w", b.g_ln, 3 * F, N, 3 * F, b.xhr, 3 * F);
      dB("lnr_b", b.g_ln, 3 * F, N, 3 * F);
    }
    KR(CAT_ELEMENTWISE, Nd * 3 * Fd * 12, launch_layernorm_bwd(b.g_ln, b.xhr, b.rstdr, W.lnr_w, N, 3 * F, b.g_feat, s));
    KR(CAT_ELEMENTWISE, 2 * nodeB + Nd * 3 * Fd * 4, launch_readout_bwd(b.X[L], b.g_feat, N, F, b.G, s));
    launch_fill(b.gd, 0.f, P1, s);

    // ---- adjoint of ChargePredict head l: g_c [N, qd] -> adds into G (the gradient wrt X[l])
    auto charge_predict_bwd = [&](int l) {
      const CpParams& cp = T2.cp[l];
      KR(CAT_ELEMENTWISE, Nd * qd * 16, launch_qeq_bwd(g, t.outq[l], batch, N, B, qd, t.FuQ[l], t.g_c, t.g_outq, s));
      NODE();
      const std::string c_ = "cp" + std::to_string(l) + ".";
      if (tc) {
        dW(c_ + "W3", t.g_outq, 2 * qd, t.h2q[l], F, N, 2 * qd, F);
        dB(c_ + "b3", t.g_outq, 2 * qd, N, 2 * qd);
      }
      gemm(s, t.g_outq, 2 * qd, cp.W3T, 2 * qd, nullptr, t.g_h2q, F, N, F, 2 * qd, GEMM_MUL_DSILU_AUX, nullptr, 0, t.a2q[l], F);
      if (tc) {
        dW(c_ + "W2", t.g_h2q, F, t.h1q[l], F, N, F, F);
        dB(c_ + "b2", t.g_h2q, F, N, F);
      }
      gemm(s, t.g_h2q, F, cp.W2T, F, nullptr, t.g_h1q, F, N, F, F, GEMM_MUL_DSILU_AUX, nullptr, 0, t.a1q[l], F);
      if (tc) {
        dW(c_ + "W1", t.g_h1q, F, t.lnq[l], 3 * F, N, F, 3 * F);
        dB(c_ + "b1", t.g_h1q, F, N, F);
      }
      gemm(s, t.g_h1q, F, cp.W1T, F, nullptr, t.g_lnq, 3 * F, N, 3 * F, F);
      if (tc) {
        dB(c_ + "ln_w", t.g_lnq, 3 * F, N, 3 * F, t.xhq[l], 3 * F);
        dB(c_ + "ln_b", t.g_lnq, 3 * F, N, 3 * F);
      }
      KR(CAT_ELEMENTWISE, Nd * 3 * Fd * 12, launch_layernorm_bwd(t.g_lnq, t.xhq[l], t.rstdq[l], cp.ln_w, N, 3 * F, t.g_featq, s));
      KR(CAT_ELEMENTWISE, 2 * nodeB + Nd * 3 * Fd * 4, launch_cp_feat_bwd(b.X[l], t.g_featq, N, F, b.G, s));
    };
    // gradient wrt the charges of head l: its Coulomb slice (+ what the edge MLP of layer l contributes, added below)
    auto coulomb_slice = [&](int l) { KR(CAT_ELEMENTWISE, Nd * qd * 8, launch_slice_cols(t.g_q, QC, l * qd, N, qd, t.g_c, s)); };

    coulomb_slice(L);
    charge_predict_bwd(L);
    for (int l = L - 1; l >= 0; --l) {
      const LayerP& q_ = W.layer[l];
      const Tn2LayerP& q2 = T2.layer[l];
      // G = gradient wrt X[l + 1]
      const std::string t_ = "l" + std::to_string(l) + ".";
      if (t9) {
        Tl9Args ta{};
        ta.A = b.G; ta.A2 = b.D[l]; ta.N = N; ta.F = F; ta.C = b.gCh;
        tlin9(s, TL9_PRO_UPDBWD, TL9_EPI_PLAIN, q_.VT + 3, ta, 3.0, "updbwd");
      } else {
      KR(CAT_ELEMENTWISE, 3 * nodeB, launch_update_bwd(b.G, b.D[l], nullptr, nullptr, N, F, b.gD, s));
      if (tc) tensor_linear_grad(b.gD, tc->Ch[l], t_ + "Vb");
      tensor_linear(s, b.gD, q_.VT + 3, b.gCh, N, F);
      }
      KR(CAT_ELEMENTWISE, 5 * nodeB, launch_message_bwd_node(b.gCh, b.Pn[l], b.Mi[l], nullptr, nullptr, o3, N, F, b.gMi, b.gPn, s));
      // per-edge weight gradient -> g_pre3, cutoff gradient slots; then the transposed sweep with w[erev[e]]
      KR(CAT_PAIR, Ed * (24 * Fd + 12) + 2 * nodeB,
         launch_tn2_edge_gw(g, N, F, b.gMi, b.Pn[l], t.pre3[l], t.Ce[l], t.g_pre3, t.gCe_slots, E, s));
      KR(CAT_MESSAGE, Ed * (12 * Fd + 12) + 3 * nodeB, launch_message_adjoint(gR, N, F, t.we[l], b.gMi, b.gPn, s));
      EDGE();
      if (tc) {
        dW(t_ + "M2", t.g_pre3, 3 * F, t.he2[l], 2 * F, Erows, 3 * F, 2 * F);
        dB(t_ + "b2", t.g_pre3, 3 * F, Erows, 3 * F);
      }
      gemm(s, t.g_pre3, 3 * F, q2.M3T, 3 * F, nullptr, t.g_pre2, 2 * F, Erows, 2 * F, 3 * F, GEMM_MUL_DSILU_AUX, nullptr, 0, t.pre2[l], 2 * F);
      if (tc) {
        dW(t_ + "M1", t.g_pre2, 2 * F, t.he1[l], F, Erows, 2 * F, F);
        dB(t_ + "b1", t.g_pre2, 2 * F, Erows, 2 * F);
      }
      gemm(s, t.g_pre2, 2 * F, q2.M2T, 2 * F, nullptr, t.g_pre1, F, Erows, F, 2 * F, GEMM_MUL_DSILU_AUX, nullptr, 0, t.pre1[l], F);
      KR(CAT_PAIR, Ed * Fd * 8 + Nd * Fd * 8 + Pd * Fd * 4, launch_tn2_edge_reduce(g, N, F, t.g_pre1, t.erev, t.gB, t.gCs, t.gAp, s));
      if (tc) {
        // first layer pre1[e] = (M1a phi + b1)[pair] + (M1b c)[target] + (M1c c)[source]: pair rows (the self pair's row is the sum
        // over the atoms' self edges), target / source atom rows
        launch_tn2_self_rows(g, N, F, t.g_pre1, tc->g1, s);
        launch_colsum(s, tc->g1, RP(F), nullptr, RP(F), nullptr, nullptr, N, F, t.gAp + (int64_t)P * F, false, tc->part);
        dW(t_ + "M0", t.gAp, F, b.phi, K, P1, F, K);
        dB(t_ + "b0", t.gAp, F, P1, F);
        dW(t_ + "M0b", t.gB, F, t.chg[l], qd, N, F, qd);
        dW(t_ + "M0c", t.gCs, F, t.chg[l], qd, N, F, qd);
      }
      KR(CAT_PAIR, Pd * Fd * 8, launch_tn2_pair_gd(g, P, F, t.gAp, t.dAp[l], t.gCe_slots, tn2_gw_slots(F), E, t.pair_edge, t.erev, b.dC, b.gd, s));
      // charges of head l enter this layer's edge MLP: g_c = Coulomb slice l + gB M1b + gCs M1c
      coulomb_slice(l);
      NODE();
      gemm(s, t.gB, F, q2.M1bT, F, nullptr, t.g_c, qd, N, qd, F, GEMM_ACCUM);
      gemm(s, t.gCs, F, q2.M1cT, F, nullptr, t.g_c, qd, N, qd, F, GEMM_ACCUM);
      // node chain down to X[l]
      if (t9) {
        Tl9Args ta{};
        ta.A = b.gPn; ta.e0 = b.X[l]; ta.e1 = b.G; ta.C = b.G; ta.N = N; ta.F = F;
        tlin9(s, TL9_PRO_PLAIN, TL9_EPI_NORMBWD, q_.VT, ta, 4.0, "normbwd");
      } else {
      if (tc) tensor_linear_grad(b.gPn, tc->Xh[l], t_ + "Va");
      tensor_linear(s, b.gPn, q_.VT, b.gXl, N, F);
      KR(CAT_ELEMENTWISE, 4 * nodeB, launch_norm_bwd(b.X[l], b.gXl, N, F, b.G, s));  // reads the residual G, writes the gradient wrt X[l]
      }
      charge_predict_bwd(l);
    }
    // ---- embedding adjoint (TensorNet's)
    KR(CAT_ELEMENTWISE, 3 * nodeB + Nd * 3 * Fd * 12, launch_embed_gate_bwd(b.G, b.UX, b.gates, b.a2, N, F, b.gUX, b.g_a2, s));
    NODE();
    gemm(s, b.g_a2, 3 * F, W.L2T, 3 * F, nullptr, b.g_a1, 2 * F, N, 2 * F, 3 * F, GEMM_MUL_DSILU_AUX, nullptr, 0, b.a1, 2 * F);
    gemm(s, b.g_a1, 2 * F, W.L1T, 2 * F, nullptr, b.g_ln0, F, N, F, 2 * F);
    KR(CAT_ELEMENTWISE, Nd * Fd * 12, launch_layernorm_bwd(b.g_ln0, b.xh0, b.rstd0, W.ln0_w, N, F, b.g_s0n, s));
    if (t9) {
      Tl9Args ta{};
      ta.A = b.gUX; ta.e0 = b.u0; ta.e1 = b.g_s0n; ta.o1 = b.gA; ta.N = N; ta.F = F;
      tlin9(s, TL9_PRO_PLAIN, TL9_EPI_EMBBWD, W.UeT, ta, 2.0 + 11.0 / 9.0, "embbwd");
    } else {
    tensor_linear(s, b.gUX, W.UeT, b.g_u0l, N, F);
    KR(CAT_ELEMENTWISE, 2 * nodeB + Nd * 11 * Fd * 4, launch_embed_bwd_atom(b.g_u0l, b.u0, b.g_s0n, N, F, b.gA, s));
    }
    if (tc) {  // the embedding is TensorNet's: same products as in tn_api.hip
      tensor_linear_grad(b.gUX, b.u0, "Ue");
      dW("L2", b.g_a2, 3 * F, b.h1, 2 * F, N, 3 * F, 2 * F);
      dB("bL2", b.g_a2, 3 * F, N, 3 * F);
      dW("L1", b.g_a1, 2 * F, b.ln0, F, N, 2 * F, F);
      dB("bL1", b.g_a1, 2 * F, N, 2 * F);
      dB("ln0_w", b.g_ln0, F, N, F, b.xh0, F);
      dB("ln0_b", b.g_ln0, F, N, F);
      const int64_t dir = (int64_t)P1 * 3 * F;
      const int Z = hp.max_z;
      launch_train_embed(g, N, F, z, W.Utab, W.Vtab, b.Q, b.C, b.gA, tc->gq, dir, tc->selfq, tc->gZu, tc->gZv, s);
      dW("Wdp", tc->gq, 3 * F, b.phi, K, P, 3 * F, K);
      dW("Wdp", tc->gq + dir, 3 * F, b.phi, K, P, 3 * F, K, true);
      launch_tn_gemm(s, tc->selfq, RP(F), b.phi + (int64_t)P * K, RP(0), nullptr, nullptr, N, F, K, tc->at("Wdp"), true, tc->part);
      dB("bdp", tc->gq, 3 * F, P, 3 * F);
      dB("bdp", tc->gq + dir, 3 * F, P, 3 * F, nullptr, 0, true);
      dB("bdp", tc->selfq, F, N, F, nullptr, 0, true);
      launch_onehot(z, N, Z, tc->onehot, s);
      dW("Utab", tc->onehot, Z, tc->gZu, F, N, Z, F);
      dW("Vtab", tc->onehot, Z, tc->gZv, F, N, Z, F);
    } else if (ntp) {
      KR(CAT_PAIR, Nd * 10 * Fd * 4 + momB, launch_embed_gm(g, N, F, K, ntp, z, W.Utab, W.Vtab, m->rb_rev, W.bdp, b.gA, b.gmom, s));
      KR(CAT_PAIR, Pd * 40 + momB, launch_embed_pair_rb(g, P, N, rbp, ntp, b.ps, b.gmom, b.gd, b.g_rhat, s, nullptr, nullptr, 0, 0));
    } else
      KR(CAT_PAIR, Pd * (24 * Fd + 24) + Nd * 10 * Fd * 4,
         launch_embed_pair_gd(g, P, F, z, W.Utab, W.Vtab, b.Q, b.dQ, b.C, b.dC, b.gA, b.gd, b.g_rhat, s));
    if (!tc) {
      KR(CAT_ELEMENTWISE, Pd * 40, launch_geom_gd(g, P, b.gd, b.g_rhat, b.g_delta, s));
      KR(CAT_ELEMENTWISE, Ed * 8 + Nd * 12, launch_force_gather(g, N, b.g_delta, perm, forces, s));
      KR(CAT_ELEMENTWISE, Nd * 24, launch_add_forces(t.fcoul, perm, N, forces, s));
    }
  }
  NODE();
  HIP_TRY(m, hipGetLastError());
  m->last = b;
  m->lastN = N;
  m->lastP = P;
  m->has_last = true;
  m->tn2_last_chg = t.chg_all;
  return TMDNET_OK;
}
