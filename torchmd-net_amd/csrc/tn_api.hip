// C-ABI entry points of libtmdnet_amd.so (include/tmdnet_amd.h): parameter packing, workspace
// carving and the kernel schedule of the TensorNet energy+force path on one MI355X.
//
// Schedule = reference call stack TorchMD_Net.forward -> TensorNet.forward -> TensorEmbedding /
// Interaction x L -> readout -> Scalar head -> reduce (torchmdnet/models/model.py:530-631,
// tensornet.py:308-402, 543-619, 729-814, output_modules.py:43-117), followed by the hand-written
// reverse pass that replaces torch.autograd.grad (model.py:618-628).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <unordered_map>
#include <string>
#include <vector>

#include "tn_model.h"
#include "tn_tlin9.h"
#include "tn_small.h"
#include "tn_interp.h"

using namespace tn;

static const char* kCatNames[CAT_COUNT] = {"graph", "gemm_edge", "gemm_node", "message", "pair_bwd", "embed_scatter", "elementwise", "edge_table"};

int fail(tmdnet_model* m, int code, const std::string& msg) {
  if (m) m->err = msg;
  return code;
}
thread_local tmdnet_model* g_cur = nullptr;
thread_local int g_gemm_cat = CAT_GEMM_NODE;
thread_local const int* g_mdev = nullptr;
thread_local int g_madd = 0;
thread_local const uint16_t* g_wsb_debug = nullptr;  // tmdnet_debug_gemm: split image supplied by the caller

void gemm(hipStream_t s, const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, float* C, int64_t ldc, int M,
          int N, int K, int flags, float* pre, int64_t ldpre, const float* aux, int64_t ldaux,
          const float* rowscale) {
  GemmArgs a{};
  a.A = A;
  a.W[0] = W;
  a.C = C;
  a.bias[0] = bias;
  a.pre = pre;
  a.aux = aux;
  a.rowscale = rowscale;
  a.lda = lda;
  a.ldw = ldw;
  a.ldc = ldc;
  a.ldpre = ldpre;
  a.ldaux = ldaux;
  a.M = M;
  a.N = N;
  a.K = K;
  a.groups = 1;
  a.flags = flags;
  a.m_dev = g_mdev;
  a.m_add = g_madd;
  if (g_cur && ldw == K) {
    auto it = g_cur->sb_of.find(W);
    if (it != g_cur->sb_of.end()) a.Wsbg[0] = it->second;
  }
  if (g_wsb_debug) a.Wsbg[0] = g_wsb_debug;
  // algorithmic traffic: A and W read once, C (and the saved pre-activation / aux operand) once
  const double bytes = 4.0 * ((double)M * K + (double)N * K + (double)M * N * (1 + (pre ? 1 : 0) + (aux ? 1 : 0) + ((flags & GEMM_ACCUM) ? 1 : 0)));
  char lab[64];
  std::snprintf(lab, sizeof(lab), "gemm %dx%dx%d", M, N, K);
  ProfScope ps_(s, g_gemm_cat, 2.0 * M * N * K, bytes, lab);
  launch_gemm(a, s);
}

// value + tangent through one weight tile (tn_gemm_dual.hip); rows = pairs (device-side count)
void gemm_dual(hipStream_t s, int kind, const float* A, const float* A2, int64_t lda, const float* W, const float* bias, float* C,
               float* C2, int64_t ldc, int M, int N, int K, const float* rs, const float* rs2,
               const uint16_t* Wsb) {
  GemmArgs a{};
  a.Wsb = Wsb;
  a.A = A;
  a.A2 = A2;
  a.W[0] = W;
  a.bias[0] = bias;
  a.C = C;
  a.C2 = C2;
  a.rowscale = rs;
  a.rowscale2 = rs2;
  a.lda = lda;
  a.ldw = K;
  a.ldc = ldc;
  a.M = M;
  a.N = N;
  a.K = K;
  a.groups = 1;
  a.m_dev = g_mdev;
  a.m_add = g_madd;
  const double bytes = 4.0 * (2.0 * M * K + (double)N * K + 2.0 * M * N);
  char lab[64];
  std::snprintf(lab, sizeof(lab), "gemm_dual<%d> %dx%dx%d", kind, M, N, K);
  ProfScope ps_(s, CAT_GEMM_EDGE, 4.0 * M * N * K, bytes, lab);
  launch_gemm_dual(a, kind, s);
}

// the three weight matrices act on the channel axis of the 1 + 3 + 5 irreducible components
// (reference tensornet.py:595-617, 752-754, 808-810): one grouped launch, 9 groups
namespace {

void build_specs(tmdnet_model* m) {
  const int F = m->hp.hidden_channels, K = m->hp.num_rbf, L = m->hp.num_layers, Z = m->hp.max_z, H = m->hp.head_hidden;
  auto& s = m->specs;
  const std::string R = "representation_model.", T = R + "tensor_embedding.";
  s.push_back({R + "distance_expansion.means", K, 1});
  s.push_back({R + "distance_expansion.betas", K, 1});
  for (int k = 1; k <= 3; ++k) {
    s.push_back({T + "distance_proj" + std::to_string(k) + ".weight", F, K});
    s.push_back({T + "distance_proj" + std::to_string(k) + ".bias", F, 1});
  }
  s.push_back({T + "emb.weight", Z, F});
  s.push_back({T + "emb2.weight", F, 2 * F});
  s.push_back({T + "emb2.bias", F, 1});
  for (int k = 0; k < 3; ++k) s.push_back({T + "linears_tensor." + std::to_string(k) + ".weight", F, F});
  s.push_back({T + "linears_scalar.0.weight", 2 * F, F});
  s.push_back({T + "linears_scalar.0.bias", 2 * F, 1});
  s.push_back({T + "linears_scalar.1.weight", 3 * F, 2 * F});
  s.push_back({T + "linears_scalar.1.bias", 3 * F, 1});
  s.push_back({T + "init_norm.weight", F, 1});
  s.push_back({T + "init_norm.bias", F, 1});
  for (int l = 0; l < L; ++l) {
    const std::string Lp = R + "layers." + std::to_string(l) + ".";
    const int64_t K1 = m->tn2 ? K + 2 * m->tn2->hp.q_dim : K;  // TensorNet2: [phi ; c_i ; c_j]
    const int64_t dims[3][2] = {{F, K1}, {2 * F, F}, {3 * F, 2 * F}};
    for (int k = 0; k < 3; ++k) {
      s.push_back({Lp + "linears_scalar." + std::to_string(k) + ".weight", dims[k][0], dims[k][1]});
      s.push_back({Lp + "linears_scalar." + std::to_string(k) + ".bias", dims[k][0], 1});
    }
    for (int k = 0; k < 6; ++k) s.push_back({Lp + "linears_tensor." + std::to_string(k) + ".weight", F, F});
  }
  s.push_back({R + "linear.weight", F, 3 * F});
  s.push_back({R + "linear.bias", F, 1});
  s.push_back({R + "out_norm.weight", 3 * F, 1});
  s.push_back({R + "out_norm.bias", 3 * F, 1});
  const std::string O = "output_model.output_network.layers.";
  s.push_back({O + "0.weight", H, F});
  s.push_back({O + "0.bias", H, 1});
  s.push_back({O + "2.weight", 1, H});
  s.push_back({O + "2.bias", 1, 1});
  s.push_back({"mean", 1, 1});
  s.push_back({"std", 1, 1});
  if (m->hp.has_atomref) s.push_back({"atomref", Z, 1});
  if (m->tn2) {
    const int qd = m->tn2->hp.q_dim;
    for (int l = 0; l <= L; ++l) {
      const std::string Cp = R + (l == 0 ? std::string("charge_predict_0.") : "charge_predicts." + std::to_string(l - 1) + ".");
      s.push_back({Cp + "q_norm.weight", 3 * F, 1});
      s.push_back({Cp + "q_norm.bias", 3 * F, 1});
      s.push_back({Cp + "q_mlp.layers.0.weight", F, 3 * F});
      s.push_back({Cp + "q_mlp.layers.0.bias", F, 1});
      s.push_back({Cp + "q_mlp.layers.2.weight", F, F});
      s.push_back({Cp + "q_mlp.layers.2.bias", F, 1});
      s.push_back({Cp + "q_mlp.layers.4.weight", 2 * qd, F});
      s.push_back({Cp + "q_mlp.layers.4.bias", 2 * qd, 1});
    }
    s.push_back({"output_model.qweights", (int64_t)(L + 1) * qd, 1});
  }
}

}  // namespace

void tensor_linear(hipStream_t s, const float* A, const float* const W3[3], float* C, int N, int F, int flags, float* pre,
                   const float* gates) {
  GemmArgs a{};
  a.A = A;
  a.C = C;
  a.pre = pre;
  a.aux = gates;
  a.lda = a.ldc = a.ldpre = 9 * (int64_t)F;
  a.ldaux = 3 * (int64_t)F;
  a.ldw = F;
  for (int c = 0; c < 9; ++c) {
    const int t = c == 0 ? 0 : (c < 4 ? 1 : 2);
    a.W[c] = W3[t];
    if (g_cur) {
      auto it = g_cur->sb_of.find(W3[t]);
      if (it != g_cur->sb_of.end()) a.Wsbg[c] = it->second;
    }
    a.bias[c] = nullptr;
    a.a_off[c] = a.c_off[c] = a.pre_off[c] = c * F;
    a.aux_off[c] = t * F;
  }
  a.M = N;
  a.N = F;
  a.K = F;
  a.groups = 9;
  a.flags = flags;
  const double bytes = 4.0 * (9.0 * N * F * (2 + (pre ? 1 : 0)) + (gates ? 3.0 * N * F : 0.0) + 3.0 * F * F);
  char lab[64];
  std::snprintf(lab, sizeof(lab), "tensor_linear 9x%dx%dx%d", N, F, F);
  ProfScope ps_(s, CAT_GEMM_NODE, 2.0 * 9 * N * (double)F * F, bytes, lab);
  launch_gemm(a, s);
}


// per-workspace record of the last graph build (tn_model.h GraphRecord)
void remember_graph(tmdnet_model* m, const void* graph_ws) {
  if (m->graph_rec.size() >= 64 && !m->graph_rec.count(graph_ws)) m->graph_rec.clear();  // workspaces regrown many times: old addresses are dead
  m->graph_rec[graph_ws] = tmdnet_model::GraphRecord{m->graph_is_cell, m->graph_cell_multi, m->graph_has_z, m->last_nt, m->lastE, m->graph_no_ghost_pairs};
}
void recall_graph(tmdnet_model* m, const void* graph_ws) {
  auto it = m->graph_rec.find(graph_ws);
  if (it == m->graph_rec.end()) return;  // never built through this handle: the fields keep the last build's values
  m->graph_is_cell = it->second.is_cell;
  m->graph_cell_multi = it->second.cell_multi;
  m->graph_has_z = it->second.has_z;
  m->last_nt = it->second.nt;
  m->lastE = it->second.lastE;
  m->graph_no_ghost_pairs = it->second.no_ghost_pairs;
}

// fused form (tn_tlin9.hip): the weights' fragment-major images must exist (F % 32 == 0); `tensors` = [N, 9, F] tensors the
// launch reads or writes (algorithmic bytes of the profile record)
bool tlin9_images(const float* const W3[3]) {
  if (!g_cur) return false;
  for (int t = 0; t < 3; ++t)
    if (g_cur->fm_of.find(W3[t]) == g_cur->fm_of.end()) return false;
  return true;
}
void tlin9(hipStream_t s, int pro, int epi, const float* const W3[3], Tl9Args a, double tensors, const char* what) {
  for (int t = 0; t < 3; ++t) a.Wfm[t] = g_cur->fm_of.at(W3[t]);
  char lab[64];
  std::snprintf(lab, sizeof(lab), "tlin9 %s 9x%dx%dx%d", what, a.N, a.F, a.F);
  ProfScope ps_(s, CAT_GEMM_NODE, 2.0 * 9 * a.N * (double)a.F * a.F, 4.0 * (tensors * 9.0 * a.N * a.F + 3.0 * a.F * a.F), lab);
  launch_tlin9(a, pro, epi, s);
}

Graph carve_graph(void* ws, int64_t N, int64_t B, int64_t ecap, size_t* total) {
  Carver c(ws);
  Graph g{};
  const int64_t pcap = ecap / 2 + 1;
  g.small_mols = N <= 96 * (B > 0 ? B : 1) ? 1 : 0;
  g.counts = c.take<int>(8);
  g.mstart = c.take<int>(B);
  g.mend = c.take<int>(B);
  g.nlow = c.take<int>(N);
  g.ntot = c.take<int>(N);
  g.rowptr = c.take<int>(N + 1);
  g.pairptr = c.take<int>(N + 1);
  g.col = c.take<int>(ecap);
  g.epair = c.take<int>(ecap);
  g.esign = c.take<float>(ecap);
  g.pair_i = c.take<int>(pcap);
  g.pair_j = c.take<int>(pcap);
  g.pd = c.take<float>(pcap + 1);
  g.pdelta = c.take<float>(pcap * 3);
  g.prhat = c.take<float>(pcap * 3);
  g.ecap = ecap;
  g.pcap = pcap;
  g.perm = c.take<int>(N);
  g.iota = c.take<int>(N);
  g.cell_key = c.take<int>(N);
  g.cell_key_sorted = c.take<int>(N);
  g.cell_start = c.take<int>(8 * N + 2);
  g.pos_s = c.take<float>(3 * N);
  g.z_s = c.take<int64_t>(N);
  g.boxd = c.take<float>(12);
  g.cgrid = c.take<int>(4);
  g.bat_s = c.take<int>(N);
  g.bat_c = c.take<int64_t>(N);
  g.z_c = c.take<int64_t>(N);
  g.tix = c.take<int>(N);
  g.tz = c.take<int>(64);
  g.sort_tmp_bytes = cell_sort_temp_bytes(N);
  g.sort_tmp = c.take<char>((int64_t)g.sort_tmp_bytes);
  if (total) *total = c.off;
  return g;
}

// the cell list applies to systems that share ONE grid: one system or several molecules, periodic in one common box
// (orthorhombic or reduced triclinic) or non-periodic (box_mode 0: a fictitious orthorhombic box around the bounding box of
// ALL positions, as the reference does with a fixed box, models/utils.py:206-212); per-molecule boxes (box_mode 2) have no
// common grid and take the brute-force sweep inside each molecule.  The host asks for it (tmdnet_set_cell_grid) when the
// molecules are large on average; several molecules are interleaved in cell order and pairs stay inside a molecule.
// cell_n < 0: the grid is computed on the device from the current box / bounding box on every call (tn_cell.hip).
namespace {
bool cell_applicable(const tmdnet_model* m, int64_t n_atoms, int64_t n_mol, int box_mode) {
  const int* n = m->cell_n;
  if (n_mol < 1 || box_mode == 2 || n_atoms < 1) return false;
  if (n[0] < 0) return true;
  if (n[0] < 1 || n[1] < 1 || n[2] < 1) return false;
  return (int64_t)n[0] * n[1] * n[2] <= 8 * n_atoms;
}
void set_cell(Graph& g, const tmdnet_model* m, bool on, int64_t n_mol = 1) {
  g.use_cell = on ? (n_mol > 1 ? 2 : 1) : 0;
  const bool explicit_grid = m->cell_n[0] > 0;
  g.ncx = explicit_grid ? m->cell_n[0] : 0;
  g.ncy = explicit_grid ? m->cell_n[1] : 0;
  g.ncz = explicit_grid ? m->cell_n[2] : 0;
}

// species count rounded up to 4 / 8 when this call takes the embedding in the radial basis (tn_embed_rb.hip), else 0:
// dynamic shapes only (the species count is read back with the pair counts), a batch-scale system, at most 8 species
}  // namespace
int rb_ntp(const tmdnet_model* m, int64_t n_atoms, int64_t n_pairs) {
  if (!m->rb_fwd || m->et || m->train || n_pairs < 0 || n_atoms < m->rb_min_atoms) return 0;
  return embed_rb_ntp(m->last_nt);
}
namespace {

// lean: 1 = no direct-evaluation buffers (phi, the edge MLP's activations and tangents), 2 = no Q / dQ rows (embedding in the
// radial basis), 4 = no w / dw rows (the sweeps interpolate them) - what option "recompute_pair_rows" drops: 12.6 -> 0.1 KB per pair
FwdBuffers carve_fwd(void* ws, const tmdnet_hparams& hp, int64_t N, int64_t B, int64_t P, bool bwd, size_t* total, int ntp = 0,
                     int lean = 0) {
  Carver c(ws);
  FwdBuffers b{};
  const int64_t F = hp.hidden_channels, K = hp.num_rbf, L = hp.num_layers, Z = hp.max_z, H = hp.head_hidden;
  const int64_t P1 = P + 1, N9 = N * 9 * F;
  b.phi = c.take<float>((lean & 1) ? 0 : P1 * K);
  b.dphi = c.take<float>((lean & 1) ? 0 : P1 * K);
  b.C = c.take<float>(P1);
  b.dC = c.take<float>(P1);
  b.Q = c.take<float>((lean & 2) ? 0 : P1 * 3 * F);
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
    b.w.push_back(c.take<float>((lean & 4) ? 0 : P1 * 3 * F));
    b.dw.push_back(c.take<float>(bwd && !(lean & 4) ? P1 * 3 * F : 0));
    b.Pn.push_back(c.take<float>(N9));
    b.Mi.push_back(c.take<float>(N9));
    b.D.push_back(c.take<float>(N9));
  }
  b.he1 = c.take<float>((lean & 1) ? 0 : P1 * F);
  b.he2 = c.take<float>((lean & 1) ? 0 : P1 * 2 * F);
  b.te1 = c.take<float>(bwd && !(lean & 1) ? P1 * F : 0);
  b.te2 = c.take<float>(bwd && !(lean & 1) ? P1 * 2 * F : 0);
  b.dQ = c.take<float>(bwd && !(lean & 2) ? P1 * 3 * F : 0);
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
  b.kap = c.take<float>(N);
  b.skeys_s = c.take<unsigned>(P1);
  b.svals_s = c.take<int>(P1);
  b.shist = c.take<int>(65536 + 2);
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
    b.gd_slots = c.take<float>(2 * P1 * (int64_t)L * ((F + 31) / 32));  // [layer][wave or channel chunk][pair][direction] partial g_d
    b.gUX = c.take<float>(N9);
    b.g_a2 = c.take<float>(N * 3 * F);
    b.g_a1 = c.take<float>(N * 2 * F);
    b.g_ln0 = c.take<float>(N * F);
    b.g_s0n = c.take<float>(N * F);
    b.g_u0l = c.take<float>(N9);
    b.gA = c.take<float>(N * 10 * F);
    b.g_rhat = c.take<float>(P1 * 3);
    b.g_delta = c.take<float>(P1 * 3);
  }
  b.mom = c.take<float>(ntp ? embed_rb_moment_elems(N, ntp, (int)K) : 0);
  b.gmom = c.take<float>(ntp && bwd ? embed_rb_gmoment_elems(N, ntp, (int)K) : 0);
  b.ps = c.take<float>(ntp ? P1 * 8 : 0);
  if (total) *total = c.off;
  return b;
}

}  // namespace

void free_radial_tables(EdgeTables& t) {
  for (float* p : t.tab)
    if (p) (void)hipFree(p);
  t = EdgeTables{};
}

// Radial tables (tn_edge_table.hip): evaluate every table's function and its d/dd on a uniform grid in fp64, pack them as
// fp32 rows (value | slope | divided difference), then verify the step's fp32 interpolation against the fp64 evaluation at
// all interval midpoints; refine the grid until the measured error is below the bound, or leave the tables off.
int build_radial_tables(tmdnet_model* m, EdgeTables& out, const std::vector<TableSpec>& specs, const float* means, const float* betas,
                        int K, double lo, double up) {
  free_radial_tables(out);
  const char* env = getenv("TMDNET_EDGE_TABLE");
  if (env && atoi(env) == 0) return TMDNET_OK;  // developer / benchmark switch: direct GEMMs every step
  int Rmax = 0, Nmax = 0;
  for (const TableSpec& sp : specs) {
    const int R = sp.chain.back().N;
    if (R % 4 || R / 4 > 256) return TMDNET_OK;  // what the interpolation kernel's thread layout covers
    Rmax = std::max(Rmax, R);
    for (const TableLayer& l : sp.chain) Nmax = std::max(Nmax, l.N);
  }
  if (specs.empty()) return TMDNET_OK;
  const double tol_value = 5e-7, tol_slope = 2e-6;  // relative to the table's largest |value| / |slope|
  hipStream_t s = nullptr;
  const int ntab = (int)specs.size();
  for (int T = 8192; T <= 65536; T *= 2) {
    const int M = T + 2;
    std::vector<void*> tmp;
    bool alloc_ok = true;
    auto D_ = [&](int64_t n) {
      double* p = nullptr;
      alloc_ok = (hipMalloc(reinterpret_cast<void**>(&p), (size_t)n * sizeof(double)) == hipSuccess) && alloc_ok;
      tmp.push_back(p);
      return p;
    };
    auto F_ = [&](int64_t n) {
      float* p = nullptr;
      alloc_ok = (hipMalloc(reinterpret_cast<void**>(&p), (size_t)n * sizeof(float)) == hipSuccess) && alloc_ok;
      tmp.push_back(p);
      return p;
    };
    double *dist = D_(M), *phi = D_((int64_t)M * K), *dphi = D_((int64_t)M * K), *C = D_(M), *dC = D_(M);
    double* hv_[2] = {D_((int64_t)M * Nmax), D_((int64_t)M * Nmax)};  // ping-pong: layer outputs (value)
    double* hs_[2] = {D_((int64_t)M * Nmax), D_((int64_t)M * Nmax)};  // (slope)
    float *ip = F_((int64_t)T * Rmax), *dip = F_((int64_t)T * Rmax);
    std::vector<float*> tab(ntab, nullptr);
    for (int t = 0; t < ntab; ++t)
      alloc_ok = (hipMalloc(reinterpret_cast<void**>(&tab[t]), (size_t)M * 3 * specs[t].chain.back().N * sizeof(float)) == hipSuccess) &&
                 alloc_ok;
    auto cleanup = [&](bool keep_tabs) {
      for (void* p : tmp)
        if (p) (void)hipFree(p);
      if (!keep_tabs)
        for (float* t : tab)
          if (t) (void)hipFree(t);
    };
    if (!alloc_ok) {
      cleanup(false);
      return TMDNET_OK;  // no memory for tables: the direct path stays
    }
    // function t at the distances described by (rows, mid), fp64: returns which ping-pong buffer holds (value, slope)
    auto evaluate = [&](int t, int rows, bool mid) {
      launch_radial_f64(rows, lo, up, T, mid, means, betas, K, dist, phi, dphi, C, dC, s);
      const double *a = phi, *a2 = phi == nullptr ? nullptr : dphi;
      int lda = K, cur = 0;
      for (const TableLayer& l : specs[t].chain) {
        launch_dense_f64(a, a2, lda, l.W, l.b, rows, l.N, l.K, l.kind, C, dC, hv_[cur], hs_[cur], l.N, s);
        a = hv_[cur];
        a2 = hs_[cur];
        lda = l.N;
        cur ^= 1;
      }
      return cur ^ 1;
    };
    double worst_v = 0.0, worst_s = 0.0;
    bool fail_ = false;
    for (int t = 0; t < ntab && !fail_; ++t) {
      const int R = specs[t].chain.back().N;
      int w = evaluate(t, M, false);
      launch_table_pack(hv_[w], hs_[w], T, R, (up - lo) / (double)T, tab[t], s);
      w = evaluate(t, T, true);  // fp64 truth at the interval midpoints
      launch_interp_list(tab[t], dist, T, R, T, (float)lo, (float)up, ip, dip, s);
      std::vector<double> hv((size_t)T * R), hs((size_t)T * R);
      std::vector<float> hip_((size_t)T * R), hdip((size_t)T * R);
      if (hipMemcpy(hv.data(), hv_[w], hv.size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess ||
          hipMemcpy(hs.data(), hs_[w], hs.size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess ||
          hipMemcpy(hip_.data(), ip, hip_.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess ||
          hipMemcpy(hdip.data(), dip, hdip.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) {
        fail_ = true;
        break;
      }
      double mv = 0.0, ms = 0.0, ev = 0.0, es = 0.0;
      for (size_t i = 0; i < hv.size(); ++i) {
        mv = std::max(mv, std::fabs(hv[i]));
        ms = std::max(ms, std::fabs(hs[i]));
        ev = std::max(ev, std::fabs(hv[i] - (double)hip_[i]));
        es = std::max(es, std::fabs(hs[i] - (double)hdip[i]));
      }
      if (!std::isfinite(ev) || !std::isfinite(es) || !std::isfinite(mv)) {
        fail_ = true;
        break;
      }
      if (mv > 0.0) worst_v = std::max(worst_v, ev / mv);
      if (ms > 0.0) worst_s = std::max(worst_s, es / ms);
    }
    const bool good = !fail_ && worst_v <= tol_value && worst_s <= tol_slope;
    out.err_value = worst_v;  // of the last grid tried (reported also when the tables stay off)
    out.err_slope = worst_s;
    if (getenv("TMDNET_DEBUG"))
      fprintf(stderr, "[tmdnet] radial tables T=%d: value err %.3e, slope err %.3e, fail=%d\n", T, worst_v, worst_s, (int)fail_);
    cleanup(good);
    if (good) {
      out.ok = true;
      out.T = T;
      out.tab = tab;
      for (const TableSpec& sp : specs) out.R.push_back(sp.chain.back().N);
      return TMDNET_OK;
    }
    if (fail_) return TMDNET_OK;
  }
  (void)m;
  return TMDNET_OK;  // bound not met even on the finest grid: direct GEMMs
}

namespace {

int build_edge_tables(tmdnet_model* m) {
  const tmdnet_hparams& hp = m->hp;
  const int F = hp.hidden_channels, K = hp.num_rbf, L = hp.num_layers;
  const DevParams& W = m->P;
  std::vector<TableSpec> specs;
  if (m->tn2) {  // TensorNet2: Q of the embedding and the pair block of every layer's first edge-MLP layer, M1a phi(d) + b1
    specs.push_back(TableSpec{{TableLayer{W.Wdp, W.bdp, 3 * F, K, 0}}});
    for (int l = 0; l < L; ++l) specs.push_back(TableSpec{{TableLayer{m->tn2->layer[l].M1a, W.layer[l].b1, F, K, 0}}});
  } else if (L + 1 <= 8) {
    specs.push_back(TableSpec{{TableLayer{W.Wdp, W.bdp, 3 * F, K, 0}}});
    for (int l = 0; l < L; ++l) {
      const LayerP& q_ = W.layer[l];
      specs.push_back(TableSpec{{TableLayer{q_.M1, q_.b1, F, K, 1}, TableLayer{q_.M2, q_.b2, 2 * F, F, 1},
                                 TableLayer{q_.M3, q_.b3, 3 * F, 2 * F, 2}}});
    }
  }
  return build_radial_tables(m, m->tabs, specs, W.means, W.betas, K, hp.cutoff_lower, hp.cutoff_upper);
}

}  // namespace

int ensure_radial_tables(tmdnet_model* m, hipStream_t s) {
  if (!m->tabs_pending) return TMDNET_OK;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (s && hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
    return fail(m, TMDNET_ERR_STATE, "the radial tables are rebuilt after a parameter upload: run one step outside the capture first");
  // the builders launch on the NULL stream and read m->dev: a device-side parameter update (tmdnet_update_params_device) may still
  // be queued on ANY stream - torch's side streams are non-blocking, the NULL stream orders nothing against them
  HIP_TRY(m, hipDeviceSynchronize());
  const int rc = m->et ? et_build_tables(m) : build_edge_tables(m);
  if (rc != TMDNET_OK) return rc;
  HIP_TRY(m, hipStreamSynchronize(nullptr));
  m->tabs_pending = false;
  return TMDNET_OK;
}

extern "C" {

const char* tmdnet_version(void) { return "tmdnet_amd 0.3 (gfx950)"; }
int tmdnet_abi_version(void) { return TMDNET_ABI_VERSION; }

int tmdnet_create(const tmdnet_hparams* hp, tmdnet_model** out) {
  if (!hp || !out) return TMDNET_ERR_INVALID;
  if (hp->hidden_channels <= 0 || hp->num_layers < 0 || hp->num_rbf <= 0 || hp->max_z <= 0 || hp->head_hidden <= 0 ||
      hp->max_num_neighbors <= 0 || !(hp->cutoff_upper > hp->cutoff_lower))
    return TMDNET_ERR_INVALID;
  tmdnet_model* m = new tmdnet_model();
  m->hp = *hp;
  if (const char* e = getenv("TMDNET_EDGE_TABLE_MIN_PAIRS")) m->tab_min_pairs = atoll(e);  // developer switch (default: tn_model.h)
  build_specs(m);
  if (getenv("TMDNET_SIDE_STREAM")) {  // opt-in (profiles/r01_notes.md): +2 % batch throughput, but the GEMMs then share the chip
    if (hipStreamCreateWithFlags(&m->side, hipStreamNonBlocking) != hipSuccess) m->side = nullptr;
    if (m->side) {
      (void)hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming);
      m->ev_join.resize(hp->num_layers);
      for (auto& e : m->ev_join) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
    }
  }
  *out = m;
  return TMDNET_OK;
}

int tmdnet_create_tn2(const tmdnet_tn2_hparams* hp, tmdnet_model** out) {
  if (!hp || !out) return TMDNET_ERR_INVALID;
  if (hp->hidden_channels <= 0 || hp->num_layers < 0 || hp->num_rbf <= 0 || hp->max_z <= 0 || hp->head_hidden <= 0 ||
      hp->max_num_neighbors <= 0 || !(hp->cutoff_upper > hp->cutoff_lower) || hp->q_dim <= 0 || hp->q_dim > 64 ||
      (hp->num_layers + 1) * hp->q_dim > 64)
    return TMDNET_ERR_INVALID;
  tmdnet_model* m = new tmdnet_model();
  m->hp = tmdnet_hparams{hp->hidden_channels, hp->num_layers, hp->num_rbf, hp->max_z, hp->max_num_neighbors, hp->group_o3,
                         hp->head_hidden, hp->has_atomref, hp->cutoff_lower, hp->cutoff_upper};
  m->tn2 = new Tn2Model();
  m->tn2->hp = *hp;
  if (const char* e = getenv("TMDNET_EDGE_TABLE_MIN_PAIRS")) m->tab_min_pairs = atoll(e);
  build_specs(m);
  *out = m;
  return TMDNET_OK;
}

int tmdnet_create_et(const tmdnet_et_hparams* hp, tmdnet_model** out) {
  if (!hp || !out) return TMDNET_ERR_INVALID;
  if (hp->hidden_channels <= 0 || hp->num_layers < 0 || hp->num_rbf <= 0 || hp->max_z <= 0 || hp->num_heads <= 0 ||
      hp->max_num_neighbors <= 0 || !(hp->cutoff_upper > hp->cutoff_lower))
    return TMDNET_ERR_INVALID;
  tmdnet_model* m = new tmdnet_model();
  m->hp = tmdnet_hparams{};  // what the graph phase reads
  m->hp.hidden_channels = hp->hidden_channels;
  m->hp.num_layers = hp->num_layers;
  m->hp.num_rbf = hp->num_rbf;
  m->hp.max_z = hp->max_z;
  m->hp.max_num_neighbors = hp->max_num_neighbors;
  m->hp.has_atomref = hp->has_atomref;
  m->hp.cutoff_lower = hp->cutoff_lower;
  m->hp.cutoff_upper = hp->cutoff_upper;
  const int rc = et_create(m, hp);
  if (rc != TMDNET_OK) {
    delete m;
    return rc;
  }
  *out = m;
  return TMDNET_OK;
}

int tmdnet_destroy(tmdnet_model* m) {
  if (!m) return TMDNET_OK;
  et_destroy(m);
  if (m->side) {
    (void)hipStreamSynchronize(m->side);
    (void)hipEventDestroy(m->ev_fork);
    for (auto& e : m->ev_join) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(m->side);
  }
  free_radial_tables(m->tabs);
  delete m->tn2;
  if (m->dev) (void)hipFree(m->dev);
  if (m->dev_sb) (void)hipFree(m->dev_sb);
  if (m->upd_tid) (void)hipFree(m->upd_tid);
  if (m->upd_ofs) (void)hipFree(m->upd_ofs);
  if (m->upd_ptrs) (void)hipFree(m->upd_ptrs);
  if (m->rb_img) (void)hipFree(m->rb_img);
  if (m->halo_rng) (void)hipFree(m->halo_rng);
  delete m;
  return TMDNET_OK;
}

const char* tmdnet_last_error(const tmdnet_model* m) { return m ? m->err.c_str() : "null model"; }

int tmdnet_num_params(const tmdnet_model* m) { return m ? (int)m->specs.size() : 0; }

const char* tmdnet_param_name(const tmdnet_model* m, int idx, int64_t* numel) {
  if (!m || idx < 0 || idx >= (int)m->specs.size()) return nullptr;
  if (numel) *numel = m->specs[idx].rows * m->specs[idx].cols;
  return m->specs[idx].name.c_str();
}

int tmdnet_set_param(tmdnet_model* m, const char* name, const float* data_host, int64_t numel) {
  if (!m || !name || !data_host) return TMDNET_ERR_INVALID;
  for (const auto& sp : m->specs) {
    if (sp.name == name) {
      if (numel != sp.rows * sp.cols)
        return fail(m, TMDNET_ERR_INVALID, std::string("parameter ") + name + ": expected " + std::to_string(sp.rows * sp.cols) +
                                               " elements, got " + std::to_string(numel));
      m->host[name].assign(data_host, data_host + numel);
      m->finalized = false;
      return TMDNET_OK;
    }
  }
  return fail(m, TMDNET_ERR_INVALID, std::string("unknown parameter: ") + name);
}

// Packing of the state dict into the engine's parameter buffer: copies, transposes, column blocks, concatenations - pure data
// movement, no arithmetic (the device-side update below relies on that: it runs this once on index tags to learn where every
// packed element comes from).
static void pack_tensornet_params(tmdnet_model* m, std::map<std::string, std::vector<float>>& h, Packer& pk,
                                  std::map<std::string, size_t>& off) {
  const int F = m->hp.hidden_channels, K = m->hp.num_rbf, L = m->hp.num_layers, H = m->hp.head_hidden;
  (void)H;
  const std::string R = "representation_model.", T = R + "tensor_embedding.", O = "output_model.output_network.layers.";
  auto put = [&](const std::string& key, const std::vector<float>& v) { off[key] = pk.add(v); };
  auto putT = [&](const std::string& key, const std::vector<float>& v, int64_t r, int64_t c) { off[key] = pk.add_T(v, r, c); };
  put("means", h[R + "distance_expansion.means"]);
  put("betas", h[R + "distance_expansion.betas"]);
  std::vector<float> Wdp, bdp;
  for (int k = 1; k <= 3; ++k) {
    auto& w = h[T + "distance_proj" + std::to_string(k) + ".weight"];
    auto& b = h[T + "distance_proj" + std::to_string(k) + ".bias"];
    Wdp.insert(Wdp.end(), w.begin(), w.end());
    bdp.insert(bdp.end(), b.begin(), b.end());
  }
  put("Wdp", Wdp);
  put("bdp", bdp);
  put("emb", h[T + "emb.weight"]);
  put("emb2_w", h[T + "emb2.weight"]);
  put("emb2_b", h[T + "emb2.bias"]);
  {  // transposed halves of emb2.weight = [Wa | Wb] for the per-type table kernel (coalesced over channels)
    const auto& w2 = h[T + "emb2.weight"];
    std::vector<float> wa((size_t)F * F), wb((size_t)F * F);
    for (int f = 0; f < F; ++f)
      for (int k = 0; k < F; ++k) {
        wa[(size_t)k * F + f] = w2[(size_t)f * 2 * F + k];
        wb[(size_t)k * F + f] = w2[(size_t)f * 2 * F + F + k];
      }
    put("emb2_waT", wa);
    put("emb2_wbT", wb);
  }
  for (int k = 0; k < 3; ++k) {
    put("Ue" + std::to_string(k), h[T + "linears_tensor." + std::to_string(k) + ".weight"]);
    putT("UeT" + std::to_string(k), h[T + "linears_tensor." + std::to_string(k) + ".weight"], F, F);
  }
  put("L1", h[T + "linears_scalar.0.weight"]);
  put("bL1", h[T + "linears_scalar.0.bias"]);
  putT("L1T", h[T + "linears_scalar.0.weight"], 2 * F, F);
  put("L2", h[T + "linears_scalar.1.weight"]);
  put("bL2", h[T + "linears_scalar.1.bias"]);
  putT("L2T", h[T + "linears_scalar.1.weight"], 3 * F, 2 * F);
  put("ln0_w", h[T + "init_norm.weight"]);
  put("ln0_b", h[T + "init_norm.bias"]);
  for (int l = 0; l < L; ++l) {
    const std::string Lp = R + "layers." + std::to_string(l) + ".", t = "l" + std::to_string(l) + ".";
    const int64_t dims[3][2] = {{F, K}, {2 * F, F}, {3 * F, 2 * F}};
    for (int k = 0; k < 3; ++k) {
      put(t + "M" + std::to_string(k), h[Lp + "linears_scalar." + std::to_string(k) + ".weight"]);
      put(t + "b" + std::to_string(k), h[Lp + "linears_scalar." + std::to_string(k) + ".bias"]);
    }
    for (int k = 0; k < 6; ++k) {
      put(t + "V" + std::to_string(k), h[Lp + "linears_tensor." + std::to_string(k) + ".weight"]);
      putT(t + "VT" + std::to_string(k), h[Lp + "linears_tensor." + std::to_string(k) + ".weight"], F, F);
    }
    if (!m->tn2) {  // transposes of the edge MLP's second and third layer: the data-gradient products of the parameter-gradient
      // pass (g_he1 = g_pre2 M2, g_he2 = g_pre3 M3) then run on the split-bf16 kernels like every other product (they were
      // transposed on the fly, without a split image: fp32 matrix pipe, 1.6 ms per step at C2)
      putT(t + "M2T", h[Lp + "linears_scalar.1.weight"], 2 * F, F);
      putT(t + "M3T", h[Lp + "linears_scalar.2.weight"], 3 * F, 2 * F);
    }
  }
  if (m->tn2) {  // TensorNet2: column blocks of linears_scalar.0, transposes for the reverse pass, ChargePredict heads
    const int qd = m->tn2->hp.q_dim, K1 = K + 2 * qd;
    for (int l = 0; l < L; ++l) {
      const std::string Lp = R + "layers." + std::to_string(l) + ".", t = "l" + std::to_string(l) + ".";
      const auto& w1 = h[Lp + "linears_scalar.0.weight"];
      std::vector<float> a((size_t)F * K), bq((size_t)F * qd), cq((size_t)F * qd);
      for (int f = 0; f < F; ++f) {
        for (int k = 0; k < K; ++k) a[(size_t)f * K + k] = w1[(size_t)f * K1 + k];
        for (int k = 0; k < qd; ++k) {
          bq[(size_t)f * qd + k] = w1[(size_t)f * K1 + K + k];
          cq[(size_t)f * qd + k] = w1[(size_t)f * K1 + K + qd + k];
        }
      }
      put(t + "M1a", a);
      put(t + "M1b", bq);
      put(t + "M1c", cq);
      putT(t + "M1bT", bq, F, qd);
      putT(t + "M1cT", cq, F, qd);
      putT(t + "M2T", h[Lp + "linears_scalar.1.weight"], 2 * F, F);
      putT(t + "M3T", h[Lp + "linears_scalar.2.weight"], 3 * F, 2 * F);
    }
    for (int l = 0; l <= L; ++l) {
      const std::string Cp = R + (l == 0 ? std::string("charge_predict_0.") : "charge_predicts." + std::to_string(l - 1) + "."),
                        t = "cp" + std::to_string(l) + ".";
      put(t + "ln_w", h[Cp + "q_norm.weight"]);
      put(t + "ln_b", h[Cp + "q_norm.bias"]);
      put(t + "W1", h[Cp + "q_mlp.layers.0.weight"]);
      put(t + "b1", h[Cp + "q_mlp.layers.0.bias"]);
      putT(t + "W1T", h[Cp + "q_mlp.layers.0.weight"], F, 3 * F);
      put(t + "W2", h[Cp + "q_mlp.layers.2.weight"]);
      put(t + "b2", h[Cp + "q_mlp.layers.2.bias"]);
      putT(t + "W2T", h[Cp + "q_mlp.layers.2.weight"], F, F);
      put(t + "W3", h[Cp + "q_mlp.layers.4.weight"]);
      put(t + "b3", h[Cp + "q_mlp.layers.4.bias"]);
      putT(t + "W3T", h[Cp + "q_mlp.layers.4.weight"], 2 * qd, F);
    }
    put("qweights", h["output_model.qweights"]);
  }
  put("lnr_w", h[R + "out_norm.weight"]);
  put("lnr_b", h[R + "out_norm.bias"]);
  put("Lin", h[R + "linear.weight"]);
  put("bLin", h[R + "linear.bias"]);
  putT("LinT", h[R + "linear.weight"], F, 3 * F);
  put("O1", h[O + "0.weight"]);
  put("bO1", h[O + "0.bias"]);
  putT("O1T", h[O + "0.weight"], H, F);
  put("O2", h[O + "2.weight"]);
  put("bO2", h[O + "2.bias"]);
  if (m->hp.has_atomref) put("atomref", h["atomref"]);
  // per-type tables U[z], V[z] of the pair embedding (weights only): filled once below by k_ztables
  put("Utab", std::vector<float>((size_t)m->hp.max_z * F, 0.f));
  put("Vtab", std::vector<float>((size_t)m->hp.max_z * F, 0.f));

}

int tmdnet_finalize_params(tmdnet_model* m) {
  if (!m) return TMDNET_ERR_INVALID;
  if (m->et) return et_finalize(m);
  for (const auto& sp : m->specs)
    if (!m->host.count(sp.name)) return fail(m, TMDNET_ERR_STATE, "missing parameter: " + sp.name);
  const int F = m->hp.hidden_channels, K = m->hp.num_rbf, L = m->hp.num_layers, H = m->hp.head_hidden;
  const std::string R = "representation_model.", T = R + "tensor_embedding.", O = "output_model.output_network.layers.";
  auto& h = m->host;
  Packer pk;
  std::map<std::string, size_t> off;
  pack_tensornet_params(m, h, pk, off);
  // device buffers are kept across uploads of the same size (a training loop re-uploads every step: hipFree / hipMalloc are
  // device-wide synchronisations)
  if (m->dev && m->dev_cap < pk.buf.size()) {
    HIP_TRY(m, hipFree(m->dev));
    m->dev = nullptr;
  }
  if (!m->dev) {
    HIP_TRY(m, hipMalloc(reinterpret_cast<void**>(&m->dev), pk.buf.size() * sizeof(float)));
    m->dev_cap = pk.buf.size();
  }
  HIP_TRY(m, hipMemcpy(m->dev, pk.buf.data(), pk.buf.size() * sizeof(float), hipMemcpyHostToDevice));
  auto D = [&](const std::string& key) -> const float* { return m->dev + off.at(key); };
  DevParams& P = m->P;
  P = DevParams{};
  P.means = D("means");
  P.betas = D("betas");
  P.Wdp = D("Wdp");
  P.bdp = D("bdp");
  P.emb = D("emb");
  P.emb2_w = D("emb2_w");
  P.emb2_b = D("emb2_b");
  P.emb2_waT = D("emb2_waT");
  P.emb2_wbT = D("emb2_wbT");
  for (int k = 0; k < 3; ++k) {
    P.Ue[k] = D("Ue" + std::to_string(k));
    P.UeT[k] = D("UeT" + std::to_string(k));
  }
  P.L1 = D("L1");
  P.bL1 = D("bL1");
  P.L1T = D("L1T");
  P.L2 = D("L2");
  P.bL2 = D("bL2");
  P.L2T = D("L2T");
  P.ln0_w = D("ln0_w");
  P.ln0_b = D("ln0_b");
  P.layer.resize(L);
  for (int l = 0; l < L; ++l) {
    const std::string t = "l" + std::to_string(l) + ".";
    LayerP& q = P.layer[l];
    q.M1 = D(t + "M0");
    q.b1 = D(t + "b0");
    q.M2 = D(t + "M1");
    q.b2 = D(t + "b1");
    q.M3 = D(t + "M2");
    q.b3 = D(t + "b2");
    q.M2T = m->tn2 ? nullptr : D(t + "M2T");
    q.M3T = m->tn2 ? nullptr : D(t + "M3T");
    for (int k = 0; k < 6; ++k) {
      q.V[k] = D(t + "V" + std::to_string(k));
      q.VT[k] = D(t + "VT" + std::to_string(k));
    }
  }
  P.lnr_w = D("lnr_w");
  P.lnr_b = D("lnr_b");
  P.Lin = D("Lin");
  P.bLin = D("bLin");
  P.LinT = D("LinT");
  P.O1 = D("O1");
  P.bO1 = D("bO1");
  P.O1T = D("O1T");
  P.O2 = D("O2");
  P.bO2 = D("bO2");
  P.atomref = m->hp.has_atomref ? D("atomref") : nullptr;
  P.mean = h["mean"][0];
  P.std = h["std"][0];
  {  // split-bf16 tile images (tn_gemm_sb.hip / tn_gemm_sb1.hip) of every GEMM weight, keyed by its fp32 device copy
    struct Img { std::string key; int64_t n, k; size_t o; };
    std::vector<Img> imgs;
    size_t sb_elems = 0;  // the images are made on the device from the uploaded fp32 copies (the host loop cost 4 - 6 ms per upload)
    auto add_sb = [&](const std::string& key, int64_t n, int64_t k) {  // packed matrix `key` is [n][k] row-major
      if ((k & 15) || (n & 3)) return;  // shapes the split kernels do not take stay on the fp32-MFMA kernels
      imgs.push_back({key, n, k, sb_elems});
      sb_elems += split_weight_elems(n, k);
    };
    add_sb("Wdp", 3 * F, K);
    for (int k = 0; k < 3; ++k) {
      add_sb("Ue" + std::to_string(k), F, F);
      add_sb("UeT" + std::to_string(k), F, F);
    }
    add_sb("L1", 2 * F, F);
    add_sb("L1T", F, 2 * F);
    add_sb("L2", 3 * F, 2 * F);
    add_sb("L2T", 2 * F, 3 * F);
    for (int l = 0; l < L; ++l) {
      const std::string t = "l" + std::to_string(l) + ".";
      add_sb(t + "M0", F, m->tn2 ? K + 2 * m->tn2->hp.q_dim : K);
      add_sb(t + "M1", 2 * F, F);
      add_sb(t + "M2", 3 * F, 2 * F);
      if (!m->tn2) {
        add_sb(t + "M2T", F, 2 * F);
        add_sb(t + "M3T", 2 * F, 3 * F);
      }
      for (int k = 0; k < 6; ++k) {
        add_sb(t + "V" + std::to_string(k), F, F);
        add_sb(t + "VT" + std::to_string(k), F, F);
      }
    }
    if (m->tn2) {
      const int qd = m->tn2->hp.q_dim;
      for (int l = 0; l < L; ++l) {
        const std::string t = "l" + std::to_string(l) + ".";
        add_sb(t + "M1a", F, K);
        add_sb(t + "M1b", F, qd);
        add_sb(t + "M1c", F, qd);
        add_sb(t + "M1bT", qd, F);
        add_sb(t + "M1cT", qd, F);
        add_sb(t + "M2T", F, 2 * F);
        add_sb(t + "M3T", 2 * F, 3 * F);
      }
      for (int l = 0; l <= L; ++l) {
        const std::string t = "cp" + std::to_string(l) + ".";
        add_sb(t + "W1", F, 3 * F);
        add_sb(t + "W1T", 3 * F, F);
        add_sb(t + "W2", F, F);
        add_sb(t + "W2T", F, F);
        add_sb(t + "W3", 2 * qd, F);
        add_sb(t + "W3T", F, 2 * qd);
      }
    }
    add_sb("Lin", F, 3 * F);
    add_sb("LinT", 3 * F, F);
    add_sb("O1", H, F);
    add_sb("O1T", F, H);
    // fragment-major images of the 9-component tensor linears' weights (tn_tlin9.hip), same buffer
    std::vector<Img> fms;
    auto add_fm = [&](const std::string& key) {
      if (F % 32) return;
      fms.push_back({key, F, F, sb_elems});
      sb_elems += split_weight_fm_elems(F, F);
    };
    for (int k = 0; k < 3; ++k) {
      add_fm("Ue" + std::to_string(k));
      add_fm("UeT" + std::to_string(k));
    }
    for (int l = 0; l < L; ++l)
      for (int k = 0; k < 6; ++k) {
        add_fm("l" + std::to_string(l) + ".V" + std::to_string(k));
        add_fm("l" + std::to_string(l) + ".VT" + std::to_string(k));
      }
    if (m->dev_sb && m->dev_sb_cap < sb_elems) {
      HIP_TRY(m, hipFree(m->dev_sb));
      m->dev_sb = nullptr;
    }
    if (!m->dev_sb) {
      HIP_TRY(m, hipMalloc(reinterpret_cast<void**>(&m->dev_sb), sb_elems * sizeof(uint16_t)));
      m->dev_sb_cap = sb_elems;
    }
    for (const auto& im : imgs) launch_split_weight_tiles(m->dev + off.at(im.key), im.n, im.k, m->dev_sb + im.o, nullptr);
    for (const auto& im : fms) launch_split_weight_fm(m->dev + off.at(im.key), im.n, im.k, m->dev_sb + im.o, nullptr);
    m->image_jobs.clear();
    for (const auto& im : imgs) m->image_jobs.push_back({off.at(im.key), im.n, im.k, im.o, 0});
    for (const auto& im : fms) m->image_jobs.push_back({off.at(im.key), im.n, im.k, im.o, 1});
    m->packed_elems = pk.buf.size();
    m->sb_of.clear();
    m->fm_of.clear();
    for (const auto& im : imgs) m->sb_of[m->dev + off.at(im.key)] = m->dev_sb + im.o;
    for (const auto& im : fms) m->fm_of[m->dev + off.at(im.key)] = m->dev_sb + im.o;
    auto sb_or_null = [&](const float* w) -> const uint16_t* {
      auto it = m->sb_of.find(w);
      return it == m->sb_of.end() ? nullptr : it->second;
    };
    P.Wdp_sb = sb_or_null(P.Wdp);
    for (int l = 0; l < L; ++l) {
      P.layer[l].M_sb[0] = sb_or_null(P.layer[l].M1);
      P.layer[l].M_sb[1] = sb_or_null(P.layer[l].M2);
      P.layer[l].M_sb[2] = sb_or_null(P.layer[l].M3);
    }
    if (m->tn2) {
      Tn2Model& t2 = *m->tn2;
      t2.layer.resize(L);
      t2.cp.resize(L + 1);
      for (int l = 0; l < L; ++l) {
        const std::string t = "l" + std::to_string(l) + ".";
        t2.layer[l] = Tn2LayerP{D(t + "M1a"), D(t + "M1b"), D(t + "M1c"), D(t + "M1bT"), D(t + "M1cT"), D(t + "M2T"), D(t + "M3T")};
      }
      for (int l = 0; l <= L; ++l) {
        const std::string t = "cp" + std::to_string(l) + ".";
        t2.cp[l] = CpParams{D(t + "ln_w"), D(t + "ln_b"), D(t + "W1"), D(t + "b1"), D(t + "W1T"), D(t + "W2"),
                            D(t + "b2"), D(t + "W2T"), D(t + "W3"), D(t + "b3"), D(t + "W3T")};
      }
      t2.qweights = D("qweights");
    }
  }
  {  // embedding in the radial basis: MFMA fragment images of the distance projections (tn_embed_rb.hip), made by a kernel from the
     // uploaded Wdp / bdp - the same kernel tmdnet_update_params_device runs after its gather
    m->rb_fwd = m->rb_rev = nullptr;
    const char* env = getenv("TMDNET_EMBED_RB");  // developer switch: 0 keeps the per-pair tables for the embedding
    if (!(env && atoi(env) == 0) && embed_rb_shape_ok(F, K)) {
      const size_t nf = embed_rb_image_elems(F, K, false), nr = embed_rb_image_elems(F, K, true);
      if (m->rb_img && m->rb_cap < nf + nr) {
        HIP_TRY(m, hipFree(m->rb_img));
        m->rb_img = nullptr;
      }
      if (!m->rb_img) {
        HIP_TRY(m, hipMalloc(reinterpret_cast<void**>(&m->rb_img), (nf + nr) * sizeof(uint16_t)));
        m->rb_cap = nf + nr;
      }
      launch_embed_rb_images(m->dev + off.at("Wdp"), m->dev + off.at("bdp"), F, K, m->rb_img, m->rb_img + nf, nullptr);
      m->rb_fwd = m->rb_img;
      m->rb_rev = m->rb_img + nf;
    }
  }
  P.Utab = m->dev + off.at("Utab");
  P.Vtab = m->dev + off.at("Vtab");
  launch_ztables(P.emb, P.emb2_waT, P.emb2_wbT, P.emb2_b, m->hp.max_z, F, m->dev + off.at("Utab"), m->dev + off.at("Vtab"), nullptr);
  HIP_TRY(m, hipStreamSynchronize(nullptr));
  free_radial_tables(m->tabs);  // rebuilt by the first call that uses them (ensure_radial_tables): fp64 evaluation + refinement,
  m->tabs_pending = true;      // tens of ms - not paid per parameter edit, and never on a stream that is being captured
  m->finalized = true;
  return TMDNET_OK;
}

// ---- device-side parameter update (training loops): the optimizer's tensors already live on the GPU
__global__ void k_param_gather(float* __restrict__ dst, const int32_t* __restrict__ tid, const int32_t* __restrict__ ofs,
                               const float* const* __restrict__ ptrs, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int t = tid[i];
  if (t < 0) return;
  const float* p = ptrs[t];
  if (p) dst[i] = p[ofs[i]];
}

int tmdnet_update_params_device(tmdnet_model* m, void* stream, int32_t count, const char* const* names, const float* const* dev_ptrs) {
  if (!m || count < 0 || (count > 0 && (!names || !dev_ptrs))) return TMDNET_ERR_INVALID;
  if (m->et) return fail(m, TMDNET_ERR_STATE, "device-side parameter update: not built for the Equivariant Transformer");
  if (!m->finalized || !m->dev || !m->packed_elems)
    return fail(m, TMDNET_ERR_STATE, "device-side parameter update needs one tmdnet_finalize_params first");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const size_t ns = m->specs.size();
  if (!m->upd_ready) {
    // where does every packed element come from?  Run the packing on INDEX TAGS (bit patterns; the packing only moves data)
    std::map<std::string, std::vector<float>> tags;
    std::vector<int64_t> base(ns + 1, 0);
    for (size_t i = 0; i < ns; ++i) base[i + 1] = base[i] + m->specs[i].rows * m->specs[i].cols;
    if (base[ns] + 1 >= (int64_t)0x7f000000) return fail(m, TMDNET_ERR_STATE, "device-side parameter update: model too large for the tag pass");
    for (size_t i = 0; i < ns; ++i) {
      std::vector<float>& v = tags[m->specs[i].name];
      v.resize((size_t)(base[i + 1] - base[i]));
      for (size_t j = 0; j < v.size(); ++j) {
        const uint32_t tag = (uint32_t)(base[i] + (int64_t)j + 1);  // 0 = "no source"
        std::memcpy(&v[j], &tag, 4);
      }
    }
    Packer pk;
    std::map<std::string, size_t> off;
    pack_tensornet_params(m, tags, pk, off);
    if (pk.buf.size() != m->packed_elems) return fail(m, TMDNET_ERR_STATE, "device-side parameter update: layout mismatch");
    std::vector<int32_t> tid(pk.buf.size(), -1), ofs(pk.buf.size(), 0);
    for (size_t i = 0; i < pk.buf.size(); ++i) {
      uint32_t tag;
      std::memcpy(&tag, &pk.buf[i], 4);
      if (!tag || (int64_t)tag > base[ns]) continue;
      const int64_t g = (int64_t)tag - 1;
      const size_t t = (size_t)(std::upper_bound(base.begin(), base.end(), g) - base.begin()) - 1;
      tid[i] = (int32_t)t;
      ofs[i] = (int32_t)(g - base[t]);
    }
    HIP_TRY(m, hipMalloc(reinterpret_cast<void**>(&m->upd_tid), tid.size() * sizeof(int32_t)));
    HIP_TRY(m, hipMalloc(reinterpret_cast<void**>(&m->upd_ofs), ofs.size() * sizeof(int32_t)));
    HIP_TRY(m, hipMalloc(reinterpret_cast<void**>(&m->upd_ptrs), ns * sizeof(float*)));
    HIP_TRY(m, hipMemcpy(m->upd_tid, tid.data(), tid.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_TRY(m, hipMemcpy(m->upd_ofs, ofs.data(), ofs.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    m->upd_ready = true;
  }
  std::vector<const float*> ptrs(ns, nullptr);
  for (int32_t c = 0; c < count; ++c) {
    size_t i = 0;
    for (; i < ns; ++i)
      if (m->specs[i].name == names[c]) break;
    if (i == ns) return fail(m, TMDNET_ERR_INVALID, std::string("unknown parameter: ") + names[c]);
    if (m->specs[i].name == "mean" || m->specs[i].name == "std")
      return fail(m, TMDNET_ERR_INVALID, "mean / std are kernel arguments read on the host: update them through tmdnet_set_param");
    ptrs[i] = dev_ptrs[c];
  }
  // the pointer table travels through the stream too (pageable host memory: the copy is staged before the call returns)
  HIP_TRY(m, hipMemcpyAsync(m->upd_ptrs, ptrs.data(), ns * sizeof(float*), hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_param_gather, dim3((unsigned)((m->packed_elems + 255) / 256)), dim3(256), 0, s, m->dev, m->upd_tid, m->upd_ofs,
                     m->upd_ptrs, (int64_t)m->packed_elems);
  for (const auto& j : m->image_jobs) {
    if (j.fm) launch_split_weight_fm(m->dev + j.src_off, j.n, j.k, m->dev_sb + j.dst_off, s);
    else launch_split_weight_tiles(m->dev + j.src_off, j.n, j.k, m->dev_sb + j.dst_off, s);
  }
  const int F = m->hp.hidden_channels;
  launch_ztables(m->P.emb, m->P.emb2_waT, m->P.emb2_wbT, m->P.emb2_b, m->hp.max_z, F, const_cast<float*>(m->P.Utab),
                 const_cast<float*>(m->P.Vtab), s);
  m->tabs_pending = true;           // radial tables: rebuilt by the first call that uses them
  if (m->rb_fwd)  // the radial-basis embedding's weight images follow the gather on the same stream
    launch_embed_rb_images(m->P.Wdp, m->P.bdp, F, m->hp.num_rbf, m->rb_img, m->rb_img + embed_rb_image_elems(F, m->hp.num_rbf, false), s);
  HIP_TRY(m, hipGetLastError());
  return TMDNET_OK;
}

int tmdnet_set_option(tmdnet_model* m, const char* name, double value) {
  if (!m || !name) return TMDNET_ERR_INVALID;
  const std::string n(name);
  if (n == "edge_table_min_pairs") {
    m->tab_min_pairs = value < 0 ? 0 : (int64_t)value;
    return TMDNET_OK;
  }
  if (n == "embed_rb_min_atoms") {
    m->rb_min_atoms = value < 0 ? 0 : (int64_t)value;
    return TMDNET_OK;
  }
  if (n == "pair_rows_bf16") {
    if (!m->et) return fail(m, TMDNET_ERR_INVALID, "pair_rows_bf16 applies to the Equivariant Transformer handle only");
    m->pair_bf16 = value != 0.0 ? 1 : 0;
    return TMDNET_OK;
  }
  if (n == "recompute_pair_rows") {
    if (m->et || m->tn2) return fail(m, TMDNET_ERR_INVALID, "recompute_pair_rows applies to the TensorNet handle only");
    m->recompute_rows = value != 0.0;
    return TMDNET_OK;
  }
  return fail(m, TMDNET_ERR_INVALID, "unknown option: " + n);
}

int tmdnet_get_info(const tmdnet_model* m, const char* name, double* value) {
  if (!m || !name || !value) return TMDNET_ERR_INVALID;
  const std::string n(name);
  if (m->tabs_pending && n.rfind("edge_table_", 0) == 0 && n != "edge_table_min_pairs")  // facts about tables not built yet
    if (const int rc = ensure_radial_tables(const_cast<tmdnet_model*>(m), nullptr)) return rc;
  if (n == "edge_table_T") *value = m->tabs.ok ? m->tabs.T : 0;
  else if (n == "edge_table_err_value") *value = m->tabs.err_value;
  else if (n == "edge_table_err_slope") *value = m->tabs.err_slope;
  else if (n == "edge_table_min_pairs") *value = (double)m->tab_min_pairs;
  else if (n == "pair_rows_bf16") *value = (double)m->pair_bf16;
  else if (n == "embed_rb_min_atoms") *value = (double)m->rb_min_atoms;
  else if (n == "recompute_pair_rows") *value = m->recompute_rows ? 1.0 : 0.0;
  else if (n == "embed_rb") *value = m->rb_fwd ? 1.0 : 0.0;
  else if (n == "species_last_build") *value = (double)m->last_nt;
  else if (n == "halo_active_first") *value = (double)m->halo_active[0];  // rows the per-atom kernels of the last step ran on
  else if (n == "halo_active_rows") *value = (double)m->halo_active[1];
  else return TMDNET_ERR_INVALID;
  return TMDNET_OK;
}

// ------------------------------------------------------------------------------------ graph
int tmdnet_set_cell_grid(tmdnet_model* m, int32_t ncx, int32_t ncy, int32_t ncz) {
  if (!m) return TMDNET_ERR_INVALID;
  if (ncx < 0 || ncy < 0 || ncz < 0) ncx = ncy = ncz = -1;  // automatic
  m->cell_n[0] = ncx;
  m->cell_n[1] = ncy;
  m->cell_n[2] = ncz;
  return TMDNET_OK;
}

int tmdnet_graph_workspace_bytes(const tmdnet_model* m, int64_t n_atoms, int64_t n_mol, size_t* bytes) {
  if (!m || !bytes || n_atoms < 0 || n_mol < 0) return TMDNET_ERR_INVALID;
  const int64_t ecap = (int64_t)m->hp.max_num_neighbors * n_atoms;
  carve_graph(nullptr, n_atoms, n_mol, ecap, bytes);
  return TMDNET_OK;
}

int tmdnet_build_graph(tmdnet_model* m, void* stream, void* graph_ws, size_t graph_ws_bytes, int64_t n_atoms, int64_t n_mol,
                       const float* pos, const int64_t* batch, const int64_t* z, const float* box, int32_t box_mode,
                       int64_t counts_host[8]) {
  if (!m || !graph_ws || !counts_host || n_atoms < 0 || n_mol < 0) return TMDNET_ERR_INVALID;
  if (n_atoms >= (int64_t)1 << 30) return fail(m, TMDNET_ERR_INVALID, "n_atoms too large for 32-bit indices");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int64_t ecap = (int64_t)m->hp.max_num_neighbors * n_atoms;
  size_t need = 0;
  Graph g = carve_graph(graph_ws, n_atoms, n_mol, ecap, &need);
  if (need > graph_ws_bytes) return fail(m, TMDNET_ERR_WORKSPACE, "graph workspace too small");
  if (box_mode != 0 && !box) return fail(m, TMDNET_ERR_INVALID, "box_mode != 0 needs a box");
  const bool cell = cell_applicable(m, n_atoms, n_mol, box_mode);
  set_cell(g, m, cell, n_mol);
  // halo exchange set before the graph is built (tmdnet_set_halo_exchange + tmdnet_set_atom_weights): ghost-ghost pairs are left out
  if (cell && m->halo_fn && m->atom_w && !m->et && !m->tn2) g.ghost_w = m->atom_w;
  m->graph_no_ghost_pairs = g.ghost_w != nullptr;
  m->graph_is_cell = cell;
  m->graph_cell_multi = cell && n_mol > 1;
  m->graph_has_z = z != nullptr;
  m->g_pos = pos;
  m->g_box = box;
  m->g_box_mode = box_mode;
  CurScope cur_(m);
  {
    ProfScope ps_(s, CAT_GRAPH, 0.0, (double)n_atoms * 20);
    if (cell) {
      launch_fill(reinterpret_cast<float*>(g.counts), 0.f, 8, s);
      launch_cell_phase1(g, pos, batch, box_mode == 1 ? box : nullptr, (int)n_atoms, m->hp.cutoff_lower, m->hp.cutoff_upper, true, s,
                         (int)n_mol);
      launch_scan_counts(g, (int)n_atoms, s);
    } else {
      launch_graph_build_phase1(g, pos, batch, box, box_mode, (int)n_atoms, (int)n_mol, m->hp.cutoff_lower, m->hp.cutoff_upper,
                                true, s);
    }
    if (z) launch_prepare_z(g, z, cell ? g.perm : nullptr, (int)n_atoms, m->hp.max_z, s);
    if (z && m->rb_fwd && n_atoms >= m->rb_min_atoms) launch_type_map(g, (int)n_atoms, m->hp.max_z, s);  // counts[6]
  }
  int counts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  HIP_TRY(m, hipMemcpyAsync(counts, g.counts, sizeof(counts), hipMemcpyDeviceToHost, s));
  HIP_TRY(m, hipStreamSynchronize(s));
  for (int k = 0; k < 8; ++k) counts_host[k] = counts[k];
  m->last_nt = (z && m->rb_fwd && n_atoms >= m->rb_min_atoms) ? counts[6] : 0;
  m->lastE = counts[1];
  remember_graph(m, graph_ws);
  if (counts[5])
    return fail(m, TMDNET_ERR_INVALID, "batch index out of range: every entry must be in [0, " + std::to_string(n_mol) + ")");
  if (counts[4])
    return fail(m, TMDNET_ERR_INVALID, "atomic number out of range: every z must be in [0, max_z = " + std::to_string(m->hp.max_z) + ")");
  if (counts[2])
    return fail(m, TMDNET_ERR_OVERFLOW, "Found num_pairs > max_num_pairs, please increase max_num_pairs (found " +
                                            std::to_string(counts[1]) + " edges, capacity " + std::to_string(ecap) + ")");
  m->lastE = counts[1];
  {
    ProfScope ps_(s, CAT_GRAPH, 0.0, (double)counts[1] * 12 + (double)counts[0] * 40);
    if (cell) {
      launch_cell_phase2(g, (int)n_atoms, m->hp.cutoff_lower, m->hp.cutoff_upper, true, s);
      launch_nbr_link_wave(g, (int)n_atoms, s);
    } else {
      launch_graph_build_phase2(g, pos, batch, box, box_mode, (int)n_atoms, (int)n_mol, m->hp.cutoff_lower, m->hp.cutoff_upper, true,
                                s);
    }
  }
  HIP_TRY(m, hipGetLastError());
  return TMDNET_OK;
}

int tmdnet_build_graph_static(tmdnet_model* m, void* stream, void* graph_ws, size_t graph_ws_bytes, int64_t n_atoms, int64_t n_mol,
                              const float* pos, const int64_t* batch, const int64_t* z, const float* box, int32_t box_mode) {
  if (!m || !graph_ws || n_atoms < 0 || n_mol < 0) return TMDNET_ERR_INVALID;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int64_t ecap = (int64_t)m->hp.max_num_neighbors * n_atoms;
  size_t need = 0;
  Graph g = carve_graph(graph_ws, n_atoms, n_mol, ecap, &need);
  if (need > graph_ws_bytes) return fail(m, TMDNET_ERR_WORKSPACE, "graph workspace too small");
  if (box_mode != 0 && !box) return fail(m, TMDNET_ERR_INVALID, "box_mode != 0 needs a box");
  const bool cell = cell_applicable(m, n_atoms, n_mol, box_mode);
  set_cell(g, m, cell, n_mol);
  m->graph_no_ghost_pairs = false;
  m->graph_is_cell = cell;
  m->graph_cell_multi = cell && n_mol > 1;
  m->graph_has_z = z != nullptr;
  m->g_pos = pos;
  m->g_box = box;
  m->g_box_mode = box_mode;
  CurScope cur_(m);
  {
    ProfScope ps_(s, CAT_GRAPH, 0.0, (double)n_atoms * 20);
    if (cell) {
      launch_fill(reinterpret_cast<float*>(g.counts), 0.f, 8, s);
      launch_cell_phase1(g, pos, batch, box_mode == 1 ? box : nullptr, (int)n_atoms, m->hp.cutoff_lower, m->hp.cutoff_upper, true, s,
                         (int)n_mol);
      launch_scan_counts(g, (int)n_atoms, s);
      launch_cell_phase2(g, (int)n_atoms, m->hp.cutoff_lower, m->hp.cutoff_upper, true, s);
      launch_nbr_link_wave(g, (int)n_atoms, s);
    } else if (graph_small_ok((int)n_atoms)) {
      launch_graph_small(g, pos, batch, box, box_mode, (int)n_atoms, (int)n_mol, m->hp.cutoff_lower, m->hp.cutoff_upper, true, z,
                         m->hp.max_z, s);
    } else {
      launch_graph_build_phase1(g, pos, batch, box, box_mode, (int)n_atoms, (int)n_mol, m->hp.cutoff_lower, m->hp.cutoff_upper,
                                true, s);
      launch_graph_build_phase2(g, pos, batch, box, box_mode, (int)n_atoms, (int)n_mol, m->hp.cutoff_lower, m->hp.cutoff_upper, true,
                                s);
    }
    if (z && (cell || !graph_small_ok((int)n_atoms))) launch_prepare_z(g, z, cell ? g.perm : nullptr, (int)n_atoms, m->hp.max_z, s);
  }
  m->lastE = ecap;
  m->last_nt = 0;  // static shapes: no read-back, the embedding keeps the per-pair tables
  remember_graph(m, graph_ws);
  HIP_TRY(m, hipGetLastError());
  return TMDNET_OK;
}

int tmdnet_graph_counts(tmdnet_model* m, void* stream, void* graph_ws, int64_t n_atoms, int64_t n_mol, int64_t counts_host[8]) {
  if (!m || !graph_ws || !counts_host) return TMDNET_ERR_INVALID;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  Graph g = carve_graph(graph_ws, n_atoms, n_mol, (int64_t)m->hp.max_num_neighbors * n_atoms, nullptr);
  int counts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  HIP_TRY(m, hipMemcpyAsync(counts, g.counts, sizeof(counts), hipMemcpyDeviceToHost, s));
  HIP_TRY(m, hipStreamSynchronize(s));
  for (int k = 0; k < 8; ++k) counts_host[k] = counts[k];
  if (counts[5]) return fail(m, TMDNET_ERR_INVALID, "batch index out of range: every entry must be in [0, " + std::to_string(n_mol) + ")");
  if (counts[4])
    return fail(m, TMDNET_ERR_INVALID, "atomic number out of range: every z must be in [0, max_z = " + std::to_string(m->hp.max_z) + ")");
  return counts[2] ? TMDNET_ERR_OVERFLOW : TMDNET_OK;
}

int tmdnet_set_atom_weights(tmdnet_model* m, const float* weights_dev) {
  if (!m) return TMDNET_ERR_INVALID;
  m->atom_w = weights_dev;
  return TMDNET_OK;
}

int tmdnet_set_halo_exchange(tmdnet_model* m, tmdnet_halo_exchange_fn fn, void* user) {
  if (!m) return TMDNET_ERR_INVALID;
  m->halo_fn = fn;
  m->halo_user = fn ? user : nullptr;
  if (fn && !m->halo_rng) HIP_TRY(m, hipMalloc(reinterpret_cast<void**>(&m->halo_rng), 4 * sizeof(int)));  // not inside the step
  return TMDNET_OK;
}

int tmdnet_graph_cell_grid(tmdnet_model* m, void* stream, void* graph_ws, int64_t n_atoms, int64_t n_mol, int64_t grid_host[4]) {
  if (!m || !graph_ws || !grid_host) return TMDNET_ERR_INVALID;
  recall_graph(m, graph_ws);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  Graph g = carve_graph(graph_ws, n_atoms, n_mol, (int64_t)m->hp.max_num_neighbors * n_atoms, nullptr);
  int grid[4] = {0, 0, 0, 0};
  if (m->graph_is_cell) {
    HIP_TRY(m, hipMemcpyAsync(grid, g.cgrid, sizeof(grid), hipMemcpyDeviceToHost, s));
    HIP_TRY(m, hipStreamSynchronize(s));
  }
  for (int k = 0; k < 3; ++k) grid_host[k] = grid[k];
  grid_host[3] = m->graph_is_cell ? 1 : 0;
  return TMDNET_OK;
}

// ------------------------------------------------------------------------------------ forward + reverse
int tmdnet_forward_workspace_bytes(const tmdnet_model* m, int64_t n_atoms, int64_t n_mol, int64_t n_pairs, int64_t n_edges,
                                   int32_t want_forces, size_t* bytes) {
  if (!m || !bytes) return TMDNET_ERR_INVALID;
  (void)n_edges;
  if (n_pairs < 0) n_pairs = ((int64_t)m->hp.max_num_neighbors * n_atoms) / 2 + 1;  // static mode: pair capacity
  if (m->et) return et_forward_workspace_bytes(m, n_atoms, n_mol, n_pairs, want_forces, bytes);
  if (m->tn2) return tn2_forward_workspace_bytes(m, n_atoms, n_mol, n_pairs, n_edges, want_forces, bytes);
  // the radial-basis embedding's buffers depend on the species count of the graph the call will run on: sized for the
  // largest padding (8) whenever that path can be taken, so the answer does not depend on which build came last
  const bool rb_possible = m->rb_fwd && !m->et && !m->train && n_pairs >= 0 && n_atoms >= m->rb_min_atoms;
  // recompute_pair_rows: the Q / dQ rows go only when the graph this handle built last takes the radial-basis embedding (its
  // species count is known by now: tmdnet_build_graph precedes this query); otherwise they stay in the plan
  carve_fwd(nullptr, m->hp, n_atoms, n_mol, n_pairs, want_forces != 0, bytes, rb_possible ? 8 : 0,
            m->recompute_rows ? (rb_possible && rb_ntp(m, n_atoms, n_pairs) > 0 ? 7 : 5) : 0);
  return TMDNET_OK;
}

int tmdnet_energy_forces(tmdnet_model* m, void* stream, void* graph_ws, void* ws, size_t ws_bytes, int64_t n_atoms,
                         int64_t n_mol, int64_t n_pairs, const int64_t* z, const int64_t* batch, const float* q,
                         int32_t want_forces, float* energy, float* forces) {
  if (!m || !graph_ws || !ws || !energy) return TMDNET_ERR_INVALID;
  if (!m->finalized) return fail(m, TMDNET_ERR_STATE, "parameters not finalised");
  if (want_forces && !forces) return TMDNET_ERR_INVALID;
  recall_graph(m, graph_ws);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const tmdnet_hparams& hp = m->hp;
  const int F = hp.hidden_channels, K = hp.num_rbf, L = hp.num_layers, Z = hp.max_z, H = hp.head_hidden;
  const int N = (int)n_atoms, B = (int)n_mol;
  const int64_t ecap = (int64_t)hp.max_num_neighbors * n_atoms;
  Graph g = carve_graph(graph_ws, n_atoms, n_mol, ecap, nullptr);
  if (n_pairs > g.pcap) return fail(m, TMDNET_ERR_INVALID, "n_pairs out of range");
  // stale radial tables are rebuilt here, by the first call that would use them (not by the parameter-gradient pass, whose
  // weights change every step, nor while option "edge_table_min_pairs" keeps this system on the direct evaluation)
  if (!m->train && (n_pairs >= 0 ? n_pairs : g.pcap) + 1 >= m->tab_min_pairs)
    if (const int rc_tab = ensure_radial_tables(m, s)) return rc_tab;
  if (m->graph_is_cell && m->graph_cell_multi) batch = g.bat_c;  // several molecules renumbered in cell order: their internal batch
  if (m->atom_w && (m->et || m->tn2 || m->train))
    return fail(m, TMDNET_ERR_INVALID, "atom weights (tmdnet_set_atom_weights) are implemented for TensorNet inference only");
  if (m->halo_fn && (m->et || m->tn2 || m->train || n_pairs < 0))
    return fail(m, TMDNET_ERR_INVALID, "the halo exchange (tmdnet_set_halo_exchange) is implemented for TensorNet inference with the "
                                       "exact pair count (tmdnet_build_graph, not the static mode)");
  if (m->et) {
    if (q) return fail(m, TMDNET_ERR_INVALID, "the Equivariant Transformer takes no total charge (reference torchmd_et.py:188-196)");
    CurScope cur_(m);
    if (m->graph_has_z) {  // validated + renumbered by the graph phase
      z = g.z_c;
    } else if (m->graph_is_cell) {
      set_cell(g, m, true, n_mol);
      launch_permute_z(g, z, (int)n_atoms, s);
      z = g.z_s;
    }
    if (!z) return fail(m, TMDNET_ERR_INVALID, "z is required (here or in tmdnet_build_graph)");
    return et_energy_forces(m, s, g, ws, ws_bytes, n_atoms, n_mol, n_pairs, z, batch, want_forces, energy, forces);
  }
  if (m->tn2) {
    CurScope cur_(m);
    if (m->graph_has_z) {
      z = g.z_c;
    } else if (m->graph_is_cell) {
      set_cell(g, m, true, n_mol);
      launch_permute_z(g, z, (int)n_atoms, s);
      z = g.z_s;
    }
    if (!z) return fail(m, TMDNET_ERR_INVALID, "z is required (here or in tmdnet_build_graph)");
    if (m->graph_is_cell) set_cell(g, m, true, n_mol);
    return tn2_energy_forces(m, s, g, ws, ws_bytes, n_atoms, n_mol, n_pairs, z, batch, q, want_forces, energy, forces);
  }
  // n_pairs >= 0: exact count read back by tmdnet_build_graph (launch grids sized exactly);
  // n_pairs <  0: static mode, grids and workspace sized by the pair capacity, true count read on the device
  const int P = n_pairs >= 0 ? (int)n_pairs : (int)g.pcap, P1 = P + 1;
  size_t need = 0;
  const int ntp = rb_ntp(m, n_atoms, n_pairs);  // > 0: embedding in the radial basis (no Q / dQ per pair)
  const bool recompute = m->recompute_rows;
  FwdBuffers b = carve_fwd(ws, hp, n_atoms, n_mol, P, want_forces != 0, &need, ntp, recompute ? (ntp ? 7 : 5) : 0);
  if (need > ws_bytes) return fail(m, TMDNET_ERR_WORKSPACE, "forward workspace too small: need " + std::to_string(need));
  const DevParams& W = m->P;
  const int o3 = hp.group_o3;

  CurScope cur_(m);
  const int* perm = nullptr;
  if (m->graph_is_cell) {  // the graph lives in cell order: renumber z here, scatter the forces back at the end
    set_cell(g, m, true, n_mol);
    if (!m->graph_has_z) {
      launch_permute_z(g, z, N, s);
      z = g.z_s;
    }
    perm = g.perm;
  }
  if (m->graph_has_z) z = g.z_c;  // validated (clamped) and renumbered by the graph phase
  if (!z) return fail(m, TMDNET_ERR_INVALID, "z is required (here or in tmdnet_build_graph)");
  const int64_t* batch_k = batch;
  if (q) {  // per-atom charge factor once; the kernels then read kap[n] (no dependent batch -> q gather, no range hazard)
    launch_kappa(q, batch, N, B, b.kap, s);
    q = b.kap;
    batch_k = nullptr;
  }
  const double E_ = (double)m->lastE, Nd = N, Pd = P, Fd = F;
  const double nodeB = Nd * 9 * Fd * 4;             // one [N,9,F] tensor
  // CSR sweep, algorithmic bytes (every distinct tensor once): the per-pair weights [P+1,3,F] are ONE tensor
  // although both rows of a pair read them; per directed edge only the indices (col, epair, esign)
  const double wB = (Pd + 1) * 12 * Fd, idxB = E_ * 12;
  auto EDGE = [&](int add) { g_gemm_cat = CAT_GEMM_EDGE; g_mdev = g.counts; g_madd = add; };
  auto NODE = [&]() { g_gemm_cat = CAT_GEMM_NODE; g_mdev = nullptr; g_madd = 0; };

  // ---- per-pair functions of the distance: Q (embedding) and every layer's w, with their d/dd when forces are wanted
  TrainCtx* const tc = m->train;  // parameter gradients wanted: direct evaluation with every pre-activation kept
  const bool use_tab = !tc && m->tabs.ok && (int64_t)P1 >= m->tab_min_pairs && (int)m->tabs.tab.size() == 1 + L && L + 1 <= 8;
  hipStream_t es = s;
  const RadialParams rbp{W.means, W.betas, K, hp.cutoff_lower, hp.cutoff_upper};
  const double momB = (double)embed_rb_moment_elems(N, ntp ? ntp : 4, K) * 4;
  // the parameter-gradient pass may run the two halves as separate calls (TrainCtx::phase): forward with everything kept in the
  // caller's workspaces, reverse once the seeds d loss / d E are known
  const bool run_fwd = !tc || tc->phase != 2, run_bwd = !tc || tc->phase != 1;
  // the slots of the merged distance gradient are what the fused reverse sweeps write: both or neither
  if (recompute && (!use_tab || (want_forces && !(message_adjoint_gd_ok(N, F) && !getenv("TMDNET_SEPARATE_PAIR_GD")))))
    return fail(m, TMDNET_ERR_STATE, "option recompute_pair_rows needs the radial tables (verified, not switched off) and the merged "
                                     "distance gradient; TensorNet inference only");
  PairRowTable rts[8];
  for (int l = 0; l < L && recompute; ++l) {
    const float h_ = (hp.cutoff_upper - hp.cutoff_lower) / (float)m->tabs.T;  // the expressions of launch_edge_interp
    rts[l] = PairRowTable{m->tabs.tab[1 + l], g.pd, g.counts, hp.cutoff_lower, h_, 1.0f / h_, m->tabs.T};
  }
  const bool use_mid = !small_fused_ok(N, F, H, L);  // 513 .. 1 024 atoms: four atoms per block (tn_mid.hip)
  // ghost rows refreshed between the kernels of a step (tmdnet_set_halo_exchange): 0 = ok
  auto HALO = [&](int stage, float* rows, int row_floats) -> int {
    if (!m->halo_fn) return 0;
    return m->halo_fn(m->halo_user, stage, rows, (int64_t)N, (int64_t)row_floats, perm, (void*)s);
  };
#define HALO_TRY(stage, rows, row_floats) \
  do { if (HALO(stage, rows, row_floats)) return fail(m, TMDNET_ERR_STATE, "halo exchange callback failed at stage " + std::to_string(stage)); } while (0)
  // Halo exchange with the owned atoms CONTIGUOUS in the engine's order (parallel.HaloExchangeEvaluator aligns its slabs with the
  // cell grid, whose major axis is x): the per-atom kernels run on the owned rows [a0, a0 + Na) only.  Nothing of a ghost is read
  // but its rows of the tensors a sweep gathers, and those come from the exchange; the sweeps, the pair kernels and the embedding's
  // neighbour sums keep all rows (a ghost's half of a pair's distance gradient is computed at the ghost's row).
  int a0 = 0, Na = N;
  if (m->halo_fn && m->halo_rng && m->atom_w && m->graph_is_cell && want_forces && !tc && (int64_t)N > 256 * (int64_t)B) {
    int h[4] = {0, 0, 0, 0};
    launch_owned_range(m->atom_w, perm, N, m->halo_rng, s);
    HIP_TRY(m, hipMemcpyAsync(h, m->halo_rng, sizeof(h), hipMemcpyDeviceToHost, s));
    HIP_TRY(m, hipStreamSynchronize(s));
    const bool images = tlin9_images(W.Ue) && tlin9_images(W.UeT) && L > 0 && tlin9_images(W.layer[0].V) && tlin9_images(W.layer[0].VT);
    if (h[2] > 0 && h[2] < N && h[1] - h[0] + 1 == h[2] && images && tlin9_ok(h[2], F, true)) {  // only the fused tensor linears take a row range
      a0 = h[0];
      Na = h[2];
    }
  }
  const bool rng = Na != N;  // then the fused tensor linears are the only schedule (checked above), whatever their last round of tiles
  const bool t9 = !tc && tlin9_ok(Na, F, rng) && tlin9_images(W.Ue) && tlin9_images(W.UeT) && (L == 0 || tlin9_images(W.layer[0].V));
  const bool t9r = !tc && L > 0 && tlin9_ok(Na, F, rng) && tlin9_images(W.UeT) && tlin9_images(W.layer[0].VT);
  m->halo_active[0] = a0;
  m->halo_active[1] = Na;
  const int64_t o1_ = a0, oF = (int64_t)a0 * F, o2F = 2 * oF, o3F = 3 * oF, o9F = 9 * oF, oH = (int64_t)a0 * H;
  const float* const q_a = q ? q + o1_ : nullptr;                    // per-atom charge factor (kappa) of the active rows
  const int64_t* const batch_a = batch_k ? batch_k + o1_ : nullptr;  // their molecule index
  const bool fused_small = !tc && !ntp && !recompute && !m->halo_fn && (small_fused_ok(N, F, H, L) || mid_fused_ok(N, F, H, L)) &&
                           (!want_forces || (message_adjoint_gd_ok(N, F) && !getenv("TMDNET_SEPARATE_PAIR_GD")));
  if (run_fwd) {
    if (use_tab) {
      // radial tables (tn_edge_table.hip): sort the pairs by distance, one streaming Hermite-interpolation kernel for all
      // tables; no basis functions, no pair-row GEMMs in the step
      const float* tabs[8];
      float* outs[8];
      float* douts[8];
      int nt_ = 0;
      if (!ntp) {  // Q(d) of the embedding: only when the per-pair form runs (the radial-basis form needs no per-pair rows)
        tabs[nt_] = m->tabs.tab[0];
        outs[nt_] = b.Q;
        douts[nt_++] = want_forces ? b.dQ : nullptr;
      }
      for (int l = 0; l < L && !recompute; ++l) {  // recompute: no per-pair rows of the layers (the sweeps interpolate)
        tabs[nt_] = m->tabs.tab[1 + l];
        outs[nt_] = b.w[l];
        douts[nt_++] = want_forces ? b.dw[l] : nullptr;
      }
      const double rowB = 12.0 * Fd;
      if (nt_ > 0)
        KR(CAT_EDGE_TABLE, (Pd + 1) * (rowB * nt_ * (want_forces ? 2 : 1) + 24) + (double)(m->tabs.T + 2) * 2 * rowB * nt_,
           (launch_pair_buckets(g, P, hp.cutoff_lower, hp.cutoff_upper, m->tabs.T, b.C, b.dC, b.shist, b.skeys_s, b.svals_s, s),
            launch_edge_interp(g, P, hp.cutoff_lower, hp.cutoff_upper, m->tabs.T, 3 * F, nt_, tabs, outs, douts, b.skeys_s, b.svals_s, s, b.C,
                               b.dC)));
    } else {
      // ---- radial functions per pair
      RadialParams rp{W.means, W.betas, K, hp.cutoff_lower, hp.cutoff_upper};
      KR(CAT_ELEMENTWISE, Pd * (2 * K + 3) * 4, launch_radial(g, P, rp, b.phi, b.dphi, b.C, b.dC, s));
      // ---- edge MLPs of all layers: functions of the pair geometry only -> side stream (tn_model.h)
      // only at batch scale: for a small system the cross-queue joins cost more than the overlap gives (graph replay of a
      // 64-atom molecule 0.36 -> 0.40 ms, profiles/r01_notes.md)
      es = (m->side && L > 0 && P >= 16384 && !tc) ? m->side : s;
      if (es != s) {
        HIP_TRY(m, hipEventRecord(m->ev_fork, s));
        HIP_TRY(m, hipStreamWaitEvent(es, m->ev_fork, 0));
      }
      for (int l = 0; l < L; ++l) {
        const LayerP& q_ = W.layer[l];
        EDGE(1);
        if (tc) {  // values only, pre-activations kept for the weight gradients
          gemm(es, b.phi, K, q_.M1, K, q_.b1, tc->he1[l], F, P1, F, K, GEMM_ACT_SILU, tc->pre1[l], F);
          gemm(es, tc->he1[l], F, q_.M2, F, q_.b2, tc->he2[l], 2 * F, P1, 2 * F, F, GEMM_ACT_SILU, tc->pre2[l], 2 * F);
          gemm(es, tc->he2[l], 2 * F, q_.M3, 2 * F, q_.b3, b.w[l], 3 * F, P1, 3 * F, 2 * F, GEMM_ACT_SILU | GEMM_ROWSCALE, tc->pre3[l], 3 * F,
               nullptr, 0, b.C);
        } else if (want_forces) {
          // edge MLP with its distance tangent carried forward (dw/dd): the reverse pass then needs no edge GEMM
          gemm_dual(es, 1, b.phi, b.dphi, K, q_.M1, q_.b1, b.he1, b.te1, F, P1, F, K, nullptr, nullptr, q_.M_sb[0]);
          gemm_dual(es, 1, b.he1, b.te1, F, q_.M2, q_.b2, b.he2, b.te2, 2 * F, P1, 2 * F, F, nullptr, nullptr, q_.M_sb[1]);
          gemm_dual(es, 2, b.he2, b.te2, 2 * F, q_.M3, q_.b3, b.w[l], b.dw[l], 3 * F, P1, 3 * F, 2 * F, b.C, b.dC, q_.M_sb[2]);
        } else {
          gemm(es, b.phi, K, q_.M1, K, q_.b1, b.he1, F, P1, F, K, GEMM_ACT_SILU);
          gemm(es, b.he1, F, q_.M2, F, q_.b2, b.he2, 2 * F, P1, 2 * F, F, GEMM_ACT_SILU);
          gemm(es, b.he2, 2 * F, q_.M3, 2 * F, q_.b3, b.w[l], 3 * F, P1, 3 * F, 2 * F, GEMM_ACT_SILU | GEMM_ROWSCALE, nullptr, 0, nullptr, 0,
               b.C);
        }
        if (es != s) HIP_TRY(m, hipEventRecord(m->ev_join[l], es));
      }
      // ---- embedding
      // per-type tables: Zij = emb2([emb(z_i), emb(z_j)]) = U[z_i] + V[z_j]   (reference tensornet.py:526-541)
      EDGE(1);
      if (ntp) {
      } else if (tc) gemm(s, b.phi, K, W.Wdp, K, W.bdp, b.Q, 3 * F, P1, 3 * F, K);
      else if (want_forces) gemm_dual(s, 0, b.phi, b.dphi, K, W.Wdp, W.bdp, b.Q, b.dQ, 3 * F, P1, 3 * F, K, nullptr, nullptr, W.Wdp_sb);  // distance projections + d/dd
      else gemm(s, b.phi, K, W.Wdp, K, W.bdp, b.Q, 3 * F, P1, 3 * F, K);
    }
    if (fused_small) {
      // small systems: the node side of the step as 1 + L (+ L) per-atom kernels cut at the neighbour sweeps (tn_small.hip)
      SmallEmbedArgs ea{};
      ea.g = g; ea.N = N; ea.F = F; ea.L = L; ea.z = z;
      ea.Utab = W.Utab; ea.Vtab = W.Vtab; ea.Q = b.Q; ea.C = b.C;
      ea.ln0_w = W.ln0_w; ea.ln0_b = W.ln0_b; ea.L1T = W.L1T; ea.bL1 = W.bL1; ea.L2T = W.L2T; ea.bL2 = W.bL2;
      for (int t = 0; t < 3; ++t) {
        ea.UeT[t] = W.UeT[t];
        ea.V0T[t] = W.layer[0].VT[t];
      }
      ea.u0 = b.u0; ea.xh0 = b.xh0; ea.rstd0 = b.rstd0; ea.a1 = b.a1; ea.a2 = b.a2; ea.gates = b.gates; ea.UX = b.UX; ea.X0 = b.X[0];
      ea.Pn0 = b.Pn[0];
      KR(CAT_SCATTER, Pd * 12 * Fd + E_ * 12 + 5 * nodeB, (use_mid ? launch_mid_embed(ea, s) : launch_small_embed(ea, s)));
      for (int l = 0; l < L; ++l) {
        const LayerP& q_ = W.layer[l];
        const bool last = l + 1 == L;
        SmallLayerArgs la{};
        la.g = g; la.N = N; la.F = F; la.H = H; la.o3 = o3; la.want_forces = want_forces ? 1 : 0; la.z = z; la.kap = q; la.atom_w = m->atom_w; la.perm = perm;
        la.w = b.w[l]; la.Pn = b.Pn[l]; la.X = b.X[l]; la.Mi = b.Mi[l]; la.D = b.D[l]; la.Xn = b.X[l + 1];
        for (int t = 0; t < 3; ++t) {
          la.VbT[t] = q_.VT[3 + t];
          la.Vb[t] = q_.V[3 + t];
          if (!last) la.VnT[t] = W.layer[l + 1].VT[t];
        }
        if (!last) la.Pn_next = b.Pn[l + 1];
        la.lnr_w = W.lnr_w; la.lnr_b = W.lnr_b; la.LinT = W.LinT; la.bLin = W.bLin; la.O1T = W.O1T; la.bO1 = W.bO1; la.O2 = W.O2;
        la.bO2 = W.bO2; la.atomref = W.atomref; la.Lin = W.Lin; la.O1 = W.O1; la.std_ = W.std;
        la.xhr = b.xhr; la.rstdr = b.rstdr; la.al = b.al; la.x = b.x; la.ea = b.ea;
        if (want_forces) { la.G = b.G; la.gMi = b.gMi; la.gPn = b.gPn; }
        KR(CAT_MESSAGE, wB + idxB + 6 * nodeB, (use_mid ? launch_mid_layer(la, last, s) : launch_small_layer(la, last, s)));
      }
      if (!want_forces) KR(CAT_ELEMENTWISE, Nd * 4, launch_mol_sum(g, b.ea, batch, N, B, W.mean, energy, s));
    } else {
    if (ntp) {
      // embedding in the radial basis (tn_embed_rb.hip): moments per (atom, species, component), then the per-atom contraction
      KR(CAT_SCATTER, E_ * 12 + Pd * 48 + momB,
         (launch_pair_scalars(g, P, hp.cutoff_lower, hp.cutoff_upper, b.ps, s), launch_embed_moments(g, N, rbp, ntp, b.ps, b.mom, s)));
      KR(CAT_SCATTER, momB + Nd * 10 * Fd * 4, launch_embed_combine(g, N, F, K, ntp, z, W.Utab, W.Vtab, m->rb_fwd, b.mom, b.u0, b.s0n, s));
    } else {
      KR(CAT_SCATTER, Pd * 12 * Fd + E_ * 12 + Nd * 10 * Fd * 4,
         launch_embed_scatter(g, N, F, z, W.Utab, W.Vtab, b.Q, b.C, b.u0, b.s0n, s));
    }
    KR(CAT_ELEMENTWISE, Nd * Fd * 12,
       launch_layernorm_fwd(b.s0n + oF, W.ln0_w, W.ln0_b, Na, F, b.ln0 + oF, b.xh0 + oF, b.rstd0 + o1_, s));
    NODE();
    gemm(s, b.ln0 + oF, F, W.L1, F, W.bL1, b.h1 + o2F, 2 * F, Na, 2 * F, F, GEMM_ACT_SILU, b.a1 + o2F, 2 * F);
    gemm(s, b.h1 + o2F, 2 * F, W.L2, 2 * F, W.bL2, b.gates + o3F, 3 * F, Na, 3 * F, 2 * F, GEMM_ACT_SILU, b.a2 + o3F, 3 * F);
    // fused 9-component tensor linears (tn_tlin9.hip) at batch scale; the parameter-gradient pass keeps the unfused schedule,
    // whose intermediates (X_hat, C_hat, g_D, g_C_hat per layer) are operands of its weight-gradient products
    if (t9) {
      Tl9Args ta{};
      ta.A = b.u0 + o9F; ta.C = b.X[0] + o9F; ta.o1 = b.UX + o9F; ta.e3 = b.gates + o3F; ta.N = Na; ta.F = F;
      tlin9(s, TL9_PRO_PLAIN, TL9_EPI_MULGATE, W.Ue, ta, 3.0 + 1.0 / 3.0, "gate");
    } else
      tensor_linear(s, b.u0, W.Ue, b.X[0], N, F, GEMM_MUL_AUX, b.UX, b.gates);
    // ---- interaction layers
    for (int l = 0; l < L; ++l) {
      const LayerP& q_ = W.layer[l];
      if (es != s) HIP_TRY(m, hipStreamWaitEvent(s, m->ev_join[l], 0));  // join: w[l] (and dw[l]) are ready
      // X_hat of layer l > 0 was written by the previous layer's update kernel (in place over its own X_hat)
      float* const Xh_l = tc ? tc->Xh[l] : b.Xh;  // kept per layer when parameter gradients are wanted
      float* const Ch_l = tc ? tc->Ch[l] : b.Ch;
      float* const Xh_n = tc && l + 1 < L ? tc->Xh[l + 1] : b.Xh;
      if (t9) {
        // X / (||X||^2 + 1) while the rows are staged; X_hat is never stored (the update recomputes it from X)
        Tl9Args ta{};
        ta.A = b.X[l] + o9F; ta.C = b.Pn[l] + o9F; ta.N = Na; ta.F = F;
        tlin9(s, TL9_PRO_NORM, TL9_EPI_PLAIN, q_.V, ta, 2.0, "norm");
        HALO_TRY(l, b.Pn[l], 9 * F);
        KR(CAT_MESSAGE, wB + idxB + nodeB * (1.0 + 2.0 * Na / Nd),  // P of every local atom is gathered; Mi, C_hat of the swept rows are written
           launch_message(g, N, F, b.w[l], b.Pn[l], q, batch_k, o3, b.Mi[l], Ch_l, s, recompute ? &rts[l] : nullptr, a0, rng ? Na : -1));
        // dX = linear(C_hat), then X_new = X_hat + dX + kappa dX.dX (and the readout invariants after the last layer) in the epilogue
        Tl9Args tb{};
        tb.A = Ch_l + o9F; tb.C = b.D[l] + o9F; tb.e0 = b.X[l] + o9F; tb.o1 = b.X[l + 1] + o9F; tb.o2 = b.feat + o3F;
        tb.want_feat = l + 1 == L; tb.kap = q_a;
        tb.N = Na; tb.F = F;
        tlin9(s, TL9_PRO_PLAIN, TL9_EPI_UPDATE, q_.V + 3, tb, 4.0 + (l + 1 == L ? 1.0 / 3.0 : 0.0), "update");
        continue;
      }
      if (l == 0) KR(CAT_ELEMENTWISE, 2 * nodeB, launch_norm_x(b.X[l], Xh_l, N, F, s));
      tensor_linear(s, Xh_l, q_.V, b.Pn[l], N, F);
      HALO_TRY(l, b.Pn[l], 9 * F);
      KR(CAT_MESSAGE, wB + idxB + 3 * nodeB, launch_message(g, N, F, b.w[l], b.Pn[l], q, batch_k, o3, b.Mi[l], Ch_l, s, recompute ? &rts[l] : nullptr));
      tensor_linear(s, Ch_l, q_.V + 3, b.D[l], N, F);
      // update fused with the next consumer of the new X: the next layer's normalisation, or the readout invariants
      KR(CAT_ELEMENTWISE, 4 * nodeB, launch_layer_update(Xh_l, b.D[l], q, batch_k, N, F, b.X[l + 1], l + 1 < L ? 1 : 2,
                                                         l + 1 < L ? Xh_n : b.feat, s));
    }
    // ---- readout + head + per-molecule sum
    if (L == 0) KR(CAT_ELEMENTWISE, nodeB + Nd * 3 * Fd * 4, launch_readout_feat(b.X[L] + o9F, Na, F, b.feat + o3F, s));
    KR(CAT_ELEMENTWISE, Nd * 3 * Fd * 12,
       launch_layernorm_fwd(b.feat + o3F, W.lnr_w, W.lnr_b, Na, 3 * F, b.lnr + o3F, b.xhr + o3F, b.rstdr + o1_, s));
    NODE();
    gemm(s, b.lnr + o3F, 3 * F, W.Lin, 3 * F, W.bLin, b.x + oF, F, Na, F, 3 * F, GEMM_ACT_SILU, b.al + oF, F);
    gemm(s, b.x + oF, F, W.O1, F, W.bO1, b.ao + oH, H, Na, H, F);
    if ((int64_t)N <= 256 * (int64_t)B) {  // small molecules: head + per-molecule sum in one launch (a block walks its molecule)
      KR(CAT_ELEMENTWISE, Nd * H * 4, launch_head_mol_sum(g, b.ao, W.O2, W.bO2, N, B, H, W.std, W.atomref, z, batch, W.mean, energy, s,
                                                          want_forces ? b.g_ao : nullptr, m->atom_w, perm));
    } else {
      if (Na != N) launch_fill(b.ea, 0.f, N, s);  // the ghosts' rows of the per-atom energies: not computed, not counted
      KR(CAT_ELEMENTWISE, Nd * H * 4,
         launch_head_energy(b.ao + oH, W.O2, W.bO2, Na, H, W.std, W.atomref, z + o1_, b.ea + o1_, s, want_forces ? b.g_ao + oH : nullptr,
                            perm ? m->atom_w : (m->atom_w ? m->atom_w + o1_ : nullptr), perm ? perm + o1_ : nullptr));
      KR(CAT_ELEMENTWISE, Nd * 4, launch_mol_sum(g, b.ea, batch, N, B, W.mean, energy, s));
    }
    }  // !fused_small
  }

  if (want_forces && run_bwd && fused_small) {
    const int gd_nw = message_adjoint_gd_waves(g, N, F, recompute);
    const int64_t gd_stride = 2 * (int64_t)P1;
    for (int l = L - 1; l >= 0; --l) {
      const LayerP& q_ = W.layer[l];
      // the adjoint sweep reads its neighbours' gMi while the same launch writes the layer below's: two buffers in turn
      float* const gMi_in = ((L - 1 - l) & 1) ? b.gCh : b.gMi;
      float* const gMi_out = ((L - 1 - l) & 1) ? b.gMi : b.gCh;
      SmallRevArgs ra{};
      ra.g = g; ra.N = N; ra.F = F; ra.B = B; ra.o3 = o3; ra.first = l == 0; ra.kap = q;
      ra.w = b.w[l]; ra.dw = b.dw[l]; ra.gMi_in = gMi_in; ra.Pn = b.Pn[l]; ra.X = b.X[l];
      ra.gPn = b.gPn; ra.G = b.G; ra.slots = b.gd_slots + (int64_t)l * gd_nw * gd_stride; ra.slot_stride = gd_stride;
      for (int t = 0; t < 3; ++t) {
        ra.Va[t] = q_.V[t];
        ra.Ue[t] = W.Ue[t];
        if (l > 0) ra.Vb_prev[t] = W.layer[l - 1].V[3 + t];
      }
      if (l > 0) { ra.D_prev = b.D[l - 1]; ra.Pn_prev = b.Pn[l - 1]; ra.Mi_prev = b.Mi[l - 1]; ra.gMi_out = gMi_out; }
      ra.UX = b.UX; ra.gates = b.gates; ra.a2 = b.a2; ra.a1 = b.a1; ra.L2 = W.L2; ra.L1 = W.L1; ra.xh0 = b.xh0; ra.rstd0 = b.rstd0;
      ra.ln0_w = W.ln0_w; ra.u0 = b.u0; ra.gA = b.gA;
      if (l == L - 1) { ra.ea = b.ea; ra.batch = batch; ra.mean = W.mean; ra.energy = energy; }
      KR(CAT_MESSAGE, 2 * wB + idxB + 6 * nodeB + 8 * (Pd + 1) * gd_nw, (use_mid ? launch_mid_rev(ra, s) : launch_small_rev(ra, s)));
    }
    KR(CAT_PAIR, Pd * (24 * Fd + 24) + Nd * 10 * Fd * 4,
       launch_embed_pair_gd(g, P, F, z, W.Utab, W.Vtab, b.Q, b.dQ, b.C, b.dC, b.gA, b.gd, b.g_rhat, s, b.g_delta, b.gd_slots, L * gd_nw,
                            gd_stride));
    KR(CAT_ELEMENTWISE, E_ * 8 + Nd * 12, launch_force_gather(g, N, b.g_delta, perm, forces, s));
  } else if (want_forces && run_bwd) {
    NODE();  // g_ao = d energy / d ao came out of the head kernel
    const RowMap rH = rows_plain(H), rF = rows_plain(F), r2F = rows_plain(2 * F), r3F = rows_plain(3 * F), rK = rows_plain(K);
    const RowMap rc_[3] = {rows_comp(F, 1), rows_comp(F, 3), rows_comp(F, 5)};  // (atom, component) rows of I / A / S in [N, 9, F]
    const int c0_[3] = {0, 1, 4}, nc_[3] = {1, 3, 5};
    // dW[t] (+)= sum over the rows of type t of gOut^T In  (the three weight sets of a 9-component tensor linear)
    auto tensor_linear_grad = [&](const float* gOut, const float* In, const std::string& key) {
      for (int t = 0; t < 3; ++t)
        launch_tn_gemm(s, gOut + (int64_t)c0_[t] * F, rc_[t], In + (int64_t)c0_[t] * F, rc_[t], nullptr, nullptr, N * nc_[t], F, F,
                       tc->at(key + std::to_string(t)), false, tc->part);
    };
    if (tc) {
      // seeds: every adjoint below is linear in g_ao, so scaling its rows by d loss / d E_mol(i) is all it takes
      launch_train_seed(b.ao, tc->gE, batch, N, H, W.std, b.g_ao, tc->head, s);
      launch_colsum(s, tc->head, rows_plain(H + 1), nullptr, rH, nullptr, nullptr, N, H, tc->at("O2"), false, tc->part);
      launch_colsum(s, tc->head + H, rows_plain(H + 1), nullptr, rH, nullptr, nullptr, N, 1, tc->at("bO2"), false, tc->part);
      launch_tn_gemm(s, b.g_ao, rH, b.x, rF, nullptr, nullptr, N, H, F, tc->at("O1"), false, tc->part);
      launch_colsum(s, b.g_ao, rH, nullptr, rH, nullptr, nullptr, N, H, tc->at("bO1"), false, tc->part);
    }
    gemm(s, b.g_ao + oH, H, W.O1T, H, nullptr, b.g_al + oF, F, Na, F, H, GEMM_MUL_DSILU_AUX, nullptr, 0, b.al + oF, F);
    gemm(s, b.g_al + oF, F, W.LinT, F, nullptr, b.g_ln + o3F, 3 * F, Na, 3 * F, F);
    if (tc) {
      launch_tn_gemm(s, b.g_al, rF, b.lnr, r3F, nullptr, nullptr, N, F, 3 * F, tc->at("Lin"), false, tc->part);
      launch_colsum(s, b.g_al, rF, nullptr, rF, nullptr, nullptr, N, F, tc->at("bLin"), false, tc->part);
      launch_colsum(s, b.g_ln, r3F, b.xhr, r3F, nullptr, nullptr, N, 3 * F, tc->at("lnr_w"), false, tc->part);
      launch_colsum(s, b.g_ln, r3F, nullptr, r3F, nullptr, nullptr, N, 3 * F, tc->at("lnr_b"), false, tc->part);
    }
    if (F % 64 == 0) {
      KR(CAT_ELEMENTWISE, 2 * nodeB + Nd * 3 * Fd * 8,
         launch_lnbwd_readout_bwd(b.g_ln + o3F, b.xhr + o3F, b.rstdr + o1_, W.lnr_w, Na, F, b.X[L] + o9F, b.G + o9F, s));
    } else {
      KR(CAT_ELEMENTWISE, Nd * 3 * Fd * 12,
         launch_layernorm_bwd(b.g_ln + o3F, b.xhr + o3F, b.rstdr + o1_, W.lnr_w, Na, 3 * F, b.g_feat + o3F, s));
      KR(CAT_ELEMENTWISE, 2 * nodeB + Nd * 3 * Fd * 4, launch_readout_bwd(b.X[L] + o9F, b.g_feat + o3F, Na, F, b.G + o9F, s));
    }
    // zero-fills are kernels, not hipMemsetAsync: memset nodes captured into a HIP graph were observed not to
    // re-execute on replay (ROCm 7.2), which silently accumulated gC / g_phi across MD steps
    const bool merged_gd = message_adjoint_gd_ok(N, F) && !getenv("TMDNET_SEPARATE_PAIR_GD") && !tc;
    const int gd_nw = message_adjoint_gd_waves(g, N, F, recompute);
    const int64_t gd_stride = 2 * (int64_t)P1;
    if (!merged_gd) launch_fill(b.gd, 0.f, P1, s);  // the per-layer pair kernels accumulate into it
    for (int l = L - 1; l >= 0; --l) {
      const LayerP& q_ = W.layer[l];
      if (t9r) {
        // update adjoint while G is staged (g_D never reaches memory), transposed linear; the adjoint of the group product stays a
        // kernel of its own: in the epilogue its 3x3 temporaries spilled and the fused launch lost to the pair (180 vs 77 + 62 us)
        Tl9Args ta{};
        ta.A = b.G + o9F; ta.A2 = b.D[l] + o9F; ta.kap = q_a; ta.N = Na; ta.F = F; ta.C = b.gCh + o9F;
        tlin9(s, TL9_PRO_UPDBWD, TL9_EPI_PLAIN, q_.VT + 3, ta, 3.0, "updbwd");
        KR(CAT_ELEMENTWISE, 5 * nodeB, launch_message_bwd_node(b.gCh + o9F, b.Pn[l] + o9F, b.Mi[l] + o9F, q_a, batch_a, o3, Na, F,
                                                               b.gMi + o9F, b.gPn + o9F, s));
      } else {
      // gD of the layers below the top one comes out of the previous iteration's fused normalisation adjoint
      if (l == L - 1) KR(CAT_ELEMENTWISE, 3 * nodeB, launch_update_bwd(b.G, b.D[l], q, batch_k, N, F, b.gD, s));
      if (tc) tensor_linear_grad(b.gD, tc->Ch[l], "l" + std::to_string(l) + ".V" /* 3..5 */ + std::string("b"));
      tensor_linear(s, b.gD, q_.VT + 3, b.gCh, N, F);
      KR(CAT_ELEMENTWISE, 5 * nodeB, launch_message_bwd_node(b.gCh, b.Pn[l], b.Mi[l], q, batch_k, o3, N, F, b.gMi, b.gPn, s));
      }
      HALO_TRY(100 + l, b.gMi, 9 * F);
      if (tc) {
        // edge MLP of this layer: g_w per pair (self pair: summed over the atoms), then back through silu(.) C, M3, M2, M1
        const std::string t_ = "l" + std::to_string(l) + ".";
        for (int k = 0; k < 3; ++k)
          launch_colsum(s, b.gMi + (int64_t)c0_[k] * F, rc_[k], b.Pn[l] + (int64_t)c0_[k] * F, rc_[k], nullptr, nullptr, N * nc_[k], F,
                        tc->self_gw + (int64_t)k * F, false, tc->part);
        launch_train_gw(g, P, F, b.gMi, b.Pn[l], tc->pre3[l], b.C, tc->self_gw, tc->g3, s);
        launch_tn_gemm(s, tc->g3, r3F, tc->he2[l], r2F, nullptr, nullptr, P1, 3 * F, 2 * F, tc->at(t_ + "M2"), false, tc->part);
        launch_colsum(s, tc->g3, r3F, nullptr, r3F, nullptr, nullptr, P1, 3 * F, tc->at(t_ + "b2"), false, tc->part);
        EDGE(1);
        // g_he2 = g_pre3 M3: the [N][K] operand is M3^T [2F][3F], kept with its split image (tmdnet_finalize_params)
        gemm(s, tc->g3, 3 * F, q_.M3T, 3 * F, nullptr, tc->g2, 2 * F, P1, 2 * F, 3 * F, GEMM_MUL_DSILU_AUX, nullptr, 0, tc->pre2[l], 2 * F);
        launch_tn_gemm(s, tc->g2, r2F, tc->he1[l], rF, nullptr, nullptr, P1, 2 * F, F, tc->at(t_ + "M1"), false, tc->part);
        launch_colsum(s, tc->g2, r2F, nullptr, r2F, nullptr, nullptr, P1, 2 * F, tc->at(t_ + "b1"), false, tc->part);
        gemm(s, tc->g2, 2 * F, q_.M2T, 2 * F, nullptr, tc->g1, F, P1, F, 2 * F, GEMM_MUL_DSILU_AUX, nullptr, 0, tc->pre1[l], F);
        launch_tn_gemm(s, tc->g1, rF, b.phi, rK, nullptr, nullptr, P1, F, K, tc->at(t_ + "M0"), false, tc->part);
        launch_colsum(s, tc->g1, rF, nullptr, rF, nullptr, nullptr, P1, F, tc->at(t_ + "b0"), false, tc->part);
        NODE();
      }
      if (merged_gd) {
        KR(CAT_MESSAGE, 2 * wB + idxB + nodeB * (2.0 + 2.0 * Na / Nd) + 8 * (Pd + 1) * gd_nw,  // w, dw, gMi, Pn, gPn (read + write: swept rows), g_d slots
           launch_message_adjoint_gd(g, N, F, b.w[l], b.dw[l], b.gMi, b.Pn[l], b.gPn, b.gd_slots + (int64_t)l * gd_nw * gd_stride,
                                     gd_stride, s, recompute ? &rts[l] : nullptr, a0, rng ? a0 + Na : -1, !m->graph_no_ghost_pairs));
      } else {
        KR(CAT_MESSAGE, wB + idxB + 3 * nodeB, launch_message_adjoint(g, N, F, b.w[l], b.gMi, b.gPn, s));
        if (!tc) KR(CAT_PAIR, Pd * (12 * Fd + 12) + 2 * nodeB, launch_pair_gd(g, P, F, b.gMi, b.Pn[l], b.dw[l], b.gd, s));
      }
      if (t9r) {
        // transposed linear + normalisation adjoint (incoming: the residual stream's G) in the epilogue; layer 0 goes on through
        // the embedding's gate adjoint there
        Tl9Args ta{};
        ta.A = b.gPn + o9F; ta.e0 = b.X[l] + o9F; ta.e1 = b.G + o9F; ta.N = Na; ta.F = F;
        if (l > 0) {
          ta.C = b.G + o9F;
          tlin9(s, TL9_PRO_PLAIN, TL9_EPI_NORMBWD, q_.VT, ta, 4.0, "normbwd");
        } else {
          ta.C = b.gUX + o9F; ta.o1 = b.g_a2 + o3F; ta.e2 = b.UX + o9F; ta.e3 = b.gates + o3F; ta.e4 = b.a2 + o3F;
          tlin9(s, TL9_PRO_PLAIN, TL9_EPI_NORMBWD_GATE, q_.VT, ta, 5.0 + 1.0, "normbwd+gate");
        }
        continue;
      }
      if (tc) tensor_linear_grad(b.gPn, tc->Xh[l], "l" + std::to_string(l) + ".V" /* 0..2 */ + std::string("a"));
      tensor_linear(s, b.gPn, q_.VT, b.gXl, N, F);
      if (l > 0)
        KR(CAT_ELEMENTWISE, 6 * nodeB, launch_norm_bwd_update_bwd(b.X[l], b.gXl, N, F, b.G, b.D[l - 1], q, batch_k, b.gD, s));
      else
        KR(CAT_ELEMENTWISE, 5 * nodeB + Nd * 3 * Fd * 12,
           launch_norm_bwd_gate_bwd(b.X[0], b.gXl, N, F, b.G, b.UX, b.gates, b.a2, b.gUX, b.g_a2, s));
    }
    if (L == 0)
      KR(CAT_ELEMENTWISE, 3 * nodeB + Nd * 3 * Fd * 12,
         launch_embed_gate_bwd(b.G + o9F, b.UX + o9F, b.gates + o3F, b.a2 + o3F, Na, F, b.gUX + o9F, b.g_a2 + o3F, s));
    NODE();
    gemm(s, b.g_a2 + o3F, 3 * F, W.L2T, 3 * F, nullptr, b.g_a1 + o2F, 2 * F, Na, 2 * F, 3 * F, GEMM_MUL_DSILU_AUX, nullptr, 0, b.a1 + o2F, 2 * F);
    gemm(s, b.g_a1 + o2F, 2 * F, W.L1T, 2 * F, nullptr, b.g_ln0 + oF, F, Na, F, 2 * F);
    KR(CAT_ELEMENTWISE, Nd * Fd * 12, launch_layernorm_bwd(b.g_ln0 + oF, b.xh0 + oF, b.rstd0 + o1_, W.ln0_w, Na, F, b.g_s0n + oF, s));
    if (!tc && tlin9_ok(Na, F, rng) && tlin9_images(W.UeT)) {
      Tl9Args ta{};
      ta.A = b.gUX + o9F; ta.e0 = b.u0 + o9F; ta.e1 = b.g_s0n + oF; ta.o1 = b.gA + 10 * oF; ta.N = Na; ta.F = F;
      tlin9(s, TL9_PRO_PLAIN, TL9_EPI_EMBBWD, W.UeT, ta, 2.0 + 11.0 / 9.0, "embbwd");
    } else {
      tensor_linear(s, b.gUX, W.UeT, b.g_u0l, N, F);
      KR(CAT_ELEMENTWISE, 2 * nodeB + Nd * 11 * Fd * 4, launch_embed_bwd_atom(b.g_u0l, b.u0, b.g_s0n, N, F, b.gA, s));
    }
    HALO_TRY(200, b.gA, 10 * F);
    if (tc) {
      // embedding: tensor linears, gate MLP, init_norm, then the edge weights W_k = C (U[z_i] + V[z_j]) (Wdp phi + bdp)_k
      tensor_linear_grad(b.gUX, b.u0, "Ue");
      launch_tn_gemm(s, b.g_a2, r3F, b.h1, r2F, nullptr, nullptr, N, 3 * F, 2 * F, tc->at("L2"), false, tc->part);
      launch_colsum(s, b.g_a2, r3F, nullptr, r3F, nullptr, nullptr, N, 3 * F, tc->at("bL2"), false, tc->part);
      launch_tn_gemm(s, b.g_a1, r2F, b.ln0, rF, nullptr, nullptr, N, 2 * F, F, tc->at("L1"), false, tc->part);
      launch_colsum(s, b.g_a1, r2F, nullptr, r2F, nullptr, nullptr, N, 2 * F, tc->at("bL1"), false, tc->part);
      launch_colsum(s, b.g_ln0, rF, b.xh0, rF, nullptr, nullptr, N, F, tc->at("ln0_w"), false, tc->part);
      launch_colsum(s, b.g_ln0, rF, nullptr, rF, nullptr, nullptr, N, F, tc->at("ln0_b"), false, tc->part);
      const int64_t dir = (int64_t)P1 * 3 * F;
      launch_train_embed(g, N, F, z, W.Utab, W.Vtab, b.Q, b.C, b.gA, tc->gq, dir, tc->selfq, tc->gZu, tc->gZv, s);
      float* dWdp = tc->at("Wdp");
      float* dbdp = tc->at("bdp");
      launch_tn_gemm(s, tc->gq, r3F, b.phi, rK, nullptr, nullptr, P, 3 * F, K, dWdp, false, tc->part);
      launch_tn_gemm(s, tc->gq + dir, r3F, b.phi, rK, nullptr, nullptr, P, 3 * F, K, dWdp, true, tc->part);
      launch_tn_gemm(s, tc->selfq, rF, b.phi + (int64_t)P * K, rows_plain(0), nullptr, nullptr, N, F, K, dWdp, true, tc->part);  // self pair: I block
      launch_colsum(s, tc->gq, r3F, nullptr, r3F, nullptr, nullptr, P, 3 * F, dbdp, false, tc->part);
      launch_colsum(s, tc->gq + dir, r3F, nullptr, r3F, nullptr, nullptr, P, 3 * F, dbdp, true, tc->part);
      launch_colsum(s, tc->selfq, rF, nullptr, rF, nullptr, nullptr, N, F, dbdp, true, tc->part);
      launch_onehot(z, N, Z, tc->onehot, s);
      launch_tn_gemm(s, tc->onehot, rows_plain(Z), tc->gZu, rF, nullptr, nullptr, N, Z, F, tc->at("Utab"), false, tc->part);
      launch_tn_gemm(s, tc->onehot, rows_plain(Z), tc->gZv, rF, nullptr, nullptr, N, Z, F, tc->at("Vtab"), false, tc->part);
    } else if (ntp) {
      KR(CAT_PAIR, Nd * 10 * Fd * 4 + momB, launch_embed_gm(g, N, F, K, ntp, z, W.Utab, W.Vtab, m->rb_rev, W.bdp, b.gA, b.gmom, s));
      KR(CAT_PAIR, Pd * 40 + momB + Pd * 8 * L * gd_nw,
         launch_embed_pair_rb(g, P, N, rbp, ntp, b.ps, b.gmom, b.gd, b.g_rhat, s, merged_gd ? b.g_delta : nullptr, b.gd_slots, L * gd_nw, gd_stride));
    } else
      KR(CAT_PAIR, Pd * (24 * Fd + 24) + Nd * 10 * Fd * 4,
         launch_embed_pair_gd(g, P, F, z, W.Utab, W.Vtab, b.Q, b.dQ, b.C, b.dC, b.gA, b.gd, b.g_rhat, s, merged_gd ? b.g_delta : nullptr,
                              b.gd_slots, L * gd_nw, gd_stride));
    if (!merged_gd && !tc) KR(CAT_ELEMENTWISE, Pd * 40, launch_geom_gd(g, P, b.gd, b.g_rhat, b.g_delta, s, nullptr, 0, gd_stride));
    if (!tc) KR(CAT_ELEMENTWISE, E_ * 8 + Nd * 12, launch_force_gather(g, N, b.g_delta, perm, forces, s));
  }
  NODE();
  HIP_TRY(m, hipGetLastError());
  m->last = b;
  m->lastN = N;
  m->lastP = P;
  m->has_last = true;
  return TMDNET_OK;
}
#undef HALO_TRY


// ------------------------------------------------------------------------------------ parameter gradients (TensorNet + Scalar)
namespace {
const std::vector<std::pair<std::string, int64_t>>& train_layout(tmdnet_model* m) {
  if (!m->train_entries.empty()) return m->train_entries;
  if (m->et) return m->train_entries = et_train_layout(m);
  if (m->tn2) return m->train_entries = tn2_train_layout(m);
  const int64_t F = m->hp.hidden_channels, K = m->hp.num_rbf, L = m->hp.num_layers, H = m->hp.head_hidden, Z = m->hp.max_z;
  auto& e = m->train_entries;
  e = {{"Wdp", 3 * F * K}, {"bdp", 3 * F}, {"Utab", Z * F}, {"Vtab", Z * F}, {"Ue0", F * F}, {"Ue1", F * F}, {"Ue2", F * F},
       {"L1", 2 * F * F}, {"bL1", 2 * F}, {"L2", 6 * F * F}, {"bL2", 3 * F}, {"ln0_w", F}, {"ln0_b", F}};
  for (int l = 0; l < L; ++l) {
    const std::string t = "l" + std::to_string(l) + ".";
    e.push_back({t + "M0", F * K});
    e.push_back({t + "b0", F});
    e.push_back({t + "M1", 2 * F * F});
    e.push_back({t + "b1", 2 * F});
    e.push_back({t + "M2", 6 * F * F});
    e.push_back({t + "b2", 3 * F});
    for (const char* ab : {"Va", "Vb"})
      for (int k = 0; k < 3; ++k) e.push_back({t + ab + std::to_string(k), F * F});
  }
  for (auto kv : std::vector<std::pair<std::string, int64_t>>{{"lnr_w", 3 * F}, {"lnr_b", 3 * F}, {"Lin", 3 * F * F}, {"bLin", F}, {"O1", H * F},
                                                               {"bO1", H}, {"O2", H}, {"bO2", 1}})
    e.push_back(kv);
  return e;
}
int64_t train_grad_floats(tmdnet_model* m) {
  int64_t n = 0;
  for (const auto& kv : train_layout(m)) n += (kv.second + 63) & ~int64_t(63);
  return n;
}
// extra activations + scratch of the parameter-gradient pass
void carve_train(void* ws, tmdnet_model* m, int64_t N, int64_t P, TrainCtx* tc, size_t* total) {
  const tmdnet_hparams& hp = m->hp;
  const int64_t F = hp.hidden_channels, K = hp.num_rbf, L = hp.num_layers, H = hp.head_hidden, Z = hp.max_z, P1 = P + 1, N9 = N * 9 * F;
  Carver c(ws);
  TrainCtx t;
  for (int l = 0; l < L; ++l) {
    t.pre1.push_back(c.take<float>(P1 * F));
    t.he1.push_back(c.take<float>(P1 * F));
    t.pre2.push_back(c.take<float>(P1 * 2 * F));
    t.he2.push_back(c.take<float>(P1 * 2 * F));
    t.pre3.push_back(c.take<float>(P1 * 3 * F));
    t.Ch.push_back(c.take<float>(N9));
    t.Xh.push_back(c.take<float>(N9));
  }
  t.g3 = c.take<float>(P1 * 3 * F);
  t.g2 = c.take<float>(P1 * 2 * F);
  t.g1 = c.take<float>(P1 * F);
  t.self_gw = c.take<float>(3 * F);
  t.gq = c.take<float>(2 * P1 * 3 * F);
  t.selfq = c.take<float>(N * F);
  t.gZu = c.take<float>(N * F);
  t.gZv = c.take<float>(N * F);
  t.onehot = c.take<float>(N * Z);
  t.head = c.take<float>(N * (H + 1));
  t.wT = c.take<float>(6 * F * F);
  t.forces = c.take<float>(N * 3);
  const int64_t big = std::max<int64_t>({6 * F * F, 3 * F * K, Z * F, H * F, 3 * F * F});
  t.part = c.take<float>((int64_t)train_part_floats((int)std::max<int64_t>(P1, 5 * N), big));
  (void)K;
  if (tc) {
    const float* ge = tc->gE;
    float* gr = tc->grads;
    *tc = t;
    tc->gE = ge;
    tc->grads = gr;
  }
  if (total) *total = c.off;
}
}  // namespace

extern "C++" const std::vector<std::pair<std::string, int64_t>>& param_grad_layout(tmdnet_model* m) { return train_layout(m); }

int tmdnet_param_grad_count(tmdnet_model* m) {
  if (!m) return 0;
  return (int)train_layout(m).size();
}
const char* tmdnet_param_grad_entry(tmdnet_model* m, int idx, int64_t* offset, int64_t* numel) {
  if (!m) return nullptr;
  const auto& e = train_layout(m);
  if (idx < 0 || idx >= (int)e.size()) return nullptr;
  int64_t off = 0;
  for (int i = 0; i < idx; ++i) off += (e[i].second + 63) & ~int64_t(63);
  if (offset) *offset = off;
  if (numel) *numel = e[idx].second;
  return e[idx].first.c_str();
}
int tmdnet_train_workspace_bytes(tmdnet_model* m, int64_t n_atoms, int64_t n_mol, int64_t n_pairs, size_t* fwd_bytes, size_t* train_bytes,
                                 int64_t* grad_floats) {
  if (!m || n_atoms < 0 || n_mol < 0 || n_pairs < 0) return TMDNET_ERR_INVALID;
  if (m->tn2) {  // per-edge arrays: every pair in both directions + one self edge per atom
    const int64_t n_edges = std::max<int64_t>(m->lastE, 2 * n_pairs + n_atoms);
    if (fwd_bytes) tn2_forward_workspace_bytes(m, n_atoms, n_mol, n_pairs, n_edges, 1, fwd_bytes);  // covers the per-pair embedding too
    if (train_bytes) tn2_carve_train(nullptr, m, n_atoms, n_pairs, n_edges, nullptr, train_bytes);
    if (grad_floats) *grad_floats = train_grad_floats(m);
    return TMDNET_OK;
  }
  if (m->et) {
    if (fwd_bytes) et_forward_workspace_bytes(m, n_atoms, n_mol, n_pairs, 1, fwd_bytes);
    if (train_bytes) et_carve_train(nullptr, m, n_atoms, n_pairs, nullptr, train_bytes);
    if (grad_floats) *grad_floats = train_grad_floats(m);
    return TMDNET_OK;
  }
  if (fwd_bytes) carve_fwd(nullptr, m->hp, n_atoms, n_mol, n_pairs, true, fwd_bytes, 0);
  if (train_bytes) carve_train(nullptr, m, n_atoms, n_pairs, nullptr, train_bytes);
  if (grad_floats) *grad_floats = train_grad_floats(m);
  return TMDNET_OK;
}
int tmdnet_energy_param_grads(tmdnet_model* m, void* stream, void* graph_ws, void* ws, size_t ws_bytes, void* train_ws, size_t train_bytes,
                              int64_t n_atoms, int64_t n_mol, int64_t n_pairs, const int64_t* z, const int64_t* batch, const float* q,
                              const float* grad_energy, float* energy, float* grads) {
  // grad_energy == NULL: forward half only (energies out, every activation kept in ws / train_ws); energy == NULL: reverse half
  // only, on the workspaces a forward-half call with the same arguments left behind; both given: one pass
  if (!m || !graph_ws || !ws || !train_ws || (!grad_energy && !energy) || (grad_energy && !grads)) return TMDNET_ERR_INVALID;
  recall_graph(m, graph_ws);
  if ((m->et || m->tn2) && (!grad_energy || !energy)) return fail(m, TMDNET_ERR_INVALID, "the two-call form is built for TensorNet only");
  if (n_pairs < 0) return fail(m, TMDNET_ERR_INVALID, "parameter gradients need the exact pair count (dynamic shapes)");
  if (m->graph_is_cell) return fail(m, TMDNET_ERR_STATE, "parameter gradients: build the graph without the cell list");
  TrainCtx tc;
  tc.gE = grad_energy;
  tc.grads = grads;
  size_t need = 0;
  if (m->et) et_carve_train(train_ws, m, n_atoms, n_pairs, &tc, &need);
  else if (m->tn2) tn2_carve_train(train_ws, m, n_atoms, n_pairs, std::max<int64_t>(m->lastE, 2 * n_pairs + n_atoms), &tc, &need);
  else carve_train(train_ws, m, n_atoms, n_pairs, &tc, &need);
  if (need > train_bytes) return fail(m, TMDNET_ERR_WORKSPACE, "training workspace too small: need " + std::to_string(need));
  int64_t off = 0;
  for (const auto& kv : train_layout(m)) {
    tc.off[kv.first] = off;
    off += (kv.second + 63) & ~int64_t(63);
  }
  tc.phase = !grad_energy ? 1 : (!energy ? 2 : 0);
  if (grads) launch_fill(grads, 0.f, off, reinterpret_cast<hipStream_t>(stream));
  m->train = &tc;
  float* e_out = energy ? energy : tc.forces;  // reverse half: the energies are not recomputed (scratch pointer, never written)
  const int rc = tmdnet_energy_forces(m, stream, graph_ws, ws, ws_bytes, n_atoms, n_mol, n_pairs, z, batch, q, 1, e_out, tc.forces);
  m->train = nullptr;
  return rc;
}

// ------------------------------------------------------------------------------------ neighbour operator
int tmdnet_neighbor_workspace_bytes(int64_t n_atoms, int64_t n_mol, int64_t max_num_pairs, size_t* bytes) {
  if (!bytes || n_atoms < 0 || n_mol < 0 || max_num_pairs < 0) return TMDNET_ERR_INVALID;
  // directed capacity: both directions + self loops of the requested undirected/directed list
  const int64_t ecap = 2 * max_num_pairs + n_atoms + 2;
  carve_graph(nullptr, n_atoms, n_mol, ecap, bytes);
  return TMDNET_OK;
}

int tmdnet_neighbor_pairs(void* stream, void* ws, size_t ws_bytes, int64_t n_atoms, int64_t n_mol, const float* pos,
                          const int64_t* batch, const float* box, int32_t box_mode, float cutoff_lower, float cutoff_upper,
                          int64_t max_num_pairs, int32_t loop, int32_t include_transpose, int32_t strategy, int64_t* neighbors,
                          float* deltas, float* distances, int32_t* num_pairs) {
  if (!ws || !neighbors || !deltas || !distances || !num_pairs) return TMDNET_ERR_INVALID;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int64_t ecap = 2 * max_num_pairs + n_atoms + 2;
  size_t need = 0;
  Graph g = carve_graph(ws, n_atoms, n_mol, ecap, &need);
  if (need > ws_bytes) return TMDNET_ERR_WORKSPACE;
  // internal CSR always carries self loops + both directions; the export selects what the caller asked for
  // strategy 1 = cell list (one box or none for all molecules; per-molecule boxes have no common grid -> brute force)
  const bool cell = strategy == 1 && box_mode != 2 && n_atoms > 0;
  if (cell) {
    g.use_cell = n_mol > 1 ? 2 : 1;
    g.ncx = g.ncy = g.ncz = 0;
    launch_fill(reinterpret_cast<float*>(g.counts), 0.f, 8, s);
    launch_cell_phase1(g, pos, batch, box_mode == 1 ? box : nullptr, (int)n_atoms, cutoff_lower, cutoff_upper, true, s);
    launch_scan_counts(g, (int)n_atoms, s);
    launch_cell_phase2(g, (int)n_atoms, cutoff_lower, cutoff_upper, true, s);
    launch_nbr_link_wave(g, (int)n_atoms, s);
  } else {
    launch_graph_build_phase1(g, pos, batch, box, box_mode, (int)n_atoms, (int)n_mol, cutoff_lower, cutoff_upper, true, s);
    launch_graph_build_phase2(g, pos, batch, box, box_mode, (int)n_atoms, (int)n_mol, cutoff_lower, cutoff_upper, true, s);
  }
  launch_export_pairs(g, (int)n_atoms, include_transpose != 0, loop != 0, max_num_pairs, cell ? g.perm : nullptr, neighbors, deltas,
                      distances, num_pairs, s);
  return hipGetLastError() == hipSuccess ? TMDNET_OK : TMDNET_ERR_HIP;
}

int tmdnet_neighbor_grad(void* stream, const int64_t* neighbors, const float* deltas, const float* distances,
                         const float* grad_deltas, const float* grad_distances, int64_t num_entries, int64_t n_atoms,
                         float* grad_positions) {
  if (!neighbors || !deltas || !distances || !grad_positions || num_entries < 0 || n_atoms < 0) return TMDNET_ERR_INVALID;
  launch_neighbor_grad(neighbors, deltas, distances, grad_deltas, grad_distances, num_entries, (int)n_atoms, grad_positions,
                       reinterpret_cast<hipStream_t>(stream));
  return hipGetLastError() == hipSuccess ? TMDNET_OK : TMDNET_ERR_HIP;
}

// ------------------------------------------------------------------------------------ profiling
int tmdnet_profile_begin(tmdnet_model* m, uint32_t category_mask) {
  if (!m) return TMDNET_ERR_INVALID;
  m->prof.on = true;
  m->prof.mask = category_mask;
  m->prof.recs.clear();
  m->prof.used = 0;
  return TMDNET_OK;
}

int tmdnet_profile_end(tmdnet_model* m, void* stream, double* ms, double* flops, double* bytes, int64_t* launches) {
  if (!m || !ms || !flops || !bytes || !launches) return TMDNET_ERR_INVALID;
  HIP_TRY(m, hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream)));
  for (int c = 0; c < CAT_COUNT; ++c) ms[c] = flops[c] = bytes[c] = 0.0, launches[c] = 0;
  for (const ProfRec& r : m->prof.recs) {
    float t = 0.f;
    HIP_TRY(m, hipEventElapsedTime(&t, r.a, r.b));
    ms[r.cat] += t;
    flops[r.cat] += r.flops;
    bytes[r.cat] += r.bytes;
    launches[r.cat] += 1;
  }
  m->prof.on = false;
  m->prof.recs.clear();
  m->prof.used = 0;
  return TMDNET_OK;
}

int tmdnet_profile_end_records(tmdnet_model* m, void* stream, int64_t cap, int32_t* cat, double* ms, double* flops, double* bytes,
                               char* labels, int64_t* n_out) {
  if (!m || !cat || !ms || !flops || !bytes || !n_out || cap < 0) return TMDNET_ERR_INVALID;
  HIP_TRY(m, hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream)));
  int64_t n = 0;
  for (const ProfRec& r : m->prof.recs) {
    if (n >= cap) break;
    float t = 0.f;
    HIP_TRY(m, hipEventElapsedTime(&t, r.a, r.b));
    cat[n] = r.cat;
    ms[n] = t;
    flops[n] = r.flops;
    bytes[n] = r.bytes;
    if (labels) std::memcpy(labels + 64 * n, r.label, 64);
    ++n;
  }
  *n_out = (int64_t)m->prof.recs.size();
  m->prof.on = false;
  m->prof.recs.clear();
  m->prof.used = 0;
  return TMDNET_OK;
}

int tmdnet_profile_num_categories(void) { return CAT_COUNT; }
const char* tmdnet_profile_category_name(int idx) { return idx >= 0 && idx < CAT_COUNT ? kCatNames[idx] : nullptr; }

// ------------------------------------------------------------------------------------ diagnostics
int tmdnet_debug_tensor(tmdnet_model* m, void* stream, const char* name, float* out, int64_t numel) {
  if (m && m->et) return et_debug_tensor(m, reinterpret_cast<hipStream_t>(stream), name, out, numel);
  if (!m || !name || !out || !m->has_last) return TMDNET_ERR_STATE;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int64_t F = m->hp.hidden_channels, K = m->hp.num_rbf, N = m->lastN, P1 = m->lastP + 1;
  const FwdBuffers& b = m->last;
  const float* src = nullptr;
  int64_t n = 0;
  std::string nm(name);
  if (nm == "X_embed") { src = b.X[0]; n = N * 9 * F; }
  else if (nm.rfind("X_layer", 0) == 0) {
    int l = std::atoi(nm.c_str() + 7);
    if (l < 0 || l >= m->hp.num_layers) return TMDNET_ERR_INVALID;
    src = b.X[l + 1]; n = N * 9 * F;
  }
  else if (nm.rfind("Pn", 0) == 0 || nm.rfind("Mi", 0) == 0 || (nm.rfind("D", 0) == 0 && nm.size() == 2)) {  // per-layer tensors of the last step
    const int l = std::atoi(nm.c_str() + (nm[0] == 'D' ? 1 : 2));
    if (l < 0 || l >= m->hp.num_layers) return TMDNET_ERR_INVALID;
    src = nm[0] == 'P' ? b.Pn[l] : (nm[0] == 'M' ? b.Mi[l] : b.D[l]);
    n = N * 9 * F;
  }
  else if (nm == "x") { src = b.x; n = N * F; }
  else if (nm == "phi") { src = b.phi; n = P1 * K; }
  else if (nm == "Q") { src = b.Q; n = P1 * 3 * F; }
  else if (nm == "u0") { src = b.u0; n = N * 9 * F; }
  else if (nm == "G_embed") { src = b.G; n = N * 9 * F; }
  else if (nm == "charges" && m->tn2) { src = m->tn2_last_chg; n = N * (int64_t)(m->hp.num_layers + 1) * m->tn2->hp.q_dim; }
  else return fail(m, TMDNET_ERR_INVALID, "unknown debug tensor");
  if (!src || numel != n) return fail(m, TMDNET_ERR_INVALID, "debug tensor size mismatch: expected " + std::to_string(n));
  HIP_TRY(m, hipMemcpyAsync(out, src, sizeof(float) * n, hipMemcpyDeviceToDevice, s));
  return TMDNET_OK;
}

int tmdnet_debug_gemm_dual(void* stream, const float* A, const float* A2, const float* W, const float* bias, float* C, float* C2,
                           int64_t M, int64_t N, int64_t K, int32_t kind, const float* rs, const float* rs2, const uint16_t* Wsb) {
  g_mdev = nullptr;
  gemm_dual(reinterpret_cast<hipStream_t>(stream), kind, A, A2, K, W, bias, C, C2, N, (int)M, (int)N, (int)K, rs, rs2, Wsb);
  return hipGetLastError() == hipSuccess ? TMDNET_OK : TMDNET_ERR_HIP;
}

int64_t tmdnet_debug_split_weight(const float* W_host, int64_t N, int64_t K, uint16_t* out_host) {
  if (out_host) split_weight_tiles(W_host, N, K, out_host);
  return (int64_t)split_weight_elems(N, K);
}

int tmdnet_debug_gemm(void* stream, const float* A, const float* W, const float* bias, float* C, int64_t M, int64_t N,
                      int64_t K, int32_t silu_, const uint16_t* Wsb) {
  g_mdev = nullptr;
  g_wsb_debug = Wsb;
  gemm(reinterpret_cast<hipStream_t>(stream), A, K, W, K, bias, C, N, (int)M, (int)N, (int)K, silu_ ? GEMM_ACT_SILU : 0);
  g_wsb_debug = nullptr;
  return hipGetLastError() == hipSuccess ? TMDNET_OK : TMDNET_ERR_HIP;
}

}  // extern "C"
