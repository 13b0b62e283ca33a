// Equivariant Transformer + EquivariantScalar head behind the same C ABI as TensorNet (include/tmdnet_amd.h,
// tmdnet_create_et): parameter packing, workspace carving and the kernel schedule of forward + hand-written reverse.
//
// Schedule = reference call stack TorchMD_Net.forward -> TorchMD_ET.forward -> NeighborEmbedding ->
// EquivariantMultiHeadAttention x L -> out_norm -> EquivariantScalar (2 GatedEquivariantBlocks) -> reduce
// (torchmdnet/models/model.py:530-631, torchmd_et.py:188-426, models/utils.py:45-117, 583-655,
// output_modules.py:43-73, 120-163), followed by the explicit adjoints specified in oracle/et_adjoint.py.
// Every dense contraction goes through gemm()/gemm_dual() (bf16 matrix pipe, exact 3-way split); the per-pair
// quantities (phi, C, dk, dv, the neighbour-embedding filter) are evaluated once per undirected pair together with
// their d/dd tangents, exactly as in the TensorNet path.
#include <cstdio>
#include <cstdlib>

#include "tn_et.h"
#include "tn_model.h"

namespace {

struct EtLayerP {
  const float *ln_w, *ln_b;
  const float *Wqkv, *bqkv;      // [5F][F]: q | k | vx | v1 | v2 (value rows permuted to thirds)
  const float *WqkvT;            // [F][5F]
  const float *Wvp, *WvpT;       // [3F][F] (no bias), [F][3F]
  const float *Wo, *bo, *WoT;    // [3F][F], [F][3F]
  const float *Wdkv, *bdkv;      // [Wd][K]: dk | dvx | dv1 | dv2 (the parts the model has)
  const uint16_t* Wdkv_sb;
};

struct EtParams {
  const float *means, *betas, *emb;
  const float *embN, *Wn, *bn, *Wc, *bc, *WcT;  // neighbour embedding (Wc [F][2F])
  const uint16_t* Wn_sb;
  std::vector<EtLayerP> layer;
  const float *lno_w, *lno_b;
  const float *W1u, *W1uT;            // [F + F/2][F] = vec1_proj | vec2_proj of block 0 ; transposed [F][F + F/2]
  const float *Wm1, *bm1, *Wm1T;      // [F][2F]
  const float *Wm2, *bm2, *Wm2T;      // [F][F]
  const float *W21, *W21T;            // [F/2][F/2]
  const float *Wn1, *bn1, *Wn1T;      // [F/2][F]
  const float *Wn2, *bn2;             // row 0 of [2][F/2], bias[0]
  const float* atomref;
  float mean, std;
};

struct EtBuffers {
  float *phi, *dphi, *C, *dC, *Wn, *dWn, *xcat;
  unsigned* skeys_s;  // radial tables: pairs in grid-interval order (tn_edge_table.hip)
  int *svals_s, *shist;
  std::vector<float*> x, vec;                               // L+1
  std::vector<float*> xt, xh, rstd, qkv, vp, o, vdot, dkv, tkv;  // per layer
  float *xagg, *vagg;
  float *xf, *xfh, *rstdf, *u12, *hcat, *pre1, *m1, *y, *hcat2, *vq, *w1, *pre2, *m2, *ea;
  // reverse
  float *g_pre2, *g_h2, *g_w1, *g_vq, *g_y, *g_m1, *g_h1, *g_u12, *g_xf, *g_x, *g_vec, *g_o, *g_vp, *g_xagg, *g_qkv, *g_xt, *g_ln,
      *g_xcat, *gd2, *gr2, *gd, *g_rhat, *g_delta;
  float* erec = nullptr;     // per-edge records of the tile sweeps
  int* tile_start = nullptr;  // [tiles + 1] first row of every tile of the tile sweeps
  int* tile_open = nullptr;  // device flag: some row has a neighbour outside its tile of 64 rows (tn_et_g16.hip)
};

}  // namespace

struct EtModel {
  tmdnet_et_hparams hp;
  EtParams P;
  EtBuffers last{};
  int64_t lastN = 0;
  bool has_last = false;
};

namespace {

int wd_of(const tmdnet_et_hparams& hp) {
  const int F = hp.hidden_channels;
  return ((hp.distance_influence & 1) ? F : 0) + ((hp.distance_influence & 2) ? 3 * F : 0);
}

void et_build_specs(tmdnet_model* m) {
  const tmdnet_et_hparams& hp = m->et->hp;
  const int F = hp.hidden_channels, K = hp.num_rbf, L = hp.num_layers, Z = hp.max_z, F2 = F / 2;
  auto& s = m->specs;
  const std::string R = "representation_model.";
  s.push_back({R + "embedding.weight", Z, F});
  s.push_back({R + "distance_expansion.means", K, 1});
  s.push_back({R + "distance_expansion.betas", K, 1});
  if (hp.neighbor_embedding) {
    const std::string Np = R + "neighbor_embedding.";
    s.push_back({Np + "embedding.weight", Z, F});
    s.push_back({Np + "distance_proj.weight", F, K});
    s.push_back({Np + "distance_proj.bias", F, 1});
    s.push_back({Np + "combine.weight", F, 2 * F});
    s.push_back({Np + "combine.bias", F, 1});
  }
  for (int l = 0; l < L; ++l) {
    const std::string Lp = R + "attention_layers." + std::to_string(l) + ".";
    s.push_back({Lp + "layernorm.weight", F, 1});
    s.push_back({Lp + "layernorm.bias", F, 1});
    s.push_back({Lp + "q_proj.weight", F, F});
    s.push_back({Lp + "q_proj.bias", F, 1});
    s.push_back({Lp + "k_proj.weight", F, F});
    s.push_back({Lp + "k_proj.bias", F, 1});
    s.push_back({Lp + "v_proj.weight", 3 * F, F});
    s.push_back({Lp + "v_proj.bias", 3 * F, 1});
    s.push_back({Lp + "o_proj.weight", 3 * F, F});
    s.push_back({Lp + "o_proj.bias", 3 * F, 1});
    s.push_back({Lp + "vec_proj.weight", 3 * F, F});
    if (hp.distance_influence & 1) {
      s.push_back({Lp + "dk_proj.weight", F, K});
      s.push_back({Lp + "dk_proj.bias", F, 1});
    }
    if (hp.distance_influence & 2) {
      s.push_back({Lp + "dv_proj.weight", 3 * F, K});
      s.push_back({Lp + "dv_proj.bias", 3 * F, 1});
    }
  }
  s.push_back({R + "out_norm.weight", F, 1});
  s.push_back({R + "out_norm.bias", F, 1});
  const std::string O0 = "output_model.output_network.0.", O1 = "output_model.output_network.1.";
  s.push_back({O0 + "vec1_proj.weight", F, F});
  s.push_back({O0 + "vec2_proj.weight", F2, F});
  s.push_back({O0 + "update_net.layers.0.weight", F, 2 * F});
  s.push_back({O0 + "update_net.layers.0.bias", F, 1});
  s.push_back({O0 + "update_net.layers.2.weight", F, F});
  s.push_back({O0 + "update_net.layers.2.bias", F, 1});
  s.push_back({O1 + "vec1_proj.weight", F2, F2});
  s.push_back({O1 + "vec2_proj.weight", 1, F2});
  s.push_back({O1 + "update_net.layers.0.weight", F2, F});
  s.push_back({O1 + "update_net.layers.0.bias", F2, 1});
  s.push_back({O1 + "update_net.layers.2.weight", 2, F2});
  s.push_back({O1 + "update_net.layers.2.bias", 2, 1});
  s.push_back({"mean", 1, 1});
  s.push_back({"std", 1, 1});
  if (hp.has_atomref) s.push_back({"atomref", Z, 1});
}

// rows of a [3F][cols] value-type matrix from the reference's per-head layout [H][3][hd] to thirds [3][F]
std::vector<float> thirds(const std::vector<float>& w, int F, int hd, int64_t cols) {
  std::vector<float> o(w.size());
  for (int t = 0; t < 3; ++t)
    for (int c = 0; c < F; ++c) {
      const int64_t src = (int64_t)(c / hd) * 3 * hd + t * hd + c % hd, dst = (int64_t)t * F + c;
      std::memcpy(o.data() + dst * cols, w.data() + src * cols, cols * sizeof(float));
    }
  return o;
}
std::vector<float> cat(std::initializer_list<const std::vector<float>*> parts) {
  std::vector<float> o;
  for (auto* p : parts) o.insert(o.end(), p->begin(), p->end());
  return o;
}

EtBuffers et_carve(void* ws, const tmdnet_et_hparams& hp, int64_t N, int64_t B, int64_t P, bool bwd, size_t* total) {
  Carver c(ws);
  EtBuffers b{};
  const int64_t F = hp.hidden_channels, K = hp.num_rbf, L = hp.num_layers, F2 = F / 2, P1 = P + 1, Wd = wd_of(hp);
  b.phi = c.take<float>(P1 * K);
  b.dphi = c.take<float>(P1 * K);
  b.C = c.take<float>(P1);
  b.dC = c.take<float>(P1);
  if (hp.neighbor_embedding) {
    b.Wn = c.take<float>(P1 * F);
    b.dWn = c.take<float>(P1 * F);
    b.xcat = c.take<float>(N * 2 * F);
  }
  for (int l = 0; l <= L; ++l) {
    b.x.push_back(c.take<float>(N * F));
    b.vec.push_back(c.take<float>(N * 3 * F));
  }
  for (int l = 0; l < L; ++l) {
    b.xt.push_back(c.take<float>(N * F));
    b.xh.push_back(c.take<float>(N * F));
    b.rstd.push_back(c.take<float>(N));
    b.qkv.push_back(c.take<float>(N * 5 * F));
    b.vp.push_back(c.take<float>(N * 9 * F));
    b.o.push_back(c.take<float>(N * 3 * F));
    b.vdot.push_back(c.take<float>(N * F));
    b.dkv.push_back(c.take<float>(P1 * Wd));
    b.tkv.push_back(c.take<float>(P1 * Wd));
  }
  b.xagg = c.take<float>(N * F);
  b.vagg = c.take<float>(N * 3 * F);
  b.xf = c.take<float>(N * F);
  b.xfh = c.take<float>(N * F);
  b.rstdf = c.take<float>(N);
  b.u12 = c.take<float>(N * 3 * (F + F2));
  b.hcat = c.take<float>(N * 2 * F);
  b.pre1 = c.take<float>(N * F);
  b.m1 = c.take<float>(N * F);
  b.y = c.take<float>(N * F);
  b.hcat2 = c.take<float>(N * F);
  b.vq = c.take<float>(N * 3 * F2);
  b.w1 = c.take<float>(N * 3 * F2);
  b.pre2 = c.take<float>(N * F2);
  b.m2 = c.take<float>(N * F2);
  b.ea = c.take<float>(N);
  b.skeys_s = c.take<unsigned>(P1);
  b.svals_s = c.take<int>(P1);
  b.shist = c.take<int>(65536 + 2);
  b.tile_open = c.take<int>(64);
  b.tile_start = c.take<int>(N + 8);
  b.erec = c.take<float>(8 * (2 * P1 + N));
  if (bwd) {
    b.g_pre2 = c.take<float>(N * F2);
    b.g_h2 = c.take<float>(N * F);
    b.g_w1 = c.take<float>(N * 3 * F2);
    b.g_vq = c.take<float>(N * 3 * F2);
    b.g_y = c.take<float>(N * F);
    b.g_m1 = c.take<float>(N * F);
    b.g_h1 = c.take<float>(N * 2 * F);
    b.g_u12 = c.take<float>(N * 3 * (F + F2));
    b.g_xf = c.take<float>(N * F);
    b.g_x = c.take<float>(N * F);
    b.g_vec = c.take<float>(N * 3 * F);
    b.g_o = c.take<float>(N * 3 * F);
    b.g_vp = c.take<float>(N * 9 * F);
    b.g_xagg = c.take<float>(N * F);
    b.g_qkv = c.take<float>(N * 5 * F);
    b.g_xt = c.take<float>(N * F);
    b.g_ln = c.take<float>(N * F);
    b.g_xcat = c.take<float>(N * 2 * F);
    b.gd2 = c.take<float>(2 * P1 * (et_sweep_waves((int)F) * L + 1));  // [layer][wave] slot arrays + one for the embedding
    b.gr2 = c.take<float>(6 * P1 * et_sweep_waves((int)F) * L);
    b.gd = c.take<float>(P1);
    b.g_rhat = c.take<float>(3 * P1);
    b.g_delta = c.take<float>(3 * P1);
  }
  (void)B;
  if (total) *total = c.off;
  return b;
}

}  // namespace

int et_create(tmdnet_model* m, const tmdnet_et_hparams* hp) {
  const int F = hp->hidden_channels, H = hp->num_heads;
  if (F % H) return TMDNET_ERR_INVALID;
  const int hd = F / H;
  if (hd > 64 || (hd & (hd - 1)) || (F & 1) || F > 1024) return TMDNET_ERR_INVALID;  // head reductions are wave shuffles
  m->et = new EtModel();
  m->et->hp = *hp;
  et_build_specs(m);
  return TMDNET_OK;
}

void et_destroy(tmdnet_model* m) {
  if (m && m->et) {
    delete m->et;
    m->et = nullptr;
  }
}

int et_finalize(tmdnet_model* m) {
  for (const auto& sp : m->specs)
    if (!m->host.count(sp.name)) return fail(m, TMDNET_ERR_STATE, "missing parameter: " + sp.name);
  const tmdnet_et_hparams& hp = m->et->hp;
  const int F = hp.hidden_channels, K = hp.num_rbf, L = hp.num_layers, F2 = F / 2, hd = F / hp.num_heads, Wd = wd_of(hp);
  const std::string R = "representation_model.", Np = R + "neighbor_embedding.";
  auto& h = m->host;
  Packer pk;
  std::map<std::string, size_t> off;
  struct Img { std::string key; int64_t n, k; };
  std::vector<Img> imgs;
  auto put = [&](const std::string& key, const std::vector<float>& v) { off[key] = pk.add(v); };
  auto putW = [&](const std::string& key, const std::vector<float>& v, int64_t n, int64_t k) {  // GEMM weight [n][k] + transpose
    off[key] = pk.add(v);
    off[key + "T"] = pk.add_T(v, n, k);
    imgs.push_back({key, n, k});
    imgs.push_back({key + "T", k, n});
  };
  put("means", h[R + "distance_expansion.means"]);
  put("betas", h[R + "distance_expansion.betas"]);
  put("emb", h[R + "embedding.weight"]);
  if (hp.neighbor_embedding) {
    put("embN", h[Np + "embedding.weight"]);
    putW("Wn", h[Np + "distance_proj.weight"], F, K);
    put("bn", h[Np + "distance_proj.bias"]);
    putW("Wc", h[Np + "combine.weight"], F, 2 * F);
    put("bc", h[Np + "combine.bias"]);
  }
  for (int l = 0; l < L; ++l) {
    const std::string Lp = R + "attention_layers." + std::to_string(l) + ".", t = "l" + std::to_string(l) + ".";
    put(t + "ln_w", h[Lp + "layernorm.weight"]);
    put(t + "ln_b", h[Lp + "layernorm.bias"]);
    const auto wv = thirds(h[Lp + "v_proj.weight"], F, hd, F), bv = thirds(h[Lp + "v_proj.bias"], F, hd, 1);
    putW(t + "Wqkv", cat({&h[Lp + "q_proj.weight"], &h[Lp + "k_proj.weight"], &wv}), 5 * F, F);
    put(t + "bqkv", cat({&h[Lp + "q_proj.bias"], &h[Lp + "k_proj.bias"], &bv}));
    putW(t + "Wvp", h[Lp + "vec_proj.weight"], 3 * F, F);
    putW(t + "Wo", h[Lp + "o_proj.weight"], 3 * F, F);
    put(t + "bo", h[Lp + "o_proj.bias"]);
    if (Wd > 0) {
      std::vector<float> w, b;
      if (hp.distance_influence & 1) {
        w = h[Lp + "dk_proj.weight"];
        b = h[Lp + "dk_proj.bias"];
      }
      if (hp.distance_influence & 2) {
        const auto wd = thirds(h[Lp + "dv_proj.weight"], F, hd, K), bd = thirds(h[Lp + "dv_proj.bias"], F, hd, 1);
        w.insert(w.end(), wd.begin(), wd.end());
        b.insert(b.end(), bd.begin(), bd.end());
      }
      putW(t + "Wdkv", w, Wd, K);
      put(t + "bdkv", b);
    }
  }
  put("lno_w", h[R + "out_norm.weight"]);
  put("lno_b", h[R + "out_norm.bias"]);
  const std::string O0 = "output_model.output_network.0.", O1 = "output_model.output_network.1.";
  putW("W1u", cat({&h[O0 + "vec1_proj.weight"], &h[O0 + "vec2_proj.weight"]}), F + F2, F);
  putW("Wm1", h[O0 + "update_net.layers.0.weight"], F, 2 * F);
  put("bm1", h[O0 + "update_net.layers.0.bias"]);
  putW("Wm2", h[O0 + "update_net.layers.2.weight"], F, F);
  put("bm2", h[O0 + "update_net.layers.2.bias"]);
  putW("W21", h[O1 + "vec1_proj.weight"], F2, F2);
  putW("Wn1", h[O1 + "update_net.layers.0.weight"], F2, F);
  put("bn1", h[O1 + "update_net.layers.0.bias"]);
  put("Wn2", h[O1 + "update_net.layers.2.weight"]);  // row 0 = the scalar output (row 1 gates a vector that is dropped)
  put("bn2", h[O1 + "update_net.layers.2.bias"]);
  if (hp.has_atomref) put("atomref", h["atomref"]);

  if (m->dev && m->dev_cap < pk.buf.size()) {
    HIP_TRY(m, hipFree(m->dev));
    m->dev = nullptr;
  }
  if (!m->dev) {
    HIP_TRY(m, hipMalloc(reinterpret_cast<void**>(&m->dev), pk.buf.size() * sizeof(float)));
    m->dev_cap = pk.buf.size();
  }
  HIP_TRY(m, hipMemcpy(m->dev, pk.buf.data(), pk.buf.size() * sizeof(float), hipMemcpyHostToDevice));
  {  // split-bf16 tile images of every GEMM weight
    std::vector<size_t> so;  // made on the device from the uploaded fp32 copies
    size_t sb_elems = 0;
    for (const auto& im : imgs) {
      so.push_back(sb_elems);
      sb_elems += split_weight_elems(im.n, im.k);
    }
    if (m->dev_sb && m->dev_sb_cap < sb_elems) {
      HIP_TRY(m, hipFree(m->dev_sb));
      m->dev_sb = nullptr;
    }
    if (!m->dev_sb) {
      HIP_TRY(m, hipMalloc(reinterpret_cast<void**>(&m->dev_sb), sb_elems * sizeof(uint16_t)));
      m->dev_sb_cap = sb_elems;
    }
    for (size_t i = 0; i < imgs.size(); ++i)
      launch_split_weight_tiles(m->dev + off.at(imgs[i].key), imgs[i].n, imgs[i].k, m->dev_sb + so[i], nullptr);
    m->sb_of.clear();
    for (size_t i = 0; i < imgs.size(); ++i) m->sb_of[m->dev + off.at(imgs[i].key)] = m->dev_sb + so[i];
  }
  auto D = [&](const std::string& key) -> const float* { return m->dev + off.at(key); };
  EtParams& P = m->et->P;
  P = EtParams{};
  P.means = D("means");
  P.betas = D("betas");
  P.emb = D("emb");
  if (hp.neighbor_embedding) {
    P.embN = D("embN");
    P.Wn = D("Wn");
    P.bn = D("bn");
    P.Wc = D("Wc");
    P.bc = D("bc");
    P.WcT = D("WcT");
    P.Wn_sb = m->sb_of.at(P.Wn);
  }
  P.layer.resize(L);
  for (int l = 0; l < L; ++l) {
    const std::string t = "l" + std::to_string(l) + ".";
    EtLayerP& q = P.layer[l];
    q.ln_w = D(t + "ln_w");
    q.ln_b = D(t + "ln_b");
    q.Wqkv = D(t + "Wqkv");
    q.bqkv = D(t + "bqkv");
    q.WqkvT = D(t + "WqkvT");
    q.Wvp = D(t + "Wvp");
    q.WvpT = D(t + "WvpT");
    q.Wo = D(t + "Wo");
    q.bo = D(t + "bo");
    q.WoT = D(t + "WoT");
    if (Wd > 0) {
      q.Wdkv = D(t + "Wdkv");
      q.bdkv = D(t + "bdkv");
      q.Wdkv_sb = m->sb_of.at(q.Wdkv);
    }
  }
  P.lno_w = D("lno_w");
  P.lno_b = D("lno_b");
  P.W1u = D("W1u");
  P.W1uT = D("W1uT");
  P.Wm1 = D("Wm1");
  P.bm1 = D("bm1");
  P.Wm1T = D("Wm1T");
  P.Wm2 = D("Wm2");
  P.bm2 = D("bm2");
  P.Wm2T = D("Wm2T");
  P.W21 = D("W21");
  P.W21T = D("W21T");
  P.Wn1 = D("Wn1");
  P.bn1 = D("bn1");
  P.Wn1T = D("Wn1T");
  P.Wn2 = D("Wn2");
  P.bn2 = D("bn2");
  P.atomref = hp.has_atomref ? D("atomref") : nullptr;
  P.mean = h["mean"][0];
  P.std = h["std"][0];
  HIP_TRY(m, hipStreamSynchronize(nullptr));  // the weight images are made by kernels on the NULL stream
  free_radial_tables(m->tabs);  // built by the first call that uses them (ensure_radial_tables -> et_build_tables)
  m->tabs_pending = true;
  m->finalized = true;
  return TMDNET_OK;
}

// radial tables (tn_edge_table.hip): every layer's distance filters silu(D phi + b) [Wd] and the neighbour-embedding filter
// (D phi + b) C(d) [F] are functions of the pair distance alone (reference torchmd_et.py:375-384, models/utils.py:99-104)
int et_build_tables(tmdnet_model* m) {
  const tmdnet_et_hparams& hp = m->et->hp;
  const EtParams& P = m->et->P;
  const int F = hp.hidden_channels, K = hp.num_rbf, L = hp.num_layers, Wd = wd_of(hp);
  std::vector<TableSpec> specs;
  if (Wd > 0)
    for (int l = 0; l < L; ++l) specs.push_back(TableSpec{{TableLayer{P.layer[l].Wdkv, P.layer[l].bdkv, Wd, K, 1}}});
  if (hp.neighbor_embedding) specs.push_back(TableSpec{{TableLayer{P.Wn, P.bn, F, K, 3}}});
  return build_radial_tables(m, m->tabs, specs, P.means, P.betas, K, hp.cutoff_lower, hp.cutoff_upper);
}

// parameter-gradient pass: gradient buffer layout and the extra buffers
std::vector<std::pair<std::string, int64_t>> et_train_layout(const tmdnet_model* m) {
  const tmdnet_et_hparams& hp = m->et->hp;
  const int64_t F = hp.hidden_channels, K = hp.num_rbf, L = hp.num_layers, Z = hp.max_z, F2 = F / 2, Wd = wd_of(hp), U = F + F2;
  std::vector<std::pair<std::string, int64_t>> e = {{"emb", Z * F}};
  if (hp.neighbor_embedding)
    for (auto kv : std::vector<std::pair<std::string, int64_t>>{{"embN", Z * F}, {"Wn", F * K}, {"bn", F}, {"Wc", 2 * F * F}, {"bc", F}})
      e.push_back(kv);
  for (int l = 0; l < L; ++l) {
    const std::string t = "l" + std::to_string(l) + ".";
    for (auto kv : std::vector<std::pair<std::string, int64_t>>{{"ln_w", F}, {"ln_b", F}, {"Wqkv", 5 * F * F}, {"bqkv", 5 * F}, {"Wvp", 3 * F * F},
                                                                 {"Wo", 3 * F * F}, {"bo", 3 * F}})
      e.push_back({t + kv.first, kv.second});
    if (Wd > 0) {
      e.push_back({t + "Wdkv", Wd * K});
      e.push_back({t + "bdkv", Wd});
    }
  }
  for (auto kv : std::vector<std::pair<std::string, int64_t>>{{"lno_w", F}, {"lno_b", F}, {"W1u", U * F}, {"Wm1", 2 * F * F}, {"bm1", F},
                                                               {"Wm2", F * F}, {"bm2", F}, {"W21", F2 * F2}, {"Wn1", F2 * F}, {"bn1", F2},
                                                               {"Wn2", F2}, {"bn2", 1}})
    e.push_back(kv);
  return e;
}
void et_carve_train(void* ws, const tmdnet_model* m, int64_t N, int64_t P, TrainCtx* tc, size_t* total) {
  const tmdnet_et_hparams& hp = m->et->hp;
  const int64_t F = hp.hidden_channels, K = hp.num_rbf, L = hp.num_layers, Z = hp.max_z, F2 = F / 2, P1 = P + 1;
  const int64_t Wd = std::max<int64_t>(wd_of(hp), F);
  Carver c(ws);
  TrainCtx t;
  for (int l = 0; l < L; ++l) {
    t.pre3.push_back(c.take<float>(P1 * Wd));  // pre-activations of the filter rows
    t.Ch.push_back(c.take<float>(N * F));      // xagg of the layer
  }
  t.g3 = c.take<float>(P1 * Wd);
  t.gq = c.take<float>(2 * P1 * Wd);
  t.selfq = c.take<float>(N * Wd);
  t.self_gw = c.take<float>(Wd);
  t.gZu = c.take<float>(N * F);
  t.onehot = c.take<float>(N * Z);
  t.head = c.take<float>(N * (F2 + 1));
  t.forces = c.take<float>(N * 3);
  const int64_t big = std::max<int64_t>({5 * F * F, Wd * K, Z * F, 2 * F * F});
  t.part = c.take<float>((int64_t)train_part_floats((int)std::max<int64_t>(P1, 3 * N), big));
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

int et_forward_workspace_bytes(const tmdnet_model* m, int64_t n_atoms, int64_t n_mol, int64_t n_pairs, int32_t want_forces,
                               size_t* bytes) {
  et_carve(nullptr, m->et->hp, n_atoms, n_mol, n_pairs, want_forces != 0, bytes);
  return TMDNET_OK;
}

int et_energy_forces(tmdnet_model* m, hipStream_t s, const Graph& g, void* ws, size_t ws_bytes, int64_t n_atoms, int64_t n_mol,
                     int64_t n_pairs, const int64_t* z, const int64_t* batch, int32_t want_forces, float* energy, float* forces) {
  const tmdnet_et_hparams& hp = m->et->hp;
  const int F = hp.hidden_channels, K = hp.num_rbf, L = hp.num_layers, F2 = F / 2, hd = F / hp.num_heads, Wd = wd_of(hp);
  const int N = (int)n_atoms, B = (int)n_mol;
  const int P = n_pairs >= 0 ? (int)n_pairs : (int)g.pcap, P1 = P + 1;
  size_t need = 0;
  EtBuffers b = et_carve(ws, hp, n_atoms, n_mol, P, want_forces != 0, &need);
  if (need > ws_bytes) return fail(m, TMDNET_ERR_WORKSPACE, "forward workspace too small: need " + std::to_string(need));
  const EtParams& W = m->et->P;
  const int* perm = m->graph_is_cell ? g.perm : nullptr;
  auto EDGE = [&](int add) { g_gemm_cat = CAT_GEMM_EDGE; g_mdev = n_pairs < 0 ? g.counts : nullptr; g_madd = add; };
  auto NODE = [&]() { g_gemm_cat = CAT_GEMM_NODE; g_mdev = nullptr; g_madd = 0; };
  const double Nd = N, Pd = P, Fd = F, Ed = (double)m->lastE;

  // ---------------- forward
  const int n_dkv = Wd > 0 ? L : 0;
  TrainCtx* const tc = m->train;  // parameter gradients: direct filters with their pre-activations kept, fp32 rows
  const bool use_tab = !tc && m->tabs.ok && (int64_t)P1 >= m->tab_min_pairs &&
                       (int)m->tabs.tab.size() == n_dkv + (hp.neighbor_embedding ? 1 : 0) && !m->tabs.tab.empty();
  // reduced-precision STORAGE of the per-pair filter rows (option "pair_rows_bf16", BASELINE configs[3] "bf16"): dkv / tkv are
  // written by the table interpolation as bf16 and widened to fp32 when the sweeps load them; every product and sum stays fp32.
  // Only with the tables (the value + tangent GEMMs write fp32 rows): otherwise the call silently keeps fp32 storage.
  const int pbf = (m->pair_bf16 && use_tab) ? 1 : 0;
  const double pB = pbf ? 2.0 : 4.0;  // bytes per stored pair-row element
  if (use_tab) {
    // all per-pair filters from the radial tables: one bucket sort of the pairs, one interpolation launch per row length
    std::vector<const float*> tabs;
    std::vector<float*> outs, douts;
    for (int l = 0; l < n_dkv; ++l) {
      tabs.push_back(m->tabs.tab[l]);
      outs.push_back(b.dkv[l]);
      douts.push_back(want_forces ? b.tkv[l] : nullptr);
    }
    const double nt = (double)m->tabs.T + 2;
    KR(CAT_EDGE_TABLE, (Pd + 1) * (pB * Wd * n_dkv * (want_forces ? 2 : 1) + 24) + nt * 12.0 * Wd * n_dkv,
       (launch_pair_buckets(g, P, hp.cutoff_lower, hp.cutoff_upper, m->tabs.T, b.C, b.dC, b.shist, b.skeys_s, b.svals_s, s),
        n_dkv ? launch_edge_interp(g, P, hp.cutoff_lower, hp.cutoff_upper, m->tabs.T, Wd, n_dkv, tabs.data(), outs.data(), douts.data(),
                                   b.skeys_s, b.svals_s, s, b.C, b.dC, pbf)
              : (void)0));
    if (hp.neighbor_embedding) {
      const float* t1[1] = {m->tabs.tab[n_dkv]};
      float* o1[1] = {b.Wn};
      float* d1[1] = {want_forces ? b.dWn : nullptr};
      KR(CAT_EDGE_TABLE, (Pd + 1) * 4.0 * Fd * (want_forces ? 2 : 1) + nt * 12.0 * Fd,
         launch_edge_interp(g, P, hp.cutoff_lower, hp.cutoff_upper, m->tabs.T, F, 1, t1, o1, d1, b.skeys_s, b.svals_s, s,
                            n_dkv ? nullptr : b.C, n_dkv ? nullptr : b.dC));
    }
  } else {
    KR(CAT_ELEMENTWISE, Pd * K * 8, launch_radial(g, P, RadialParams{W.means, W.betas, K, hp.cutoff_lower, hp.cutoff_upper}, b.phi, b.dphi, b.C, b.dC, s));
  }
  if (hp.neighbor_embedding) {
    EDGE(1);
    if (use_tab) {
    } else if (want_forces && !tc) gemm_dual(s, 3, b.phi, b.dphi, K, W.Wn, W.bn, b.Wn, b.dWn, F, P1, F, K, b.C, b.dC, W.Wn_sb);
    else gemm(s, b.phi, K, W.Wn, K, W.bn, b.Wn, F, P1, F, K, GEMM_ROWSCALE, nullptr, 0, nullptr, 0, b.C);  // energies only: no tangents
    KR(CAT_SCATTER, Ed * Fd * 8, launch_et_nbr_embed(g, N, F, z, W.emb, W.embN, b.Wn, b.xcat, s));
    NODE();
    gemm(s, b.xcat, 2 * F, W.Wc, 2 * F, W.bc, b.x[0], F, N, F, 2 * F);
  } else {
    KR(CAT_ELEMENTWISE, Nd * Fd * 4, launch_et_embed(z, W.emb, N, F, b.x[0], s));
  }
  KR(CAT_ELEMENTWISE, Nd * Fd * 12, launch_fill(b.vec[0], 0.f, (int64_t)N * 3 * F, s));
  // which generation of the attention sweeps runs (decided on the device) + the per-edge records of the tile sweeps
  KR(CAT_ELEMENTWISE, Ed * 64, launch_et_tile_prep(g, N, B, batch, b.C, b.dC, 2 * (int64_t)P1 + N, pbf, b.tile_open, b.tile_start, b.erec, s));
  std::vector<EtAttnArgs> aa(L);
  for (int l = 0; l < L; ++l) {
    const EtLayerP& q = W.layer[l];
    NODE();
    KR(CAT_ELEMENTWISE, Nd * Fd * 12, launch_layernorm_fwd(b.x[l], q.ln_w, q.ln_b, N, F, b.xt[l], b.xh[l], b.rstd[l], s));
    gemm(s, b.xt[l], F, q.Wqkv, F, q.bqkv, b.qkv[l], 5 * F, N, 5 * F, F);
    gemm(s, b.vec[l], F, q.Wvp, F, nullptr, b.vp[l], 3 * F, 3 * N, 3 * F, F);
    if (Wd > 0 && !use_tab) {
      EDGE(1);
      if (tc) gemm(s, b.phi, K, q.Wdkv, K, q.bdkv, b.dkv[l], Wd, P1, Wd, K, GEMM_ACT_SILU, tc->pre3[l], Wd);  // pre-activation kept
      else if (want_forces) gemm_dual(s, 1, b.phi, b.dphi, K, q.Wdkv, q.bdkv, b.dkv[l], b.tkv[l], Wd, P1, Wd, K, nullptr, nullptr, q.Wdkv_sb);
      else gemm(s, b.phi, K, q.Wdkv, K, q.bdkv, b.dkv[l], Wd, P1, Wd, K, GEMM_ACT_SILU);
    }
    EtAttnArgs& a = aa[l];
    a = EtAttnArgs{b.qkv[l], b.vec[l], b.dkv[l], b.tkv[l], b.C, b.dC, F, hd, Wd,
                   (hp.distance_influence & 1) ? 0 : -1, (hp.distance_influence & 2) ? ((hp.distance_influence & 1) ? F : 0) : -1,
                   hp.vector_cutoff, 2 * (int64_t)P1, pbf, b.tile_open, b.tile_start, et_g16_max_tiles(N, B), b.erec};
    // algorithmic bytes (every distinct tensor once): dkv [P+1, Wd], qkv [N,5F], vec [N,3F] in; xagg [N,F],
    // vagg [N,3F] out; edge indices
    float* const xagg_l = tc ? tc->Ch[l] : b.xagg;  // kept per layer when parameter gradients are wanted
    KR(CAT_MESSAGE, (Pd + 1) * Wd * pB + Nd * Fd * 4 * 12 + Ed * 12, launch_et_attn_fwd(g, N, a, xagg_l, b.vagg, s));
    NODE();
    gemm(s, xagg_l, F, q.Wo, F, q.bo, b.o[l], 3 * F, N, 3 * F, F);
    KR(CAT_ELEMENTWISE, Nd * Fd * 4 * 20,
       launch_et_update(b.x[l], b.vec[l], b.vp[l], b.o[l], b.vagg, N, F, b.x[l + 1], b.vec[l + 1], b.vdot[l], s));
  }
  NODE();
  KR(CAT_ELEMENTWISE, Nd * Fd * 12, launch_layernorm_fwd(b.x[L], W.lno_w, W.lno_b, N, F, b.xf, b.xfh, b.rstdf, s));
  const int U = F + F2;  // u12 row width: vec1_proj | vec2_proj of head block 0
  gemm(s, b.vec[L], F, W.W1u, F, nullptr, b.u12, U, 3 * N, U, F);
  KR(CAT_ELEMENTWISE, Nd * Fd * 24, launch_et_cat_norm(b.xf, F, b.u12, U, F, N, b.hcat, s));
  gemm(s, b.hcat, 2 * F, W.Wm1, 2 * F, W.bm1, b.m1, F, N, F, 2 * F, GEMM_ACT_SILU, b.pre1, F);
  gemm(s, b.m1, F, W.Wm2, F, W.bm2, b.y, F, N, F, F);
  KR(CAT_ELEMENTWISE, Nd * Fd * 16, launch_et_head_mid(b.y, b.u12 + F, U, F2, N, b.hcat2, b.vq, s));
  gemm(s, b.vq, F2, W.W21, F2, nullptr, b.w1, F2, 3 * N, F2, F2);
  KR(CAT_ELEMENTWISE, Nd * Fd * 8, launch_et_cat_norm(nullptr, F2, b.w1, F2, F2, N, b.hcat2, s));
  gemm(s, b.hcat2, F, W.Wn1, F, W.bn1, b.m2, F2, N, F2, F, GEMM_ACT_SILU, b.pre2, F2);
  if ((int64_t)N <= 256 * (int64_t)B) {
    KR(CAT_ELEMENTWISE, Nd * F2 * 4,
       launch_head_mol_sum(g, b.pre2, W.Wn2, W.bn2, N, B, F2, W.std, W.atomref, z, batch, W.mean, energy, s,
                           want_forces ? b.g_pre2 : nullptr));
  } else {
    KR(CAT_ELEMENTWISE, Nd * F2 * 4, launch_head_energy(b.pre2, W.Wn2, W.bn2, N, F2, W.std, W.atomref, z, b.ea, s, want_forces ? b.g_pre2 : nullptr));
    KR(CAT_ELEMENTWISE, Nd * 12, launch_mol_sum(g, b.ea, batch, N, B, W.mean, energy, s));
  }

  // ---------------- reverse (oracle/et_adjoint.py)
  if (want_forces) {
    NODE();
    // parameter gradients: dW (+)= g_out^T in, db = colsum(g_out) wherever an adjoint meets its layer's input (tn_train.hip)
    auto RP = [](int64_t ld) { return rows_plain(ld); };
    auto dW = [&](const char* key, const float* gOut, int64_t ldg, const float* In, int64_t ldi, int R, int Nout, int Kin,
                  const float* rs = nullptr, bool acc = false) {
      launch_tn_gemm(s, gOut, RP(ldg), In, RP(ldi), rs, nullptr, R, Nout, Kin, tc->at(key), acc, tc->part);
    };
    auto dB = [&](const char* key, const float* gOut, int64_t ldg, int R, int ncol, const float* mul = nullptr, int64_t ldm = 0,
                  const float* rs = nullptr, bool acc = false) {
      launch_colsum(s, gOut, RP(ldg), mul, RP(ldm), rs, nullptr, R, ncol, tc->at(key), acc, tc->part);
    };
    if (tc) {
      launch_train_seed(b.pre2, tc->gE, batch, N, F2, W.std, b.g_pre2, tc->head, s);  // rows of g_pre2 scaled by d loss / d E_mol
      dB("Wn2", tc->head, F2 + 1, N, F2);
      dB("bn2", tc->head + F2, F2 + 1, N, 1);
      dW("Wn1", b.g_pre2, F2, b.hcat2, F, N, F2, F);
      dB("bn1", b.g_pre2, F2, N, F2);
    }
    gemm(s, b.g_pre2, F2, W.Wn1T, F2, nullptr, b.g_h2, F, N, F, F2);                       // (g_xs | g_n2)
    KR(CAT_ELEMENTWISE, Nd * Fd * 12, launch_et_norm_bwd(b.g_h2 + F2, F, b.w1, F2, F2, N, b.g_w1, F2, s));
    if (tc) dW("W21", b.g_w1, F2, b.vq, F2, 3 * N, F2, F2);
    gemm(s, b.g_w1, F2, W.W21T, F2, nullptr, b.g_vq, F2, 3 * N, F2, F2);
    KR(CAT_ELEMENTWISE, Nd * Fd * 20,
       launch_et_head_mid_bwd(b.y, b.u12 + F, U, b.g_h2, b.g_vq, F2, N, b.g_y, b.g_u12 + F, U, s));
    if (tc) {
      dW("Wm2", b.g_y, F, b.m1, F, N, F, F);
      dB("bm2", b.g_y, F, N, F);
    }
    gemm(s, b.g_y, F, W.Wm2T, F, nullptr, b.g_m1, F, N, F, F, GEMM_MUL_DSILU_AUX, nullptr, 0, b.pre1, F);
    if (tc) {
      dW("Wm1", b.g_m1, F, b.hcat, 2 * F, N, F, 2 * F);
      dB("bm1", b.g_m1, F, N, F);
    }
    gemm(s, b.g_m1, F, W.Wm1T, F, nullptr, b.g_h1, 2 * F, N, 2 * F, F);                    // (g_xf | g_n1)
    KR(CAT_ELEMENTWISE, Nd * Fd * 24, launch_et_norm_bwd(b.g_h1 + F, 2 * F, b.u12, U, F, N, b.g_u12, U, s));
    if (tc) dW("W1u", b.g_u12, U, b.vec[L], F, 3 * N, U, F);
    gemm(s, b.g_u12, U, W.W1uT, U, nullptr, b.g_vec, F, 3 * N, F, U);
    // LayerNorm adjoint of out_norm: its input gradient is the first F columns of g_h1 (row stride 2F) -> compact copy
    KR(CAT_ELEMENTWISE, Nd * Fd * 8, launch_et_copy2d(b.g_h1, 2 * F, b.g_xf, F, N, F, s));
    if (tc) {
      dB("lno_w", b.g_xf, F, N, F, b.xfh, F);
      dB("lno_b", b.g_xf, F, N, F);
    }
    KR(CAT_ELEMENTWISE, Nd * Fd * 16, launch_layernorm_bwd(b.g_xf, b.xfh, b.rstdf, W.lno_w, N, F, b.g_x, s));
    const int nwv = et_sweep_waves(F);
    const int64_t sstride = 2 * (int64_t)P1;   // one slot array: [pair][direction]
    float* gd_emb = b.gd2 + sstride * nwv * L;   // the neighbour-embedding term of g_d gets its own array
    KR(CAT_ELEMENTWISE, Pd * 8, launch_fill(gd_emb, 0.f, sstride, s));
    for (int l = L - 1; l >= 0; --l) {
      const EtLayerP& q = W.layer[l];
      NODE();
      KR(CAT_ELEMENTWISE, Nd * Fd * 4 * 28,
         launch_et_update_bwd(b.g_x, b.g_vec, b.vp[l], b.o[l], b.vdot[l], N, F, b.g_o, b.g_vp, s));
      const std::string t_ = "l" + std::to_string(l) + ".";
      if (tc) {
        dW((t_ + "Wo").c_str(), b.g_o, 3 * F, tc->Ch[l], F, N, 3 * F, F);
        dB((t_ + "bo").c_str(), b.g_o, 3 * F, N, 3 * F);
        dW((t_ + "Wvp").c_str(), b.g_vp, 3 * F, b.vec[l], F, 3 * N, 3 * F, F);
      }
      gemm(s, b.g_o, 3 * F, q.WoT, 3 * F, nullptr, b.g_xagg, F, N, F, 3 * F);
      // g_vagg = g_vec (read by both sweeps before sweep "s" adds the source terms into it: snapshot in vagg)
      KR(CAT_ELEMENTWISE, Nd * Fd * 24, launch_et_copy2d(b.g_vec, 3 * F, b.vagg, 3 * F, N, 3 * F, s));
      // dkv + tkv, qkv, vec, g_xagg, g_vagg in; g_qkv out, g_vec read + written; the (pair, direction) slots of g_d, g_rhat
      KR(CAT_PAIR, 2 * (Pd + 1) * Wd * pB + Nd * Fd * 4 * 23 + Ed * 12 + (Pd + 1) * 32 * nwv,
         launch_et_attn_bwd(g, N, aa[l], b.g_xagg, b.vagg, b.g_qkv, b.g_vec, b.gd2 + sstride * nwv * l, b.gr2 + 3 * sstride * nwv * l, s));
      if (tc) {
        if (Wd > 0) {  // filter rows: adjoint per directed edge -> per pair (self pair: summed over the atoms) -> dk_proj / dv_proj
          const int64_t dir = (int64_t)P1 * Wd;
          launch_et_train_filter(g, N, aa[l], b.g_xagg, b.vagg, tc->gq, dir, tc->selfq, s);
          launch_colsum(s, tc->selfq, RP(Wd), nullptr, RP(Wd), nullptr, nullptr, N, Wd, tc->self_gw, false, tc->part);
          launch_et_train_gpre(g, P, Wd, tc->gq, dir, tc->self_gw, tc->pre3[l], tc->g3, s);
          dW((t_ + "Wdkv").c_str(), tc->g3, Wd, b.phi, K, P1, Wd, K);
          dB((t_ + "bdkv").c_str(), tc->g3, Wd, P1, Wd);
        }
        dW((t_ + "Wqkv").c_str(), b.g_qkv, 5 * F, b.xt[l], F, N, 5 * F, F);
        dB((t_ + "bqkv").c_str(), b.g_qkv, 5 * F, N, 5 * F);
      }
      gemm(s, b.g_vp, 3 * F, q.WvpT, 3 * F, nullptr, b.g_vec, F, 3 * N, F, 3 * F, GEMM_ACCUM);
      gemm(s, b.g_qkv, 5 * F, q.WqkvT, 5 * F, nullptr, b.g_xt, F, N, F, 5 * F);
      if (tc) {
        dB((t_ + "ln_w").c_str(), b.g_xt, F, N, F, b.xh[l], F);
        dB((t_ + "ln_b").c_str(), b.g_xt, F, N, F);
      }
      KR(CAT_ELEMENTWISE, Nd * Fd * 16, launch_layernorm_bwd(b.g_xt, b.xh[l], b.rstd[l], q.ln_w, N, F, b.g_ln, s));
      KR(CAT_ELEMENTWISE, Nd * Fd * 12, launch_et_add(b.g_ln, b.g_x, (int64_t)N * F, s));
    }
    if (hp.neighbor_embedding) {
      gemm(s, b.g_x, F, W.WcT, F, nullptr, b.g_xcat, 2 * F, N, 2 * F, F);
      if (!tc) KR(CAT_PAIR, Pd * Fd * 12, launch_et_nbr_embed_bwd(g, P, F, z, W.embN, b.g_xcat, b.dWn, gd_emb, s));
    }
    if (tc) {
      const int Z = hp.max_z;
      launch_onehot(z, N, Z, tc->onehot, s);
      if (hp.neighbor_embedding) {
        dW("Wc", b.g_x, F, b.xcat, 2 * F, N, F, 2 * F);
        dB("bc", b.g_x, F, N, F);
        dW("emb", tc->onehot, Z, b.g_xcat, 2 * F, N, Z, F);  // first half of g_xcat
        const int64_t dir = (int64_t)P1 * F;
        launch_et_train_nbr(g, N, F, z, W.embN, b.Wn, b.g_xcat, tc->gq, dir, tc->gZu, s);
        dW("embN", tc->onehot, Z, tc->gZu, F, N, Z, F);
        // Wn[p] = (distance_proj phi + b) C(d): both direction halves, rows scaled by the cutoff
        dW("Wn", tc->gq, F, b.phi, K, P, F, K, b.C);
        dW("Wn", tc->gq + dir, F, b.phi, K, P, F, K, b.C, true);
        dB("bn", tc->gq, F, P, F, nullptr, 0, b.C);
        dB("bn", tc->gq + dir, F, P, F, nullptr, 0, b.C, true);
      } else {
        dW("emb", tc->onehot, Z, b.g_x, F, N, Z, F);
      }
    } else {
      KR(CAT_PAIR, Pd * 48, launch_et_pair_combine(g, P, b.gd2, b.gr2, nwv * L, sstride, gd_emb, b.gd, b.g_rhat, s));
      KR(CAT_PAIR, Pd * 40, launch_geom_gd(g, P, b.gd, b.g_rhat, b.g_delta, s));
      KR(CAT_PAIR, Ed * 12, launch_force_gather(g, N, b.g_delta, perm, forces, s));
    }
  }
  m->et->last = b;
  m->et->lastN = N;
  m->et->has_last = true;
  HIP_TRY(m, hipGetLastError());
  return TMDNET_OK;
}

int et_debug_tensor(tmdnet_model* m, hipStream_t s, const char* name, float* out, int64_t numel) {
  if (!m->et->has_last || !name || !out) return TMDNET_ERR_STATE;
  const EtBuffers& b = m->et->last;
  const int64_t N = m->et->lastN, F = m->et->hp.hidden_channels;
  const int L = m->et->hp.num_layers;
  const std::string nm = name;
  const float* src = nullptr;
  int64_t n = 0;
  if (nm == "x_embed") src = b.x[0], n = N * F;
  else if (nm == "x_out") src = b.xf, n = N * F;
  else if (nm == "g_x") src = b.g_x, n = N * F;
  else if (nm == "g_vec") src = b.g_vec, n = N * 3 * F;
  else if (nm == "tile_meta") src = reinterpret_cast<const float*>(b.tile_open), n = 2;  // int bits: [0] open flag, [1] number of tiles (tn_et_g16.hip)
  else if (nm.rfind("x_layer", 0) == 0 || nm.rfind("vec_layer", 0) == 0) {
    const bool isx = nm[0] == 'x';
    const int l = std::atoi(nm.c_str() + (isx ? 7 : 9));
    if (l < 0 || l >= L) return fail(m, TMDNET_ERR_INVALID, "layer out of range");
    src = isx ? b.x[l + 1] : b.vec[l + 1];
    n = isx ? N * F : N * 3 * F;
  }
  if (!src) return fail(m, TMDNET_ERR_INVALID, "unknown tensor: " + nm);
  if (numel != n) return fail(m, TMDNET_ERR_INVALID, "wrong size for " + nm);
  HIP_TRY(m, hipMemcpyAsync(out, src, sizeof(float) * n, hipMemcpyDeviceToDevice, s));
  return TMDNET_OK;
}

// ------------------------------------------------------------------------------------ analytic second-order pass (force matching)
// d/d theta and d/d pos of  s = v . d(sum_m E_m)/d pos  for the Equivariant Transformer: the forward-mode tangent, along v, of the
// forward + reverse program above (reference: a second autograd pass, model.py:618-628 over torchmd_et.py:188-426).  One
// self-contained pass in the packed layouts of this file; statement by statement oracle/et_second_order.py, launch by launch
// tests/et_hvp_host_mirror.py (the same kernel bodies on the host).  Gradient buffer: et_train_layout.
#include "tn_et_hvp_math.h"
#include "tn_hvp.h"

namespace {

struct EtHvpLayer {
  float *xt, *xt_t, *xh, *xh_t, *rstd, *rstd_t, *qkv, *qkv_t, *vp, *vp_t, *ekv, *ekv_t, *dkv, *dkv_t, *xagg, *xagg_t, *o, *o_t, *vdot, *vdot_t;
};
struct EtHvpBuffers {
  float *phi, *dphi, *C, *dC, *d_t, *rhat_t, *phi_t, *C_t, *d2phi, *d2C, *g_dphi, *g_dphi_t;
  float *en, *en_t, *WnC, *WnC_t, *xcat, *xcat_t;
  std::vector<float*> x, x_t, vec, vec_t;  // L + 1
  std::vector<EtHvpLayer> lay;
  float *vagg, *vagg_t;
  float *xf, *xfh, *rstdf, *xf_t, *xfh_t, *rstdf_t, *u12, *u12_t, *hcat, *hcat_t, *pre1, *pre1_t, *m1, *m1_t, *y, *y_t, *hcat2, *hcat2_t, *vq,
      *vq_t, *w1, *w1_t, *pre2, *pre2_t, *g_pre2, *g_pre2_t, *headv;
  // reverse
  float *g_h2, *g_h2_t, *g_w1, *g_w1_t, *g_vq, *g_vq_t, *g_y, *g_y_t, *g_u12, *g_u12_t, *g_m1h, *g_m1h_t, *g_m1, *g_m1_t, *g_h1, *g_h1_t,
      *g_vec, *g_vec_t, *g_xf, *g_xf_t, *g_x, *g_x_t;
  float *g_o, *g_o_t, *g_vp, *g_vp_t, *g_xagg, *g_xagg_t, *g_vagg, *g_vagg_t, *g_qkv, *g_qkv_t, *g_vin, *g_vin_t, *gq, *gq_t, *selfq, *selfq_t,
      *self_g, *self_g_t, *g_e, *g_e_t, *de, *d2e, *g_xt, *g_xt_t, *g_ln, *g_ln_t, *slots, *slots_t;
  float *g_xcat, *g_xcat_t, *gZ_t, *g_Wn, *g_Wn_t, *g_en, *g_en_t, *g_cutn, *g_cutn_t, *g_cut, *g_cut_t, *g_rh, *g_rh_t, *gdel, *gdel_t, *onehot,
      *part;
};

// last call's buffers (developer / test hook tmdnet_hvp_debug_tensor)
thread_local EtHvpBuffers g_et_last;
thread_local int64_t g_et_lastN = -1, g_et_lastP = 0;
thread_local const tmdnet_model* g_et_last_model = nullptr;

EtHvpBuffers et_carve_hvp(void* ws, const tmdnet_et_hparams& hp, int64_t N, int64_t P, size_t* total) {
  const int64_t F = hp.hidden_channels, K = hp.num_rbf, L = hp.num_layers, Z = hp.max_z, F2 = F / 2, U = F + F2, H = hp.num_heads, P1 = P + 1;
  const int64_t Wd = std::max<int64_t>(wd_of(hp), 1), NF = N * F;
  Carver c(ws);
  EtHvpBuffers b;
  auto f = [&](int64_t n) { return c.take<float>(n); };
  b.phi = f(P1 * K); b.dphi = f(P1 * K); b.C = f(P1); b.dC = f(P1); b.d_t = f(P1); b.rhat_t = f(P1 * 3); b.phi_t = f(P1 * K); b.C_t = f(P1);
  b.d2phi = f(P1 * K); b.d2C = f(P1); b.g_dphi = f(P1); b.g_dphi_t = f(P1);
  b.en = f(P1 * F); b.en_t = f(P1 * F); b.WnC = f(P1 * F); b.WnC_t = f(P1 * F); b.xcat = f(2 * NF); b.xcat_t = f(2 * NF);
  for (int l = 0; l <= L; ++l) {
    b.x.push_back(f(NF)); b.x_t.push_back(f(NF)); b.vec.push_back(f(3 * NF)); b.vec_t.push_back(f(3 * NF));
  }
  for (int l = 0; l < L; ++l) {
    EtHvpLayer y;
    y.xt = f(NF); y.xt_t = f(NF); y.xh = f(NF); y.xh_t = f(NF); y.rstd = f(N); y.rstd_t = f(N); y.qkv = f(5 * NF); y.qkv_t = f(5 * NF);
    y.vp = f(9 * NF); y.vp_t = f(9 * NF); y.ekv = f(P1 * Wd); y.ekv_t = f(P1 * Wd); y.dkv = f(P1 * Wd); y.dkv_t = f(P1 * Wd);
    y.xagg = f(NF); y.xagg_t = f(NF); y.o = f(3 * NF); y.o_t = f(3 * NF); y.vdot = f(NF); y.vdot_t = f(NF);
    b.lay.push_back(y);
  }
  b.vagg = f(3 * NF); b.vagg_t = f(3 * NF);
  b.xf = f(NF); b.xfh = f(NF); b.rstdf = f(N); b.xf_t = f(NF); b.xfh_t = f(NF); b.rstdf_t = f(N); b.u12 = f(3 * N * U); b.u12_t = f(3 * N * U);
  b.hcat = f(2 * NF); b.hcat_t = f(2 * NF); b.pre1 = f(NF); b.pre1_t = f(NF); b.m1 = f(NF); b.m1_t = f(NF); b.y = f(NF); b.y_t = f(NF);
  b.hcat2 = f(NF); b.hcat2_t = f(NF); b.vq = f(3 * N * F2); b.vq_t = f(3 * N * F2); b.w1 = f(3 * N * F2); b.w1_t = f(3 * N * F2);
  b.pre2 = f(N * F2); b.pre2_t = f(N * F2); b.g_pre2 = f(N * F2); b.g_pre2_t = f(N * F2); b.headv = f(N * F2);
  b.g_h2 = f(NF); b.g_h2_t = f(NF); b.g_w1 = f(3 * N * F2); b.g_w1_t = f(3 * N * F2); b.g_vq = f(3 * N * F2); b.g_vq_t = f(3 * N * F2);
  b.g_y = f(NF); b.g_y_t = f(NF); b.g_u12 = f(3 * N * U); b.g_u12_t = f(3 * N * U); b.g_m1h = f(NF); b.g_m1h_t = f(NF); b.g_m1 = f(NF);
  b.g_m1_t = f(NF); b.g_h1 = f(2 * NF); b.g_h1_t = f(2 * NF); b.g_vec = f(3 * NF); b.g_vec_t = f(3 * NF); b.g_xf = f(NF); b.g_xf_t = f(NF);
  b.g_x = f(NF); b.g_x_t = f(NF);
  b.g_o = f(3 * NF); b.g_o_t = f(3 * NF); b.g_vp = f(9 * NF); b.g_vp_t = f(9 * NF); b.g_xagg = f(NF); b.g_xagg_t = f(NF); b.g_vagg = f(3 * NF);
  b.g_vagg_t = f(3 * NF); b.g_qkv = f(5 * NF); b.g_qkv_t = f(5 * NF); b.g_vin = f(3 * NF); b.g_vin_t = f(3 * NF);
  b.gq = f(2 * P1 * std::max<int64_t>(Wd, F)); b.gq_t = f(2 * P1 * std::max<int64_t>(Wd, F)); b.selfq = f(N * Wd); b.selfq_t = f(N * Wd);
  b.self_g = f(Wd); b.self_g_t = f(Wd); b.g_e = f(P1 * Wd); b.g_e_t = f(P1 * Wd); b.de = f(P1 * std::max<int64_t>(Wd, F));
  b.d2e = f(P1 * std::max<int64_t>(Wd, F)); b.g_xt = f(NF); b.g_xt_t = f(NF); b.g_ln = f(NF); b.g_ln_t = f(NF);
  b.slots = f(L * 2 * P1 * H * 4); b.slots_t = f(L * 2 * P1 * H * 4);
  b.g_xcat = f(2 * NF); b.g_xcat_t = f(2 * NF); b.gZ_t = f(NF); b.g_Wn = f(P1 * F); b.g_Wn_t = f(P1 * F); b.g_en = f(P1 * F); b.g_en_t = f(P1 * F);
  b.g_cutn = f(P1); b.g_cutn_t = f(P1); b.g_cut = f(P1); b.g_cut_t = f(P1); b.g_rh = f(P1 * 3); b.g_rh_t = f(P1 * 3); b.gdel = f(P1 * 3);
  b.gdel_t = f(P1 * 3); b.onehot = f(N * Z);
  const int64_t big = std::max<int64_t>({5 * F * F, Wd * K, Z * F, 2 * F * F, U * F});
  b.part = f((int64_t)train_part_floats((int)std::max<int64_t>(P1, 3 * N), big));
  if (total) *total = c.off;
  return b;
}

}  // namespace

int et_force_param_workspace_bytes(const tmdnet_model* m, int64_t n_atoms, int64_t n_pairs, size_t* bytes) {
  et_carve_hvp(nullptr, m->et->hp, n_atoms, n_pairs, bytes);
  return TMDNET_OK;
}

int et_force_param_grads(tmdnet_model* m, hipStream_t s, const Graph& g, void* ws, size_t ws_bytes, int64_t n_atoms, int64_t n_mol,
                         int64_t n_pairs, const int64_t* z, const int64_t* batch, const float* v, const float* ge, float* grads, float* hv) {
  (void)n_mol;
  if (m->et->hp.hidden_channels > 512) return fail(m, TMDNET_ERR_INVALID, "second-order pass: hidden_channels <= 512 (a block per atom, a thread per channel)");
  const tmdnet_et_hparams& hp = m->et->hp;
  const int F = hp.hidden_channels, K = hp.num_rbf, L = hp.num_layers, Z = hp.max_z, F2 = F / 2, U = F + F2, H = hp.num_heads, hd = F / H;
  const int Wd = wd_of(hp), N = (int)n_atoms, P = (int)n_pairs, P1 = P + 1;
  const int ok = (hp.distance_influence & 1) ? 0 : -1, ov = (hp.distance_influence & 2) ? ((hp.distance_influence & 1) ? F : 0) : -1;
  size_t need = 0;
  EtHvpBuffers b = et_carve_hvp(ws, hp, n_atoms, n_pairs, &need);
  if (need > ws_bytes) return fail(m, TMDNET_ERR_WORKSPACE, "second-order workspace too small: need " + std::to_string(need));
  const EtParams& W = m->et->P;
  g_gemm_cat = CAT_GEMM_NODE;
  g_mdev = nullptr;
  g_madd = 0;
  std::map<std::string, int64_t> off;
  int64_t total = 0;
  for (const auto& kv : et_train_layout(m)) {
    off[kv.first] = total;
    total += (kv.second + 63) & ~int64_t(63);
  }
  launch_fill(grads, 0.f, total, s);
  auto at = [&](const std::string& k) { return grads + off.at(k); };
  auto RP = [](int64_t ld) { return rows_plain(ld); };
  // d W = g_y_t^T x + g_y^T x_t, d b = colsum(g_y_t) of a dense layer y = x W^T + b
  auto dense = [&](const std::string& wkey, const std::string& bkey, const float* gy, const float* gy_t, int64_t ldg, const float* x,
                   const float* x_t, int64_t ldx, int R, int Nout, int Kin) {
    launch_tn_gemm_pair(s, gy_t, x, gy, x_t, RP(ldg), RP(ldx), nullptr, R, Nout, Kin, at(wkey), false, b.part);  // one launch, one reduction
    if (!bkey.empty()) launch_colsum(s, gy_t, RP(ldg), nullptr, RP(ldg), nullptr, nullptr, R, Nout, at(bkey), false, b.part);
  };
  auto ln_grad = [&](const std::string& wkey, const std::string& bkey, const float* gy, const float* gy_t, const float* xh, const float* xh_t) {
    launch_colsum(s, gy_t, RP(F), xh, RP(F), nullptr, nullptr, N, F, at(wkey), false, b.part);
    launch_colsum(s, gy, RP(F), xh_t, RP(F), nullptr, nullptr, N, F, at(wkey), true, b.part);
    launch_colsum(s, gy_t, RP(F), nullptr, RP(F), nullptr, nullptr, N, F, at(bkey), false, b.part);
  };

  // ---- radial functions and the tangent of the geometry
  launch_radial(g, P, RadialParams{W.means, W.betas, K, hp.cutoff_lower, hp.cutoff_upper}, b.phi, b.dphi, b.C, b.dC, s);
  hvp::launch_pair_tangent(g, P, K, v, b.dphi, b.dC, b.d_t, b.rhat_t, b.phi_t, b.C_t, s);
  hvp::launch_radial2(g, P, K, W.means, W.betas, hp.cutoff_lower, hp.cutoff_upper, b.d2phi, b.d2C, s);
  launch_fill(b.g_dphi, 0.f, P1, s);
  launch_fill(b.g_dphi_t, 0.f, P1, s);
  launch_fill(b.g_cutn, 0.f, P1, s);
  launch_fill(b.g_cutn_t, 0.f, P1, s);

  // ---- embedding
  if (hp.neighbor_embedding) {
    gemm(s, b.phi, K, W.Wn, K, W.bn, b.en, F, P1, F, K);
    gemm(s, b.phi_t, K, W.Wn, K, nullptr, b.en_t, F, P1, F, K);
    hvp::launch_rowscale_dual(P1, F, b.en, b.en_t, b.C, b.C_t, b.WnC, b.WnC_t, s);
    hvp::launch_et_nbr_embed_dual(g, N, F, P, z, W.emb, W.embN, b.WnC, b.WnC_t, b.xcat, b.xcat_t, s);
    gemm(s, b.xcat, 2 * F, W.Wc, 2 * F, W.bc, b.x[0], F, N, F, 2 * F);
    gemm(s, b.xcat_t, 2 * F, W.Wc, 2 * F, nullptr, b.x_t[0], F, N, F, 2 * F);
  } else {
    hvp::launch_et_embed_dual(N, F, z, W.emb, b.x[0], b.x_t[0], s);
  }
  launch_fill(b.vec[0], 0.f, (int64_t)N * 3 * F, s);
  launch_fill(b.vec_t[0], 0.f, (int64_t)N * 3 * F, s);
  std::vector<hvp::EtAttn> aa(L);
  for (int l = 0; l < L; ++l) {
    const EtLayerP& q = W.layer[l];
    EtHvpLayer& y = b.lay[l];
    hvp::launch_ln_dual(N, F, b.x[l], b.x_t[l], q.ln_w, q.ln_b, y.xt, y.xh, y.rstd, y.xt_t, y.xh_t, y.rstd_t, s);
    gemm(s, y.xt, F, q.Wqkv, F, q.bqkv, y.qkv, 5 * F, N, 5 * F, F);
    gemm(s, y.xt_t, F, q.Wqkv, F, nullptr, y.qkv_t, 5 * F, N, 5 * F, F);
    gemm(s, b.vec[l], F, q.Wvp, F, nullptr, y.vp, 3 * F, 3 * N, 3 * F, F);
    gemm(s, b.vec_t[l], F, q.Wvp, F, nullptr, y.vp_t, 3 * F, 3 * N, 3 * F, F);
    if (Wd > 0) {
      gemm(s, b.phi, K, q.Wdkv, K, q.bdkv, y.dkv, Wd, P1, Wd, K, GEMM_ACT_SILU, y.ekv, Wd);
      gemm(s, b.phi_t, K, q.Wdkv, K, nullptr, y.ekv_t, Wd, P1, Wd, K);
      hvp::launch_silu_tangent((int64_t)P1 * Wd, y.ekv, y.ekv_t, y.dkv_t, s);
    }
    aa[l] = hvp::EtAttn{y.qkv, y.qkv_t, b.vec[l], b.vec_t[l], y.dkv, y.dkv_t, b.C, b.C_t, g.prhat, b.rhat_t, F, hd, std::max(Wd, 1), ok, ov,
                        hp.vector_cutoff ? 1 : 0, P};
    hvp::launch_et_attn_fwd_dual(g, N, H, aa[l], y.xagg, y.xagg_t, b.vagg, b.vagg_t, s);
    gemm(s, y.xagg, F, q.Wo, F, q.bo, y.o, 3 * F, N, 3 * F, F);
    gemm(s, y.xagg_t, F, q.Wo, F, nullptr, y.o_t, 3 * F, N, 3 * F, F);
    hvp::launch_et_update_dual(N, F, b.x[l], b.x_t[l], b.vec[l], b.vec_t[l], y.vp, y.vp_t, y.o, y.o_t, b.vagg, b.vagg_t, b.x[l + 1],
                               b.x_t[l + 1], b.vec[l + 1], b.vec_t[l + 1], y.vdot, y.vdot_t, s);
  }

  // ---- out_norm and the EquivariantScalar head
  hvp::launch_ln_dual(N, F, b.x[L], b.x_t[L], W.lno_w, W.lno_b, b.xf, b.xfh, b.rstdf, b.xf_t, b.xfh_t, b.rstdf_t, s);
  gemm(s, b.vec[L], F, W.W1u, F, nullptr, b.u12, U, 3 * N, U, F);
  gemm(s, b.vec_t[L], F, W.W1u, F, nullptr, b.u12_t, U, 3 * N, U, F);
  hvp::launch_et_cat_norm_dual(N, F, b.xf, b.xf_t, F, b.u12, b.u12_t, U, F, 2 * F, b.hcat, b.hcat_t, s);
  gemm(s, b.hcat, 2 * F, W.Wm1, 2 * F, W.bm1, b.m1, F, N, F, 2 * F, GEMM_ACT_SILU, b.pre1, F);
  gemm(s, b.hcat_t, 2 * F, W.Wm1, 2 * F, nullptr, b.pre1_t, F, N, F, 2 * F);
  hvp::launch_silu_tangent((int64_t)N * F, b.pre1, b.pre1_t, b.m1_t, s);
  gemm(s, b.m1, F, W.Wm2, F, W.bm2, b.y, F, N, F, F);
  gemm(s, b.m1_t, F, W.Wm2, F, nullptr, b.y_t, F, N, F, F);
  hvp::launch_et_head_mid_dual(N, F2, b.y, b.y_t, b.u12 + F, b.u12_t + F, U, b.hcat2, b.hcat2_t, b.vq, b.vq_t, s);
  gemm(s, b.vq, F2, W.W21, F2, nullptr, b.w1, F2, 3 * N, F2, F2);
  gemm(s, b.vq_t, F2, W.W21, F2, nullptr, b.w1_t, F2, 3 * N, F2, F2);
  hvp::launch_et_cat_norm_dual(N, F2, nullptr, nullptr, F2, b.w1, b.w1_t, F2, F2, F, b.hcat2, b.hcat2_t, s);
  gemm(s, b.hcat2, F, W.Wn1, F, W.bn1, b.pre2, F2, N, F2, F);
  gemm(s, b.hcat2_t, F, W.Wn1, F, nullptr, b.pre2_t, F2, N, F2, F);
  hvp::launch_head_dual(N, F2, b.pre2, b.pre2_t, W.Wn2, W.std, ge, batch, b.g_pre2, b.g_pre2_t, b.headv, s);

  // ---- reverse with tangents: head
  launch_colsum(s, b.headv, RP(F2), nullptr, RP(F2), nullptr, nullptr, N, F2, at("Wn2"), false, b.part);  // d s / d bn2 = 0
  if (ge) hvp::launch_head_bias_seed(N, W.std, ge, batch, at("bn2"), s);  // energy seed: d S / d bn2 = - std sum_n ge[molecule(n)]
  dense("Wn1", "bn1", b.g_pre2, b.g_pre2_t, F2, b.hcat2, b.hcat2_t, F, N, F2, F);
  gemm(s, b.g_pre2, F2, W.Wn1T, F2, nullptr, b.g_h2, F, N, F, F2);
  gemm(s, b.g_pre2_t, F2, W.Wn1T, F2, nullptr, b.g_h2_t, F, N, F, F2);
  hvp::launch_et_norm_bwd_dual(N, F2, b.g_h2 + F2, b.g_h2_t + F2, F, b.w1, b.w1_t, F2, b.g_w1, b.g_w1_t, F2, s);
  dense("W21", "", b.g_w1, b.g_w1_t, F2, b.vq, b.vq_t, F2, 3 * N, F2, F2);
  gemm(s, b.g_w1, F2, W.W21T, F2, nullptr, b.g_vq, F2, 3 * N, F2, F2);
  gemm(s, b.g_w1_t, F2, W.W21T, F2, nullptr, b.g_vq_t, F2, 3 * N, F2, F2);
  hvp::launch_et_head_mid_bwd_dual(N, F2, b.y, b.y_t, b.u12 + F, b.u12_t + F, U, b.g_h2, b.g_h2_t, b.g_vq, b.g_vq_t, b.g_y, b.g_y_t, b.g_u12 + F,
                                   b.g_u12_t + F, U, s);
  dense("Wm2", "bm2", b.g_y, b.g_y_t, F, b.m1, b.m1_t, F, N, F, F);
  gemm(s, b.g_y, F, W.Wm2T, F, nullptr, b.g_m1h, F, N, F, F);
  gemm(s, b.g_y_t, F, W.Wm2T, F, nullptr, b.g_m1h_t, F, N, F, F);
  hvp::launch_dsilu_dual((int64_t)N * F, b.g_m1h, b.g_m1h_t, b.pre1, b.pre1_t, b.g_m1, b.g_m1_t, s);
  dense("Wm1", "bm1", b.g_m1, b.g_m1_t, F, b.hcat, b.hcat_t, 2 * F, N, F, 2 * F);
  gemm(s, b.g_m1, F, W.Wm1T, F, nullptr, b.g_h1, 2 * F, N, 2 * F, F);
  gemm(s, b.g_m1_t, F, W.Wm1T, F, nullptr, b.g_h1_t, 2 * F, N, 2 * F, F);
  hvp::launch_et_norm_bwd_dual(N, F, b.g_h1 + F, b.g_h1_t + F, 2 * F, b.u12, b.u12_t, U, b.g_u12, b.g_u12_t, U, s);
  dense("W1u", "", b.g_u12, b.g_u12_t, U, b.vec[L], b.vec_t[L], F, 3 * N, U, F);
  gemm(s, b.g_u12, U, W.W1uT, U, nullptr, b.g_vec, F, 3 * N, F, U);
  gemm(s, b.g_u12_t, U, W.W1uT, U, nullptr, b.g_vec_t, F, 3 * N, F, U);
  launch_et_copy2d(b.g_h1, 2 * F, b.g_xf, F, N, F, s);
  launch_et_copy2d(b.g_h1_t, 2 * F, b.g_xf_t, F, N, F, s);
  ln_grad("lno_w", "lno_b", b.g_xf, b.g_xf_t, b.xfh, b.xfh_t);
  hvp::launch_lnbwd_dual(N, F, b.g_xf, b.g_xf_t, b.xfh, b.xfh_t, b.rstdf, b.rstdf_t, W.lno_w, b.g_x, b.g_x_t, s);

  // ---- reverse with tangents: attention layers
  const int64_t slot_dir = (int64_t)P1 * H * 4, Wdx = std::max(Wd, 1), qdir = (int64_t)P1 * Wdx;
  for (int l = L - 1; l >= 0; --l) {
    const EtLayerP& q = W.layer[l];
    EtHvpLayer& y = b.lay[l];
    const std::string t_ = "l" + std::to_string(l) + ".";
    hvp::launch_et_update_bwd_dual(N, F, b.g_x, b.g_x_t, b.g_vec, b.g_vec_t, y.vp, y.vp_t, y.o, y.o_t, y.vdot, y.vdot_t, b.g_o, b.g_o_t, b.g_vp,
                                   b.g_vp_t, s);
    dense(t_ + "Wo", t_ + "bo", b.g_o, b.g_o_t, 3 * F, y.xagg, y.xagg_t, F, N, 3 * F, F);
    dense(t_ + "Wvp", "", b.g_vp, b.g_vp_t, 3 * F, b.vec[l], b.vec_t[l], F, 3 * N, 3 * F, F);
    gemm(s, b.g_o, 3 * F, q.WoT, 3 * F, nullptr, b.g_xagg, F, N, F, 3 * F);
    gemm(s, b.g_o_t, 3 * F, q.WoT, 3 * F, nullptr, b.g_xagg_t, F, N, F, 3 * F);
    // g_vagg = g_vec: snapshot, the sweeps read it while the source terms are added to g_vec
    launch_et_copy2d(b.g_vec, 3 * F, b.g_vagg, 3 * F, N, 3 * F, s);
    launch_et_copy2d(b.g_vec_t, 3 * F, b.g_vagg_t, 3 * F, N, 3 * F, s);
    hvp::launch_et_attn_bwd_dual(g, N, H, aa[l], b.g_xagg, b.g_xagg_t, b.g_vagg, b.g_vagg_t, b.g_qkv, b.g_qkv_t, b.g_vin, b.g_vin_t, b.gq,
                                 b.gq_t, qdir, b.selfq, b.selfq_t, b.slots + (int64_t)l * 2 * slot_dir, b.slots_t + (int64_t)l * 2 * slot_dir,
                                 slot_dir, s);
    if (Wd > 0) {  // filter rows: per pair (self pair: summed over the atoms) -> dk_proj / dv_proj, and the distance gradient through phi
      launch_colsum(s, b.selfq, RP(Wd), nullptr, RP(Wd), nullptr, nullptr, N, Wd, b.self_g, false, b.part);
      launch_colsum(s, b.selfq_t, RP(Wd), nullptr, RP(Wd), nullptr, nullptr, N, Wd, b.self_g_t, false, b.part);
      hvp::launch_et_filter_gpre_dual(P, Wd, b.gq, b.gq_t, qdir, b.self_g, b.self_g_t, y.ekv, y.ekv_t, b.g_e, b.g_e_t, s);
      dense(t_ + "Wdkv", t_ + "bdkv", b.g_e, b.g_e_t, Wd, b.phi, b.phi_t, K, P1, Wd, K);
      gemm(s, b.dphi, K, q.Wdkv, K, nullptr, b.de, Wd, P1, Wd, K);
      gemm(s, b.d2phi, K, q.Wdkv, K, nullptr, b.d2e, Wd, P1, Wd, K);
      hvp::launch_pair_rowdot(P1, Wd, b.g_e, b.g_e_t, b.de, b.d2e, b.d_t, true, b.g_dphi, b.g_dphi_t, s);
    }
    dense(t_ + "Wqkv", t_ + "bqkv", b.g_qkv, b.g_qkv_t, 5 * F, y.xt, y.xt_t, F, N, 5 * F, F);
    hvp::launch_add2((int64_t)N * 3 * F, b.g_vin, b.g_vec, s);
    hvp::launch_add2((int64_t)N * 3 * F, b.g_vin_t, b.g_vec_t, s);
    gemm(s, b.g_vp, 3 * F, q.WvpT, 3 * F, nullptr, b.g_vec, F, 3 * N, F, 3 * F, GEMM_ACCUM);
    gemm(s, b.g_vp_t, 3 * F, q.WvpT, 3 * F, nullptr, b.g_vec_t, F, 3 * N, F, 3 * F, GEMM_ACCUM);
    gemm(s, b.g_qkv, 5 * F, q.WqkvT, 5 * F, nullptr, b.g_xt, F, N, F, 5 * F);
    gemm(s, b.g_qkv_t, 5 * F, q.WqkvT, 5 * F, nullptr, b.g_xt_t, F, N, F, 5 * F);
    ln_grad(t_ + "ln_w", t_ + "ln_b", b.g_xt, b.g_xt_t, y.xh, y.xh_t);
    hvp::launch_lnbwd_dual(N, F, b.g_xt, b.g_xt_t, y.xh, y.xh_t, y.rstd, y.rstd_t, q.ln_w, b.g_ln, b.g_ln_t, s);
    hvp::launch_add2((int64_t)N * F, b.g_ln, b.g_x, s);
    hvp::launch_add2((int64_t)N * F, b.g_ln_t, b.g_x_t, s);
  }

  // ---- embeddings
  launch_onehot(z, N, Z, b.onehot, s);
  if (hp.neighbor_embedding) {
    dense("Wc", "bc", b.g_x, b.g_x_t, F, b.xcat, b.xcat_t, 2 * F, N, F, 2 * F);
    gemm(s, b.g_x, F, W.WcT, F, nullptr, b.g_xcat, 2 * F, N, 2 * F, F);
    gemm(s, b.g_x_t, F, W.WcT, F, nullptr, b.g_xcat_t, 2 * F, N, 2 * F, F);
    launch_tn_gemm(s, b.onehot, RP(Z), b.g_xcat_t, RP(2 * F), nullptr, nullptr, N, Z, F, at("emb"), false, b.part);  // first half of g_xcat
    const int64_t ndir = (int64_t)P1 * F;
    hvp::launch_et_nbr_bwd_dual(g, N, F, P, z, W.embN, b.WnC, b.WnC_t, b.g_xcat, b.g_xcat_t, b.gq, b.gq_t, ndir, b.gZ_t, s);
    launch_tn_gemm(s, b.onehot, RP(Z), b.gZ_t, RP(F), nullptr, nullptr, N, Z, F, at("embN"), false, b.part);
    hvp::launch_et_nbr_pair_dual(P, F, b.gq, b.gq_t, ndir, b.C, b.C_t, b.g_Wn, b.g_Wn_t, b.g_en, b.g_en_t, s);
    dense("Wn", "bn", b.g_en, b.g_en_t, F, b.phi, b.phi_t, K, P, F, K);
    gemm(s, b.dphi, K, W.Wn, K, nullptr, b.de, F, P1, F, K);
    gemm(s, b.d2phi, K, W.Wn, K, nullptr, b.d2e, F, P1, F, K);
    hvp::launch_pair_rowdot(P, F, b.g_en, b.g_en_t, b.de, b.d2e, b.d_t, true, b.g_dphi, b.g_dphi_t, s);
    hvp::launch_pair_rowdot2(P, F, b.g_Wn, b.g_Wn_t, b.en, b.en_t, true, b.g_cutn, b.g_cutn_t, s);
  } else {
    launch_tn_gemm(s, b.onehot, RP(Z), b.g_x_t, RP(F), nullptr, nullptr, N, Z, F, at("emb"), false, b.part);
  }

  // ---- geometry: H v
  if (hv) {
    hvp::launch_et_pair_slots_dual(P, H, L, b.slots, b.slots_t, 2 * slot_dir, slot_dir, b.g_cut, b.g_cut_t, b.g_rh, b.g_rh_t, s);
    hvp::launch_add2(P, b.g_cutn, b.g_cut, s);
    hvp::launch_add2(P, b.g_cutn_t, b.g_cut_t, s);
    hvp::launch_et_geom_dual(g, P, b.d_t, b.rhat_t, b.dC, b.d2C, b.g_cut, b.g_cut_t, b.g_dphi, b.g_dphi_t, b.g_rh, b.g_rh_t, b.gdel, b.gdel_t, s);
    hvp::launch_pair_to_atom(g, N, P, b.gdel_t, hv, s);
  }
  HIP_TRY(m, hipGetLastError());
  g_et_last = b;
  g_et_lastN = N;
  g_et_lastP = P;
  g_et_last_model = m;
  return TMDNET_OK;
}

// intermediates of the last et_force_param_grads call by name (tests/test_gpu_hvp.py walks them against tests/et_hvp_host_mirror.py):
// "x{l}" "x_t{l}" "vec{l}" "vec_t{l}" (l = 0 .. L), "l{l}.qkv" ".qkv_t" ".dkv" ".dkv_t" ".xagg" ".xagg_t" ".o_t" ".vdot_t", the head's
// "pre2" "pre2_t" "g_pre2_t" "headv", the reverse sweep's scratch as its LAST layer (l = 0) left it: "g_qkv" "g_qkv_t" "g_vin" "g_vin_t"
// "gq" "gq_t" "selfq" "selfq_t", and "slots" "slots_t" [L][2][P + 1][H][4], "g_x" "g_x_t" (at the embedding), "g_cut" "g_rh" "gdel" (+ "_t")
int et_hvp_debug_tensor(tmdnet_model* m, hipStream_t s, const char* name, float* out, int64_t numel) {
  if (g_et_last_model != m || g_et_lastN < 0) return fail(m, TMDNET_ERR_STATE, "no second-order pass has run on this handle (this thread)");
  const tmdnet_et_hparams& hp = m->et->hp;
  const int64_t F = hp.hidden_channels, L = hp.num_layers, H = hp.num_heads, N = g_et_lastN, P = g_et_lastP, P1 = P + 1, NF = N * F, F2 = F / 2;
  const int64_t Wd = std::max<int64_t>(wd_of(hp), 1);
  const EtHvpBuffers& b = g_et_last;
  std::map<std::string, std::pair<const float*, int64_t>> t;
  for (int l = 0; l <= L; ++l) {
    const std::string k = std::to_string(l);
    t["x" + k] = {b.x[l], NF};
    t["x_t" + k] = {b.x_t[l], NF};
    t["vec" + k] = {b.vec[l], 3 * NF};
    t["vec_t" + k] = {b.vec_t[l], 3 * NF};
  }
  for (int l = 0; l < L; ++l) {
    const EtHvpLayer& y = b.lay[l];
    const std::string q = "l" + std::to_string(l) + ".";
    t[q + "qkv"] = {y.qkv, 5 * NF};
    t[q + "qkv_t"] = {y.qkv_t, 5 * NF};
    t[q + "dkv"] = {y.dkv, P1 * Wd};
    t[q + "dkv_t"] = {y.dkv_t, P1 * Wd};
    t[q + "xagg"] = {y.xagg, NF};
    t[q + "xagg_t"] = {y.xagg_t, NF};
    t[q + "o_t"] = {y.o_t, 3 * NF};
    t[q + "vdot_t"] = {y.vdot_t, NF};
  }
#define T_(field, n) t[#field] = {b.field, (n)}
  T_(pre2, N * F2); T_(pre2_t, N * F2); T_(g_pre2_t, N * F2); T_(headv, N * F2); T_(g_qkv, 5 * NF); T_(g_qkv_t, 5 * NF); T_(g_vin, 3 * NF);
  T_(g_vin_t, 3 * NF); T_(gq, 2 * P1 * Wd); T_(gq_t, 2 * P1 * Wd); T_(selfq, N * Wd); T_(selfq_t, N * Wd); T_(slots, L * 2 * P1 * H * 4);
  T_(slots_t, L * 2 * P1 * H * 4); T_(g_x, NF); T_(g_x_t, NF); T_(g_cut, P1); T_(g_cut_t, P1); T_(g_rh, P1 * 3); T_(g_rh_t, P1 * 3);
  T_(gdel, P * 3); T_(gdel_t, P * 3);
#undef T_
  auto it = t.find(name);
  if (it == t.end()) return fail(m, TMDNET_ERR_INVALID, std::string("unknown second-order tensor: ") + name);
  if (!out) return (int)it->second.second;
  if (numel != it->second.second) return fail(m, TMDNET_ERR_INVALID, "second-order tensor: size mismatch");
  HIP_TRY(m, hipMemcpyAsync(out, it->second.first, (size_t)numel * sizeof(float), hipMemcpyDeviceToDevice, s));
  return TMDNET_OK;
}
