// Host-side model state and launch helpers shared by the TensorNet schedule (tn_api.hip) and the Equivariant
// Transformer schedule (tn_et_api.hip).  Internal to libtmdnet_amd.so; the public surface is include/tmdnet_amd.h.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/tmdnet_amd.h"
#include "tn_gemm.h"
#include "tn_kernels.h"
#include "tn_train.h"

using namespace tn;


struct LayerP {
  const float *M1, *b1, *M2, *b2, *M3, *b3;
  const float *M2T, *M3T;  // transposes of M2, M3 (TensorNet: the parameter-gradient pass's data-gradient products)
  const float* V[6];
  const float* VT[6];
  const uint16_t* M_sb[3];  // split-bf16 tile images of M1..M3 (tn_gemm_sb.hip)
};

struct DevParams {
  const float *means, *betas;
  const float *Wdp, *bdp;
  const float *emb, *emb2_w, *emb2_b, *emb2_waT, *emb2_wbT;
  const float* Ue[3];
  const float* UeT[3];
  const float *L1, *bL1, *L1T, *L2, *bL2, *L2T;
  const float *ln0_w, *ln0_b;
  std::vector<LayerP> layer;
  const float *lnr_w, *lnr_b, *Lin, *bLin, *LinT;
  const float *O1, *bO1, *O1T, *O2, *bO2;
  const float* atomref;
  const uint16_t* Wdp_sb;
  const float *Utab, *Vtab;  // per-type pair-embedding tables (k_ztables at finalize)
  float mean, std;
};

struct ParamSpec {
  std::string name;
  int64_t rows, cols;  // cols = 1 for vectors
};

// carve helper: 256-byte aligned sub-buffers of one caller-owned allocation
struct Carver {
  char* base;
  size_t off = 0;
  explicit Carver(void* p) : base(reinterpret_cast<char*>(p)) {}
  template <typename T>
  T* take(int64_t n) {
    size_t bytes = (size_t)(n > 0 ? n : 0) * sizeof(T);
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += (bytes + 255) & ~size_t(255);
    return p;
  }
};

struct FwdBuffers {
  float *phi, *dphi, *C, *dC;
  float *Q, *u0, *s0n, *ln0, *xh0, *rstd0, *a1, *h1, *a2, *gates, *UX;
  std::vector<float*> X;                               // L+1
  std::vector<float*> w, dw, Pn, Mi, D;                 // per layer (dw = d w / d d, forward tangent)
  float *he1, *he2, *te1, *te2, *dQ, *Xh, *Ch;
  float *feat, *lnr, *xhr, *rstdr, *al, *x, *ao, *ea, *kap;
  // reverse
  unsigned* skeys_s;  // radial tables: fp32 distance bits of the pairs in grid-interval order
  int *svals_s, *shist;  // pair ids in that order; bucket counters [T + 2]
  float *g_ao, *g_al, *g_ln, *g_feat, *G, *gD, *gCh, *gMi, *gPn, *gXl, *gd, *gd_slots;
  float *gUX, *g_a2, *g_a1, *g_ln0, *g_s0n, *g_u0l, *gA, *g_rhat, *g_delta;
  float *mom, *gmom, *ps;  // embedding in the radial basis (tn_embed_rb.hip): moments [N][NTP][10][K + 4], their gradient, pair scalars [P + 1][8]
};


enum ProfCat { CAT_GRAPH = 0, CAT_GEMM_EDGE, CAT_GEMM_NODE, CAT_MESSAGE, CAT_PAIR, CAT_SCATTER, CAT_ELEMENTWISE, CAT_EDGE_TABLE, CAT_COUNT };


struct ProfRec {
  int cat;
  hipEvent_t a, b;
  double flops, bytes;
  char label[64];  // kernel + shape, e.g. "k_gemm_dual_sb2<2> 193710x384x256" (empty: the class name says it all)
};
struct Profiler {
  bool on = false;
  unsigned mask = 0;
  std::vector<ProfRec> recs;
  std::vector<hipEvent_t> pool;
  size_t used = 0;
  hipEvent_t get() {
    if (used == pool.size()) {
      hipEvent_t e;
      (void)hipEventCreate(&e);
      pool.push_back(e);
    }
    return pool[used++];
  }
};

struct EtModel;  // tn_et_api.hip

// TensorNet2 extras (tn_tn2_api.hip); everything TensorNet has lives in DevParams
struct CpParams {  // one ChargePredict head (tensornet2.py:49-66): LayerNorm(3F) + MLP 3F -> F -> F -> 2 q_dim
  const float *ln_w, *ln_b, *W1, *b1, *W1T, *W2, *b2, *W2T, *W3, *b3, *W3T;
};
struct Tn2LayerP {  // blocks of linears_scalar.0 = [M1a (F x K) | M1b (F x q) | M1c (F x q)] and transposes for the reverse pass
  const float *M1a, *M1b, *M1c, *M1bT, *M1cT, *M2T, *M3T;
};
struct Tn2Model {
  tmdnet_tn2_hparams hp;
  std::vector<CpParams> cp;      // num_layers + 1
  std::vector<Tn2LayerP> layer;
  const float* qweights = nullptr;  // [(L + 1) q_dim]
};

// radial tables of the per-pair functions (tn_edge_table.hip): built by tmdnet_finalize_params, verified against the direct
// evaluation; tab[0] = distance projections Q (embedding), tab[1 + l] = edge MLP of layer l; each [T + 2][2][3F]
struct EdgeTables {
  bool ok = false;
  int T = 0;
  std::vector<float*> tab;
  std::vector<int> R;                       // row length of each table
  double err_value = 0.0, err_slope = 0.0;  // measured at the interval midpoints, relative to the table's largest entry
};
// one table = a chain of dense layers applied to the radial basis; kind: 0 plain, 1 silu, 2 silu * C(d), 3 plain * C(d)
struct TableLayer {
  const float* W;  // [N][K] fp32 (device)
  const float* b;  // [N] or null
  int N, K, kind;
};
struct TableSpec {
  std::vector<TableLayer> chain;  // the last layer's N is the table's row length
};

// parameter-gradient pass (tmdnet_energy_param_grads): seeds, output and the extra activations the reverse pass keeps per layer
struct TrainCtx {
  int phase = 0;              // 0: forward + reverse in one call, 1: forward only (activations kept), 2: reverse only
  const float* gE = nullptr;  // [B] d loss / d E_m (device)
  float* grads = nullptr;     // flat gradient buffer (device), layout = train_layout(m)
  std::map<std::string, int64_t> off;
  std::vector<float*> pre1, he1, pre2, he2, pre3, Ch, Xh;  // per layer: edge-MLP pre-/post-activations, group product, normalised X
  float *g3 = nullptr, *g2 = nullptr, *g1 = nullptr;       // adjoints of the edge-MLP pre-activations [P + 1, 3F / 2F / F]
  float *self_gw = nullptr, *gq = nullptr, *selfq = nullptr, *gZu = nullptr, *gZv = nullptr, *onehot = nullptr, *head = nullptr;
  float *part = nullptr, *wT = nullptr, *forces = nullptr;
  float* at(const std::string& k) const { return grads + off.at(k); }
};

struct tmdnet_model {
  bool recompute_rows = false;  // option "recompute_pair_rows": the sweeps interpolate the per-pair rows themselves (no w / dw workspace)
  const float* atom_w = nullptr;  // per-atom weights of the energy sum (tmdnet_set_atom_weights), caller's atom order
  // per-layer halo exchange (tmdnet_set_halo_exchange): called between the per-atom kernels and the neighbour sweeps of a step
  int (*halo_fn)(void*, int32_t, float*, int64_t, int64_t, const int32_t*, void*) = nullptr;
  void* halo_user = nullptr;
  int* halo_rng = nullptr;          // device scratch: first / last / count of the owned rows in the engine's order
  bool graph_no_ghost_pairs = false;  // the last graph was built with the exchange set: no pairs of two ghosts in it
  int halo_active[2] = {0, 0};      // [first row, rows] the per-atom kernels of the last step ran on ("halo_active_rows")
  tmdnet_hparams hp;
  TrainCtx* train = nullptr;  // non-null while tmdnet_energy_param_grads drives tmdnet_energy_forces
  std::vector<std::pair<std::string, int64_t>> train_entries;  // gradient buffer layout (name, numel), built on first use
  // optional (TMDNET_SIDE_STREAM=1) second stream + events: the edge MLPs of the interaction layers depend on the pair geometry only, so they are
  // enqueued on `side` (fork after the radial kernel, one join per layer before its message sweep) and run beside the
  // per-atom chain; under HIP-graph capture the fork/join pattern becomes parallel branches of the graph
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr;
  std::vector<hipEvent_t> ev_join;
  EtModel* et = nullptr;  // non-null: Equivariant Transformer handle (hp then only carries what the graph phase reads)
  Tn2Model* tn2 = nullptr;  // non-null: TensorNet2 handle (hp carries the shared TensorNet hyper-parameters)
  // geometry of the last graph build (caller-owned device pointers; the TensorNet2 Coulomb head needs positions again)
  const float* g_pos = nullptr;
  const float* g_box = nullptr;
  int g_box_mode = 0;
  Profiler prof;
  int64_t lastE = 0;
  int cell_n[3] = {0, 0, 0};  // cell grid set by tmdnet_set_cell_grid (0 = brute force, < 0 = from the box, on the device)
  bool graph_is_cell = false; // last build used the cell list (atoms internally renumbered)
  bool graph_cell_multi = false;  // ... with several molecules in the common grid (their atoms interleaved: Graph::bat_c)
  bool graph_has_z = false;   // last build validated z into Graph::z_c (internal order)
  // what each graph workspace was last built as (host side, keyed by its address): a call that takes a workspace restores the four
  // fields above / last_nt from its record, so builds on several workspaces may be interleaved with their evaluations
  struct GraphRecord { bool is_cell, cell_multi, has_z; int nt; int64_t lastE; bool no_ghost_pairs; };
  std::unordered_map<const void*, GraphRecord> graph_rec;
  std::vector<ParamSpec> specs;
  std::map<std::string, std::vector<float>> host;
  float* dev = nullptr;  // packed parameters
  size_t dev_cap = 0, dev_sb_cap = 0, rb_cap = 0;  // element capacities of dev / dev_sb / rb_img (kept across re-uploads)
  uint16_t* dev_sb = nullptr;  // split-bf16 weight tile images
  std::unordered_map<const float*, const uint16_t*> sb_of;  // fp32 device weight -> its split image
  std::unordered_map<const float*, const uint16_t*> fm_of;  // fp32 device weight -> its fragment-major split image (tn_tlin9.hip)
  // device-side parameter update (tmdnet_update_params_device): the derived images to rebuild and, built on first use, where every
  // element of the packed buffer comes from (state-dict tensor, offset)
  struct ImageJob { size_t src_off; int64_t n, k; size_t dst_off; int fm; };
  std::vector<ImageJob> image_jobs;
  size_t packed_elems = 0;
  int32_t* upd_tid = nullptr;   // [packed_elems] index into specs (-1: no source: zero fill / derived table)
  int32_t* upd_ofs = nullptr;   // [packed_elems]
  const float** upd_ptrs = nullptr;  // [specs.size()] device table of the caller's tensors for one update
  bool upd_ready = false;
  DevParams P;
  EdgeTables tabs;
  // embedding in the radial basis (tn_embed_rb.hip): weight fragment images (null: shape not covered / switched off),
  // species count of the last dynamic graph build (0: unknown -> the per-pair tables run), atom threshold
  uint16_t* rb_img = nullptr;
  const uint16_t *rb_fwd = nullptr, *rb_rev = nullptr;
  int last_nt = 0;
  int64_t rb_min_atoms = 1024;
  int pair_bf16 = 0;          // option "pair_rows_bf16" (Equivariant Transformer): per-pair filter rows stored as bf16
  bool tabs_pending = false;  // parameters changed since the radial tables were built: rebuilt by the next call that uses them
  int64_t tab_min_pairs = 1;  // developer / test switch (option "edge_table_min_pairs"): fewer pairs take the value + tangent GEMMs
  bool finalized = false;
  std::string err;
  // last-call bookkeeping for tmdnet_debug_tensor
  FwdBuffers last{};
  int64_t lastN = 0, lastP = 0;
  bool has_last = false;
  const float* tn2_last_chg = nullptr;  // TensorNet2: [N, (L + 1) q_dim] charge channels of the last call (debug tensor "charges")
};


int fail(tmdnet_model* m, int code, const std::string& msg);
#define HIP_TRY(m, expr)                                                                    \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess) return fail(m, TMDNET_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

// host-side packer: appends a tensor (optionally transposed) to the staging buffer, 64-float aligned
struct Packer {
  std::vector<float> buf;
  size_t add(const std::vector<float>& v) {
    size_t off = (buf.size() + 63) & ~size_t(63);
    buf.resize(off + v.size());
    std::memcpy(buf.data() + off, v.data(), v.size() * sizeof(float));
    return off;
  }
  size_t add_T(const std::vector<float>& v, int64_t rows, int64_t cols) {
    std::vector<float> t(v.size());
    for (int64_t r = 0; r < rows; ++r)
      for (int64_t c = 0; c < cols; ++c) t[c * rows + r] = v[r * cols + c];
    return add(t);
  }
};

// ---- launch-time context (thread local: a handle is not thread-safe, different handles on different threads are)
extern thread_local tmdnet_model* g_cur;
extern thread_local int g_gemm_cat;
extern thread_local const int* g_mdev;  // device-side row count of pair-row GEMMs (see GemmArgs::m_dev)
extern thread_local int g_madd;
struct CurScope {  // g_cur is valid exactly while an entry point is enqueueing for that model (also on error returns)
  explicit CurScope(tmdnet_model* m) { g_cur = m; }
  ~CurScope() { g_cur = nullptr; }
};
struct ProfScope {
  int idx = -1;
  hipStream_t s;
  ProfScope(hipStream_t s_, int cat, double flops, double bytes, const char* label = nullptr) : s(s_) {
    tmdnet_model* m = g_cur;
    if (!m || !m->prof.on || !((m->prof.mask >> cat) & 1u)) return;
    ProfRec r{cat, m->prof.get(), m->prof.get(), flops, bytes, {0}};
    if (label) std::snprintf(r.label, sizeof(r.label), "%s", label);
    (void)hipEventRecord(r.a, s);
    idx = (int)m->prof.recs.size();
    m->prof.recs.push_back(r);
  }
  ~ProfScope() {
    if (idx >= 0) (void)hipEventRecord(g_cur->prof.recs[idx].b, s);
  }
};
#define KR(cat, bytes, call)                             \
  do {                                                   \
    ProfScope ps_(s, cat, 0.0, (double)(bytes), #call);  \
    call;                                                \
  } while (0)

// C = epilogue(A W^T + bias) through the path's MFMA GEMMs (split-bf16 kernels when W has a registered image)
void gemm(hipStream_t s, const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, float* C, int64_t ldc, int M,
          int N, int K, int flags = 0, float* pre = nullptr, int64_t ldpre = 0, const float* aux = nullptr, int64_t ldaux = 0,
          const float* rowscale = nullptr);
// value + d/dd tangent of a pair-row layer (kind: tn_gemm_dual.hip)
void gemm_dual(hipStream_t s, int kind, const float* A, const float* A2, int64_t lda, const float* W, const float* bias, float* C,
               float* C2, int64_t ldc, int M, int N, int K, const float* rs = nullptr, const float* rs2 = nullptr,
               const uint16_t* Wsb = nullptr);
Graph carve_graph(void* ws, int64_t N, int64_t B, int64_t ecap, size_t* total);
// restores the per-workspace record of the last graph build (is_cell / has_z / species count) into the handle
void recall_graph(tmdnet_model* m, const void* graph_ws);
// (name, numel) of every entry of the flat parameter-gradient buffer (tmdnet_param_grad_entry); entries are 64-float aligned
const std::vector<std::pair<std::string, int64_t>>& param_grad_layout(tmdnet_model* m);
// species count rounded up to 4 / 8 when this call takes the embedding in the radial basis (tn_embed_rb.hip), else 0
int rb_ntp(const tmdnet_model* m, int64_t n_atoms, int64_t n_pairs);
// the three weight matrices act on the channel axis of the 1 + 3 + 5 irreducible components: one grouped launch, 9 groups
void tensor_linear(hipStream_t s, const float* A, const float* const W3[3], float* C, int N, int F, int flags = 0, float* pre = nullptr,
                   const float* gates = nullptr);
// fused form of the 9-component tensor linears (tn_tlin9.hip) and whether the three weights have their fragment-major images
namespace tn { struct Tl9Args; }
bool tlin9_images(const float* const W3[3]);
void tlin9(hipStream_t s, int pro, int epi, const float* const W3[3], tn::Tl9Args a, double tensors, const char* what);
// radial tables (tn_api.hip): fp64 build + midpoint verification; `out.ok` says whether they may be used
// builds the radial tables if a parameter upload left them pending (blocking, NULL stream); refuses while `s` is being captured
int ensure_radial_tables(tmdnet_model* m, hipStream_t s);
int et_build_tables(tmdnet_model* m);  // tn_et_api.hip
std::vector<std::pair<std::string, int64_t>> et_train_layout(const tmdnet_model* m);
void et_carve_train(void* ws, const tmdnet_model* m, int64_t N, int64_t P, TrainCtx* tc, size_t* total);
std::vector<std::pair<std::string, int64_t>> tn2_train_layout(const tmdnet_model* m);  // tn_tn2_api.hip
void tn2_carve_train(void* ws, const tmdnet_model* m, int64_t N, int64_t P, int64_t E, TrainCtx* tc, size_t* total);
int build_radial_tables(tmdnet_model* m, EdgeTables& out, const std::vector<TableSpec>& specs, const float* means, const float* betas,
                        int K, double lo, double up);
void free_radial_tables(EdgeTables& t);

// TensorNet2 (tn_tn2_api.hip)
int tn2_forward_workspace_bytes(const tmdnet_model* m, int64_t n_atoms, int64_t n_mol, int64_t n_pairs, int64_t n_edges, int32_t want_forces,
                                size_t* bytes);
int tn2_energy_forces(tmdnet_model* m, hipStream_t s, const Graph& g, void* ws, size_t ws_bytes, int64_t n_atoms, int64_t n_mol,
                      int64_t n_pairs, const int64_t* z, const int64_t* batch, const float* q, int32_t want_forces, float* energy,
                      float* forces);

// Equivariant Transformer (tn_et_api.hip)
int et_create(tmdnet_model* m, const tmdnet_et_hparams* hp);
void et_destroy(tmdnet_model* m);
int et_finalize(tmdnet_model* m);
int et_forward_workspace_bytes(const tmdnet_model* m, int64_t n_atoms, int64_t n_mol, int64_t n_pairs, int32_t want_forces, size_t* bytes);
int et_energy_forces(tmdnet_model* m, hipStream_t s, const Graph& g, void* ws, size_t ws_bytes, int64_t n_atoms, int64_t n_mol,
                     int64_t n_pairs, const int64_t* z, const int64_t* batch, int32_t want_forces, float* energy, float* forces);
int et_debug_tensor(tmdnet_model* m, hipStream_t s, const char* name, float* out, int64_t numel);
// analytic second-order pass of force matching (the Equivariant Transformer's half of tmdnet_force_param_grads)
int et_force_param_workspace_bytes(const tmdnet_model* m, int64_t n_atoms, int64_t n_pairs, size_t* bytes);
int et_hvp_debug_tensor(tmdnet_model* m, hipStream_t s, const char* name, float* out, int64_t numel);
int et_force_param_grads(tmdnet_model* m, hipStream_t s, const Graph& g, void* ws, size_t ws_bytes, int64_t n_atoms, int64_t n_mol,
                         int64_t n_pairs, const int64_t* z, const int64_t* batch, const float* v, const float* ge, float* grads, float* hv);
