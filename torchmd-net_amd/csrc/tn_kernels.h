// Launch wrappers of the non-GEMM HIP kernels of the TensorNet energy+force path (gfx950).
// Every wrapper enqueues on `s` and returns immediately.  Layouts: see tn_common.h / DESIGN.md.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tn {

struct Graph {  // device pointers into the graph workspace
  int* mstart;   // [B]   first atom of each molecule (valid when batch is sorted)
  int* mend;     // [B]
  int* nlow;     // [N]   number of neighbours j < i
  int* ntot;     // [N]   row length (lower + self + upper)
  int* rowptr;   // [N+1]
  int* pairptr;  // [N+1]
  int* col;      // [Ecap] neighbour index, ascending within a row
  int* epair;    // [Ecap] pair id of the edge (P = self pair)
  float* esign;  // [Ecap] +1: row atom is the pair's i, -1: it is the pair's j, 0: self edge
  int* pair_i;   // [Pcap]
  int* pair_j;   // [Pcap]
  float* pd;     // [Pcap+1] distance (self pair: 0)
  float* pdelta; // [Pcap,3] pos_i - pos_j (+ minimum image)
  float* prhat;  // [Pcap,3] unit vector
  int* counts;   // [8]: 0 = P, 1 = E, 2 = overflow, 3 = batch unsorted, 4 = z out of range, 5 = batch out of range, 6 = species present
  int* tix;      // [N]  index of the atom's atomic number among the species present in the batch (k_type_map, tn_embed_rb.hip)
  int* tz;       // [64] atomic number of each species index
  int64_t ecap, pcap;
  // ---- cell-list path (tn_cell.hip): atoms renumbered in cell order
  int* perm;             // [N] internal index -> caller's atom index
  int* iota;             // [N]
  int* cell_key;         // [N] cell id per caller atom
  int* cell_key_sorted;  // [N] cell id per internal atom
  int* cell_start;       // [ncells+1]
  float* pos_s;          // [N,3] positions in internal order
  int64_t* z_s;          // [N] atomic numbers in internal order
  float* boxd;           // [9] device copy of the box (rows a, b, c); fictitious box of a non-periodic cell search
  int* cgrid;            // [4] cells per axis + their product, computed on the device from the current box
  int* bat_s;            // [N] molecule index per internal atom (several molecules in one box)
  int64_t* bat_c;        // [N] the same as int64 (validated): the `batch` the model kernels see when the cell list renumbered several molecules
  int64_t* z_c;          // [N] validated (clamped to [0, max_z)) atomic numbers in internal order (tmdnet_build_graph with z)
  void* sort_tmp;
  size_t sort_tmp_bytes;
  int ncx, ncy, ncz;     // explicit grid (tmdnet_set_cell_grid); 0 = from the box
  int use_cell;          // 0: brute force, 1: cell list / one molecule, 2: cell list / several molecules
  const float* ghost_w;  // halo exchange (cell list): the caller's atom weights; pairs of two weight-0 atoms (ghosts) are left out
  int max_z;
  int small_mols;        // host hint: on average <= 96 atoms per molecule, i.e. the column window of a 64-row tile fits LDS
};

struct RadialParams {
  const float* means;
  const float* betas;
  int K;
  float lo, up;
};

// ---- graph construction (reference models/utils.py:233-313, warp_kernels/neighbors_brute.py:40-205,
//      warp_ops/graph_transform.py:160-179)
void launch_graph_build_phase1(const Graph& g, const float* pos, const int64_t* batch, const float* box, int box_mode, int N,
                               int B, float lo, float up, bool loop, hipStream_t s);
void launch_graph_build_phase2(const Graph& g, const float* pos, const int64_t* batch, const float* box, int box_mode, int N,
                               int B, float lo, float up, bool loop, hipStream_t s);
// COO export in the reference operator's format (warp_ops/neighbors.py:34-148)
void launch_export_pairs(const Graph& g, int N, bool include_transpose, bool loop, int64_t max_pairs, const int* perm,
                         int64_t* neighbors, float* deltas, float* distances, int* num_pairs, hipStream_t s);

void launch_neighbor_grad(const int64_t* nb, const float* deltas, const float* dist, const float* g_delta, const float* g_dist,
                          int64_t M, int N, float* out, hipStream_t s);

// ---- radial basis + cutoff per pair (reference models/utils.py:402-407, 506-528)
void launch_radial(const Graph& g, int P, RadialParams rp, float* phi, float* dphi, float* C, float* dC, hipStream_t s);

// ---- embedding (reference tensornet.py:543-619, 405-445)
void launch_ztables(const float* emb, const float* WaT, const float* WbT, const float* b2, int Z, int F, float* Utab, float* Vtab,
                    hipStream_t s);
void launch_embed_scatter(const Graph& g, int N, int F, const int64_t* z, const float* Utab, const float* Vtab, const float* Q,
                          const float* C, float* u0, float* s0n, hipStream_t s);
// ---- LayerNorm over rows of length R (torch.nn.LayerNorm semantics, eps 1e-5, biased variance)
void launch_layernorm_fwd(const float* x, const float* w, const float* b, int rows, int R, float* y, float* xhat, float* rstd,
                          hipStream_t s);
void launch_layernorm_bwd(const float* g, const float* xhat, const float* rstd, const float* w, int rows, int R, float* gx,
                          hipStream_t s);
// ---- interaction layer (reference tensornet.py:729-814, 622-679)
void launch_norm_x(const float* X, float* Xh, int N, int F, hipStream_t s);
// mode 0: forward message + O(3)/SO(3) product + normalisation -> Mi, Ch ; mode 1: dst[i] += sum_e w * src[j] (adjoint)
struct PairRowTable;  // tn_interp.h: a layer's radial table for sweeps that evaluate the per-pair rows themselves
void launch_message_dual(const Graph& g, int N, int F, const float* w, const float* w_t, const float* src, const float* src_t,
                         float* out, float* out_t, bool accumulate, hipStream_t s);  // value + tangent sweep (second-order pass)
void launch_message(const Graph& g, int N, int F, const float* w, const float* src, const float* q, const int64_t* batch, int o3,
                    float* Mi, float* Ch, hipStream_t s, const PairRowTable* rt = nullptr, int row0 = 0, int nrows = -1);  // nrows >= 0: rows [row0, row0 + nrows) only
void launch_message_adjoint(const Graph& g, int N, int F, const float* w, const float* gMi, float* gPn, hipStream_t s);
// adjoint sweep + the layer's per-pair distance gradient in one pass (replaces launch_message_adjoint + launch_pair_gd when
// message_adjoint_gd_ok): partial sums go to slots[wave][2 * pair + direction], summed by launch_geom_gd
bool message_adjoint_gd_ok(int N, int F);
int message_adjoint_gd_waves(const Graph& g, int N, int F, bool rows_from_table = false);  // slot arrays the sweep writes per layer
bool message_adjoint_pair_ok(const Graph& g, int N, int F);  // tn_message_pair.hip: the reverse sweep in the tile layout
void launch_message_adjoint_pair(const Graph& g, int N, int F, const float* w, const float* dw, const float* gMi, const float* Pn,
                                 float* gPn, float* slots, int64_t slot_stride, hipStream_t s);
void launch_message_adjoint_gd(const Graph& g, int N, int F, const float* w, const float* dw, const float* gMi, const float* Pn,
                               float* gPn, float* slots, int64_t slot_stride, hipStream_t s, const PairRowTable* rt = nullptr, int own0 = 0,
                               int own1 = -1, bool narrow = true);  // own1 >= 0: rows outside [own0, own1) are ghosts (no gPn; narrow: their pairs
                                                                    // with owned atoms are searched for in the row - the graph holds ghost-ghost pairs)
// next = 0: plain; 1: nxt = X_hat of the new X (next layer's k_norm_x); 2: nxt = readout invariants of the new X
void launch_layer_update(const float* Xh, const float* D, const float* q, const int64_t* batch, int N, int F, float* Xn, int next,
                         float* nxt, hipStream_t s);
// ---- readout (reference tensornet.py:384-398, output_modules.py:43-117, model.py:591-607)
void launch_readout_feat(const float* X, int N, int F, float* feat, hipStream_t s);
void launch_head_energy(const float* ao, const float* O2, const float* bO2, int N, int H, float std, const float* atomref,
                        const int64_t* z, float* ea, hipStream_t s, float* g_ao = nullptr,  // g_ao: also d e / d ao (launch_head_bwd)
                        const float* atom_w = nullptr, const int* perm = nullptr);  // weights of the energy sum, caller's order
void launch_mol_sum(const Graph& g, const float* ea, const int64_t* batch, int N, int B, float mean, float* energy, hipStream_t s);

// ---- reverse pass
void launch_head_bwd(const float* ao, const float* O2, int N, int H, float std, float* g_ao, hipStream_t s);
void launch_readout_bwd(const float* X, const float* g_feat, int N, int F, float* G, hipStream_t s);
void launch_update_bwd(const float* G, const float* D, const float* q, const int64_t* batch, int N, int F, float* gD,
                       hipStream_t s);
void launch_message_bwd_node(const float* gCh, const float* Pn, const float* Mi, const float* q, const int64_t* batch, int o3, int N,
                             int F, float* gMi, float* gPn, hipStream_t s);
void launch_norm_bwd(const float* X, const float* gXh_lin, int N, int F, float* G, hipStream_t s);
// the same followed in-register by the update adjoint of the layer below / by the embedding gate adjoint (G then not stored)
void launch_norm_bwd_update_bwd(const float* X, const float* gXh_lin, int N, int F, float* G, const float* Dn, const float* q,
                                const int64_t* batch, float* gD, hipStream_t s);
void launch_norm_bwd_gate_bwd(const float* X, const float* gXh_lin, int N, int F, float* G, const float* UX, const float* gates,
                              const float* a2, float* gUX, float* g_a2, hipStream_t s);
void launch_lnbwd_readout_bwd(const float* g, const float* xhat, const float* rstd, const float* w, int N, int F, const float* X,
                              float* G, hipStream_t s);
void launch_head_mol_sum(const Graph& g, const float* ao, const float* O2, const float* bO2, int N, int B, int H, float std,
                         const float* atomref, const int64_t* z, const int64_t* batch, float mean, float* energy, hipStream_t s,
                         float* g_ao = nullptr, const float* atom_w = nullptr, const int* perm = nullptr);
void launch_embed_gate_bwd(const float* G, const float* UX, const float* gates, const float* a2, int N, int F, float* gUX, float* g_a2,
                           hipStream_t s);
// out[0..2] = first row, last row, number of rows n with aw[perm ? perm[n] : n] != 0 (halo exchange: are the owned atoms one
// contiguous range of the engine's order?)
void launch_owned_range(const float* aw, const int* perm, int N, int* out, hipStream_t s);
void launch_embed_bwd_atom(const float* g_u0_lin, const float* u0, const float* g_s0n, int N, int F, float* gA, hipStream_t s);
void launch_force_gather(const Graph& g, int N, const float* g_delta, const int* perm, float* forces, hipStream_t s);
void launch_fill(float* p, float v, int64_t n, hipStream_t s);
void launch_kappa(const float* q, const int64_t* batch, int N, int B, float* kap, hipStream_t s);

// ---- wave-per-atom neighbour kernels (tn_graph_wave.hip)
void launch_nbr_count_wave(const Graph& g, const float* pos, const int64_t* batch, const float* box, int box_mode, int N, int B,
                           float lo, float up, bool loop, hipStream_t s);
void launch_nbr_fill_link_wave(const Graph& g, const float* pos, const int64_t* batch, const float* box, int box_mode, int N, int B,
                               float lo, float up, bool loop, hipStream_t s);

void launch_nbr_link_wave(const Graph& g, int N, hipStream_t s);
// small systems: reset + molecule ranges + count + scan + fill + link + atomic-number check in one launch (one block)
bool graph_small_ok(int N);
void launch_graph_small(const Graph& g, const float* pos, const int64_t* batch, const float* box, int box_mode, int N, int B, float lo,
                        float up, bool loop, const int64_t* z, int max_z, hipStream_t s);
void launch_scan_counts(const Graph& g, int N, hipStream_t s);

// ---- O(N) cell list for one periodic orthorhombic system (tn_cell.hip)
size_t cell_sort_temp_bytes(int64_t n);
void launch_cell_phase1(const Graph& g, const float* pos, const int64_t* batch, const float* box, int N, float lo, float up, bool loop,
                        hipStream_t s, int B = 0);  // B > 0: validate the molecule indices (several molecules in one box)
void launch_cell_phase2(const Graph& g, int N, float lo, float up, bool loop, hipStream_t s);
void launch_permute_z(const Graph& g, const int64_t* z, int N, hipStream_t s);
// z_c[i] = clamp(z[perm ? perm[i] : i], 0, max_z - 1); counts[4] = 1 when any value was out of range (reference: nn.Embedding raises)
void launch_prepare_z(const Graph& g, const int64_t* z, const int* perm, int N, int max_z, hipStream_t s);

// ---- per-pair reverse kernels of the forward-tangent formulation (tn_pairgrad.hip)
void launch_pair_gd(const Graph& g, int Pcap, int F, const float* gMi, const float* Pn, const float* dw, float* gd, hipStream_t s);
void launch_embed_pair_gd(const Graph& g, int Pcap, int F, const int64_t* z, const float* Utab, const float* Vtab, const float* Q,
                          const float* dQ, const float* C, const float* dC, const float* gA, float* gd, float* g_rhat, hipStream_t s,
                          float* g_delta = nullptr, const float* slots = nullptr, int n_slots = 0, int64_t slot_stride = 0);
// g_delta != null: the embedding is the only writer of g_d besides the sweeps' slots, so the kernel finishes the pair
// (launch_geom_gd's arithmetic) itself: gd / g_rhat are not touched and need neither a zero fill nor a second launch
// g_d[p] = gd[p] + sum of the n_slots slot arrays (each [2 * slot_stride/2]: pair, direction) when slots != null
void launch_geom_gd(const Graph& g, int Pcap, const float* gd, const float* g_rhat, float* g_delta, hipStream_t s,
                    const float* slots = nullptr, int n_slots = 0, int64_t slot_stride = 0);

// ---- radial tables of the per-pair functions (tn_edge_table.hip)
void launch_radial_f64(int rows, double lo, double up, int T, bool mid, const float* means, const float* betas, int K, double* dist,
                       double* phi, double* dphi, double* C, double* dC, hipStream_t s);
void launch_dense_f64(const double* A, const double* A2, int lda, const float* W, const float* bias, int M, int N, int K, int kind,
                      const double* rs, const double* rs2, double* C, double* C2, int ldc, hipStream_t s);
void launch_table_pack(const double* f, const double* sl, int T, int R, double h, float* tab, hipStream_t s);
void launch_pair_buckets(const Graph& g, int Pcap, float lo, float up, int T, float* C, float* dC, int* hist, unsigned* keys_s,
                         int* vals_s, hipStream_t s);
// C / dC: the step's cutoff arrays; written by the first call of a step when the pair list is small enough for the
// one-launch path (edge_interp_direct: launch_pair_buckets is a no-op then), ignored otherwise
bool edge_interp_direct(int Pcap);
void launch_edge_interp(const Graph& g, int Pcap, float lo, float up, int T, int R, int ntab, const float* const* tabs,
                        float* const* outs, float* const* douts, const unsigned* keys_s, const int* vals_s, hipStream_t s, float* C,
                        float* dC, int out_bf16 = 0);  // out_bf16: rows stored as bf16 (the ET's reduced-precision pair storage)
// ---- embedding in the radial basis (tn_embed_rb.hip; reverse pair kernel in tn_pairgrad.hip)
void launch_type_map(const Graph& g, int N, int max_z, hipStream_t s);
bool embed_rb_shape_ok(int F, int K);
int embed_rb_ntp(int nt);  // species count rounded up to 4 / 8; 0: more than 8 species (the per-pair tables run)
int64_t embed_rb_moment_elems(int64_t N, int ntp, int K);
int64_t embed_rb_gmoment_elems(int64_t N, int ntp, int K);
size_t embed_rb_image_elems(int F, int K, bool reverse);
void launch_embed_rb_images(const float* Wdp_dev, const float* bdp_dev, int F, int K, uint16_t* fwd_dev, uint16_t* rev_dev, hipStream_t s);
void launch_pair_scalars(const Graph& g, int Pcap, float lo, float up, float* ps, hipStream_t s);  // [P + 1][8]: C, C', C0, C0', u
void launch_embed_moments(const Graph& g, int N, RadialParams rp, int ntp, const float* ps, float* m, hipStream_t s);
void launch_embed_combine(const Graph& g, int N, int F, int K, int ntp, const int64_t* z, const float* Utab, const float* Vtab,
                          const uint16_t* Bimg, const float* m, float* u0, float* s0n, hipStream_t s);
void launch_embed_gm(const Graph& g, int N, int F, int K, int ntp, const int64_t* z, const float* Utab, const float* Vtab,
                     const uint16_t* B2img, const float* bdp, const float* gA, float* gm, hipStream_t s);
void launch_embed_pair_rb(const Graph& g, int Pcap, int N, RadialParams rp, int ntp, const float* ps, const float* gm, float* gd,
                          float* g_rhat, hipStream_t s, float* g_delta, const float* slots, int n_slots, int64_t slot_stride);
void launch_interp_list(const float* tab, const double* dist, int M, int R, int T, float lo, float up, float* out, float* dout,
                        hipStream_t s);

// ---- LDS-staged forward sweep (tn_message_pair.hip): 8 lanes per row x 16 bytes per lane, the tile's source window in LDS
bool message_pair_ok(int N, int F);
void launch_message_pair(const Graph& g, int N, int F, const float* w, const float* src, const float* q, const int64_t* batch,
                         int o3, float* Mi, float* Ch, hipStream_t s);

// 16-byte-per-lane form of the per-pair kernels (tn_pairgrad.hip)
bool gather_v4_ok(int F);

}  // namespace tn
