/* tmdnet_amd.h -- C ABI of the MI355X-native TensorNet energy+force engine (libtmdnet_amd.so).
 *
 * Drop-in boundary.  The reference has no C FFI of its own on this path:
 * its "native" seam is a set of torch custom ops written in NVIDIA Warp
 * (torchmdnet/extensions/ops.py:14-106, the files under torchmdnet/extensions/warp_ops/) below the Python
 * surface create_model / load_model / TorchMD_Net.forward (torchmdnet/models/model.py:21,208,530).
 * This header is what a binding for that path binds instead: plain pointers and sizes, no torch
 * types.  The host side in torchmd-net_amd/torchmdnet_amd/ calls it through ctypes (see
 * INTEGRATION.md for the stub a reference maintainer would add).
 *
 * Conventions
 *   - every `const float*` / `const int64_t*` data pointer is a DEVICE pointer (HIP, gfx950) unless
 *     the parameter name ends in `_host`; all tensors are dense row-major fp32 / int64.
 *   - `stream` is a hipStream_t passed as void* (0 = default stream).  Calls enqueue work and return;
 *     only tmdnet_build_graph synchronises the stream (it reads back three integers, the analogue
 *     of the reference's `resize_to_fit` host sync, torchmdnet/models/utils.py:303-307).
 *   - no hidden device allocation happens inside build_graph / energy_forces: the caller owns both
 *     workspaces (size them with the *_workspace_bytes queries).
 *   - a handle is re-entrant across handles but not thread-safe on one handle.
 *   - return value: 0 on success, a TMDNET_ERR_* code otherwise; tmdnet_last_error() gives text.
 */
#ifndef TMDNET_AMD_H
#define TMDNET_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TMDNET_OK 0
#define TMDNET_ERR_INVALID 1   /* bad argument / unknown parameter / wrong size; also: atomic number or molecule
                                  index out of range (the reference's nn.Embedding / scatter raise there) */
#define TMDNET_ERR_HIP 2       /* a HIP runtime call failed */
#define TMDNET_ERR_OVERFLOW 3  /* more neighbour pairs than max_num_neighbors * n_atoms:
                                  the reference raises RuntimeError here (models/utils.py:297-300) */
#define TMDNET_ERR_WORKSPACE 4 /* caller workspace too small */
#define TMDNET_ERR_STATE 5     /* call order violated (e.g. parameters not finalised) */

typedef struct tmdnet_model tmdnet_model; /* opaque */

/* Hyper-parameters of TensorNet + Scalar head: the subset of the reference's create_model argument
 * dict that the path depends on (torchmdnet/models/model.py:35-60,96-105,134-152). */
typedef struct tmdnet_hparams {
  int32_t hidden_channels;   /* embedding_dimension (F) */
  int32_t num_layers;        /* interaction layers (L) */
  int32_t num_rbf;           /* ExpNormal radial basis size (K) */
  int32_t max_z;             /* rows of the atom-type embedding */
  int32_t max_num_neighbors; /* pair capacity = max_num_neighbors * n_atoms (tensornet.py:281-290) */
  int32_t group_o3;          /* 1: O(3) (Y M + M Y), 0: SO(3) (2 Y M)   (tensornet.py:788-793) */
  int32_t head_hidden;       /* Scalar head hidden width (F/2, output_modules.py:96-103) */
  int32_t has_atomref;       /* 1: an "atomref" table [max_z] is added per atom (priors/atomref.py:93-96) */
  float cutoff_lower;
  float cutoff_upper;
} tmdnet_hparams;

/* Hyper-parameters of the Equivariant Transformer + EquivariantScalar head (reference
 * torchmdnet/models/torchmd_et.py:84-103, model.py:62-81,134-135).  Activations are SiLU (activation, attn_activation). */
typedef struct tmdnet_et_hparams {
  int32_t hidden_channels;    /* embedding_dimension (F) */
  int32_t num_layers;         /* attention layers */
  int32_t num_rbf;            /* ExpNormal radial basis size (K) */
  int32_t max_z;
  int32_t max_num_neighbors;  /* pair capacity = max_num_neighbors * n_atoms */
  int32_t num_heads;          /* F / num_heads must be a power of two <= 64 */
  int32_t neighbor_embedding; /* 1: NeighborEmbedding (models/utils.py:45-117) */
  int32_t vector_cutoff;      /* 1: cutoff on the values, 0: on the attention weights (torchmd_et.py:395-401) */
  int32_t distance_influence; /* bit 0: keys (dk_proj), bit 1: values (dv_proj) */
  int32_t has_atomref;
  float cutoff_lower;
  float cutoff_upper;
} tmdnet_et_hparams;

/* Hyper-parameters of TensorNet2 + ScalarPlusWeightedCoulomb (AceFF-2.0; reference
 * torchmdnet/models/tensornet2.py:160-330, output_modules.py:344-441, model.py:106-152).  TensorNet with per-atom charge
 * channels: a ChargePredict head after the embedding and after every layer (invariants -> LayerNorm -> MLP -> charge
 * equilibration to the molecule's total charge `q`), the next layer's edge MLP takes [phi(d), c_i, c_j], and the head adds
 * the damped pair Coulomb energy of all (num_layers + 1) * q_dim charge channels, weighted by `output_model.qweights`. */
typedef struct tmdnet_tn2_hparams {
  int32_t hidden_channels, num_layers, num_rbf, max_z, max_num_neighbors, group_o3, head_hidden, has_atomref;
  int32_t q_dim;                 /* charge channels per ChargePredict head, (num_layers + 1) * q_dim <= 64 */
  float cutoff_lower, cutoff_upper;
  float coulomb_cutoff;          /* <= 0: all pairs of a molecule (no PBC); > 0: reaction field inside this cutoff */
  float coulomb_epsilon_solvent; /* reaction-field dielectric constant (reference default 78.3) */
} tmdnet_tn2_hparams;

/* ---- lifecycle ------------------------------------------------------------------------------- */
int tmdnet_create(const tmdnet_hparams* hp, tmdnet_model** out);
/* TensorNet2 handle: parameters by the reference's state-dict keys, every other entry point shared.  tmdnet_build_graph must
 * be given the positions / box it should use for the Coulomb sum as well (it keeps the pointers until tmdnet_energy_forces). */
int tmdnet_create_tn2(const tmdnet_tn2_hparams* hp, tmdnet_model** out);
/* Equivariant Transformer handle: every other entry point (parameters by state-dict key, graph, energy_forces with
 * q = NULL, workspaces, profiling) is shared with the TensorNet handle. */
int tmdnet_create_et(const tmdnet_et_hparams* hp, tmdnet_model** out);
int tmdnet_destroy(tmdnet_model* m);
const char* tmdnet_last_error(const tmdnet_model* m);
const char* tmdnet_version(void);
/* ABI revision of this header: bumped whenever an exported signature or struct layout changes (3: `z` in
 * tmdnet_build_graph[_static], `strategy` in tmdnet_neighbor_pairs).  A binding compares its compile-time
 * TMDNET_ABI_VERSION with the loaded library's tmdnet_abi_version() before its first call. */
#define TMDNET_ABI_VERSION 9
int tmdnet_abi_version(void);

/* Parameters are addressed by the reference's state-dict keys without the "model." prefix
 * (SURVEY.md Appendix A), e.g. "representation_model.layers.0.linears_scalar.2.weight", plus
 * "mean", "std" and "atomref".  `data_host` is HOST memory, row-major, `numel` floats; it is copied.
 * tmdnet_finalize_params checks that every tensor is present with the expected size, builds the
 * transposed copies used by the reverse pass and uploads one packed device buffer.  It marks the radial
 * tables stale: the first tmdnet_energy_forces after it rebuilds them (blocking, NULL stream, tens of ms)
 * and returns TMDNET_ERR_STATE if its stream is being captured at that moment. */
int tmdnet_set_param(tmdnet_model* m, const char* name, const float* data_host, int64_t numel);
int tmdnet_finalize_params(tmdnet_model* m);
/* Device-side parameter update (training loops; TensorNet and TensorNet2): `dev_ptrs[c]` is the fp32, contiguous DEVICE copy of
 * state-dict tensor `names[c]` (any subset; "mean" / "std" excluded - they are host-read kernel arguments).  Needs one
 * tmdnet_set_param + tmdnet_finalize_params round first (shapes, layout, host-built images).  Enqueues on `stream`: one gather
 * kernel rewrites the packed parameter buffer from the caller's tensors (the mapping is learnt once by running the packing on
 * index tags), then the split-bf16 images and the per-species tables are rebuilt by kernels; nothing crosses PCIe except a
 * pointer table, no synchronisation.  The caller's tensors must stay valid until the stream reaches this point.  Afterwards
 * the radial tables are stale (rebuilt by the first call that uses them) and the radial-basis embedding stays off until the
 * next full tmdnet_finalize_params (its weight images are made on the host).  Evaluations that used the handle on OTHER
 * streams must have finished (same rule as tmdnet_finalize_params, which additionally blocks the host). */
int tmdnet_update_params_device(tmdnet_model* m, void* stream, int32_t count, const char* const* names, const float* const* dev_ptrs);
/* number of parameter tensors the model expects; name of the idx-th one and its element count */
int tmdnet_num_params(const tmdnet_model* m);
const char* tmdnet_param_name(const tmdnet_model* m, int idx, int64_t* numel);

/* ---- radial tables of the per-pair functions --------------------------------------------------------------------
 * Q_c(d) = P_c phi(d) + b_c and every layer's edge MLP w^l(d) (reference tensornet.py:558-560, 738-743) depend on the pair
 * distance only.  tmdnet_finalize_params evaluates them and their d/dd once on a uniform grid in double precision,
 * stores fp32 rows, and verifies the step's cubic Hermite interpolation against the fp64 evaluation at every interval
 * midpoint (bounds: 5e-7 of the table's largest value, 2e-6 of its largest slope; the grid is refined 8192 -> 65536
 * intervals until they hold, else the tables stay off) and tmdnet_energy_forces then interpolates per pair instead of running the pair-row
 * GEMMs, when the system has at least `edge_table_min_pairs` pairs (default 1024: single systems from about 100 atoms on).  TMDNET_EDGE_TABLE=0 in the
 * environment disables the tables (direct GEMMs every step).
 * Options: "edge_table_min_pairs"; "pair_rows_bf16" (Equivariant Transformer handle only; 1: the per-pair distance-filter
 * rows silu(dk_proj phi) | silu(dv_proj phi) and their d/dd - reference torchmd_et.py:375-415 - are written by the table
 * interpolation as bf16 and widened when the attention sweeps load them; products and sums stay fp32; default 0).  Info: "edge_table_T" (0 = off), "edge_table_err_value", "edge_table_err_slope"
 * (measured at the midpoints), "edge_table_min_pairs".
 * "recompute_pair_rows" (TensorNet handle; default 0): 1 = the message sweeps interpolate a layer's per-pair row from its table
 * themselves (12 table loads per edge instead of 3, or 6 in the reverse sweep) and the rows w^l, d w^l / dd, the distance
 * projections' rows and the direct evaluation's activations get NO workspace: tmdnet_forward_workspace_bytes drops from
 * ~12.6 KB to ~0.1 KB per pair, which is what lets a 10^6-atom periodic box fit one 288 GB device.  Same arithmetic as the stored
 * rows (one definition, csrc/tn_interp.h): energies and forces are bit-identical.  Needs the tables (verified, not switched
 * off); tmdnet_energy_forces reports TMDNET_ERR_STATE otherwise.  tmdnet_forward_workspace_bytes plans for the embedding form
 * of the graph this handle built LAST (query it after tmdnet_build_graph of the same system, as the Python host does); a plan made
 * for another graph is refused with TMDNET_ERR_WORKSPACE, never overrun. */
int tmdnet_set_option(tmdnet_model* m, const char* name, double value);
int tmdnet_get_info(const tmdnet_model* m, const char* name, double* value);

/* ---- phase A: neighbour graph ------------------------------------------------------------------
 * Replaces OptimizedDistance.forward + get_neighbor_pairs_kernel + graph_transform
 * (torchmdnet/models/utils.py:233-313, extensions/ops.py:14, warp_ops/graph_transform.py:160-179).
 * Brute-force pair search restricted to each molecule, self loops included, both directions,
 * optional triclinic minimum image (box_mode 1: one [3,3] box, 2: one box per molecule [B,3,3]).
 * Produces a deterministic pair list + symmetric CSR inside `graph_ws`.
 * counts_host[0] = number of undirected pairs P, [1] = number of directed edges incl. self loops E,
 * [2] = overflow flag, [3] = 1 if `batch` was not sorted (slow path), [4] = 1 if an atomic number was outside
 * [0, max_z), [5] = 1 if a molecule index was outside [0, n_mol); [6..7] reserved.
 * `z` (int64 [n_atoms], may be NULL) is validated here, in the same read-back: out-of-range values return
 * TMDNET_ERR_INVALID before any table is indexed with them (the reference's nn.Embedding raises IndexError,
 * tensornet.py:473); a validated copy in the graph's internal atom order is kept in `graph_ws` and used by
 * tmdnet_energy_forces.  With z = NULL nothing is checked and tmdnet_energy_forces uses its own `z` argument as is.
 * Returns TMDNET_ERR_OVERFLOW when E > max_num_neighbors * n_atoms. */
int tmdnet_graph_workspace_bytes(const tmdnet_model* m, int64_t n_atoms, int64_t n_mol, size_t* bytes);
/* O(N) cell list (reference "cell" strategy: extensions/neighbor_utils.py:89-150, warp_kernels/neighbors_cell.py:17-153;
 * the reference handles orthorhombic boxes only; here any box in the reduced lower-triangular form a=(ax,0,0),
 * b=(bx,by,0), c=(cx,cy,cz) is taken, one box for the whole system).
 *   (0,0,0)   : always brute force (default);
 *   negative  : cell list whenever n_mol == 1 and box_mode == 1, with the grid n_axis = floor(w_axis / cutoff_upper)
 *               (w_axis = perpendicular width of the box along that axis) computed ON THE DEVICE from the box of
 *               each call -- no host copy of the box, and a captured HIP graph stays valid when the box changes (NPT);
 *   positive  : the same with an explicit grid (n_x*n_y*n_z <= 8*n_atoms, else brute force).
 * Any grid gives the same pair set as brute force (axes with fewer than 3 cells visit each cell once); atoms are
 * renumbered in cell order internally and forces are returned in the caller's order. */
int tmdnet_set_cell_grid(tmdnet_model* m, int32_t ncx, int32_t ncy, int32_t ncz);

/* Per-atom weights of the energy sum (ABI 5): E_mol = sum_i w_i e_i + mean, and the forces are minus the gradient of THAT sum.
 * `weights_dev` is a device vector of n_atoms floats in the caller's atom order (read by every later tmdnet_energy_forces on this
 * handle until it is reset with NULL; it must stay valid for as long as a captured HIP graph replays the call).  This is what a
 * domain decomposition of one large system needs (parallel.SpatialEvaluator): a rank evaluates its owned atoms (w = 1) inside a
 * halo of ghost copies (w = 0), and the ranks' forces add up to the whole system's.  The reference has no such argument: its
 * multi-GPU story stops at data parallelism over molecules.  TensorNet only (ET / TensorNet2 and the
 * parameter-gradient pass refuse a handle with weights set). */
int tmdnet_set_atom_weights(tmdnet_model* m, const float* weights_dev);

/* Per-layer halo exchange (ABI 8) for a domain decomposition whose halo is ONE cutoff deep (parallel.HaloExchangeEvaluator; the
 * reference has no counterpart).  A rank's local system is its owned atoms (weight 1) plus the ghost
 * copies within one cutoff of its domain (weight 0).  Every per-atom kernel of the step is local to a row, and a neighbour sweep
 * only gathers rows of neighbours, so the result on the owned atoms is exact as long as the ghost rows of the tensor each sweep
 * gathers hold their owners' values.  With a callback set, tmdnet_energy_forces (TensorNet inference, exact pair count) calls
 *     fn(user, stage, rows, n_rows, row_floats, perm, stream)
 * on the host, between enqueueing the kernel that writes `rows` ([n_rows][row_floats] floats on the device, n_rows = n_atoms) and
 * the sweep that gathers it: stage l = P_l of layer l (forward, row_floats = 9 F), 100 + l = the adjoint of layer l's message
 * (reverse, 9 F), 200 = the adjoint of the embedding sum (reverse, 10 F); 2 L + 1 calls per step.  The callback overwrites the
 * ghost rows with their owners' rows, ordered on `stream`.  Row r of `rows` belongs to the caller's atom perm[r] (perm == NULL:
 * to atom r; a device vector, valid during the call).  A non-zero return aborts the step with TMDNET_ERR_STATE.  The forces come
 * back for every local atom; those of the owned atoms are complete (no reduction over ranks), those of the ghosts are not.
 * fn == NULL switches the exchange off.  Not capturable: the callback runs at enqueue time.
 * Set together with the atom weights BEFORE tmdnet_build_graph, the cell list leaves out pairs of two weight-0 atoms (nothing of a
 * ghost's own neighbourhood is used).  If the weight-1 atoms are one contiguous range of the engine's cell order (cells are numbered
 * x-major; a slab along x whose faces are cell faces of tmdnet_set_cell_grid's grid is such a range), the per-atom kernels and the
 * forward sweeps run on that range only: tmdnet_get_info "halo_active_first" / "halo_active_rows" report it. */
typedef int (*tmdnet_halo_exchange_fn)(void* user, int32_t stage, float* rows, int64_t n_rows, int64_t row_floats,
                                       const int32_t* perm, void* stream);
int tmdnet_set_halo_exchange(tmdnet_model* m, tmdnet_halo_exchange_fn fn, void* user);
int tmdnet_build_graph(tmdnet_model* m, void* stream, void* graph_ws, size_t graph_ws_bytes, int64_t n_atoms, int64_t n_mol,
                       const float* pos, const int64_t* batch, const int64_t* z, const float* box, int32_t box_mode,
                       int64_t counts_host[8]);

/* Static (HIP-graph capturable) variant: same kernels, NO read-back and no synchronisation.  The pair count
 * stays in device memory; call tmdnet_energy_forces with n_pairs = -1 and workspaces sized for the pair
 * capacity (tmdnet_forward_workspace_bytes with n_pairs = -1).  On overflow the kernels of both phases skip
 * their work (outputs undefined) and the flag is left in the graph workspace: poll it with
 * tmdnet_graph_counts (synchronises; returns TMDNET_ERR_OVERFLOW) whenever convenient -- the analogue of the
 * reference's torch._assert_async (torchmdnet/models/utils.py:297-300). */
int tmdnet_build_graph_static(tmdnet_model* m, void* stream, void* graph_ws, size_t graph_ws_bytes, int64_t n_atoms,
                              int64_t n_mol, const float* pos, const int64_t* batch, const int64_t* z, const float* box,
                              int32_t box_mode);
/* out-of-range z / batch (flags [4], [5]) return TMDNET_ERR_INVALID here; in static mode the kernels ran on clamped
 * atomic numbers (memory-safe, results meaningless) */
int tmdnet_graph_counts(tmdnet_model* m, void* stream, void* graph_ws, int64_t n_atoms, int64_t n_mol, int64_t counts_host[8]);
/* cell grid of the last build on this workspace (synchronises): grid_host = {n_x, n_y, n_z, 1 if the cell list ran else 0} */
int tmdnet_graph_cell_grid(tmdnet_model* m, void* stream, void* graph_ws, int64_t n_atoms, int64_t n_mol, int64_t grid_host[4]);

/* ---- phase B: energies and forces ---------------------------------------------------------------
 * Replaces TorchMD_Net.forward for TensorNet + Scalar (torchmdnet/models/model.py:530-631):
 * energy[n_mol] = sum over atoms of the per-atom scalar (* std, + atomref[z]) + mean, and
 * forces[n_atoms,3] = -d(sum_m energy[m])/d(pos) from the hand-written reverse pass
 * (want_forces = 0 skips it).  `q` = total charge per molecule [n_mol] or NULL (tensornet.py:341-344).
 * `z` may be NULL when tmdnet_build_graph[_static] received (and validated) it.
 * Must be called after tmdnet_build_graph on the same graph_ws (which holds the pair geometry);
 * n_pairs = counts_host[0] of that call (or -1 after tmdnet_build_graph_static).  Enqueues only: no
 * synchronisation, no allocation.  The handle remembers per graph_ws address what it was last built as (cell list or
 * not, validated z, species count): builds on several workspaces may be interleaved with their evaluations.
 * tmdnet_forward_workspace_bytes does not depend on which build came last (the species-dependent buffers are sized for
 * their largest padding). */
int tmdnet_forward_workspace_bytes(const tmdnet_model* m, int64_t n_atoms, int64_t n_mol, int64_t n_pairs, int64_t n_edges,
                                   int32_t want_forces, size_t* bytes);
int tmdnet_energy_forces(tmdnet_model* m, void* stream, void* graph_ws, void* ws, size_t ws_bytes, int64_t n_atoms,
                         int64_t n_mol, int64_t n_pairs, const int64_t* z, const int64_t* batch, const float* q,
                         int32_t want_forces, float* energy, float* forces);

/* ---- fine-grained operator: the reference's neighbour op --------------------------------------
 * Same outputs as torch.ops.torchmdnet.warp_neighbor_brute_fwd (warp_ops/neighbors.py:34-148):
 * neighbors int64 [2,max_num_pairs] padded with -1, deltas [max_num_pairs,3], distances
 * [max_num_pairs], num_pairs int32[1] (the true count, may exceed max_num_pairs).  Pair order is
 * deterministic: lower pairs (i>j) sorted by (i,j), then their transposes, then self loops.
 * strategy 0 = brute force inside each molecule; 1 = O(N) cell list (tn_cell.hip) over all molecules at once, pairs kept
 * inside a molecule: one periodic box (box_mode 1) or, without a box, a fictitious one around the bounding box of the
 * positions (the reference: models/utils.py:206-212); per-molecule boxes (box_mode 2) always take strategy 0.  Same
 * pair set either way (order differs: cell order).
 * `ws` must hold tmdnet_neighbor_workspace_bytes(n_atoms, n_mol, max_num_pairs). */
int tmdnet_neighbor_workspace_bytes(int64_t n_atoms, int64_t n_mol, int64_t max_num_pairs, size_t* bytes);
int tmdnet_neighbor_pairs(void* stream, void* ws, size_t ws_bytes, int64_t n_atoms, int64_t n_mol, const float* pos,
                          const int64_t* batch, const float* box, int32_t box_mode, float cutoff_lower, float cutoff_upper,
                          int64_t max_num_pairs, int32_t loop, int32_t include_transpose, int32_t strategy, int64_t* neighbors,
                          float* deltas, float* distances, int32_t* num_pairs);

/* Gradient of (deltas, distances) wrt the positions: the reference's neighbor_grad_positions
 * (torchmdnet/extensions/neighbor_utils.py:11-46), registered as the backward of its neighbour ops
 * (warp_ops/neighbors.py:105-148).  neighbors int64 [2, num_entries] (padding -1 is skipped, d = 0 contributes nothing),
 * grad_deltas [num_entries,3] / grad_distances [num_entries] may be NULL (= zero); grad_positions [n_atoms,3] is overwritten. */
int tmdnet_neighbor_grad(void* stream, const int64_t* neighbors, const float* deltas, const float* distances,
                         const float* grad_deltas, const float* grad_distances, int64_t num_entries, int64_t n_atoms,
                         float* grad_positions);

/* The same operator for DOUBLE positions (ABI 9).  The reference's neighbour kernels are generic over the position dtype
 * (torchmdnet/extensions/warp_kernels/neighbors_brute.py:27-175 and neighbors_cell.py:17-153 are instantiated for float32 and
 * float64; torchmdnet/extensions/warp_ops/neighbors.py:34-148 dispatches on positions.dtype; tests/test_neighbors.py:83,157,281
 * run both).  Outputs, padding, overflow reporting and the order (lower pairs by (i, j), their transposes, self loops) as above;
 * every strategy takes the brute-force search (same pair set); `ws` as for tmdnet_neighbor_pairs.  The model path stays fp32. */
int tmdnet_neighbor_pairs_f64(void* stream, void* ws, size_t ws_bytes, int64_t n_atoms, int64_t n_mol, const double* pos,
                              const int64_t* batch, const double* box, int32_t box_mode, double cutoff_lower, double cutoff_upper,
                              int64_t max_num_pairs, int32_t loop, int32_t include_transpose, int64_t* neighbors, double* deltas,
                              double* distances, int32_t* num_pairs);
int tmdnet_neighbor_grad_f64(void* stream, const int64_t* neighbors, const double* deltas, const double* distances,
                             const double* grad_deltas, const double* grad_distances, int64_t num_entries, int64_t n_atoms,
                             double* grad_positions);

/* ---- per-kernel-class timing (HIP events recorded on the launch stream around every launch of the
 * selected classes; bit c of category_mask selects class c).  tmdnet_profile_end synchronises the
 * stream and returns, per class: summed milliseconds, algorithmic FLOPs, algorithmic bytes (each distinct
 * input/output tensor of a launch counted once) and the number of launches.
 * Arrays must hold tmdnet_profile_num_categories() entries.  Used by bench.py for `roofline`. */
int tmdnet_profile_begin(tmdnet_model* m, uint32_t category_mask);
int tmdnet_profile_end(tmdnet_model* m, void* stream, double* ms, double* flops, double* bytes, int64_t* launches);
/* the same per launch, in launch order (up to `cap` records are written, *n_out = number recorded); `labels` (may be
 * NULL) receives one 64-byte NUL-terminated string per record naming the launch (kernel wrapper + shape): bench.py groups
 * by it to single out the dominant KERNEL of a class */
int tmdnet_profile_end_records(tmdnet_model* m, void* stream, int64_t cap, int32_t* cat, double* ms, double* flops, double* bytes,
                               char* labels, int64_t* n_out);
int tmdnet_profile_num_categories(void);
const char* tmdnet_profile_category_name(int idx);

/* ---- diagnostics ---------------------------------------------------------------------------------
 * Copy an intermediate of the last tmdnet_energy_forces call out of the workspace (device -> device).
 * Names: "X_embed", "X_layer<l>", "x", "phi", "Q", "u0", "G_embed".  Used by the parity tests. */
int tmdnet_debug_tensor(tmdnet_model* m, void* stream, const char* name, float* out, int64_t numel);
/* plain dense contraction through the path's MFMA GEMM: C[M,N] = A[M,K] @ W[N,K]^T (+bias) (silu != 0: silu of it); Wsb as
 * for tmdnet_debug_gemm_dual (NULL: fp32-MFMA kernels); for unit tests */
/* value + tangent GEMM of the edge MLP (kind 0 plain, 1 silu, 2 silu * rs with rs2 = d rs): C = f(A W^T + b), C2 = d/dd.
 * Wsb: optional DEVICE copy of the split-bf16 tile image of W (tmdnet_debug_split_weight); when given and the shape
 * qualifies, the bf16-MFMA kernel runs instead of the fp32-MFMA one. */
int tmdnet_debug_gemm_dual(void* stream, const float* A, const float* A2, const float* W, const float* bias, float* C, float* C2,
                           int64_t M, int64_t N, int64_t K, int32_t kind, const float* rs, const float* rs2, const uint16_t* Wsb);
/* host-side 3 x bf16 split of W[N,K] into the kernel's tile image; returns the number of uint16 elements
 * (out_host may be NULL to query the size) */
int64_t tmdnet_debug_split_weight(const float* W_host, int64_t N, int64_t K, uint16_t* out_host);
int tmdnet_debug_gemm(void* stream, const float* A, const float* W, const float* bias, float* C, int64_t M, int64_t N,
                      int64_t K, int32_t silu, const uint16_t* Wsb);

/* ---- First-order parameter gradients (TensorNet, TensorNet2 and Equivariant Transformer handles; energy-only training).
 * Replaces what autograd does in the reference for `loss(E).backward()` over torchmdnet/models/tensornet.py:543-619, 729-814,
 * 384-398 and output_modules.py:108-117: given d loss / d E_m per molecule, one call evaluates the energies (direct evaluation
 * of the radial functions, no tables) and the gradient of sum_m grad_energy[m] E_m with respect to every weight, into one flat
 * device buffer whose layout tmdnet_param_grad_entry enumerates (name, offset and element count; offsets are 64-float
 * aligned).  Entry names are the engine's: "Wdp"/"bdp" = distance_proj1..3 stacked, "Utab"/"Vtab" = the per-species tables
 * U[z] = emb(z) Wa^T + b, V[z] = emb(z) Wb^T of emb2([emb(z_i), emb(z_j)]) (chain to emb / emb2 on the caller's side),
 * "Ue{k}", "L1", "bL1", "L2", "bL2", "ln0_w", "ln0_b" = tensor_embedding linears_tensor / linears_scalar / init_norm,
 * "l{l}.M{k}", "l{l}.b{k}" = layers.l.linears_scalar.k, "l{l}.Va{k}" / "l{l}.Vb{k}" = layers.l.linears_tensor.k / .(3+k),
 * "lnr_w", "lnr_b" = out_norm, "Lin", "bLin" = linear, "O1", "bO1", "O2", "bO2" = output_network.layers.0 / .2.
 * TensorNet2 handles add "l{l}.M0b" / "l{l}.M0c" (charge blocks of linears_scalar.0) and "cp{h}.ln_w", "cp{h}.W1" ... (ChargePredict).
 * Equivariant Transformer handles enumerate their own entries ("emb", "embN", "Wn", "bn", "Wc", "bc", "l{l}.ln_w", "l{l}.Wqkv" =
 * q | k | v rows as packed by the engine, "l{l}.Wvp", "l{l}.Wo", "l{l}.Wdkv" = dk_proj | dv_proj, "lno_w", "W1u", "Wm1", "Wm2", "W21",
 * "Wn1", "Wn2", ...; torchmdnet_amd/models/model.py::_et_grads maps them back) and take the one-call form only.
 * Needs a graph built with the exact pair count (tmdnet_build_graph, no cell list); deterministic; no position gradient.
 * Two-call form for autograd: grad_energy == NULL runs the forward half only (energies out, activations kept in ws / train_ws),
 * a later call with energy == NULL and the same other arguments runs the reverse half on those workspaces. */
int tmdnet_param_grad_count(tmdnet_model* m);
const char* tmdnet_param_grad_entry(tmdnet_model* m, int idx, int64_t* offset, int64_t* numel);
int tmdnet_train_workspace_bytes(tmdnet_model* m, int64_t n_atoms, int64_t n_mol, int64_t n_pairs, size_t* fwd_bytes,
                                 size_t* train_bytes, int64_t* grad_floats);
int tmdnet_energy_param_grads(tmdnet_model* m, void* stream, void* graph_ws, void* ws, size_t ws_bytes, void* train_ws,
                              size_t train_bytes, int64_t n_atoms, int64_t n_mol, int64_t n_pairs, const int64_t* z,
                              const int64_t* batch, const float* q, const float* grad_energy, float* energy, float* grads);

/* Second-order pass of force-matching training (TensorNet + Scalar, Equivariant Transformer): the gradient, with respect to every weight, of
 *     s(theta) = v . d(sum_m E_m)/d pos = - v . F          v [n_atoms, 3] = d loss / d F  (device, caller's atom order)
 * so that d loss / d theta through the forces is  - grads.  Replaces the reference's second autograd pass over its own graph
 * (torchmdnet/models/model.py:618-628, create_graph = self.training) and the *_bwd_bwd kernels behind it
 * (torchmdnet/extensions/warp_ops/tensornet_mp.py:538-548 and siblings).  Analytic: the forward-mode tangent, along v, of the
 * forward + reverse program (no difference quotient); one self-contained pass that evaluates the radial functions directly and
 * keeps its own activations in `ws` (tmdnet_force_param_workspace_bytes: about 2 KB per atom-channel plus 0.4 KB per
 * pair-channel, 14 GiB for 256 molecules of 64 atoms at 128 channels - nothing is aliased).  `grads` has the layout of tmdnet_energy_param_grads (tmdnet_param_grad_entry; d s / d bO2 = 0).  Needs a graph
 * built with the exact pair count and without the cell list; `z` may be NULL when tmdnet_build_graph saw it; deterministic.
 * `hv` (device, [n_atoms, 3], or NULL): d s / d pos = H v, the Hessian of the summed energy applied to v - the position gradient
 * of a loss that depends on the forces is  - H (d loss / d F)  (the reference gets it from the same second autograd pass). */
int tmdnet_force_param_workspace_bytes(tmdnet_model* m, int64_t n_atoms, int64_t n_mol, int64_t n_pairs, size_t* bytes);
int tmdnet_force_param_grads(tmdnet_model* m, void* stream, void* graph_ws, void* ws, size_t ws_bytes, int64_t n_atoms,
                             int64_t n_mol, int64_t n_pairs, const int64_t* z, const int64_t* batch, const float* q, const float* v,
                             float* grads, float* hv);
/* The whole gradient of a loss(E, F) in ONE pass: tmdnet_force_param_grads with an energy seed.  ge [n_mol] = d loss / d E (device;
 * NULL = tmdnet_force_param_grads): grads = d S / d theta and hv = d S / d pos of
 *     S = v . d(sum_m E_m)/d pos - sum_m ge_m E_m ,
 * so d loss / d theta = - grads and d loss / d pos = - hv, without the separate tmdnet_energy_param_grads pass (the reference gets
 * both terms from one backward over its autograd graph, torchmdnet/models/model.py:618-628).  How: the tangent adjoint minus the
 * adjoint of sum_m ge_m E_m obeys the tangent adjoint's own recursion, so only its seed at the head differs.  Terms outside the
 * engine's parameter set (an Atomref prior's table: index_add of ge[molecule] over z) stay with the caller. */
int tmdnet_loss_param_grads(tmdnet_model* m, void* stream, void* graph_ws, void* ws, size_t ws_bytes, int64_t n_atoms, int64_t n_mol,
                            int64_t n_pairs, const int64_t* z, const int64_t* batch, const float* q, const float* v, const float* ge,
                            float* grads, float* hv);
/* Developer / test hook: copies one intermediate of the LAST tmdnet_force_param_grads call on this handle (same thread, workspace
 * untouched since) into `out` (device); names are the buffer names of csrc/tn_hvp_api.hip ("u0_t", "l0.Mi_t", "g_Pn", ...).
 * out == NULL: returns the element count instead of a status. */
int tmdnet_hvp_debug_tensor(tmdnet_model* m, void* stream, const char* name, float* out, int64_t numel);

#ifdef __cplusplus
}
#endif
#endif /* TMDNET_AMD_H */
