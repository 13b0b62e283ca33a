// Launch wrappers of the kernels only the TensorNet2 + ScalarPlusWeightedCoulomb path needs (tn_tn2.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tn_kernels.h"

namespace tn {

// erev[e] = reverse edge of e, eid[e] = e, pair_edge[p] = lower edge of pair p        (all [E] / [P])
void launch_edge_reverse(const Graph& g, int N, int* erev, int* eid, int* pair_edge, hipStream_t s);
// ChargePredict invariants [I ; |A|^2 ; |S|^2] and their adjoint (G += ...)           (tensornet2.py:139-150)
void launch_cp_feat(const float* X, int N, int F, float* feat, hipStream_t s);
void launch_cp_feat_bwd(const float* X, const float* g_feat, int N, int F, float* G, hipStream_t s);
// per-molecule charge equilibration                                                    (tensornet2.py:99-138)
void launch_qeq_fwd(const Graph& g, const float* out, const float* Qmol, const int64_t* batch, int N, int B, int qd, float* charges,
                    float* FuQ, hipStream_t s);
void launch_qeq_bwd(const Graph& g, const float* out, const int64_t* batch, int N, int B, int qd, const float* FuQ, const float* g_c,
                    float* g_out, hipStream_t s);
// first edge-MLP layer per directed edge: pre1 = Ap[pair] + Bt[i] + Cs[j], he1 = silu(pre1), Ce = C[pair]
void launch_tn2_edge_pre1(const Graph& g, int N, int F, const float* Ap, const float* Bt, const float* Cs, const float* C, float* pre1,
                          float* he1, float* Ce, hipStream_t s);
int tn2_gw_slots(int F);
void launch_tn2_edge_gw(const Graph& g, int N, int F, const float* gMi, const float* Pn, const float* pre3, const float* Ce,
                        float* g_pre3, float* gCe_slots, int64_t slot_stride, hipStream_t s);
void launch_tn2_edge_reduce(const Graph& g, int N, int F, const float* g_pre1, const int* erev, float* gB, float* gCs, float* gAp,
                            hipStream_t s);
void launch_tn2_pair_gd(const Graph& g, int Pcap, int F, const float* gAp, const float* dAp, const float* gCe_slots, int nslots,
                        int64_t slot_stride, const int* pair_edge, const int* erev, const float* dC, float* gd, hipStream_t s);
// Coulomb head (output_modules.py:440-606): cut <= 0: all pairs of a molecule; else reaction field inside the cutoff.
// ec [N] per-atom energy; with g_q != null also g_q [N, QC] and the pair forces fpos [N, 3] (for d E_total / d ... with d E = std)
int launch_coulomb(const Graph& g, const float* pos, const int64_t* batch, const float* box, int box_mode, int N, int B, int QC,
                   const float* charges, const float* wq, float cut, float eps_solvent, float scale, float* ec, float* g_q, float* fpos,
                   hipStream_t s);
void launch_tn2_head_energy(const float* ao, const float* O2, const float* bO2, const float* ec, int N, int H, float std,
                            const float* atomref, const int64_t* z, float* ea, hipStream_t s);
void launch_add_forces(const float* fc, const int* perm, int N, float* forces, hipStream_t s);
void launch_add(const float* x, float* y, int64_t n, hipStream_t s);
void launch_slice_cols(const float* src, int ld, int off, int rows, int cols, float* dst, hipStream_t s);
void launch_scale_rows(float* x, const float* r, float mul, int rows, int cols, hipStream_t s);
void launch_tn2_self_rows(const Graph& g, int N, int F, const float* rows, float* out, hipStream_t s);
void launch_put_cols(const float* src, int rows, int cols, float* dst, int ld, int off, hipStream_t s);

}  // namespace tn
