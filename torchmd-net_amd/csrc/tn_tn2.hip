// Kernels of the TensorNet2 + ScalarPlusWeightedCoulomb path that TensorNet does not have (gfx950).
// Reference: torchmdnet/models/tensornet2.py:49-157 (ChargePredict: invariants -> MLP -> per-molecule charge
// equilibration), :466-626 (Interaction: the edge MLP takes [phi(d), c_i, c_j], so its weights are per DIRECTED edge),
// torchmdnet/models/output_modules.py:323-609 (damped pair Coulomb energy of all charge channels, all-to-all inside a
// molecule or reaction field inside a cutoff).  Everything else of the path (graph, embedding, tensor linears, CSR sweeps,
// 3x3 algebra, readout) is TensorNet's code: the sweeps take their weight rows through an index array (Graph::epair), which
// here is the identity (forward: w[e]) or the reverse-edge map (adjoint: w[erev[e]]).
//
// Thread mapping as everywhere: channel fastest (coalesced 4F-byte runs), per-edge scalars wave-uniform, sums over a CSR row in
// the registers of the owning thread, per-molecule sums by one block per molecule in a fixed order: deterministic, no atomics.
#include "tn_tn2.h"

#include "tn_common.h"

namespace tn {

static inline int cdiv2(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------------------------------------- graph extras
// erev[e] = index of the edge (j <- i) for e = (i <- j) (binary search of i in row j; rows are sorted); eid[e] = e;
// pair_edge[p] = the lower edge (i > j) of pair p
__global__ __launch_bounds__(256) void k_edge_reverse(Graph g, int N, int* __restrict__ erev, int* __restrict__ eid,
                                                      int* __restrict__ pair_edge) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= N || g.counts[2]) return;
  const int e0 = g.rowptr[i], e1 = g.rowptr[i + 1];
  for (int e = e0 + lane; e < e1; e += 64) {
    const int j = g.col[e];
    int lo = g.rowptr[j], hi = g.rowptr[j + 1] - 1;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (g.col[mid] < i) lo = mid + 1; else hi = mid;
    }
    erev[e] = lo;
    eid[e] = e;
    if (j < i) pair_edge[g.epair[e]] = e;
  }
}
void launch_edge_reverse(const Graph& g, int N, int* erev, int* eid, int* pair_edge, hipStream_t s) {
  if (N <= 0) return;
  hipLaunchKernelGGL(k_edge_reverse, dim3(cdiv2(N, 4)), dim3(256), 0, s, g, N, erev, eid, pair_edge);
}

// ---------------------------------------------------------------------------------------------- charge prediction
// feat = [I ; ||A||^2 ; ||S||^2]  (tensornet2.py:141: I itself, unlike the readout's 3 I^2)
__global__ void k_cp_feat(const float* __restrict__ X, int N, int F, float* __restrict__ feat) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)N * F) return;
  const int n = (int)(idx / F), f = (int)(idx - (int64_t)n * F);
  float u[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) u[c] = X[(int64_t)n * 9 * F + c * F + f];
  float* o = feat + (int64_t)n * 3 * F + f;
  const float t = u[4] + u[7];
  o[0] = u[0];
  o[F] = 2.0f * (u[1] * u[1] + u[2] * u[2] + u[3] * u[3]);
  o[2 * F] = u[4] * u[4] + u[7] * u[7] + t * t + 2.0f * (u[5] * u[5] + u[6] * u[6] + u[8] * u[8]);
}
// G += d feat / d X . g_feat
__global__ void k_cp_feat_bwd(const float* __restrict__ X, const float* __restrict__ g_feat, int N, int F, float* __restrict__ G) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)N * F) return;
  const int n = (int)(idx / F), f = (int)(idx - (int64_t)n * F);
  float u[9], dq[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) u[c] = X[(int64_t)n * 9 * F + c * F + f];
  dquad(u, dq);
  const float* gf = g_feat + (int64_t)n * 3 * F + f;
  const float gI = gf[0], gA = gf[F], gS = gf[2 * F];
  float* o = G + (int64_t)n * 9 * F + f;
  o[0] += gI;  // feat_I = I = trace / 3 is component 0 itself
#pragma unroll
  for (int c = 1; c < 4; ++c) o[c * F] += dq[c] * gA;
#pragma unroll
  for (int c = 4; c < 9; ++c) o[c * F] += dq[c] * gS;
}
void launch_cp_feat(const float* X, int N, int F, float* feat, hipStream_t s) {
  if (N <= 0) return;
  hipLaunchKernelGGL(k_cp_feat, dim3(cdiv2((int64_t)N * F, 256)), dim3(256), 0, s, X, N, F, feat);
}
void launch_cp_feat_bwd(const float* X, const float* g_feat, int N, int F, float* G, hipStream_t s) {
  if (N <= 0) return;
  hipLaunchKernelGGL(k_cp_feat_bwd, dim3(cdiv2((int64_t)N * F, 256)), dim3(256), 0, s, X, g_feat, N, F, G);
}

// charge equilibration per molecule (tensornet2.py:99-138): out = [r | f] per atom,
//   fu = f^2, Fu = sum_mol fu + 1e-6, Qu = sum_mol r, c = r + fu / Fu (Q - Qu)
// one block per molecule; thread t owns channel t % qd and every (256 / qd)-th atom; fixed-order tree over the groups
template <bool BWD>
__global__ __launch_bounds__(256) void k_qeq(Graph g, const float* __restrict__ out, const float* __restrict__ Qmol,
                                             const int64_t* __restrict__ batch, int N, int qd, float* __restrict__ charges,
                                             float* __restrict__ FuQ /* [B][2 qd]: Fu | Q - Qu */,
                                             const float* __restrict__ g_c, float* __restrict__ g_out) {
  __shared__ float red[2][256];
  const int m = blockIdx.x, tid = threadIdx.x;
  const int groups = 256 / qd, grp = tid / qd, k = tid - grp * qd;
  const bool live = grp < groups;
  const bool sorted = !g.counts[3];
  const int i0 = sorted ? g.mstart[m] : 0, i1 = sorted ? g.mend[m] : N;
  const float Q = Qmol ? Qmol[m] : 0.f;
  auto block_sum2 = [&](float a, float b, float& sa, float& sb) {
    red[0][tid] = live ? a : 0.f;
    red[1][tid] = live ? b : 0.f;
    __syncthreads();
    sa = sb = 0.f;
    for (int gq = 0; gq < groups; ++gq) {  // fixed order
      sa += red[0][gq * qd + k];
      sb += red[1][gq * qd + k];
    }
    __syncthreads();
  };
  if (!BWD) {
    float sf = 0.f, sr = 0.f;
    if (live)
      for (int i = i0 + grp; i < i1; i += groups) {
        if (!sorted && batch[i] != m) continue;
        const float r = out[(int64_t)i * 2 * qd + k], f = out[(int64_t)i * 2 * qd + qd + k];
        sf += f * f;
        sr += r;
      }
    float Fu, Qu;
    block_sum2(sf, sr, Fu, Qu);
    Fu += 1.0e-6f;
    const float dQ = Q - Qu;
    if (tid < qd) {
      FuQ[(int64_t)m * 2 * qd + k] = Fu;
      FuQ[(int64_t)m * 2 * qd + qd + k] = dQ;
    }
    if (live)
      for (int i = i0 + grp; i < i1; i += groups) {
        if (!sorted && batch[i] != m) continue;
        const float r = out[(int64_t)i * 2 * qd + k], f = out[(int64_t)i * 2 * qd + qd + k];
        charges[(int64_t)i * qd + k] = r + f * f / Fu * dQ;
      }
  } else {
    // g_r = g_c - S1 ; g_f = 2 f dQ / Fu (g_c - S1) ;  S1 = sum_mol g_c fu / Fu
    const float Fu = FuQ[(int64_t)m * 2 * qd + k], dQ = FuQ[(int64_t)m * 2 * qd + qd + k];
    float s1 = 0.f;
    if (live)
      for (int i = i0 + grp; i < i1; i += groups) {
        if (!sorted && batch[i] != m) continue;
        const float f = out[(int64_t)i * 2 * qd + qd + k];
        s1 += g_c[(int64_t)i * qd + k] * f * f;
      }
    float S1, dummy;
    block_sum2(s1, 0.f, S1, dummy);
    S1 /= Fu;
    if (live)
      for (int i = i0 + grp; i < i1; i += groups) {
        if (!sorted && batch[i] != m) continue;
        const float f = out[(int64_t)i * 2 * qd + qd + k];
        const float gr = g_c[(int64_t)i * qd + k] - S1;
        g_out[(int64_t)i * 2 * qd + k] = gr;
        g_out[(int64_t)i * 2 * qd + qd + k] = 2.0f * f * dQ / Fu * gr;
      }
  }
}
void launch_qeq_fwd(const Graph& g, const float* out, const float* Qmol, const int64_t* batch, int N, int B, int qd, float* charges,
                    float* FuQ, hipStream_t s) {
  if (B <= 0) return;
  hipLaunchKernelGGL((k_qeq<false>), dim3(B), dim3(256), 0, s, g, out, Qmol, batch, N, qd, charges, FuQ, nullptr, nullptr);
}
void launch_qeq_bwd(const Graph& g, const float* out, const int64_t* batch, int N, int B, int qd, const float* FuQ, const float* g_c,
                    float* g_out, hipStream_t s) {
  if (B <= 0) return;
  hipLaunchKernelGGL((k_qeq<true>), dim3(B), dim3(256), 0, s, g, out, nullptr, batch, N, qd, nullptr, const_cast<float*>(FuQ), g_c,
                     g_out);
}

// ---------------------------------------------------------------------------------------------- edge MLP, first layer
// pre1[e] = Ap[pair(e)] + Bt[i] + Cs[j]  (M1 [phi ; c_i ; c_j] + b split into its pair / target / source blocks),
// he1 = silu(pre1), Ce[e] = C[pair(e)].  Block per row, thread per channel.
__global__ void k_tn2_edge_pre1(Graph g, int N, int F, const float* __restrict__ Ap, const float* __restrict__ Bt,
                                const float* __restrict__ Cs, const float* __restrict__ C, float* __restrict__ pre1,
                                float* __restrict__ he1, float* __restrict__ Ce) {
  const int i = xcd_chunk(blockIdx.x, gridDim.x);
  if (g.counts[2]) return;
  const int e0 = g.rowptr[i], e1 = g.rowptr[i + 1];
  for (int f = threadIdx.x; f < F; f += blockDim.x) {
    const float bt = Bt[(int64_t)i * F + f];
    int e = e0;
    for (; e + 4 <= e1; e += 4) {  // four edges' rows in flight
      int j[4], p[4];
      float ap[4], cs[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        j[u] = g.col[e + u];
        p[u] = g.epair[e + u];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        ap[u] = Ap[(int64_t)p[u] * F + f];
        cs[u] = Cs[(int64_t)j[u] * F + f];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float v = ap[u] + bt + cs[u];
        pre1[(int64_t)(e + u) * F + f] = v;
        he1[(int64_t)(e + u) * F + f] = silu(v);
        if (f == 0) Ce[e + u] = C[p[u]];
      }
    }
    for (; e < e1; ++e) {
      const int j = g.col[e], p = g.epair[e];
      const float v = Ap[(int64_t)p * F + f] + bt + Cs[(int64_t)j * F + f];
      pre1[(int64_t)e * F + f] = v;
      he1[(int64_t)e * F + f] = silu(v);
      if (f == 0) Ce[e] = C[p];
    }
  }
}
void launch_tn2_edge_pre1(const Graph& g, int N, int F, const float* Ap, const float* Bt, const float* Cs, const float* C, float* pre1,
                          float* he1, float* Ce, hipStream_t s) {
  if (N <= 0) return;
  const int t = ((F + 63) / 64) * 64;
  hipLaunchKernelGGL(k_tn2_edge_pre1, dim3(N), dim3(t > 256 ? 256 : t), 0, s, g, N, F, Ap, Bt, Cs, C, pre1, he1, Ce);
}

// ---------------------------------------------------------------------------------------------- reverse: per-edge weight gradient
// g_w[e, k, f] = sum_{c in k} gM[i, c, f] Pn[j, c, f]  for e = (i <- j);  w = silu(pre3) Ce:
//   g_pre3 = g_w Ce silu'(pre3) ;  g_Ce[e] = sum_{k, f} g_w silu(pre3)  (one partial per wave: slots[wave][e])
__global__ void k_tn2_edge_gw(Graph g, int N, int F, const float* __restrict__ gMi, const float* __restrict__ Pn,
                              const float* __restrict__ pre3, const float* __restrict__ Ce, float* __restrict__ g_pre3,
                              float* __restrict__ gCe_slots, int64_t slot_stride) {
  const int i = xcd_chunk(blockIdx.x, gridDim.x);
  if (g.counts[2]) return;
  const int f = threadIdx.x, lane = f & 63, wave = f >> 6;  // blockDim.x = F rounded up to whole waves
  const bool act = f < F;
  const int fc_ = act ? f : 0;
  const int e0 = g.rowptr[i], e1 = g.rowptr[i + 1];
  const int F3 = 3 * F, F9 = 9 * F;
  float gm[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) gm[c] = act ? gMi[(int64_t)i * F9 + c * F + fc_] : 0.f;
  // one edge: g_pre3 row of the edge, and the channel product whose wave sum is g_C[e]
  auto edge = [&](int e, float& h) {
    const int j = g.col[e];
    const float* sp = Pn + (int64_t)j * F9 + fc_;
    const float gw0 = gm[0] * sp[0];
    const float gw1 = gm[1] * sp[F] + gm[2] * sp[2 * F] + gm[3] * sp[3 * F];
    const float gw2 = gm[4] * sp[4 * F] + gm[5] * sp[5 * F] + gm[6] * sp[6 * F] + gm[7] * sp[7 * F] + gm[8] * sp[8 * F];
    const float* pp = pre3 + (int64_t)e * F3 + fc_;
    const float p0 = pp[0], p1 = pp[F], p2 = pp[2 * F];
    const float ce = Ce[e];
    if (act) {
      float* o = g_pre3 + (int64_t)e * F3 + f;
      o[0] = gw0 * ce * silu_grad(p0);
      o[F] = gw1 * ce * silu_grad(p1);
      o[2 * F] = gw2 * ce * silu_grad(p2);
    }
    h = gw0 * silu(p0) + gw1 * silu(p1) + gw2 * silu(p2);  // inactive lanes: gm = 0 -> h = 0
  };
  // four edges per trip: independent loads in flight together, one wave_sum4 for the four channel sums
  int e = e0;
  for (; e + 4 <= e1; e += 4) {
    float h0, h1, h2, h3;
    edge(e, h0);
    edge(e + 1, h1);
    edge(e + 2, h2);
    edge(e + 3, h3);
    const float tot = wave_sum4(h0, h1, h2, h3, lane);
    if ((lane & 15) == 0) gCe_slots[(int64_t)wave * slot_stride + e + (lane >> 4)] = tot;
  }
  for (; e < e1; ++e) {
    float h;
    edge(e, h);
    h = wave_sum(h);
    if (lane == 0) gCe_slots[(int64_t)wave * slot_stride + e] = h;
  }
}
int tn2_gw_slots(int F) { return (F + 63) / 64; }
void launch_tn2_edge_gw(const Graph& g, int N, int F, const float* gMi, const float* Pn, const float* pre3, const float* Ce,
                        float* g_pre3, float* gCe_slots, int64_t slot_stride, hipStream_t s) {
  if (N <= 0) return;
  hipLaunchKernelGGL(k_tn2_edge_gw, dim3(N), dim3(tn2_gw_slots(F) * 64), 0, s, g, N, F, gMi, Pn, pre3, Ce, g_pre3, gCe_slots, slot_stride);
}

// gB[i] = sum_{e in row i} g_pre1[e] ; gCs[i] = sum_{e in row i} g_pre1[erev[e]] (the edges whose SOURCE is i) ;
// gAp[p] = g_pre1[e] + g_pre1[erev[e]] for the lower edge e of pair p
__global__ void k_tn2_edge_reduce(Graph g, int N, int F, const float* __restrict__ g_pre1, const int* __restrict__ erev,
                                  float* __restrict__ gB, float* __restrict__ gCs, float* __restrict__ gAp) {
  const int i = xcd_chunk(blockIdx.x, gridDim.x);
  if (g.counts[2]) return;
  const int e0 = g.rowptr[i], e1 = g.rowptr[i + 1];
  for (int f = threadIdx.x; f < F; f += blockDim.x) {
    float sb = 0.f, sc = 0.f;
    auto edge = [&](int e, float& a, float& b) {
      const int r = erev[e];
      a = g_pre1[(int64_t)e * F + f];
      b = g_pre1[(int64_t)r * F + f];
    };
    auto put = [&](int e, float a, float b) {
      sb += a;
      sc += b;
      if (g.col[e] < i) gAp[(int64_t)g.epair[e] * F + f] = a + b;
    };
    int e = e0;
    for (; e + 4 <= e1; e += 4) {  // four edges' loads in flight together, summed in list order
      float a0, b0, a1, b1, a2, b2, a3, b3;
      edge(e, a0, b0);
      edge(e + 1, a1, b1);
      edge(e + 2, a2, b2);
      edge(e + 3, a3, b3);
      put(e, a0, b0);
      put(e + 1, a1, b1);
      put(e + 2, a2, b2);
      put(e + 3, a3, b3);
    }
    for (; e < e1; ++e) {
      float a, b;
      edge(e, a, b);
      put(e, a, b);
    }
    gB[(int64_t)i * F + f] = sb;
    gCs[(int64_t)i * F + f] = sc;
  }
}
void launch_tn2_edge_reduce(const Graph& g, int N, int F, const float* g_pre1, const int* erev, float* gB, float* gCs, float* gAp,
                            hipStream_t s) {
  if (N <= 0) return;
  const int t = ((F + 63) / 64) * 64;
  hipLaunchKernelGGL(k_tn2_edge_reduce, dim3(N), dim3(t > 256 ? 256 : t), 0, s, g, N, F, g_pre1, erev, gB, gCs, gAp);
}

// g_d[p] += sum_f gAp[p, f] dAp[p, f] + (g_Ce[e] + g_Ce[erev[e]]) dC[p]   (wave per pair; Ap(d) carried its tangent forward)
__global__ __launch_bounds__(256) void k_tn2_pair_gd(Graph g, int Pcap, int F, const float* __restrict__ gAp,
                                                     const float* __restrict__ dAp, const float* __restrict__ gCe_slots, int nslots,
                                                     int64_t slot_stride, const int* __restrict__ pair_edge,
                                                     const int* __restrict__ erev, const float* __restrict__ dC,
                                                     float* __restrict__ gd) {
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (p >= g.counts[0] || p >= Pcap || g.counts[2]) return;
  float s = 0.f;
  for (int f = lane; f < F; f += 64) s += gAp[(int64_t)p * F + f] * dAp[(int64_t)p * F + f];
  s = wave_sum(s);
  if (lane == 0) {
    const int e = pair_edge[p], r = erev[e];
    float gc = 0.f;
    for (int k = 0; k < nslots; ++k) gc += gCe_slots[(int64_t)k * slot_stride + e] + gCe_slots[(int64_t)k * slot_stride + r];
    gd[p] += s + gc * dC[p];
  }
}
void launch_tn2_pair_gd(const Graph& g, int Pcap, int F, const float* gAp, const float* dAp, const float* gCe_slots, int nslots,
                        int64_t slot_stride, const int* pair_edge, const int* erev, const float* dC, float* gd, hipStream_t s) {
  if (Pcap <= 0) return;
  hipLaunchKernelGGL(k_tn2_pair_gd, dim3(cdiv2(Pcap, 4)), dim3(256), 0, s, g, Pcap, F, gAp, dAp, gCe_slots, nslots, slot_stride,
                     pair_edge, erev, dC, gd);
}

// ---------------------------------------------------------------------------------------------- Coulomb head
// e_i = sum_{j != i, same molecule [, d < cut]} F0 fc(d) g(d) s_ij ,  s_ij = sum_k wq_k q_ik q_jk / W   (each pair in both atoms:
// output_modules.py:504-534, 546-603); fc = 1 - exp(1 - 1/(1 - u^2)), u = clamp(d / 4.6); g = 1/d (all-to-all) or
// 1/d + k_rf d^2 - c_rf (reaction field).  With forces: g_q[i,k] = 2 std F0 wq_k/W sum_j fc g q_jk and
// fpos[i] = - 2 std F0 sum_j s_ij (fc g)' (pos_i - pos_j)/d.  Wave per atom, lanes over the molecule's atoms.
constexpr int kMaxQC = 64;  // (num_layers + 1) * q_dim
template <int QC8>  // charge channels handled in chunks of 8 registers
__global__ __launch_bounds__(256) void k_coulomb(Graph g, const float* __restrict__ pos, const int64_t* __restrict__ batch,
                                                 const float* __restrict__ box, int box_mode, int N, int B, int QC,
                                                 const float* __restrict__ charges, const float* __restrict__ wq, float cut,
                                                 float k_rf, float c_rf, float scale /* std */, float* __restrict__ ec,
                                                 float* __restrict__ g_q, float* __restrict__ fpos) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= N) return;
  const float F0 = 0.5f * 27.211386024367243f * 0.5291772105638411f;
  int64_t b = batch ? batch[i] : 0;
  int j0 = 0, j1 = N;
  if (b < 0 || b >= B) {
    j1 = 0;
    b = 0;
  } else if (!g.counts[3]) {
    j0 = g.mstart[b];
    j1 = g.mend[b];
  }
  const float* bx = box_mode == 0 ? nullptr : (box_mode == 1 ? box : box + b * 9);
  float W = 0.f;
  for (int k = 0; k < QC; ++k) W += wq[k];
  const float xi = pos[i * 3], yi = pos[i * 3 + 1], zi = pos[i * 3 + 2];
  float e_acc = 0.f, fx = 0.f, fy = 0.f, fz = 0.f;
  float gq[QC8 * 8];
#pragma unroll
  for (int k = 0; k < QC8 * 8; ++k) gq[k] = 0.f;
  const float* qi = charges + (int64_t)i * QC;
  for (int base = j0; base < j1; base += 64) {
    const int j = base + lane;
    if (j >= j1 || j == i) continue;
    if (batch && batch[j] != b) continue;
    float dx = xi - pos[j * 3], dy = yi - pos[j * 3 + 1], dz = zi - pos[j * 3 + 2];
    if (bx) {
      const float s3 = roundf(dz / bx[8]);
      dx -= s3 * bx[6]; dy -= s3 * bx[7]; dz -= s3 * bx[8];
      const float s2 = roundf(dy / bx[4]);
      dx -= s2 * bx[3]; dy -= s2 * bx[4];
      const float s1 = roundf(dx / bx[0]);
      dx -= s1 * bx[0];
    }
    const float d2 = dx * dx + dy * dy + dz * dz;
    if (cut > 0.f && !(d2 < cut * cut)) continue;
    const float d = sqrtf(d2), inv = 1.0f / d;
    const float u = fminf(d * (1.0f / 4.6f), 1.0f - 1e-6f);
    const float om = 1.0f - u * u;
    const float hexp = expf(1.0f - 1.0f / om);  // exp(-1/(1-u^2)) / exp(-1)
    const float fc = 1.0f - hexp;
    const float dfc = (d * (1.0f / 4.6f) < 1.0f - 1e-6f) ? 2.0f * u * hexp / (om * om) * (1.0f / 4.6f) : 0.f;
    float gg = inv, dg = -inv * inv;
    if (cut > 0.f) {
      gg += k_rf * d2 - c_rf;
      dg += 2.0f * k_rf * d;
    }
    const float a = F0 * fc * gg;
    const float* qj = charges + (int64_t)j * QC;
    float sij = 0.f;
#pragma unroll
    for (int k = 0; k < QC8 * 8; ++k)
      if (k < QC) {
        const float qjk = qj[k];
        sij += wq[k] * qi[k] * qjk;
        gq[k] += a * qjk;
      }
    sij /= W;
    e_acc += a * sij;
    const float fr = -2.0f * scale * F0 * sij * (dfc * gg + fc * dg) * inv;  // force on i along (pos_i - pos_j)
    fx += fr * dx; fy += fr * dy; fz += fr * dz;
  }
  e_acc = wave_sum(e_acc);
  if (lane == 0) ec[i] = e_acc;
  if (g_q) {
    fx = wave_sum(fx); fy = wave_sum(fy); fz = wave_sum(fz);
    if (lane == 0) {
      fpos[i * 3] = fx; fpos[i * 3 + 1] = fy; fpos[i * 3 + 2] = fz;
    }
#pragma unroll
    for (int k = 0; k < QC8 * 8; ++k)
      if (k < QC) {
        const float v = wave_sum(gq[k]);
        if (lane == 0) g_q[(int64_t)i * QC + k] = 2.0f * scale * wq[k] / W * v;
      }
  }
}
int launch_coulomb(const Graph& g, const float* pos, const int64_t* batch, const float* box, int box_mode, int N, int B, int QC,
                   const float* charges, const float* wq, float cut, float eps_solvent, float scale, float* ec, float* g_q, float* fpos,
                   hipStream_t s) {
  if (N <= 0) return 0;
  if (QC > kMaxQC) return 1;
  float k_rf = 0.f, c_rf = 0.f;
  if (cut > 0.f) {
    k_rf = (1.0f / (cut * cut * cut)) * (eps_solvent - 1.0f) / (2.0f * eps_solvent + 1.0f);
    c_rf = (1.0f / cut) * (3.0f * eps_solvent) / (2.0f * eps_solvent + 1.0f);
  }
  const dim3 grid(cdiv2(N, 4)), block(256);
#define CL(n8) hipLaunchKernelGGL((k_coulomb<n8>), grid, block, 0, s, g, pos, batch, box, box_mode, N, B, QC, charges, wq, cut, k_rf, c_rf, scale, ec, g_q, fpos)
  const int n8 = (QC + 7) / 8;
  switch (n8) {
    case 1: CL(1); break;
    case 2: CL(2); break;
    case 3: CL(3); break;
    case 4: CL(4); break;
    case 5: CL(5); break;
    case 6: CL(6); break;
    case 7: CL(7); break;
    default: CL(8); break;
  }
#undef CL
  return 0;
}

// per-atom energy: (silu(ao) . O2 + b + e_coulomb) std + atomref[z]
__global__ __launch_bounds__(256) void k_tn2_head_energy(const float* __restrict__ ao, const float* __restrict__ O2,
                                                         const float* __restrict__ bO2, const float* __restrict__ ec, int N, int H,
                                                         float std, const float* __restrict__ atomref,
                                                         const int64_t* __restrict__ z, float* __restrict__ ea) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (n >= N) return;
  float sacc = 0.f;
  for (int k = lane; k < H; k += 64) sacc += silu(ao[(int64_t)n * H + k]) * O2[k];
  sacc = wave_sum(sacc);
  if (lane == 0) {
    float e = (sacc + bO2[0] + ec[n]) * std;
    if (atomref) e += atomref[z[n]];
    ea[n] = e;
  }
}
void launch_tn2_head_energy(const float* ao, const float* O2, const float* bO2, const float* ec, int N, int H, float std,
                            const float* atomref, const int64_t* z, float* ea, hipStream_t s) {
  if (N <= 0) return;
  hipLaunchKernelGGL(k_tn2_head_energy, dim3(cdiv2(N, 4)), dim3(256), 0, s, ao, O2, bO2, ec, N, H, std, atomref, z, ea);
}

// forces -= fcoul permuted back (the Coulomb pair forces are computed in the graph's internal atom order)
__global__ void k_add_forces(const float* __restrict__ fc, const int* __restrict__ perm, int N, float* __restrict__ forces) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int o = perm ? perm[i] : i;
  forces[o * 3] += fc[i * 3];
  forces[o * 3 + 1] += fc[i * 3 + 1];
  forces[o * 3 + 2] += fc[i * 3 + 2];
}
void launch_add_forces(const float* fc, const int* perm, int N, float* forces, hipStream_t s) {
  if (N <= 0) return;
  hipLaunchKernelGGL(k_add_forces, dim3(cdiv2(N, 256)), dim3(256), 0, s, fc, perm, N, forces);
}

// y[i, :] += x[i, :] (row-major, same shapes)
__global__ void k_axpy1(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] += x[i];
}
void launch_add(const float* x, float* y, int64_t n, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_axpy1, dim3(cdiv2(n, 256)), dim3(256), 0, s, x, y, n);
}
// dst[i, 0:cols] = src[i, off : off + cols]
__global__ void k_slice_cols(const float* __restrict__ src, int ld, int off, int rows, int cols, float* __restrict__ dst) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)rows * cols) return;
  const int r = (int)(idx / cols), c = (int)(idx - (int64_t)r * cols);
  dst[idx] = src[(int64_t)r * ld + off + c];
}
void launch_slice_cols(const float* src, int ld, int off, int rows, int cols, float* dst, hipStream_t s) {
  if (rows <= 0) return;
  hipLaunchKernelGGL(k_slice_cols, dim3(cdiv2((int64_t)rows * cols, 256)), dim3(256), 0, s, src, ld, off, rows, cols, dst);
}
// dst[i, off : off + cols] = src[i, 0:cols]
__global__ void k_put_cols(const float* __restrict__ src, int rows, int cols, float* __restrict__ dst, int ld, int off) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)rows * cols) return;
  const int r = (int)(idx / cols), c = (int)(idx - (int64_t)r * cols);
  dst[(int64_t)r * ld + off + c] = src[idx];
}
void launch_put_cols(const float* src, int rows, int cols, float* dst, int ld, int off, hipStream_t s) {
  if (rows <= 0) return;
  hipLaunchKernelGGL(k_put_cols, dim3(cdiv2((int64_t)rows * cols, 256)), dim3(256), 0, s, src, rows, cols, dst, ld, off);
}


// ---- parameter gradients
// x[i, :] *= r[i] * mul
__global__ void k_scale_rows(float* __restrict__ x, const float* __restrict__ r, float mul, int rows, int cols) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)rows * cols) return;
  x[idx] *= r[idx / cols] * mul;
}
void launch_scale_rows(float* x, const float* r, float mul, int rows, int cols, hipStream_t s) {
  if (rows <= 0 || cols <= 0) return;
  hipLaunchKernelGGL(k_scale_rows, dim3(cdiv2((int64_t)rows * cols, 256)), dim3(256), 0, s, x, r, mul, rows, cols);
}
// out[i, :] = rows[e, :] of atom i's self edge (every atom has one: the pair list carries self loops); zeros if it has none
__global__ void k_tn2_self_rows(Graph g, int N, int F, const float* __restrict__ rows, float* __restrict__ out) {
  const int i = blockIdx.x;
  if (g.counts[2]) return;
  int es = -1;
  for (int e = g.rowptr[i]; e < g.rowptr[i + 1]; ++e)
    if (g.col[e] == i) es = e;
  for (int f = threadIdx.x; f < F; f += blockDim.x) out[(int64_t)i * F + f] = es >= 0 ? rows[(int64_t)es * F + f] : 0.f;
}
void launch_tn2_self_rows(const Graph& g, int N, int F, const float* rows, float* out, hipStream_t s) {
  if (N <= 0) return;
  hipLaunchKernelGGL(k_tn2_self_rows, dim3(N), dim3(((F + 63) / 64) * 64 > 256 ? 256 : ((F + 63) / 64) * 64), 0, s, g, N, F, rows, out);
}

}  // namespace tn
