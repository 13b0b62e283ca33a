// Per-element arithmetic of the analytic second-order pass of TensorNet2 + ScalarPlusWeightedCoulomb (force-matching training): what
// the architecture adds to TensorNet's pass (tn_hvp_math.h) - the ChargePredict heads with their per-molecule equilibration, the
// edge MLP per DIRECTED edge on [phi(d), c_i, c_j], the damped pair Coulomb energy of all charge channels.  Specification:
// oracle/tn2_second_order.py (pinned to autograd-of-autograd); reference tensornet2.py:49-626, output_modules.py:323-609 and its
// second autograd pass model.py:618-628.  Same rules as tn_hvp_math.h: one logical thread per function call, raw pointers,
// __host__ __device__, run on the host by tests/hvp_host.hip.
#pragma once
#include "tn_hvp_math.h"

namespace tn {
namespace hvp {

// ------------------------------------------------------------------------------------------------ ChargePredict (tensornet2.py:49-157)
// invariants [I ; |A|^2 ; |S|^2] of X (the trace component itself, not 3 I^2) -> feat [N][3F]
HVP_FN void cp_feat_dual(int n, int f, int F, const float* X, const float* X_t, float* feat, float* feat_t) {
  const int64_t o = (int64_t)n * 9 * F + f, of = (int64_t)n * 3 * F + f;
  float x[9], xt[9], dq[9];
  ld9(X + o, F, x);
  ld9(X_t + o, F, xt);
  dquad(x, dq);
  const float t = x[4] + x[7];
  feat[of] = x[0];
  feat_t[of] = xt[0];
  feat[of + F] = 2.0f * (x[1] * x[1] + x[2] * x[2] + x[3] * x[3]);
  feat_t[of + F] = dq[1] * xt[1] + dq[2] * xt[2] + dq[3] * xt[3];
  feat[of + 2 * F] = x[4] * x[4] + x[7] * x[7] + t * t + 2.0f * (x[5] * x[5] + x[6] * x[6] + x[8] * x[8]);
  feat_t[of + 2 * F] = dq[4] * xt[4] + dq[5] * xt[5] + dq[6] * xt[6] + dq[7] * xt[7] + dq[8] * xt[8];
}
// adjoint: G += (g_I ; dquad_A g_A ; dquad_S g_S)
HVP_FN void cp_feat_bwd_dual(int n, int f, int F, const float* X, const float* X_t, const float* g_feat, const float* g_feat_t, float* G,
                             float* G_t) {
  const int64_t o = (int64_t)n * 9 * F + f, of = (int64_t)n * 3 * F + f;
  float x[9], xt[9], dq[9], dqt[9];
  ld9(X + o, F, x);
  ld9(X_t + o, F, xt);
  dquad(x, dq);
  dquad(xt, dqt);
  G[o] += g_feat[of];
  G_t[o] += g_feat_t[of];
  for (int c = 1; c < 9; ++c) {
    const int k = c < 4 ? 1 : 2;
    const float g = g_feat[of + k * F], gt = g_feat_t[of + k * F];
    G[o + c * F] += dq[c] * g;
    G_t[o + c * F] += dqt[c] * g + dq[c] * gt;
  }
}
// molecule m, channel q: Fu = sum_n f^2 + 1e-6, Qu = sum_n c over the head's output out [N][2 qd] = (c | f); sums [B][4][qd] =
// (Fu, Fu_t, Qu, Qu_t)
HVP_FN void cp_mol_sums(int m, int q, int qd, const int* mstart, const int* mend, const float* out, const float* out_t, float* sums) {
  float Fu = 1.0e-6f, Fut = 0.f, Qu = 0.f, Qut = 0.f;
  for (int n = mstart[m]; n < mend[m]; ++n) {
    const float c = out[(int64_t)n * 2 * qd + q], ct = out_t[(int64_t)n * 2 * qd + q];
    const float f = out[(int64_t)n * 2 * qd + qd + q], ft = out_t[(int64_t)n * 2 * qd + qd + q];
    Fu += f * f;
    Fut += 2.0f * f * ft;
    Qu += c;
    Qut += ct;
  }
  float* s = sums + ((int64_t)m * 4) * qd + q;
  s[0] = Fu;
  s[qd] = Fut;
  s[2 * qd] = Qu;
  s[3 * qd] = Qut;
}
// charges = c + f^2 / Fu (Q_mol - Qu)        (atom n, channel q; written into column block `off` of a [N][ldc] buffer)
HVP_FN void cp_qeq_dual(int n, int q, int qd, const int64_t* batch, const float* Qmol, const float* out, const float* out_t,
                        const float* sums, float* ch, float* ch_t, int ldc, int off) {
  const int m = (int)batch[n];
  const float* s = sums + ((int64_t)m * 4) * qd + q;
  const float Fu = s[0], Fut = s[qd], Qu = s[2 * qd], Qut = s[3 * qd];
  const float c = out[(int64_t)n * 2 * qd + q], ct = out_t[(int64_t)n * 2 * qd + q];
  const float f = out[(int64_t)n * 2 * qd + qd + q], ft = out_t[(int64_t)n * 2 * qd + qd + q];
  const float fu = f * f, fut = 2.0f * f * ft;
  const float dQ = (Qmol ? Qmol[m] : 0.0f) - Qu, dQt = -Qut;
  const float r = fu / Fu, rt = fut / Fu - fu * Fut / (Fu * Fu);
  ch[(int64_t)n * ldc + off + q] = c + r * dQ;
  ch_t[(int64_t)n * ldc + off + q] = ct + rt * dQ + r * dQt;
}
// per-atom pieces of the equilibration's adjoint (atom n, channel q): r, dQ and the summands of the two molecule sums
HVP_FN void cp_qeq_terms(int n, int q, int qd, const int64_t* batch, const float* Qmol, const float* out, const float* out_t,
                         const float* sums, const float* g_ch, const float* g_ch_t, int ldg, int off, float& f, float& ft, float& Fu,
                         float& Fut, float& g_r, float& g_rt, float& g_dQ, float& g_dQt, float& hF, float& hFt) {
  const int m = (int)batch[n];
  const float* s = sums + ((int64_t)m * 4) * qd + q;
  Fu = s[0];
  Fut = s[qd];
  const float Qu = s[2 * qd], Qut = s[3 * qd];
  f = out[(int64_t)n * 2 * qd + qd + q];
  ft = out_t[(int64_t)n * 2 * qd + qd + q];
  const float fu = f * f, fut = 2.0f * f * ft;
  const float dQ = (Qmol ? Qmol[m] : 0.0f) - Qu, dQt = -Qut;
  const float r = fu / Fu, rt = fut / Fu - fu * Fut / (Fu * Fu);
  const float g = g_ch[(int64_t)n * ldg + off + q], gt = g_ch_t[(int64_t)n * ldg + off + q];
  g_r = g * dQ;
  g_rt = gt * dQ + g * dQt;
  g_dQ = g * r;
  g_dQt = gt * r + g * rt;
  hF = g_r * fu / (Fu * Fu);
  hFt = (g_rt * fu + g_r * fut) / (Fu * Fu) - 2.0f * g_r * fu * Fut / (Fu * Fu * Fu);
}
// molecule sums of the adjoint: bs [B][4][qd] = (sum g_dQ, its tangent, sum hF, its tangent)
HVP_FN void cp_mol_sums_bwd(int m, int q, int qd, const int* mstart, const int* mend, const int64_t* batch, const float* Qmol, const float* out,
                            const float* out_t, const float* sums, const float* g_ch, const float* g_ch_t, int ldg, int off, float* bs) {
  float a = 0.f, at = 0.f, b = 0.f, bt = 0.f;
  for (int n = mstart[m]; n < mend[m]; ++n) {
    float f, ft, Fu, Fut, g_r, g_rt, g_dQ, g_dQt, hF, hFt;
    cp_qeq_terms(n, q, qd, batch, Qmol, out, out_t, sums, g_ch, g_ch_t, ldg, off, f, ft, Fu, Fut, g_r, g_rt, g_dQ, g_dQt, hF, hFt);
    a += g_dQ;
    at += g_dQt;
    b += hF;
    bt += hFt;
  }
  float* s = bs + ((int64_t)m * 4) * qd + q;
  s[0] = a;
  s[qd] = at;
  s[2 * qd] = b;
  s[3 * qd] = bt;
}
// g_out = ( g_c | g_f ):  g_c = g_ch - sum_mol g_dQ ;  g_f = 2 f ( g_r / Fu - sum_mol hF )
HVP_FN void cp_qeq_bwd_dual(int n, int q, int qd, const int64_t* batch, const float* Qmol, const float* out, const float* out_t,
                            const float* sums, const float* bs, const float* g_ch, const float* g_ch_t, int ldg, int off, float* g_out,
                            float* g_out_t) {
  float f, ft, Fu, Fut, g_r, g_rt, g_dQ, g_dQt, hF, hFt;
  cp_qeq_terms(n, q, qd, batch, Qmol, out, out_t, sums, g_ch, g_ch_t, ldg, off, f, ft, Fu, Fut, g_r, g_rt, g_dQ, g_dQt, hF, hFt);
  const float* s = bs + ((int64_t)batch[n] * 4) * qd + q;
  const float g = g_ch[(int64_t)n * ldg + off + q], gt = g_ch_t[(int64_t)n * ldg + off + q];
  g_out[(int64_t)n * 2 * qd + q] = g - s[0];
  g_out_t[(int64_t)n * 2 * qd + q] = gt - s[qd];
  const float g_fu = g_r / Fu - s[2 * qd], g_fut = g_rt / Fu - g_r * Fut / (Fu * Fu) - s[3 * qd];
  g_out[(int64_t)n * 2 * qd + qd + q] = 2.0f * f * g_fu;
  g_out_t[(int64_t)n * 2 * qd + qd + q] = 2.0f * (ft * g_fu + f * g_fut);
}

// ------------------------------------------------------------------------------------------------ edge MLP per directed edge
// first edge-MLP layer per directed edge e = (i <- j), decomposed as the engine does (tn_tn2_api.hip): with the first weight matrix
// split by columns M1 = [M1a | M1b | M1c] (tensornet2.py:548-566),  pre1[e] = Ap[pair] + Bt[i] + Cs[j],  Ap = phi M1a^T + b1 [P + 1][F],
// Bt = c M1b^T, Cs = c M1c^T [N][F];  he1 = silu(pre1).                                                           (idx over E * F)
HVP_FN void tn2_edge_pre1_dual(int64_t idx, int N, int F, const int* rowptr, const int* col, const int* epair, const float* Ap,
                               const float* Ap_t, const float* Bt, const float* Bt_t, const float* Cs, const float* Cs_t, float* pre1,
                               float* e1_t, float* he1, float* he1_t) {
  const int e = (int)(idx / F), f = (int)(idx - (int64_t)e * F);
  const int i = edge_geom_row(e, N, rowptr), j = col[e], p = epair[e];
  const float a = Ap[(int64_t)p * F + f] + Bt[(int64_t)i * F + f] + Cs[(int64_t)j * F + f];
  const float at = Ap_t[(int64_t)p * F + f] + Bt_t[(int64_t)i * F + f] + Cs_t[(int64_t)j * F + f];
  pre1[idx] = a;
  e1_t[idx] = at;
  he1[idx] = silu0(a);
  he1_t[idx] = silu1(a) * at;
}
// adjoint of that sum: gB[i] = sum_{e in row i} g1[e] (target block), gCs[i] = sum_{e in row i} g1[erev[e]] (the edges whose SOURCE is
// i), gself[i] = g1[self edge of i] (its column sum is the self pair's row of gAp)                                   ((atom, channel))
HVP_FN void tn2_edge_reduce_dual(int i, int f, int F, const int* rowptr, const int* col, const int* erev, const float* g1, const float* g1_t,
                                 float* gB, float* gB_t, float* gCs, float* gCs_t, float* gself, float* gself_t) {
  float b = 0.f, bt = 0.f, c = 0.f, ct = 0.f, sf = 0.f, sft = 0.f;
  for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) {
    const int r = erev[e];
    b += g1[(int64_t)e * F + f];
    bt += g1_t[(int64_t)e * F + f];
    c += g1[(int64_t)r * F + f];
    ct += g1_t[(int64_t)r * F + f];
    if (col[e] == i) {
      sf = g1[(int64_t)e * F + f];
      sft = g1_t[(int64_t)e * F + f];
    }
  }
  const int64_t o = (int64_t)i * F + f;
  gB[o] = b;
  gB_t[o] = bt;
  gCs[o] = c;
  gCs_t[o] = ct;
  gself[o] = sf;
  gself_t[o] = sft;
}
// gAp[p] = g1[e] + g1[erev[e]] for the lower edge e of pair p < P (row P = the self pair: filled from the column sum of gself)   (idx over P * F)
HVP_FN void tn2_pair_reduce_dual(int64_t idx, int F, const int* pair_edge, const int* erev, const float* g1, const float* g1_t, float* gAp,
                                 float* gAp_t) {
  const int p = (int)(idx / F), f = (int)(idx - (int64_t)p * F);
  const int e = pair_edge[p], r = erev[e];
  gAp[idx] = g1[(int64_t)e * F + f] + g1[(int64_t)r * F + f];
  gAp_t[idx] = g1_t[(int64_t)e * F + f] + g1_t[(int64_t)r * F + f];
}
// w = silu(e3) C(d) per directed edge row (width 3F)
HVP_FN void tn2_w_dual(int64_t i, int F3, const int* epair, const float* e3, const float* e3_t, const float* C, const float* C_t, float* w,
                       float* w_t) {
  const int p = epair[i / F3];
  const float f3 = silu0(e3[i]);
  w[i] = f3 * C[p];
  w_t[i] = silu1(e3[i]) * e3_t[i] * C[p] + f3 * C_t[p];
}
// out[i, c] = init + sum_{e in row i} ( wA[e', type(c)] srcA[col(e), c] + wB[e', type(c)] srcB[col(e), c] ),  e' = emap ? emap[e] : e
// (forward: own edge's weights; adjoint wrt the source: emap = reverse-edge map, the weights of (col <- row))
HVP_FN void edge_sweep2(int i, int f, int F, const int* rowptr, const int* col, const int* emap, const float* wA, const float* srcA,
                        const float* wB, const float* srcB, const float* init, float* out) {
  const int F3 = 3 * F, F9 = 9 * F;
  float acc[9];
  for (int c = 0; c < 9; ++c) acc[c] = init ? init[(int64_t)i * F9 + c * F + f] : 0.f;
  for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) {
    const int j = col[e], ew = emap ? emap[e] : e;
    for (int c = 0; c < 9; ++c) {
      float t = wA[(int64_t)ew * F3 + type_of(c) * F + f] * srcA[(int64_t)j * F9 + c * F + f];
      if (wB) t += wB[(int64_t)ew * F3 + type_of(c) * F + f] * srcB[(int64_t)j * F9 + c * F + f];
      acc[c] += t;
    }
  }
  st9(out + (int64_t)i * F9 + f, F, acc);
}
// weight adjoint per directed edge g_w[e, k] = sum_{c in k} gMi[row] Pn[col], carried through w = silu(e3) C to g_e3 = g_w C silu'(e3);
// gcp[e, f] = this channel's part of the adjoint of C (position gradient; may be null)                         (idx over E * F)
HVP_FN void tn2_edge_gw_dual(int64_t idx, int N, int F, const int* rowptr, const int* col, const int* epair, const float* g_Mi,
                             const float* g_Mi_t, const float* Pn, const float* Pn_t, const float* e3, const float* e3_t, const float* C,
                             const float* C_t, float* g_e3, float* g_e3_t, float* gcp, float* gcp_t) {
  const int e = (int)(idx / F), f = (int)(idx - (int64_t)e * F);
  const int i = edge_geom_row(e, N, rowptr), j = col[e], p = epair[e];
  const int64_t oi = (int64_t)i * 9 * F + f, oj = (int64_t)j * 9 * F + f;
  float gw[3] = {0.f, 0.f, 0.f}, gwt[3] = {0.f, 0.f, 0.f};
  for (int c = 0; c < 9; ++c) {
    const int k = type_of(c);
    gw[k] += g_Mi[oi + c * F] * Pn[oj + c * F];
    gwt[k] += g_Mi_t[oi + c * F] * Pn[oj + c * F] + g_Mi[oi + c * F] * Pn_t[oj + c * F];
  }
  const float c_ = C[p], ct = C_t[p];
  float gc = 0.f, gct = 0.f;
  for (int k = 0; k < 3; ++k) {
    const int64_t o = (int64_t)e * 3 * F + k * F + f;
    const float d1 = silu1(e3[o]);
    g_e3[o] = gw[k] * c_ * d1;
    g_e3_t[o] = gwt[k] * c_ * d1 + gw[k] * ct * d1 + gw[k] * c_ * silu2(e3[o]) * e3_t[o];
    gc += gw[k] * silu0(e3[o]);
    gct += gwt[k] * silu0(e3[o]) + gw[k] * d1 * e3_t[o];
  }
  if (gcp) {
    gcp[idx] = gc;
    gcp_t[idx] = gct;
  }
}
// per directed edge: val[e] (+)= sum_j x[e, j] y[pair(e), j] ; val_t[e] (+)= sum_j ( x_t y + x y2 d_t[pair] )       (rows of x have stride ldx)
HVP_FN void edge_rowdot(int e, int W, int ldx, const int* epair, const float* x, const float* x_t, const float* y, const float* y2,
                        const float* d_t, int accumulate, float* val, float* val_t) {
  const int p = epair[e];
  float s = 0.f, st = 0.f;
  for (int j = 0; j < W; ++j) {
    const float yy = y ? y[(int64_t)p * W + j] : 1.0f;
    s += x[(int64_t)e * ldx + j] * yy;
    st += x_t[(int64_t)e * ldx + j] * yy;
    if (y2) st += x[(int64_t)e * ldx + j] * y2[(int64_t)p * W + j] * d_t[p];
  }
  val[e] = (accumulate ? val[e] : 0.f) + s;
  val_t[e] = (accumulate ? val_t[e] : 0.f) + st;
}
// pair p: out[p] += val[e] + val[erev[e]] for the pair's lower edge e = pair_edge[p]
HVP_FN void pair_from_edges(int p, const int* pair_edge, const int* erev, const float* val, const float* val_t, float* out, float* out_t) {
  const int e = pair_edge[p], r = erev[e];
  out[p] += val[e] + val[r];
  out_t[p] += val_t[e] + val_t[r];
}

// ------------------------------------------------------------------------------------------------ Coulomb head (output_modules.py:323-606)
// exp(-1 / (1 - x^2)) e with x = clamp(d / rc, 0, 1 - 1e-6): value and first two derivatives in d
HVP_FN void exp_cutoff_d2(float d, float rc, float& f, float& f1, float& f2) {
  const float x = d / rc, xc = fminf(fmaxf(x, 0.f), 1.0f - 1e-6f), u = 1.0f - xc * xc;
  f = expf(-1.0f / u) / 0.36787944117144233f;
  const bool in = x < 1.0f - 1e-6f;
  f1 = in ? f * (-2.0f * xc / (u * u)) / rc : 0.f;
  f2 = in ? f * (4.0f * xc * xc / (u * u * u * u) - 2.0f / (u * u) - 8.0f * xc * xc / (u * u * u)) / (rc * rc) : 0.f;
}
// atom i against every other atom j of its molecule (all pairs, or inside `cut` with the reaction-field terms and the triclinic
// minimum image z -> y -> x): e_atom = sum_j h(d) S_ij, S = sum_q wq c_i c_j / sum wq; with the tangent along v, the adjoint of the
// charges g_q[i] = sum_j 2 h wq c_j (seed 1 on every atom's energy: a pair counts for both of its atoms) and the position gradient
// g_pos[i] = sum_j 2 S h' r_ij, all with tangents.  scale = COULOMB_FACTOR.
HVP_FN void coulomb_atom_dual(int i, int QC, const int* mstart, const int* mend, const int64_t* batch, const float* pos, const float* v,
                              const float* box_all, int box_per_mol, const float* ch, const float* ch_t, const float* wq, float wsum, float cut, float eps,
                              float scale, float* e_atom, float* e_atom_t, float* g_q, float* g_q_t, float* g_pos, float* hv) {
  const int m = (int)batch[i];
  const float* box = box_all ? box_all + (box_per_mol ? (int64_t)m * 9 : 0) : nullptr;
  float ea = 0.f, eat = 0.f, gp[3] = {0.f, 0.f, 0.f}, gpt[3] = {0.f, 0.f, 0.f};
  for (int q = 0; q < QC; ++q) g_q[(int64_t)i * QC + q] = g_q_t[(int64_t)i * QC + q] = 0.f;
  const float k_rf = cut > 0.f ? (1.0f / (cut * cut * cut)) * (eps - 1.0f) / (2.0f * eps + 1.0f) : 0.f;
  const float c_rf = cut > 0.f ? (1.0f / cut) * (3.0f * eps) / (2.0f * eps + 1.0f) : 0.f;
  for (int j = mstart[m]; j < mend[m]; ++j) {
    if (j == i) continue;
    float dl[3] = {pos[i * 3] - pos[j * 3], pos[i * 3 + 1] - pos[j * 3 + 1], pos[i * 3 + 2] - pos[j * 3 + 2]};
    if (box) {
      float s3 = -roundf(dl[2] / box[8]);
      dl[0] += s3 * box[6]; dl[1] += s3 * box[7]; dl[2] += s3 * box[8];
      float s2 = -roundf(dl[1] / box[4]);
      dl[0] += s2 * box[3]; dl[1] += s2 * box[4];
      float s1 = -roundf(dl[0] / box[0]);
      dl[0] += s1 * box[0];
    }
    const float d2 = dl[0] * dl[0] + dl[1] * dl[1] + dl[2] * dl[2];
    if (cut > 0.f && !(d2 < cut * cut)) continue;
    const float d = sqrtf(d2), inv = 1.0f / d;
    float r[3], rt[3], dt = 0.f;
    const float e_[3] = {v[i * 3] - v[j * 3], v[i * 3 + 1] - v[j * 3 + 1], v[i * 3 + 2] - v[j * 3 + 2]};
    for (int x = 0; x < 3; ++x) {
      r[x] = dl[x] * inv;
      dt += r[x] * e_[x];
    }
    for (int x = 0; x < 3; ++x) rt[x] = (e_[x] - r[x] * dt) * inv;
    float g0 = inv, g1 = -inv * inv, g2 = 2.0f * inv * inv * inv;
    if (cut > 0.f) {
      g0 += k_rf * d2 - c_rf;
      g1 += 2.0f * k_rf * d;
      g2 += 2.0f * k_rf;
    }
    float f, f1, f2;
    exp_cutoff_d2(d, 4.6f, f, f1, f2);
    const float h0 = scale * (1.0f - f) * g0, h1 = scale * (-f1 * g0 + (1.0f - f) * g1),
                h2 = scale * (-f2 * g0 - 2.0f * f1 * g1 + (1.0f - f) * g2);
    float S = 0.f, St = 0.f;
    for (int q = 0; q < QC; ++q) {
      const float ci = ch[(int64_t)i * QC + q], cit = ch_t[(int64_t)i * QC + q], cj = ch[(int64_t)j * QC + q], cjt = ch_t[(int64_t)j * QC + q];
      const float w = wq[q] / wsum;
      S += ci * cj * w;
      St += (cit * cj + ci * cjt) * w;
      g_q[(int64_t)i * QC + q] += 2.0f * h0 * cj * w;
      g_q_t[(int64_t)i * QC + q] += 2.0f * (h1 * dt * cj + h0 * cjt) * w;
    }
    ea += h0 * S;
    eat += h1 * dt * S + h0 * St;
    const float gd = 2.0f * S * h1, gdt = 2.0f * (St * h1 + S * h2 * dt);
    for (int x = 0; x < 3; ++x) {
      gp[x] += gd * r[x];
      gpt[x] += gdt * r[x] + gd * rt[x];
    }
  }
  e_atom[i] = ea;
  e_atom_t[i] = eat;
  for (int x = 0; x < 3; ++x) {
    g_pos[i * 3 + x] = gp[x];
    hv[i * 3 + x] = gpt[x];
  }
}

// dst[n, off + q] += src[n, q]     (a [N][qd] block into a column block of a [N][ld] buffer)      ; y += a x
HVP_FN void add_cols(int64_t idx, int qd, const float* src, float* dst, int ld, int off) {
  const int64_t n = idx / qd, q = idx - n * qd;
  dst[n * ld + off + q] += src[idx];
}
HVP_FN void axpy1(int64_t i, float a, const float* x, float* y) { y[i] += a * x[i]; }
HVP_FN void scale1(int64_t i, float a, const float* x, float* y) { y[i] = a * x[i]; }

}  // namespace hvp
}  // namespace tn
