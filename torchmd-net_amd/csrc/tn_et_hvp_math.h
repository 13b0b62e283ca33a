// Per-element arithmetic of the analytic second-order pass of the Equivariant Transformer (force-matching training):
//     d/d theta and d/d pos of  s = v . d(sum_m E_m)/d pos      (reference: autograd twice over torchmd_et.py:188-426,
//                                                               models/utils.py:83-117, 583-655; model.py:618-628)
// as the forward-mode tangent, along v, of the hand-written forward + reverse program (specification: oracle/et_second_order.py,
// pinned to autograd-of-autograd).  Same construction and the same rules as tn_hvp_math.h: every function is the work of ONE
// logical thread on raw pointers, __host__ __device__, and tests/hvp_host.hip runs the same bodies in host loops.
//
// Layouts (the engine's packed ones, tn_et_api.hip):
//   qkv [N][5F] = q | k | vx | v1 | v2 (value thirds as three contiguous blocks)      vec [N][3][F]      vp [N][3][3F] = vec1 | vec2 | vec3
//   dkv [P + 1][Wd] = dk | dvx | dv1 | dv2 (the parts the model has; ok = offset of dk or -1, ov = offset of dvx or -1), per PAIR
//   o [N][3F] = o1 | o2 | o3      u12 [N][3][U = F + F/2] = u1 | u2      y [N][F] = ys | gate      hcat [N][2F] = xf | n1      hcat2 [N][F] = xs | n2
// Edge (t <- s) in row t of the symmetric CSR: unit vector r = (pos_s - pos_t) / d = -esign[e] * prhat[pair(e)], 0 for the self edge.
#pragma once
#include "tn_hvp_math.h"

namespace tn {
namespace hvp {

// ------------------------------------------------------------------------------------------------ embeddings
// Wn = e_n C(d) per pair row (neighbour-embedding filter, models/utils.py:100-106): i over (P + 1) * F
HVP_FN void rowscale_dual(int64_t i, int W, const float* e, const float* e_t, const float* C, const float* C_t, float* o, float* o_t) {
  const int64_t p = i / W;
  o[i] = e[i] * C[p];
  o_t[i] = e_t[i] * C[p] + e[i] * C_t[p];
}
// xcat[i] = ( emb[z_i] | sum_{j in row i, j != i} Wn[pair] embN[z_j] )   [N][2F]
HVP_FN void et_nbr_embed_dual(int i, int f, int F, int P, const int* rowptr, const int* col, const int* epair, const float* esign,
                              const int64_t* z, const float* emb, const float* embN, const float* Wn, const float* Wn_t, float* xcat,
                              float* xcat_t) {
  float s = 0.f, st = 0.f;
  for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) {
    const int p = epair[e];
    if (esign[e] == 0.f || p >= P) continue;
    const float en = embN[z[col[e]] * F + f];
    s += Wn[(int64_t)p * F + f] * en;
    st += Wn_t[(int64_t)p * F + f] * en;
  }
  const int64_t o = (int64_t)i * 2 * F + f;
  xcat[o] = emb[z[i] * F + f];
  xcat_t[o] = 0.f;
  xcat[o + F] = s;
  xcat_t[o + F] = st;
}
// x0 = emb[z] without the neighbour embedding
HVP_FN void et_embed_dual(int i, int f, int F, const int64_t* z, const float* emb, float* x, float* x_t) {
  x[(int64_t)i * F + f] = emb[z[i] * F + f];
  x_t[(int64_t)i * F + f] = 0.f;
}

// ------------------------------------------------------------------------------------------------ attention, forward
struct EtAttn {  // operands of the attention sweeps of one layer (all with their tangents)
  const float *qkv, *qkv_t, *vec, *vec_t, *dkv, *dkv_t, *C, *C_t, *prhat, *rhat_t;
  int F, hd, Wd, ok, ov, vc, P;  // vc: the cutoff scales the values (vector_cutoff) instead of the attention weights
};
// The attention sweeps in LANE-GROUP form: a logical thread holds `nc` channels c0 .. c0 + nc - 1 of ONE head and red(v) is the sum of
// v over the head's channels.  Kernels: nc = 1 (the lane's own channel), red = sum over the head's hd lanes (a block = an atom, so
// the edge loop is uniform); host harness: nc = hd (the whole head in one thread), red = identity.  Same statements in both; NC is
// the capacity of the per-channel arrays.  Accumulators live in registers and every output is written once.
constexpr int ET_MAXHD = 64;
struct EtRedOne {
  HVP_FN float operator()(float v) const { return v; }
};
// channel c's term of the attention logit of edge (t <- s) and of its tangent: a = sum_{c in h} q_t k_s dk
HVP_FN void et_edge_weight_c(const EtAttn& A_, int t, int s, int p, int c, float& a, float& a_t) {
  const int F = A_.F, F5 = 5 * A_.F;
  const float q = A_.qkv[(int64_t)t * F5 + c], qt = A_.qkv_t[(int64_t)t * F5 + c];
  const float k = A_.qkv[(int64_t)s * F5 + F + c], kt = A_.qkv_t[(int64_t)s * F5 + F + c];
  const float dk = A_.ok >= 0 ? A_.dkv[(int64_t)p * A_.Wd + A_.ok + c] : 1.0f;
  const float dkt = A_.ok >= 0 ? A_.dkv_t[(int64_t)p * A_.Wd + A_.ok + c] : 0.0f;
  a += q * k * dk;
  a_t += qt * k * dk + q * kt * dk + q * k * dkt;
}
// value filters of channel c (three thirds) and the scaled values s_j = v_j[s] cv f_j
HVP_FN void et_edge_values(const EtAttn& A_, int s, int p, int c, float cv, float cvt, float sv[3], float svt[3], float v3[3], float v3t[3],
                           float f3[3], float f3t[3]) {
  const int F = A_.F, F5 = 5 * A_.F;
  for (int j = 0; j < 3; ++j) {
    v3[j] = A_.qkv[(int64_t)s * F5 + (2 + j) * F + c];
    v3t[j] = A_.qkv_t[(int64_t)s * F5 + (2 + j) * F + c];
    f3[j] = A_.ov >= 0 ? A_.dkv[(int64_t)p * A_.Wd + A_.ov + j * F + c] : 1.0f;
    f3t[j] = A_.ov >= 0 ? A_.dkv_t[(int64_t)p * A_.Wd + A_.ov + j * F + c] : 0.0f;
    sv[j] = v3[j] * cv * f3[j];
    svt[j] = v3t[j] * cv * f3[j] + v3[j] * cvt * f3[j] + v3[j] * cv * f3t[j];
  }
}
HVP_FN void et_edge_geometry(const EtAttn& A_, int p, float sg, float r[3], float rt[3], float& cv, float& cvt, float& ca, float& cat) {
  const bool self = sg == 0.f || p >= A_.P;
  for (int x = 0; x < 3; ++x) {
    r[x] = self ? 0.f : -sg * A_.prhat[p * 3 + x];
    rt[x] = self ? 0.f : -sg * A_.rhat_t[p * 3 + x];
  }
  const float c = A_.C[p], ct = A_.C_t[p];
  cv = A_.vc ? c : 1.0f;
  cvt = A_.vc ? ct : 0.0f;
  ca = A_.vc ? 1.0f : c;
  cat = A_.vc ? 0.0f : ct;
}
// target atom t, the channels c0 .. of one head: xagg[t, c] = sum_e s_x A ;  vagg[t, :, c] = sum_e vec[s] s_1 + s_2 r   (torchmd_et.py:368-426)
template <int NC, class Red>
HVP_FN void et_attn_fwd_dual_g(int t, int c0, int nc, Red red, const EtAttn& A_, const int* rowptr, const int* col, const int* epair,
                               const float* esign, float* xagg, float* xagg_t, float* vagg, float* vagg_t) {
  const int F = A_.F;
  float xa[NC], xat[NC], va[NC][3], vat[NC][3];
  for (int i = 0; i < nc; ++i) {
    xa[i] = xat[i] = 0.f;
    for (int x = 0; x < 3; ++x) va[i][x] = vat[i][x] = 0.f;
  }
  for (int e = rowptr[t]; e < rowptr[t + 1]; ++e) {
    const int s = col[e], p = epair[e];
    float r[3], rt[3], cv, cvt, ca, cat, a = 0.f, a_t = 0.f;
    et_edge_geometry(A_, p, esign[e], r, rt, cv, cvt, ca, cat);
    for (int i = 0; i < nc; ++i) et_edge_weight_c(A_, t, s, p, c0 + i, a, a_t);
    a = red(a);
    a_t = red(a_t);
    const float Aw = silu0(a) * ca, Awt = silu1(a) * a_t * ca + silu0(a) * cat;
    for (int i = 0; i < nc; ++i) {
      const int c = c0 + i;
      float sv[3], svt[3], v3[3], v3t[3], f3[3], f3t[3];
      et_edge_values(A_, s, p, c, cv, cvt, sv, svt, v3, v3t, f3, f3t);
      xa[i] += sv[0] * Aw;
      xat[i] += svt[0] * Aw + sv[0] * Awt;
      for (int x = 0; x < 3; ++x) {
        const float ve = A_.vec[((int64_t)s * 3 + x) * F + c], vet = A_.vec_t[((int64_t)s * 3 + x) * F + c];
        va[i][x] += ve * sv[1] + sv[2] * r[x];
        vat[i][x] += vet * sv[1] + ve * svt[1] + svt[2] * r[x] + sv[2] * rt[x];
      }
    }
  }
  for (int i = 0; i < nc; ++i) {
    const int c = c0 + i;
    xagg[(int64_t)t * F + c] = xa[i];
    xagg_t[(int64_t)t * F + c] = xat[i];
    for (int x = 0; x < 3; ++x) {
      vagg[((int64_t)t * 3 + x) * F + c] = va[i][x];
      vagg_t[((int64_t)t * 3 + x) * F + c] = vat[i][x];
    }
  }
}
HVP_FN void et_attn_fwd_dual(int t, int h, const EtAttn& A_, const int* rowptr, const int* col, const int* epair, const float* esign,
                             float* xagg, float* xagg_t, float* vagg, float* vagg_t) {  // one thread = a whole head (host harness)
  et_attn_fwd_dual_g<ET_MAXHD>(t, h * A_.hd, A_.hd, EtRedOne{}, A_, rowptr, col, epair, esign, xagg, xagg_t, vagg, vagg_t);
}
// x_new = x + vdot o2 + o3, vec_new = vec + vec3 o1 + vagg, vdot = sum_a vec1 vec2                      (torchmd_et.py:345-366)
HVP_FN void et_update_dual(int n, int f, int F, const float* x, const float* x_t, const float* vec, const float* vec_t, const float* vp,
                           const float* vp_t, const float* o, const float* o_t, const float* vagg, const float* vagg_t, float* xn,
                           float* xn_t, float* vecn, float* vecn_t, float* vdot, float* vdot_t) {
  const int F3 = 3 * F;
  float vd = 0.f, vdt = 0.f;
  for (int a = 0; a < 3; ++a) {
    const int64_t b = ((int64_t)n * 3 + a) * F3 + f;
    vd += vp[b] * vp[b + F];
    vdt += vp_t[b] * vp[b + F] + vp[b] * vp_t[b + F];
  }
  const int64_t on = (int64_t)n * F3 + f, xi = (int64_t)n * F + f;
  const float o1 = o[on], o2 = o[on + F], o3 = o[on + 2 * F], o1t = o_t[on], o2t = o_t[on + F], o3t = o_t[on + 2 * F];
  vdot[xi] = vd;
  vdot_t[xi] = vdt;
  xn[xi] = x[xi] + vd * o2 + o3;
  xn_t[xi] = x_t[xi] + vdt * o2 + vd * o2t + o3t;
  for (int a = 0; a < 3; ++a) {
    const int64_t vi = ((int64_t)n * 3 + a) * F + f, b = ((int64_t)n * 3 + a) * F3 + 2 * F + f;
    vecn[vi] = vec[vi] + vp[b] * o1 + vagg[vi];
    vecn_t[vi] = vec_t[vi] + vp_t[b] * o1 + vp[b] * o1t + vagg_t[vi];
  }
}

// ------------------------------------------------------------------------------------------------ head, forward
// out[n, off + f] = | u[n, :, f] |  (0-safe, models/utils.py:640-646); optionally out[n, f] = x[n, f] for f < Fx (the scalar half)
HVP_FN void et_cat_norm_dual(int n, int f, const float* x, const float* x_t, int Fx, const float* u, const float* u_t, int ldu, int Fn,
                             int ldo, float* out, float* out_t) {
  if (x && f < Fx) {
    out[(int64_t)n * ldo + f] = x[(int64_t)n * Fx + f];
    out_t[(int64_t)n * ldo + f] = x_t[(int64_t)n * Fx + f];
  }
  if (f < Fn) {
    float s = 0.f, st = 0.f;
    for (int a = 0; a < 3; ++a) {
      const float w = u[((int64_t)n * 3 + a) * ldu + f], wt = u_t[((int64_t)n * 3 + a) * ldu + f];
      s += w * w;
      st += w * wt;
    }
    const float nn = sqrtf(s);
    out[(int64_t)n * ldo + Fx + f] = nn;
    out_t[(int64_t)n * ldo + Fx + f] = nn > 0.f ? st / nn : 0.f;
  }
}
// xs = silu(ys) -> hcat2[n, f], vq[n, :, f] = gate u2          (f < F/2; y [N][F] = ys | gate; u2 = u12[..., F + f])
HVP_FN void et_head_mid_dual(int n, int f, int F2, const float* y, const float* y_t, const float* u2, const float* u2_t, int ldu,
                             float* hcat2, float* hcat2_t, float* vq, float* vq_t) {
  const int F = 2 * F2;
  const float ys = y[(int64_t)n * F + f], yst = y_t[(int64_t)n * F + f], gate = y[(int64_t)n * F + F2 + f], gatet = y_t[(int64_t)n * F + F2 + f];
  hcat2[(int64_t)n * F + f] = silu0(ys);
  hcat2_t[(int64_t)n * F + f] = silu1(ys) * yst;
  for (int a = 0; a < 3; ++a) {
    const float w = u2[((int64_t)n * 3 + a) * ldu + f], wt = u2_t[((int64_t)n * 3 + a) * ldu + f];
    vq[((int64_t)n * 3 + a) * F2 + f] = gate * w;
    vq_t[((int64_t)n * 3 + a) * F2 + f] = gatet * w + gate * wt;
  }
}

// ------------------------------------------------------------------------------------------------ head, reverse
// g_u[n, a, f] = g_n[n, f] u / |u|   (0 where |u| = 0)
HVP_FN void et_norm_bwd_dual(int n, int f, const float* g_n, const float* g_n_t, int ldg, const float* u, const float* u_t, int ldu,
                             float* g_u, float* g_u_t, int ldgu) {
  float s = 0.f, st = 0.f, w[3], wt[3];
  for (int a = 0; a < 3; ++a) {
    w[a] = u[((int64_t)n * 3 + a) * ldu + f];
    wt[a] = u_t[((int64_t)n * 3 + a) * ldu + f];
    s += w[a] * w[a];
    st += w[a] * wt[a];
  }
  const float nn = sqrtf(s), nt = nn > 0.f ? st / nn : 0.f;
  const float g = g_n[(int64_t)n * ldg + f], gt = g_n_t[(int64_t)n * ldg + f];
  for (int a = 0; a < 3; ++a) {
    const float d = nn > 0.f ? w[a] / nn : 0.f, dt = nn > 0.f ? (wt[a] - d * nt) / nn : 0.f;
    g_u[((int64_t)n * 3 + a) * ldgu + f] = g * d;
    g_u_t[((int64_t)n * 3 + a) * ldgu + f] = gt * d + g * dt;
  }
}
// g_y = ( g_xs silu'(ys) | sum_a g_vq u2 ), g_u2 = g_vq gate                  (f < F/2; g_h2 [N][F] = g_xs | g_n2)
HVP_FN void et_head_mid_bwd_dual(int n, int f, int F2, const float* y, const float* y_t, const float* u2, const float* u2_t, int ldu,
                                 const float* g_h2, const float* g_h2_t, const float* g_vq, const float* g_vq_t, float* g_y, float* g_y_t,
                                 float* g_u2, float* g_u2_t, int ldgu) {
  const int F = 2 * F2;
  const float ys = y[(int64_t)n * F + f], yst = y_t[(int64_t)n * F + f], gate = y[(int64_t)n * F + F2 + f], gatet = y_t[(int64_t)n * F + F2 + f];
  const float gx = g_h2[(int64_t)n * F + f], gxt = g_h2_t[(int64_t)n * F + f];
  g_y[(int64_t)n * F + f] = gx * silu1(ys);
  g_y_t[(int64_t)n * F + f] = gxt * silu1(ys) + gx * silu2(ys) * yst;
  float gg = 0.f, ggt = 0.f;
  for (int a = 0; a < 3; ++a) {
    const int64_t i = ((int64_t)n * 3 + a) * F2 + f;
    const float w = u2[((int64_t)n * 3 + a) * ldu + f], wt = u2_t[((int64_t)n * 3 + a) * ldu + f];
    gg += g_vq[i] * w;
    ggt += g_vq_t[i] * w + g_vq[i] * wt;
    g_u2[((int64_t)n * 3 + a) * ldgu + f] = g_vq[i] * gate;
    g_u2_t[((int64_t)n * 3 + a) * ldgu + f] = g_vq_t[i] * gate + g_vq[i] * gatet;
  }
  g_y[(int64_t)n * F + F2 + f] = gg;
  g_y_t[(int64_t)n * F + F2 + f] = ggt;
}

// ------------------------------------------------------------------------------------------------ attention layer, reverse
// adjoint of et_update: g_o = ( sum_a g_vec vec3 | g_x vdot | g_x ), g_vp = ( g_vdot vec2 | g_vdot vec1 | g_vec o1 ), g_vdot = g_x o2
HVP_FN void et_update_bwd_dual(int n, int f, int F, const float* g_x, const float* g_x_t, const float* g_vec, const float* g_vec_t,
                               const float* vp, const float* vp_t, const float* o, const float* o_t, const float* vdot, const float* vdot_t,
                               float* g_o, float* g_o_t, float* g_vp, float* g_vp_t) {
  const int F3 = 3 * F;
  const int64_t xi = (int64_t)n * F + f, on = (int64_t)n * F3 + f;
  const float gx = g_x[xi], gxt = g_x_t[xi], o1 = o[on], o1t = o_t[on], o2 = o[on + F], o2t = o_t[on + F];
  const float gvd = gx * o2, gvdt = gxt * o2 + gx * o2t;
  float go1 = 0.f, go1t = 0.f;
  for (int a = 0; a < 3; ++a) {
    const int64_t vi = ((int64_t)n * 3 + a) * F + f, b = ((int64_t)n * 3 + a) * F3 + f;
    const float gv = g_vec[vi], gvt = g_vec_t[vi];
    go1 += gv * vp[b + 2 * F];
    go1t += gvt * vp[b + 2 * F] + gv * vp_t[b + 2 * F];
    g_vp[b] = gvd * vp[b + F];
    g_vp_t[b] = gvdt * vp[b + F] + gvd * vp_t[b + F];
    g_vp[b + F] = gvd * vp[b];
    g_vp_t[b + F] = gvdt * vp[b] + gvd * vp_t[b];
    g_vp[b + 2 * F] = gv * o1;
    g_vp_t[b + 2 * F] = gvt * o1 + gv * o1t;
  }
  g_o[on] = go1;
  g_o_t[on] = go1t;
  g_o[on + F] = gx * vdot[xi];
  g_o_t[on + F] = gxt * vdot[xi] + gx * vdot_t[xi];
  g_o[on + 2 * F] = gx;
  g_o_t[on + 2 * F] = gxt;
}

// the adjoint quantities of one edge (t <- s) and head h that both reverse sweeps need: everything up to g_a (per head) and the
// head quantities of one edge in the reverse sweeps
struct EtEdgeAdj {
  float a, a_t, Aw, Awt, g_a, g_a_t, g_ca, g_ca_t;  // g_ca: this head's part of the adjoint of the attention-side cutoff factor
};
// from the head sums a = sum_c q k dk and g_A = sum_c g_xagg[t, c] s_x (and their tangents)
HVP_FN void et_edge_adjoint_of(float a, float at, float gA, float gAt, float ca, float cat, EtEdgeAdj& E_) {
  E_.a = a;
  E_.a_t = at;
  E_.Aw = silu0(a) * ca;
  E_.Awt = silu1(a) * at * ca + silu0(a) * cat;
  E_.g_a = gA * silu1(a) * ca;
  E_.g_a_t = gAt * silu1(a) * ca + gA * silu2(a) * at * ca + gA * silu1(a) * cat;
  E_.g_ca = gA * silu0(a);
  E_.g_ca_t = gAt * silu0(a) + gA * silu1(a) * at;
}
// the head sums of edge (t <- s) over this thread's channels (values kept per channel for the statements that follow), then reduced
template <int NC, class Red>
HVP_FN void et_edge_head(const EtAttn& A_, int t, int s, int p, int c0, int nc, Red red, float cv, float cvt, float ca, float cat,
                         const float* g_xagg, const float* g_xagg_t, float sv[][3], float svt[][3], float v3[][3], float v3t[][3],
                         float f3[][3], float f3t[][3], EtEdgeAdj& E_) {
  const int F = A_.F;
  float a = 0.f, at = 0.f, gA = 0.f, gAt = 0.f;
  for (int i = 0; i < nc; ++i) {
    const int c = c0 + i;
    et_edge_weight_c(A_, t, s, p, c, a, at);
    et_edge_values(A_, s, p, c, cv, cvt, sv[i], svt[i], v3[i], v3t[i], f3[i], f3t[i]);
    gA += g_xagg[(int64_t)t * F + c] * sv[i][0];
    gAt += g_xagg_t[(int64_t)t * F + c] * sv[i][0] + g_xagg[(int64_t)t * F + c] * svt[i][0];
  }
  a = red(a);
  at = red(at);
  gA = red(gA);
  gAt = red(gAt);
  et_edge_adjoint_of(a, at, gA, gAt, ca, cat, E_);
}
// per-channel adjoints of the scaled values of edge (t <- s): g_s = ( g_xagg[t] A , sum_x g_vagg[t, x] vec[s, x] , sum_x g_vagg[t, x] r_x )
HVP_FN void et_edge_gs(const EtAttn& A_, int t, int s, int c, const float r[3], const float rt[3], float Aw, float Awt, const float* g_xagg,
                       const float* g_xagg_t, const float* g_vagg, const float* g_vagg_t, float gs[3], float gst[3]) {
  const int F = A_.F;
  const float gx = g_xagg[(int64_t)t * F + c], gxt = g_xagg_t[(int64_t)t * F + c];
  gs[0] = gx * Aw;
  gst[0] = gxt * Aw + gx * Awt;
  gs[1] = gst[1] = gs[2] = gst[2] = 0.f;
  for (int x = 0; x < 3; ++x) {
    const float gv = g_vagg[((int64_t)t * 3 + x) * F + c], gvt = g_vagg_t[((int64_t)t * 3 + x) * F + c];
    const float ve = A_.vec[((int64_t)s * 3 + x) * F + c], vet = A_.vec_t[((int64_t)s * 3 + x) * F + c];
    gs[1] += gv * ve;
    gst[1] += gvt * ve + gv * vet;
    gs[2] += gv * r[x];
    gst[2] += gvt * r[x] + gv * rt[x];
  }
}
// TARGET role, atom t, channels c0 .. of head h: g_q[t, c] over the row's edges; per directed edge the filter adjoints g_dk, g_dv
// (gq[dir][pair][Wd], self edge: selfq[t][Wd]) and this head's parts of the cutoff / unit-vector adjoints:
// slots[dir][pair][h][4] = (g_cutoff, g_r[3]), written by the head's `leader` thread
//   g_cutoff = g_cv (vector_cutoff: the channels' sum g_s v f) or g_ca;  g_r = sum_c g_vagg[t, :, c] s_2
template <int NC, class Red>
HVP_FN void et_attn_bwd_tgt_dual_g(int t, int h, int c0, int nc, bool leader, Red red, int H, const EtAttn& A_, const int* rowptr,
                                   const int* col, const int* epair, const float* esign, const float* g_xagg, const float* g_xagg_t,
                                   const float* g_vagg, const float* g_vagg_t, float* g_qkv, float* g_qkv_t, float* gq, float* gq_t,
                                   int64_t dir_stride, float* selfq, float* selfq_t, float* slots, float* slots_t,
                                   int64_t slot_dir_stride) {
  const int F = A_.F, F5 = 5 * A_.F, Wd = A_.Wd;
  float gqa[NC], gqat[NC];
  for (int i = 0; i < nc; ++i) gqa[i] = gqat[i] = 0.f;
  for (int e = rowptr[t]; e < rowptr[t + 1]; ++e) {
    const int s = col[e], p = epair[e];
    const float sg = esign[e];
    const bool self = sg == 0.f || p >= A_.P;
    float r[3], rt[3], cv, cvt, ca, cat;
    et_edge_geometry(A_, p, sg, r, rt, cv, cvt, ca, cat);
    float sv[NC][3], svt[NC][3], v3[NC][3], v3t[NC][3], f3[NC][3], f3t[NC][3];
    EtEdgeAdj E_;
    et_edge_head<NC>(A_, t, s, p, c0, nc, red, cv, cvt, ca, cat, g_xagg, g_xagg_t, sv, svt, v3, v3t, f3, f3t, E_);
    float* fo = self ? selfq + (int64_t)t * Wd : gq + (sg > 0.f ? 0 : dir_stride) + (int64_t)p * Wd;
    float* fot = self ? selfq_t + (int64_t)t * Wd : gq_t + (sg > 0.f ? 0 : dir_stride) + (int64_t)p * Wd;
    float gcut = 0.f, gcutt = 0.f, gr[3] = {0.f, 0.f, 0.f}, grt[3] = {0.f, 0.f, 0.f};  // this thread's channels' part
    for (int i = 0; i < nc; ++i) {
      const int c = c0 + i;
      const float q = A_.qkv[(int64_t)t * F5 + c], qt = A_.qkv_t[(int64_t)t * F5 + c];
      const float k = A_.qkv[(int64_t)s * F5 + F + c], kt = A_.qkv_t[(int64_t)s * F5 + F + c];
      const float dk = A_.ok >= 0 ? A_.dkv[(int64_t)p * Wd + A_.ok + c] : 1.0f, dkt = A_.ok >= 0 ? A_.dkv_t[(int64_t)p * Wd + A_.ok + c] : 0.0f;
      gqa[i] += E_.g_a * k * dk;
      gqat[i] += E_.g_a_t * k * dk + E_.g_a * kt * dk + E_.g_a * k * dkt;
      if (A_.ok >= 0) {
        fo[A_.ok + c] = E_.g_a * q * k;
        fot[A_.ok + c] = E_.g_a_t * q * k + E_.g_a * qt * k + E_.g_a * q * kt;
      }
      float gs[3], gst[3];
      et_edge_gs(A_, t, s, c, r, rt, E_.Aw, E_.Awt, g_xagg, g_xagg_t, g_vagg, g_vagg_t, gs, gst);
      for (int j = 0; j < 3; ++j) {
        if (A_.ov >= 0) {
          fo[A_.ov + j * F + c] = gs[j] * v3[i][j] * cv;
          fot[A_.ov + j * F + c] = gst[j] * v3[i][j] * cv + gs[j] * v3t[i][j] * cv + gs[j] * v3[i][j] * cvt;
        }
        if (A_.vc) {
          gcut += gs[j] * v3[i][j] * f3[i][j];
          gcutt += gst[j] * v3[i][j] * f3[i][j] + gs[j] * v3t[i][j] * f3[i][j] + gs[j] * v3[i][j] * f3t[i][j];
        }
      }
      for (int x = 0; x < 3; ++x) {
        const float gv = g_vagg[((int64_t)t * 3 + x) * F + c], gvt = g_vagg_t[((int64_t)t * 3 + x) * F + c];
        gr[x] += gv * sv[i][2];
        grt[x] += gvt * sv[i][2] + gv * svt[i][2];
      }
    }
    gcut = red(gcut);
    gcutt = red(gcutt);
    for (int x = 0; x < 3; ++x) {
      gr[x] = red(gr[x]);
      grt[x] = red(grt[x]);
    }
    if (!A_.vc) {
      gcut = E_.g_ca;
      gcutt = E_.g_ca_t;
    }
    if (!self && leader) {
      const int64_t o = (sg > 0.f ? 0 : slot_dir_stride) + ((int64_t)p * H + h) * 4;
      slots[o] = gcut;
      slots_t[o] = gcutt;
      for (int x = 0; x < 3; ++x) {
        slots[o + 1 + x] = gr[x];
        slots_t[o + 1 + x] = grt[x];
      }
    }
  }
  for (int i = 0; i < nc; ++i) {
    g_qkv[(int64_t)t * F5 + c0 + i] = gqa[i];
    g_qkv_t[(int64_t)t * F5 + c0 + i] = gqat[i];
  }
}
HVP_FN void et_attn_bwd_tgt_dual(int t, int h, int H, const EtAttn& A_, const int* rowptr, const int* col, const int* epair, const float* esign,
                                 const float* g_xagg, const float* g_xagg_t, const float* g_vagg, const float* g_vagg_t, float* g_qkv,
                                 float* g_qkv_t, float* gq, float* gq_t, int64_t dir_stride, float* selfq, float* selfq_t, float* slots,
                                 float* slots_t, int64_t slot_dir_stride) {  // one thread = a whole head (host harness)
  et_attn_bwd_tgt_dual_g<ET_MAXHD>(t, h, h * A_.hd, A_.hd, true, EtRedOne{}, H, A_, rowptr, col, epair, esign, g_xagg, g_xagg_t, g_vagg,
                                   g_vagg_t, g_qkv, g_qkv_t, gq, gq_t, dir_stride, selfq, selfq_t, slots, slots_t, slot_dir_stride);
}
// SOURCE role, atom s, channels c0 .. of one head: over the edges (t <- s) of the row (the graph is symmetric: row s lists the targets t):
//   g_k[s, c] += g_a q_t dk ;  g_v_j[s, c] += g_s_j cv f_j ;  g_vec[s, :, c] += g_vagg[t, :, c] s_1
template <int NC, class Red>
HVP_FN void et_attn_bwd_src_dual_g(int s, int c0, int nc, Red red, const EtAttn& A_, const int* rowptr, const int* col, const int* epair,
                                   const float* esign, const float* g_xagg, const float* g_xagg_t, const float* g_vagg,
                                   const float* g_vagg_t, float* g_qkv, float* g_qkv_t, float* g_vec_in, float* g_vec_in_t) {
  const int F = A_.F, F5 = 5 * A_.F;
  float gk[NC], gkt[NC], gv3[NC][3], gv3t[NC][3], gve[NC][3], gvet[NC][3];
  for (int i = 0; i < nc; ++i) {
    gk[i] = gkt[i] = 0.f;
    for (int x = 0; x < 3; ++x) gv3[i][x] = gv3t[i][x] = gve[i][x] = gvet[i][x] = 0.f;
  }
  for (int e = rowptr[s]; e < rowptr[s + 1]; ++e) {
    const int t = col[e], p = epair[e];
    // the edge (t <- s) seen from row s: its sign is the one of row t's entry, i.e. minus this row's
    float r[3], rt[3], cv, cvt, ca, cat;
    et_edge_geometry(A_, p, -esign[e], r, rt, cv, cvt, ca, cat);
    float sv[NC][3], svt[NC][3], v3[NC][3], v3t[NC][3], f3[NC][3], f3t[NC][3];
    EtEdgeAdj E_;
    et_edge_head<NC>(A_, t, s, p, c0, nc, red, cv, cvt, ca, cat, g_xagg, g_xagg_t, sv, svt, v3, v3t, f3, f3t, E_);
    for (int i = 0; i < nc; ++i) {
      const int c = c0 + i;
      const float q = A_.qkv[(int64_t)t * F5 + c], qt = A_.qkv_t[(int64_t)t * F5 + c];
      const float dk = A_.ok >= 0 ? A_.dkv[(int64_t)p * A_.Wd + A_.ok + c] : 1.0f, dkt = A_.ok >= 0 ? A_.dkv_t[(int64_t)p * A_.Wd + A_.ok + c] : 0.0f;
      gk[i] += E_.g_a * q * dk;
      gkt[i] += E_.g_a_t * q * dk + E_.g_a * qt * dk + E_.g_a * q * dkt;
      float gs[3], gst[3];
      et_edge_gs(A_, t, s, c, r, rt, E_.Aw, E_.Awt, g_xagg, g_xagg_t, g_vagg, g_vagg_t, gs, gst);
      for (int j = 0; j < 3; ++j) {
        gv3[i][j] += gs[j] * cv * f3[i][j];
        gv3t[i][j] += gst[j] * cv * f3[i][j] + gs[j] * cvt * f3[i][j] + gs[j] * cv * f3t[i][j];
      }
      for (int x = 0; x < 3; ++x) {
        const float gv = g_vagg[((int64_t)t * 3 + x) * F + c], gvt = g_vagg_t[((int64_t)t * 3 + x) * F + c];
        gve[i][x] += gv * sv[i][1];
        gvet[i][x] += gvt * sv[i][1] + gv * svt[i][1];
      }
    }
  }
  for (int i = 0; i < nc; ++i) {
    const int c = c0 + i;
    g_qkv[(int64_t)s * F5 + F + c] = gk[i];
    g_qkv_t[(int64_t)s * F5 + F + c] = gkt[i];
    for (int j = 0; j < 3; ++j) {
      g_qkv[(int64_t)s * F5 + (2 + j) * F + c] = gv3[i][j];
      g_qkv_t[(int64_t)s * F5 + (2 + j) * F + c] = gv3t[i][j];
    }
    for (int x = 0; x < 3; ++x) {
      g_vec_in[((int64_t)s * 3 + x) * F + c] = gve[i][x];
      g_vec_in_t[((int64_t)s * 3 + x) * F + c] = gvet[i][x];
    }
  }
}
HVP_FN void et_attn_bwd_src_dual(int s, int h, const EtAttn& A_, const int* rowptr, const int* col, const int* epair, const float* esign,
                                 const float* g_xagg, const float* g_xagg_t, const float* g_vagg, const float* g_vagg_t, float* g_qkv,
                                 float* g_qkv_t, float* g_vec_in, float* g_vec_in_t) {  // one thread = a whole head (host harness)
  et_attn_bwd_src_dual_g<ET_MAXHD>(s, h * A_.hd, A_.hd, EtRedOne{}, A_, rowptr, col, epair, esign, g_xagg, g_xagg_t, g_vagg, g_vagg_t, g_qkv,
                                   g_qkv_t, g_vec_in, g_vec_in_t);
}
// filter rows: adjoint of dkv per PAIR = the two directions' rows (self pair: the column sums over the atoms' self edges), through
// dkv = silu(ekv):  g_e = g_dkv silu'(ekv)                                                      (idx over (P + 1) * Wd)
HVP_FN void et_filter_gpre_dual(int64_t idx, int P, int Wd, const float* gq, const float* gq_t, int64_t dir_stride, const float* self_g,
                                const float* self_g_t, const float* ekv, const float* ekv_t, float* g_e, float* g_e_t) {
  const int p = (int)(idx / Wd), c = (int)(idx - (int64_t)p * Wd);
  float g, gt;
  if (p < P) {
    g = gq[idx] + gq[dir_stride + idx];
    gt = gq_t[idx] + gq_t[dir_stride + idx];
  } else {
    g = self_g[c];
    gt = self_g_t[c];
  }
  g_e[idx] = g * silu1(ekv[idx]);
  g_e_t[idx] = gt * silu1(ekv[idx]) + g * silu2(ekv[idx]) * ekv_t[idx];
}

// ------------------------------------------------------------------------------------------------ neighbour embedding, reverse
// g_xn = g_xcat[:, F:]; per directed edge (i, j != i): g_Wn[dir][pair, f] = g_xn[i] embN[z_j]; gZ[i, f] = sum_e g_xn[j] Wn[pair] (atom i
// as the neighbour j's source of embN)
HVP_FN void et_nbr_bwd_dual(int i, int f, int F, int P, const int* rowptr, const int* col, const int* epair, const float* esign,
                            const int64_t* z, const float* embN, const float* Wn, const float* Wn_t, const float* g_xcat,
                            const float* g_xcat_t, float* gq, float* gq_t, int64_t dir_stride, float* gZ_t) {
  const float gi = g_xcat[(int64_t)i * 2 * F + F + f], git = g_xcat_t[(int64_t)i * 2 * F + F + f];
  float zt = 0.f;
  for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) {
    const int j = col[e], p = epair[e];
    const float sg = esign[e];
    if (sg == 0.f || p >= P) continue;
    const int64_t o = (sg > 0.f ? 0 : dir_stride) + (int64_t)p * F + f;
    const float en = embN[z[j] * F + f];
    gq[o] = gi * en;
    gq_t[o] = git * en;
    const float gj = g_xcat[(int64_t)j * 2 * F + F + f], gjt = g_xcat_t[(int64_t)j * 2 * F + F + f];
    zt += gjt * Wn[(int64_t)p * F + f] + gj * Wn_t[(int64_t)p * F + f];
  }
  gZ_t[(int64_t)i * F + f] = zt;
}
// g_Wn per pair = both directions; g_en = g_Wn C (adjoint of the projection rows), with tangents            (idx over P * F)
HVP_FN void et_nbr_pair_dual(int64_t idx, int F, const float* gq, const float* gq_t, int64_t dir_stride, const float* C, const float* C_t,
                             float* g_Wn, float* g_Wn_t, float* g_en, float* g_en_t) {
  const int64_t p = idx / F;
  const float g = gq[idx] + gq[dir_stride + idx], gt = gq_t[idx] + gq_t[dir_stride + idx];
  g_Wn[idx] = g;
  g_Wn_t[idx] = gt;
  g_en[idx] = g * C[p];
  g_en_t[idx] = gt * C[p] + g * C_t[p];
}

// ------------------------------------------------------------------------------------------------ geometry
// out[p] (+)= sum_j x y ;  out_t[p] (+)= sum_j ( x_t y + x y_t )
HVP_FN void pair_rowdot2(int p, int W, const float* x, const float* x_t, const float* y, const float* y_t, int accumulate, float* out,
                         float* out_t) {
  const int64_t b = (int64_t)p * W;
  float s = 0.f, st = 0.f;
  for (int j = 0; j < W; ++j) {
    s += x[b + j] * y[b + j];
    st += x_t[b + j] * y[b + j] + x[b + j] * y_t[b + j];
  }
  out[p] = (accumulate ? out[p] : 0.f) + s;
  out_t[p] = (accumulate ? out_t[p] : 0.f) + st;
}
// pair p: the attention sweeps' slots (per layer, direction and head) -> g_cut (adjoint of the cutoff factor) and g_r (adjoint of the
// pair's unit vector prhat: the edge vectors are -esign prhat, direction 0 has esign = +1)
HVP_FN void et_pair_slots_dual(int p, int H, int nslot_sets, const float* slots, const float* slots_t, int64_t set_stride,
                               int64_t slot_dir_stride, float* g_cut, float* g_cut_t, float* g_rh, float* g_rh_t) {
  float c = 0.f, ct = 0.f, r[3] = {0.f, 0.f, 0.f}, rt[3] = {0.f, 0.f, 0.f};
  for (int k = 0; k < nslot_sets; ++k)
    for (int dir = 0; dir < 2; ++dir)
      for (int h = 0; h < H; ++h) {
        const int64_t o = (int64_t)k * set_stride + (dir ? slot_dir_stride : 0) + ((int64_t)p * H + h) * 4;
        const float sgn = dir ? 1.0f : -1.0f;  // r_e = -esign prhat
        c += slots[o];
        ct += slots_t[o];
        for (int x = 0; x < 3; ++x) {
          r[x] += sgn * slots[o + 1 + x];
          rt[x] += sgn * slots_t[o + 1 + x];
        }
      }
  g_cut[p] = c;
  g_cut_t[p] = ct;
  for (int x = 0; x < 3; ++x) {
    g_rh[p * 3 + x] = r[x];
    g_rh_t[p * 3 + x] = rt[x];
  }
}
// pair p < P: g_d = g_cut C' + g_dphi (the row-dots of the filter / neighbour-embedding adjoints with the distance tangents) and the
// geometry of neighbor_utils.py:11-46, differentiated: g_delta (value: minus the pair's force contribution) and g_delta_t
HVP_FN void et_geom_dual(int p, const float* pd, const float* prhat, const float* d_t, const float* rhat_t, const float* dC, const float* d2C,
                         const float* g_cut, const float* g_cut_t, const float* g_dphi, const float* g_dphi_t, const float* g_rh,
                         const float* g_rh_t, float* gdel, float* gdel_t) {
  const float d = pd[p], dt = d_t[p], inv = 1.0f / d;
  const float gd = g_cut[p] * dC[p] + g_dphi[p];
  const float gdt = g_cut_t[p] * dC[p] + g_cut[p] * d2C[p] * dt + g_dphi_t[p];
  float r[3], rt[3], a_ = 0.f, at_ = 0.f;
  for (int x = 0; x < 3; ++x) {
    r[x] = prhat[p * 3 + x];
    rt[x] = rhat_t[p * 3 + x];
    a_ += g_rh[p * 3 + x] * r[x];
    at_ += g_rh_t[p * 3 + x] * r[x] + g_rh[p * 3 + x] * rt[x];
  }
  for (int x = 0; x < 3; ++x) {
    const float t1 = (g_rh[p * 3 + x] - a_ * r[x]) * inv;
    const float t1t = (g_rh_t[p * 3 + x] - at_ * r[x] - a_ * rt[x]) * inv - t1 * dt * inv;
    gdel[p * 3 + x] = t1 + gd * r[x];
    gdel_t[p * 3 + x] = t1t + gdt * r[x] + gd * rt[x];
  }
}

}  // namespace hvp
}  // namespace tn
