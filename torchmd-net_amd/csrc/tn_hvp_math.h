// Per-element arithmetic of the analytic second-order pass (force-matching training, tn_hvp.hip):
//
//     s(theta) = v . d(sum_m E_m)/d pos ,   wanted: d s / d theta          (reference: autograd twice, model.py:618-628 and the
//                                                                            *_bwd_bwd kernels of warp_ops/)
//
// taken as the forward-mode tangent (direction v in the positions) of the hand-written E + F program: every forward
// intermediate a gets a tangent a_t, every adjoint g_a a tangent g_a_t, and a weight gradient g_W = g_y^T x becomes
// g_y_t^T x + g_y^T x_t.  The specification, statement by statement, is oracle/tensornet_second_order.py (pinned to
// autograd-of-autograd in fp64); the names below are its names.
//
// Every function here is the work of ONE logical thread (an (atom, channel), a pair row element, a row of a LayerNorm) on raw
// pointers, __host__ __device__ and free of wave intrinsics: the kernels of tn_hvp.hip are index arithmetic around them, and
// tests/hvp_host.hip (compiled host-only, test infrastructure) calls the same bodies in plain loops so that the arithmetic is
// checked against the oracle on a machine without a GPU.  The product never runs them on the host.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#define HVP_FN __host__ __device__ inline

namespace tn {
namespace hvp {

struct Mat3 {
  float a[3][3];
};

HVP_FN float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }
HVP_FN float silu0(float x) { return x * sigm(x); }
HVP_FN float silu1(float x) {
  const float s = sigm(x);
  return s * (1.0f + x * (1.0f - s));
}
HVP_FN float silu2(float x) {
  const float s = sigm(x);
  return s * (1.0f - s) * (2.0f + x * (1.0f - 2.0f * s));
}
HVP_FN int type_of(int c) { return c == 0 ? 0 : (c < 4 ? 1 : 2); }

// 9 irreducible components <-> 3x3 (tn_common.h conventions; reference tensornet.py:18-39, 84-103)
HVP_FN Mat3 compose(const float u[9]) {
  Mat3 m;
  m.a[0][0] = u[0] + u[4];
  m.a[0][1] = u[5] - u[3];
  m.a[0][2] = u[6] + u[2];
  m.a[1][0] = u[5] + u[3];
  m.a[1][1] = u[0] + u[7];
  m.a[1][2] = u[8] - u[1];
  m.a[2][0] = u[6] - u[2];
  m.a[2][1] = u[8] + u[1];
  m.a[2][2] = u[0] - u[4] - u[7];
  return m;
}
HVP_FN void decomp(const Mat3& m, float u[9]) {
  const float I = (m.a[0][0] + m.a[1][1] + m.a[2][2]) * (1.0f / 3.0f);
  u[0] = I;
  u[1] = 0.5f * (m.a[2][1] - m.a[1][2]);
  u[2] = 0.5f * (m.a[0][2] - m.a[2][0]);
  u[3] = 0.5f * (m.a[1][0] - m.a[0][1]);
  u[4] = m.a[0][0] - I;
  u[5] = 0.5f * (m.a[0][1] + m.a[1][0]);
  u[6] = 0.5f * (m.a[0][2] + m.a[2][0]);
  u[7] = m.a[1][1] - I;
  u[8] = 0.5f * (m.a[1][2] + m.a[2][1]);
}
HVP_FN void compose_T(const Mat3& g, float u[9]) {
  u[0] = g.a[0][0] + g.a[1][1] + g.a[2][2];
  u[1] = g.a[2][1] - g.a[1][2];
  u[2] = g.a[0][2] - g.a[2][0];
  u[3] = g.a[1][0] - g.a[0][1];
  u[4] = g.a[0][0] - g.a[2][2];
  u[5] = g.a[0][1] + g.a[1][0];
  u[6] = g.a[0][2] + g.a[2][0];
  u[7] = g.a[1][1] - g.a[2][2];
  u[8] = g.a[1][2] + g.a[2][1];
}
HVP_FN Mat3 decomp_T(const float g[9]) {
  Mat3 m;
  const float t = (g[0] - g[4] - g[7]) * (1.0f / 3.0f);
  m.a[0][0] = t + g[4];
  m.a[0][1] = 0.5f * (g[5] - g[3]);
  m.a[0][2] = 0.5f * (g[6] + g[2]);
  m.a[1][0] = 0.5f * (g[5] + g[3]);
  m.a[1][1] = t + g[7];
  m.a[1][2] = 0.5f * (g[8] - g[1]);
  m.a[2][0] = 0.5f * (g[6] - g[2]);
  m.a[2][1] = 0.5f * (g[8] + g[1]);
  m.a[2][2] = t;
  return m;
}
HVP_FN float quad(const float u[9]) {
  const float t = u[4] + u[7];
  return 3.0f * u[0] * u[0] + 2.0f * (u[1] * u[1] + u[2] * u[2] + u[3] * u[3]) + u[4] * u[4] + u[7] * u[7] + t * t +
         2.0f * (u[5] * u[5] + u[6] * u[6] + u[8] * u[8]);
}
HVP_FN void dquad(const float u[9], float g[9]) {  // linear in u
  g[0] = 6.0f * u[0];
  g[1] = 4.0f * u[1];
  g[2] = 4.0f * u[2];
  g[3] = 4.0f * u[3];
  g[4] = 4.0f * u[4] + 2.0f * u[7];
  g[5] = 4.0f * u[5];
  g[6] = 4.0f * u[6];
  g[7] = 4.0f * u[7] + 2.0f * u[4];
  g[8] = 4.0f * u[8];
}
HVP_FN Mat3 mul(const Mat3& x, const Mat3& y) {
  Mat3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.a[i][j] = x.a[i][0] * y.a[0][j] + x.a[i][1] * y.a[1][j] + x.a[i][2] * y.a[2][j];
  return r;
}
HVP_FN Mat3 tr(const Mat3& x) {
  Mat3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.a[i][j] = x.a[j][i];
  return r;
}
HVP_FN Mat3 add(const Mat3& x, const Mat3& y) {
  Mat3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.a[i][j] = x.a[i][j] + y.a[i][j];
  return r;
}
HVP_FN Mat3 add4(const Mat3& a, const Mat3& b, const Mat3& c, const Mat3& d) { return add(add(a, b), add(c, d)); }
HVP_FN Mat3 scl(const Mat3& x, float s) {
  Mat3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.a[i][j] = x.a[i][j] * s;
  return r;
}
HVP_FN float dot33(const Mat3& x, const Mat3& y) {
  float s = 0.f;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) s += x.a[i][j] * y.a[i][j];
  return s;
}
HVP_FN void ld9(const float* p, int F, float u[9]) {
  for (int c = 0; c < 9; ++c) u[c] = p[c * F];
}
HVP_FN void st9(float* p, int F, const float u[9]) {
  for (int c = 0; c < 9; ++c) p[c * F] = u[c];
}
HVP_FN float dot9(const float a[9], const float b[9]) {
  float s = 0.f;
  for (int c = 0; c < 9; ++c) s += a[c] * b[c];
  return s;
}

// ------------------------------------------------------------------------------------------------ geometry
// pair p < P: d_t = r_hat . (v_i - v_j), r_hat_t = ((v_i - v_j) - r_hat d_t) / d, phi_t = phi' d_t, C_t = C' d_t; self pair (p = P): 0
HVP_FN void pair_tangent(int p, int P, int K, const int* pair_i, const int* pair_j, const float* prhat, const float* pd, const float* v,
                         const float* dphi, const float* dC, float* d_t, float* rhat_t, float* phi_t, float* C_t) {
  float dt = 0.f, rt[3] = {0.f, 0.f, 0.f};
  if (p < P) {
    const int i = pair_i[p], j = pair_j[p];
    const float r[3] = {prhat[p * 3], prhat[p * 3 + 1], prhat[p * 3 + 2]};
    const float e[3] = {v[i * 3] - v[j * 3], v[i * 3 + 1] - v[j * 3 + 1], v[i * 3 + 2] - v[j * 3 + 2]};
    dt = r[0] * e[0] + r[1] * e[1] + r[2] * e[2];
    const float inv = 1.0f / pd[p];
    for (int a = 0; a < 3; ++a) rt[a] = (e[a] - r[a] * dt) * inv;
  }
  d_t[p] = dt;
  for (int a = 0; a < 3; ++a) rhat_t[p * 3 + a] = rt[a];
  for (int k = 0; k < K; ++k) phi_t[(int64_t)p * K + k] = p < P ? dphi[(int64_t)p * K + k] * dt : 0.f;
  C_t[p] = p < P ? dC[p] * dt : 0.f;
}

// ------------------------------------------------------------------------------------------------ embedding, forward
// (atom i, channel f): I0 = sum W0, vec = sum W1 r, T = sum W2 r r^T over the row's edges, W_k = C Zij Q_k; u0, s0n = ||u0||^2;
// tangents with C_t, Q_t, r_t.   (reference tensornet.py:543-619, 405-445)
HVP_FN void u0_of(const float I0, const float v[3], const float T[6], float u[9]) {
  const float trT = (T[0] + T[3] + T[5]) * (1.0f / 3.0f);
  u[0] = I0;
  u[1] = v[0];
  u[2] = v[1];
  u[3] = v[2];
  u[4] = T[0] - trT;
  u[5] = T[1];
  u[6] = T[2];
  u[7] = T[3] - trT;
  u[8] = T[4];
}
HVP_FN void embed_scatter_dual(int i, int f, int F, int P, const int* rowptr, const int* col, const int* epair, const float* esign,
                               const int64_t* z, const float* Utab, const float* Vtab, const float* Q, const float* Q_t, const float* C,
                               const float* C_t, const float* prhat, const float* rhat_t, float* u0, float* u0_t, float* s0n,
                               float* s0n_t) {
  const int F3 = 3 * F;
  const float Ui = Utab[z[i] * F + f];
  float I0 = 0.f, I0t = 0.f, vv[3] = {0.f, 0.f, 0.f}, vt[3] = {0.f, 0.f, 0.f}, T[6], Tt[6];
  for (int k = 0; k < 6; ++k) T[k] = Tt[k] = 0.f;
  for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) {
    const int j = col[e], p = epair[e];
    const float sg = esign[e];
    float r[3] = {0.f, 0.f, 0.f}, rt[3] = {0.f, 0.f, 0.f};
    if (sg != 0.f && p < P)
      for (int a = 0; a < 3; ++a) {
        r[a] = sg * prhat[p * 3 + a];
        rt[a] = sg * rhat_t[p * 3 + a];
      }
    const float zij = Ui + Vtab[z[j] * F + f];
    const float cz = C[p] * zij, czt = C_t[p] * zij;
    const float* q = Q + (int64_t)p * F3 + f;
    const float* qt = Q_t + (int64_t)p * F3 + f;
    float W[3], Wt[3];
    for (int k = 0; k < 3; ++k) {
      W[k] = cz * q[k * F];
      Wt[k] = czt * q[k * F] + cz * qt[k * F];
    }
    I0 += W[0];
    I0t += Wt[0];
    for (int a = 0; a < 3; ++a) {
      vv[a] += W[1] * r[a];
      vt[a] += Wt[1] * r[a] + W[1] * rt[a];
    }
    const float r6[6] = {r[0] * r[0], r[0] * r[1], r[0] * r[2], r[1] * r[1], r[1] * r[2], r[2] * r[2]};
    const float r6t[6] = {2.f * r[0] * rt[0],           rt[0] * r[1] + r[0] * rt[1], rt[0] * r[2] + r[0] * rt[2],
                          2.f * r[1] * rt[1],           rt[1] * r[2] + r[1] * rt[2], 2.f * r[2] * rt[2]};
    for (int k = 0; k < 6; ++k) {
      T[k] += W[2] * r6[k];
      Tt[k] += Wt[2] * r6[k] + W[2] * r6t[k];
    }
  }
  float u[9], ut[9], dq[9];
  u0_of(I0, vv, T, u);
  u0_of(I0t, vt, Tt, ut);
  dquad(u, dq);
  const int64_t o = (int64_t)i * 9 * F + f;
  st9(u0 + o, F, u);
  st9(u0_t + o, F, ut);
  s0n[(int64_t)i * F + f] = quad(u);
  s0n_t[(int64_t)i * F + f] = dot9(dq, ut);
}

// ------------------------------------------------------------------------------------------------ LayerNorm (one row of width W)
// `nl` lanes share a row (lane = 0 .. nl - 1, columns strided by nl); red(v) = the sum of v over the row's lanes.  The kernels give
// a row to a WAVE (nl = 64, red = wave_sum), the host harness to one thread (nl = 1, red = identity): the same statements run in both.
struct RedOne {
  HVP_FN float operator()(float v) const { return v; }
};
template <class Red>
HVP_FN void ln_dual_lanes(int row, int W, int lane, int nl, Red red, const float* x, const float* x_t, const float* w, const float* b,
                          float* y, float* xh, float* rstd, float* y_t, float* xh_t, float* rstd_t) {
  const float* xr = x + (int64_t)row * W;
  const float* xtr = x_t + (int64_t)row * W;
  float mu = 0.f, mt = 0.f;
  for (int k = lane; k < W; k += nl) {
    mu += xr[k];
    mt += xtr[k];
  }
  mu = red(mu) / W;
  mt = red(mt) / W;
  float var = 0.f, m = 0.f;  // m: mean(xh * x_t)
  for (int k = lane; k < W; k += nl) {
    const float dx = xr[k] - mu;
    var += dx * dx;
    m += dx * xtr[k];
  }
  var = red(var) / W;
  const float rs = 1.0f / sqrtf(var + 1e-5f);
  m = red(m) * rs / W;
  for (int k = lane; k < W; k += nl) {
    const float h = (xr[k] - mu) * rs, ht = rs * (xtr[k] - mt - h * m);
    const int64_t o = (int64_t)row * W + k;
    xh[o] = h;
    xh_t[o] = ht;
    y[o] = h * w[k] + b[k];
    y_t[o] = ht * w[k];
  }
  if (lane == 0) {
    rstd[row] = rs;
    rstd_t[row] = -rs * rs * m;
  }
}
HVP_FN void ln_dual(int row, int W, const float* x, const float* x_t, const float* w, const float* b, float* y, float* xh, float* rstd,
                    float* y_t, float* xh_t, float* rstd_t) {
  ln_dual_lanes(row, W, 0, 1, RedOne{}, x, x_t, w, b, y, xh, rstd, y_t, xh_t, rstd_t);
}
// adjoint of the normalisation (g = gradient wrt y) and its tangent
template <class Red>
HVP_FN void lnbwd_dual_lanes(int row, int W, int lane, int nl, Red red, const float* g, const float* g_t, const float* xh, const float* xh_t,
                             const float* rstd, const float* rstd_t, const float* w, float* o, float* o_t) {
  const int64_t b = (int64_t)row * W;
  float c1 = 0.f, c2 = 0.f, c1t = 0.f, c2t = 0.f;
  for (int k = lane; k < W; k += nl) {
    const float gw = g[b + k] * w[k], gwt = g_t[b + k] * w[k];
    c1 += gw;
    c2 += gw * xh[b + k];
    c1t += gwt;
    c2t += gwt * xh[b + k] + gw * xh_t[b + k];
  }
  c1 = red(c1) / W;
  c2 = red(c2) / W;
  c1t = red(c1t) / W;
  c2t = red(c2t) / W;
  const float rs = rstd[row], rst = rstd_t[row];
  for (int k = lane; k < W; k += nl) {
    const float gw = g[b + k] * w[k], gwt = g_t[b + k] * w[k];
    const float core = gw - c1 - xh[b + k] * c2;
    o[b + k] = core * rs;
    o_t[b + k] = (gwt - c1t - xh_t[b + k] * c2 - xh[b + k] * c2t) * rs + core * rst;
  }
}
HVP_FN void lnbwd_dual(int row, int W, const float* g, const float* g_t, const float* xh, const float* xh_t, const float* rstd,
                       const float* rstd_t, const float* w, float* o, float* o_t) {
  lnbwd_dual_lanes(row, W, 0, 1, RedOne{}, g, g_t, xh, xh_t, rstd, rstd_t, w, o, o_t);
}

// ------------------------------------------------------------------------------------------------ elementwise
HVP_FN void silu_tangent(int64_t i, const float* a, const float* a_t, float* h_t) { h_t[i] = silu1(a[i]) * a_t[i]; }
// o = g silu'(a) and its tangent
HVP_FN void dsilu_dual(int64_t i, const float* g, const float* g_t, const float* a, const float* a_t, float* o, float* o_t) {
  const float d1 = silu1(a[i]);
  o[i] = g[i] * d1;
  o_t[i] = g_t[i] * d1 + g[i] * silu2(a[i]) * a_t[i];
}
// X = UX gates[type]   (gates [N][3F])
HVP_FN void gate_mul_dual(int n, int f, int F, const float* UX, const float* UX_t, const float* gates, const float* gates_t, float* X,
                          float* X_t) {
  const int64_t o = (int64_t)n * 9 * F + f, og = (int64_t)n * 3 * F + f;
  for (int c = 0; c < 9; ++c) {
    const float gt = gates[og + type_of(c) * F], gtt = gates_t[og + type_of(c) * F];
    X[o + c * F] = UX[o + c * F] * gt;
    X_t[o + c * F] = UX_t[o + c * F] * gt + UX[o + c * F] * gtt;
  }
}
// w = silu(e3) C(d) per pair row (rows of width 3F)
HVP_FN void w_dual(int64_t i, int F3, const float* e3, const float* e3_t, const float* C, const float* C_t, float* w, float* w_t) {
  const int64_t p = i / F3;
  const float f3 = silu0(e3[i]);
  w[i] = f3 * C[p];
  w_t[i] = silu1(e3[i]) * e3_t[i] * C[p] + f3 * C_t[p];
}
// X_hat = X / (||X||^2 + 1)
HVP_FN void norm_dual(int n, int f, int F, const float* X, const float* X_t, float* Xh, float* Xh_t) {
  const int64_t o = (int64_t)n * 9 * F + f;
  float x[9], xt[9], dq[9], h[9], ht[9];
  ld9(X + o, F, x);
  ld9(X_t + o, F, xt);
  dquad(x, dq);
  const float s = quad(x), st = dot9(dq, xt), inv = 1.0f / (s + 1.0f);
  for (int c = 0; c < 9; ++c) {
    h[c] = x[c] * inv;
    ht[c] = xt[c] * inv - x[c] * st * inv * inv;
  }
  st9(Xh + o, F, h);
  st9(Xh_t + o, F, ht);
}

// ------------------------------------------------------------------------------------------------ group product
// Cm = kappa (Y Mf + Mf Y)  [O(3)]  or  2 Y Mf  [SO(3)] and its tangent     (reference tensornet.py:42-51, 789)
HVP_FN void group_cm(const Mat3& Y, const Mat3& Yt, const Mat3& M, const Mat3& Mt, float kap, int o3, Mat3& Cm, Mat3& Cmt) {
  if (o3) {
    Cm = scl(add(mul(Y, M), mul(M, Y)), kap);
    Cmt = scl(add4(mul(Yt, M), mul(Y, Mt), mul(Mt, Y), mul(M, Yt)), kap);
  } else {
    Cm = scl(mul(Y, M), 2.0f);
    Cmt = scl(add(mul(Yt, M), mul(Y, Mt)), 2.0f);
  }
}
// C_hat = dec(Cm) / (||Cm||^2 + 1)
HVP_FN void group_dual(int n, int f, int F, const float* Pn, const float* Pn_t, const float* Mi, const float* Mi_t, const float* kap,
                       int o3, float* Ch, float* Ch_t) {
  const int64_t o = (int64_t)n * 9 * F + f;
  float a[9], b[9];
  ld9(Pn + o, F, a);
  ld9(Pn_t + o, F, b);
  const Mat3 Y = compose(a), Yt = compose(b);
  ld9(Mi + o, F, a);
  ld9(Mi_t + o, F, b);
  const Mat3 M = compose(a), Mt = compose(b);
  Mat3 Cm, Cmt;
  group_cm(Y, Yt, M, Mt, kap ? kap[n] : 1.0f, o3, Cm, Cmt);
  float uC[9], uCt[9], h[9], ht[9];
  decomp(Cm, uC);
  decomp(Cmt, uCt);
  const float t = dot33(Cm, Cm), tt = 2.0f * dot33(Cm, Cmt), inv = 1.0f / (t + 1.0f);
  for (int c = 0; c < 9; ++c) {
    h[c] = uC[c] * inv;
    ht[c] = uCt[c] * inv - uC[c] * tt * inv * inv;
  }
  st9(Ch + o, F, h);
  st9(Ch_t + o, F, ht);
}
// X_new = X_hat + D + kappa dec(dX dX),  dX = compose(D)      (reference tensornet.py:745, 812)
HVP_FN void update_dual(int n, int f, int F, const float* Xh, const float* Xh_t, const float* D, const float* D_t, const float* kap,
                        float* Xn, float* Xn_t) {
  const int64_t o = (int64_t)n * 9 * F + f;
  float xh[9], xht[9], d[9], dt[9], q[9], qt[9], r[9], rt[9];
  ld9(Xh + o, F, xh);
  ld9(Xh_t + o, F, xht);
  ld9(D + o, F, d);
  ld9(D_t + o, F, dt);
  const Mat3 dX = compose(d), dXt = compose(dt);
  const float k = kap ? kap[n] : 1.0f;
  decomp(scl(mul(dX, dX), k), q);
  decomp(scl(add(mul(dXt, dX), mul(dX, dXt)), k), qt);
  for (int c = 0; c < 9; ++c) {
    r[c] = xh[c] + d[c] + q[c];
    rt[c] = xht[c] + dt[c] + qt[c];
  }
  st9(Xn + o, F, r);
  st9(Xn_t + o, F, rt);
}
// readout invariants per type (3 I^2, ||A||^2, ||S||^2) -> feat [N][3F]       (reference tensornet.py:384-396)
HVP_FN void feat_dual(int n, int f, int F, const float* X, const float* X_t, float* feat, float* feat_t) {
  const int64_t o = (int64_t)n * 9 * F + f, of = (int64_t)n * 3 * F + f;
  float x[9], xt[9], dq[9];
  ld9(X + o, F, x);
  ld9(X_t + o, F, xt);
  dquad(x, dq);
  const float t = x[4] + x[7];
  feat[of] = 3.0f * x[0] * x[0];
  feat[of + F] = 2.0f * (x[1] * x[1] + x[2] * x[2] + x[3] * x[3]);
  feat[of + 2 * F] = x[4] * x[4] + x[7] * x[7] + t * t + 2.0f * (x[5] * x[5] + x[6] * x[6] + x[8] * x[8]);
  feat_t[of] = dq[0] * xt[0];
  feat_t[of + F] = dq[1] * xt[1] + dq[2] * xt[2] + dq[3] * xt[3];
  feat_t[of + 2 * F] = dq[4] * xt[4] + dq[5] * xt[5] + dq[6] * xt[6] + dq[7] * xt[7] + dq[8] * xt[8];
}

// ------------------------------------------------------------------------------------------------ head, reverse pass
// seed: g_ao = std O2 silu'(ao); headv = std silu'(ao) ao_t is the summand of d s / d O2   (i over N * H)
// ge != null (one-pass training): everything downstream is the gradient of  S = s - sum_m ge_m E_m  - the tangent adjoint minus the
// adjoint of sum_m ge_m E_m obeys the tangent adjoint's recursion, so only its seed changes: - ge[molecule] at every atom's energy
HVP_FN void head_dual(int64_t i, int H, const float* ao, const float* ao_t, const float* O2, float std_, const float* ge,
                      const int64_t* batch, float* g_ao, float* g_ao_t, float* headv) {
  const float o2 = std_ * O2[i % H], d1 = silu1(ao[i]);
  const float w = ge ? ge[batch ? batch[i / H] : 0] : 0.f;
  g_ao[i] = o2 * d1;
  g_ao_t[i] = o2 * silu2(ao[i]) * ao_t[i] - w * o2 * d1;
  headv[i] = std_ * d1 * ao_t[i] - w * std_ * silu0(ao[i]);
}
// d S / d (the head's last bias) = - std sum_n ge[molecule(n)]   (one logical thread; the kernel sums in blocks of 256)
HVP_FN float head_bias_seed_term(int n, const float* ge, const int64_t* batch) { return ge[batch ? batch[n] : 0]; }
// per atom row (width W): x_t -= ge[molecule] x  - the Coulomb head's adjoints under the energy seed (its pairs lie inside a molecule)
HVP_FN void row_seed(int64_t i, int W, const float* ge, const int64_t* batch, const float* x, float* x_t) {
  x_t[i] -= ge[batch ? batch[i / W] : 0] * x[i];
}
// G = dquad(X) g_feat[type]
HVP_FN void readout_bwd_dual(int n, int f, int F, const float* X, const float* X_t, const float* g_feat, const float* g_feat_t, float* G,
                             float* G_t) {
  const int64_t o = (int64_t)n * 9 * F + f, of = (int64_t)n * 3 * F + f;
  float x[9], xt[9], dq[9], dqt[9], g[9], gt[9];
  ld9(X + o, F, x);
  ld9(X_t + o, F, xt);
  dquad(x, dq);
  dquad(xt, dqt);
  for (int c = 0; c < 9; ++c) {
    const float gf = g_feat[of + type_of(c) * F], gft = g_feat_t[of + type_of(c) * F];
    g[c] = dq[c] * gf;
    gt[c] = dqt[c] * gf + dq[c] * gft;
  }
  st9(G + o, F, g);
  st9(G_t + o, F, gt);
}
// adjoint of the update wrt D: g_D = compose_T( Gf + kappa (Gf dX^T + dX^T Gf) ),  Gf = dec_T(G)
HVP_FN void update_bwd_dual(int n, int f, int F, const float* G, const float* G_t, const float* D, const float* D_t, const float* kap,
                            float* g_D, float* g_D_t) {
  const int64_t o = (int64_t)n * 9 * F + f;
  float a[9], b[9];
  ld9(G + o, F, a);
  ld9(G_t + o, F, b);
  const Mat3 Gf = decomp_T(a), Gft = decomp_T(b);
  ld9(D + o, F, a);
  ld9(D_t + o, F, b);
  const Mat3 dXT = tr(compose(a)), dXtT = tr(compose(b));
  const float k = kap ? kap[n] : 1.0f;
  const Mat3 g = add(Gf, scl(add(mul(Gf, dXT), mul(dXT, Gf)), k));
  const Mat3 gt = add(Gft, scl(add4(mul(Gft, dXT), mul(dXT, Gft), mul(Gf, dXtT), mul(dXtT, Gf)), k));
  compose_T(g, a);
  compose_T(gt, b);
  st9(g_D + o, F, a);
  st9(g_D_t + o, F, b);
}
// adjoint of the group product: from g_Ch to g_Mi and the direct part of g_Pn (= compose_T(g_Y)); forward values are recomputed
HVP_FN void group_bwd_dual(int n, int f, int F, const float* g_Ch, const float* g_Ch_t, const float* Pn, const float* Pn_t,
                           const float* Mi, const float* Mi_t, const float* kap, int o3, float* g_Mi, float* g_Mi_t, float* g_PnY,
                           float* g_PnY_t) {
  const int64_t o = (int64_t)n * 9 * F + f;
  float a[9], b[9];
  ld9(Pn + o, F, a);
  ld9(Pn_t + o, F, b);
  const Mat3 Y = compose(a), Yt = compose(b);
  ld9(Mi + o, F, a);
  ld9(Mi_t + o, F, b);
  const Mat3 M = compose(a), Mt = compose(b);
  const float k = kap ? kap[n] : 1.0f;
  Mat3 Cm, Cmt;
  group_cm(Y, Yt, M, Mt, k, o3, Cm, Cmt);
  float uC[9], uCt[9], gc[9], gct[9], gu[9], gut[9];
  decomp(Cm, uC);
  decomp(Cmt, uCt);
  ld9(g_Ch + o, F, gc);
  ld9(g_Ch_t + o, F, gct);
  const float t = dot33(Cm, Cm), tt = 2.0f * dot33(Cm, Cmt), inv = 1.0f / (t + 1.0f);
  const float aa = dot9(gc, uC), aat = dot9(gct, uC) + dot9(gc, uCt);
  const float gt_ = -aa * inv * inv, gtt = -aat * inv * inv + 2.0f * aa * tt * inv * inv * inv;
  for (int c = 0; c < 9; ++c) {
    gu[c] = gc[c] * inv;
    gut[c] = gct[c] * inv - gc[c] * tt * inv * inv;
  }
  const Mat3 gCm = add(decomp_T(gu), scl(Cm, 2.0f * gt_));
  const Mat3 gCmt = add(decomp_T(gut), add(scl(Cmt, 2.0f * gt_), scl(Cm, 2.0f * gtt)));
  Mat3 gY, gYt, gM, gMt;
  const Mat3 MT = tr(M), MtT = tr(Mt), YT = tr(Y), YtT = tr(Yt);
  if (o3) {
    gY = scl(add(mul(gCm, MT), mul(MT, gCm)), k);
    gYt = scl(add4(mul(gCmt, MT), mul(MT, gCmt), mul(gCm, MtT), mul(MtT, gCm)), k);
    gM = scl(add(mul(YT, gCm), mul(gCm, YT)), k);
    gMt = scl(add4(mul(YtT, gCm), mul(gCm, YtT), mul(YT, gCmt), mul(gCmt, YT)), k);
  } else {
    gY = scl(mul(gCm, MT), 2.0f);
    gYt = scl(add(mul(gCmt, MT), mul(gCm, MtT)), 2.0f);
    gM = scl(mul(YT, gCm), 2.0f);
    gMt = scl(add(mul(YtT, gCm), mul(YT, gCmt)), 2.0f);
  }
  compose_T(gM, a);
  compose_T(gMt, b);
  st9(g_Mi + o, F, a);
  st9(g_Mi_t + o, F, b);
  compose_T(gY, a);
  compose_T(gYt, b);
  st9(g_PnY + o, F, a);
  st9(g_PnY_t + o, F, b);
}
// per-pair weight gradient g_w[p, k] = sum_{c in type k} (gMi[i] Pn[j] + gMi[j] Pn[i]) (self pair: the column sums self_gw), carried
// through w = silu(e3) C to the adjoint of e3:  g_e3 = g_w C silu'(e3)                 (i over (P + 1) * F; rows of width 3F)
HVP_FN void pair_gw_dual(int64_t idx, int P, int F, const int* pair_i, const int* pair_j, const float* g_Mi, const float* g_Mi_t,
                         const float* Pn, const float* Pn_t, const float* self_gw, const float* self_gw_t, const float* e3,
                         const float* e3_t, const float* C, const float* C_t, float* g_e3, float* g_e3_t, float* gcp, float* gcp_t) {
  const int p = (int)(idx / F), f = (int)(idx - (int64_t)p * F);
  float gw[3] = {0.f, 0.f, 0.f}, gwt[3] = {0.f, 0.f, 0.f};
  if (p < P) {
    const int64_t oi = (int64_t)pair_i[p] * 9 * F + f, oj = (int64_t)pair_j[p] * 9 * F + f;
    for (int c = 0; c < 9; ++c) {
      const int k = type_of(c);
      gw[k] += g_Mi[oi + c * F] * Pn[oj + c * F] + g_Mi[oj + c * F] * Pn[oi + c * F];
      gwt[k] += g_Mi_t[oi + c * F] * Pn[oj + c * F] + g_Mi_t[oj + c * F] * Pn[oi + c * F] + g_Mi[oi + c * F] * Pn_t[oj + c * F] +
                g_Mi[oj + c * F] * Pn_t[oi + c * F];
    }
  } else {
    for (int k = 0; k < 3; ++k) {
      gw[k] = self_gw[k * F + f];
      gwt[k] = self_gw_t[k * F + f];
    }
  }
  const float c = C[p], ct = C_t[p];
  float gc = 0.f, gct = 0.f;  // this channel's part of g_C[p] = sum g_w silu(e3) (position gradient only: gcp may be null)
  for (int k = 0; k < 3; ++k) {
    const int64_t o = (int64_t)p * 3 * F + k * F + f;
    const float d1 = silu1(e3[o]);
    g_e3[o] = gw[k] * c * d1;
    g_e3_t[o] = gwt[k] * c * d1 + gw[k] * ct * d1 + gw[k] * c * silu2(e3[o]) * e3_t[o];
    gc += gw[k] * silu0(e3[o]);
    gct += gwt[k] * silu0(e3[o]) + gw[k] * d1 * e3_t[o];
  }
  if (gcp) {
    gcp[(int64_t)p * F + f] = gc;
    gcp_t[(int64_t)p * F + f] = gct;
  }
}
// adjoint of X_hat = X / (s + 1) with the residual stream's G:  g_Xh = G + gL ;  G_new = g_Xh / (s + 1) + dquad(X) g_s
HVP_FN void norm_bwd_dual(int n, int f, int F, const float* X, const float* X_t, const float* G, const float* G_t, const float* gL,
                          const float* gL_t, float* Gn, float* Gn_t) {
  const int64_t o = (int64_t)n * 9 * F + f;
  float x[9], xt[9], dq[9], dqt[9], g[9], gt[9], a[9], r[9], rt[9];
  ld9(X + o, F, x);
  ld9(X_t + o, F, xt);
  dquad(x, dq);
  dquad(xt, dqt);
  ld9(G + o, F, g);
  ld9(gL + o, F, a);
  for (int c = 0; c < 9; ++c) g[c] += a[c];
  ld9(G_t + o, F, gt);
  ld9(gL_t + o, F, a);
  for (int c = 0; c < 9; ++c) gt[c] += a[c];
  const float s = quad(x), st = dot9(dq, xt), inv = 1.0f / (s + 1.0f);
  const float bb = dot9(g, x), bt = dot9(gt, x) + dot9(g, xt);
  const float gs = -bb * inv * inv, gst = -bt * inv * inv + 2.0f * bb * st * inv * inv * inv;
  for (int c = 0; c < 9; ++c) {
    r[c] = g[c] * inv + dq[c] * gs;
    rt[c] = gt[c] * inv - g[c] * st * inv * inv + dqt[c] * gs + dq[c] * gst;
  }
  st9(Gn + o, F, r);
  st9(Gn_t + o, F, rt);
}
// adjoint of X = UX gates[type], gates = silu(a2):  g_UX = G gates ;  g_a2[t] = silu'(a2) sum_{c in t} G UX        ([N][3F])
HVP_FN void gate_bwd_dual(int n, int f, int F, const float* G, const float* G_t, const float* UX, const float* UX_t, const float* gates,
                          const float* gates_t, const float* a2, const float* a2_t, float* g_UX, float* g_UX_t, float* g_a2,
                          float* g_a2_t) {
  const int64_t o = (int64_t)n * 9 * F + f, og = (int64_t)n * 3 * F + f;
  float gg[3] = {0.f, 0.f, 0.f}, ggt[3] = {0.f, 0.f, 0.f};
  for (int c = 0; c < 9; ++c) {
    const int k = type_of(c);
    const float g = G[o + c * F], gt = G_t[o + c * F], ux = UX[o + c * F], uxt = UX_t[o + c * F];
    g_UX[o + c * F] = g * gates[og + k * F];
    g_UX_t[o + c * F] = gt * gates[og + k * F] + g * gates_t[og + k * F];
    gg[k] += g * ux;
    ggt[k] += gt * ux + g * uxt;
  }
  for (int k = 0; k < 3; ++k) {
    const float a = a2[og + k * F], d1 = silu1(a);
    g_a2[og + k * F] = gg[k] * d1;
    g_a2_t[og + k * F] = ggt[k] * d1 + gg[k] * silu2(a) * a2_t[og + k * F];
  }
}
// g_u0 = gL + dquad(u0) g_s0n, then the adjoint of u0_of: gA = d/d(I0, vec[3], T00, T01, T02, T11, T12, T22)      ([N][10][F])
HVP_FN void embed_bwd_atom_dual(int n, int f, int F, const float* gL, const float* gL_t, const float* u0, const float* u0_t,
                                const float* g_s0n, const float* g_s0n_t, float* gA, float* gA_t) {
  const int64_t o = (int64_t)n * 9 * F + f, oa = (int64_t)n * 10 * F + f;
  float u[9], ut[9], dq[9], dqt[9], g[9], gt[9];
  ld9(u0 + o, F, u);
  ld9(u0_t + o, F, ut);
  dquad(u, dq);
  dquad(ut, dqt);
  ld9(gL + o, F, g);
  ld9(gL_t + o, F, gt);
  const float gs = g_s0n[(int64_t)n * F + f], gst = g_s0n_t[(int64_t)n * F + f];
  for (int c = 0; c < 9; ++c) {
    g[c] += dq[c] * gs;
    gt[c] += dqt[c] * gs + dq[c] * gst;
  }
  const float third = (g[4] + g[7]) * (1.0f / 3.0f), thirdt = (gt[4] + gt[7]) * (1.0f / 3.0f);
  const float A[10] = {g[0], g[1], g[2], g[3], g[4] - third, g[5], g[6], g[7] - third, g[8], -third};
  const float At[10] = {gt[0], gt[1], gt[2], gt[3], gt[4] - thirdt, gt[5], gt[6], gt[7] - thirdt, gt[8], -thirdt};
  for (int c = 0; c < 10; ++c) {
    gA[oa + c * F] = A[c];
    gA_t[oa + c * F] = At[c];
  }
}
// (gW_0, gW_1, gW_2) of an edge from the row atom's gA and the edge's unit vector, with the tangent
HVP_FN void edge_gw(const float a[10], const float at[10], const float r[3], const float rt[3], float w[3], float wt[3]) {
  const float r6[6] = {r[0] * r[0], r[0] * r[1], r[0] * r[2], r[1] * r[1], r[1] * r[2], r[2] * r[2]};
  const float r6t[6] = {2.f * r[0] * rt[0],           rt[0] * r[1] + r[0] * rt[1], rt[0] * r[2] + r[0] * rt[2],
                        2.f * r[1] * rt[1],           rt[1] * r[2] + r[1] * rt[2], 2.f * r[2] * rt[2]};
  w[0] = a[0];
  wt[0] = at[0];
  w[1] = a[1] * r[0] + a[2] * r[1] + a[3] * r[2];
  wt[1] = at[1] * r[0] + at[2] * r[1] + at[3] * r[2] + a[1] * rt[0] + a[2] * rt[1] + a[3] * rt[2];
  w[2] = wt[2] = 0.f;
  for (int k = 0; k < 6; ++k) {
    w[2] += a[4 + k] * r6[k];
    wt[2] += at[4 + k] * r6[k] + a[4 + k] * r6t[k];
  }
}
// per directed edge of row i (the layout of k_train_embed, tn_train.hip):
//   gq[dir][p, k] = gW_k C Zij (dir 0: i is the pair's first atom, 1: its second) ; self edge: selfq[i] = gW_0 C Zij
//   gZu_t[i] = sum_e d/dt( C sum_k gW_k Q_k ) ; gZv_t[i] = the same sum over the REVERSE edges (j <- i), where i is the V-side atom
HVP_FN void embed_edge_dual(int i, int f, int F, int P, const int* rowptr, const int* col, const int* epair, const float* esign,
                            const int64_t* z, const float* Utab, const float* Vtab, const float* Q, const float* Q_t, const float* C,
                            const float* C_t, const float* prhat, const float* rhat_t, const float* gA, const float* gA_t, float* gq,
                            float* gq_t, int64_t dir_stride, float* selfq, float* selfq_t, float* gZu_t, float* gZv_t) {
  const int F3 = 3 * F, F10 = 10 * F;
  const float Ui = Utab[z[i] * F + f];
  float ai[10], ait[10];
  for (int c = 0; c < 10; ++c) {
    ai[c] = gA[(int64_t)i * F10 + c * F + f];
    ait[c] = gA_t[(int64_t)i * F10 + c * F + f];
  }
  float zu = 0.f, zv = 0.f, sq = 0.f, sqt = 0.f;
  for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) {
    const int j = col[e], p = epair[e];
    const float sg = esign[e];
    const bool self = sg == 0.f || p >= P;
    float r[3] = {0.f, 0.f, 0.f}, rt[3] = {0.f, 0.f, 0.f};
    if (!self)
      for (int a = 0; a < 3; ++a) {
        r[a] = sg * prhat[p * 3 + a];
        rt[a] = sg * rhat_t[p * 3 + a];
      }
    const float c = C[p], ct = C_t[p];
    const float zij = Ui + Vtab[z[j] * F + f];
    const float cz = c * zij, czt = ct * zij;
    float q[3], qt[3], w[3], wt[3];
    for (int k = 0; k < 3; ++k) {
      q[k] = Q[(int64_t)p * F3 + k * F + f];
      qt[k] = Q_t[(int64_t)p * F3 + k * F + f];
    }
    edge_gw(ai, ait, r, rt, w, wt);
    float acc = 0.f;
    for (int k = 0; k < 3; ++k) acc += wt[k] * c * q[k] + w[k] * ct * q[k] + w[k] * c * qt[k];
    zu += acc;
    if (self) {
      sq = w[0] * cz;
      sqt = wt[0] * cz + w[0] * czt;
      zv += acc;  // the self edge is its own reverse (only k = 0 is non-zero: r = 0)
    } else {
      const int64_t o = (sg > 0.f ? 0 : dir_stride) + (int64_t)p * F3 + f;
      for (int k = 0; k < 3; ++k) {
        gq[o + k * F] = w[k] * cz;
        gq_t[o + k * F] = wt[k] * cz + w[k] * czt;
      }
      float aj[10], ajt[10], rr[3], rrt[3], u[3], ut[3];
      for (int cc = 0; cc < 10; ++cc) {
        aj[cc] = gA[(int64_t)j * F10 + cc * F + f];
        ajt[cc] = gA_t[(int64_t)j * F10 + cc * F + f];
      }
      for (int a = 0; a < 3; ++a) {
        rr[a] = -r[a];
        rrt[a] = -rt[a];
      }
      edge_gw(aj, ajt, rr, rrt, u, ut);
      for (int k = 0; k < 3; ++k) zv += ut[k] * c * q[k] + u[k] * ct * q[k] + u[k] * c * qt[k];
    }
  }
  selfq[(int64_t)i * F + f] = sq;
  selfq_t[(int64_t)i * F + f] = sqt;
  gZu_t[(int64_t)i * F + f] = zu;
  gZv_t[(int64_t)i * F + f] = zv;
}

// ------------------------------------------------------------------------------------------------ H v in the positions
// second derivatives of the radial functions (ExpNormalSmearing models/utils.py:402-407, CosineCutoff :506-528), element (p, k)
HVP_FN void cutoff_d2(float d, float lo, float up, float& c2) {
  const float PI = 3.14159265358979323846f;
  if (lo > 0.f) {
    const float k = 2.0f * PI / (up - lo), arg = PI * (2.0f * (d - lo) / (up - lo) + 1.0f);
    c2 = (d < up && d > lo) ? -0.5f * cosf(arg) * k * k : 0.f;
  } else {
    const float k = PI / up;
    c2 = d < up ? -0.5f * cosf(d * k) * k * k : 0.f;
  }
}
HVP_FN void radial2(int64_t idx, int P, int K, const float* pd, const float* means, const float* betas, float lo, float up, float* d2phi,
                    float* d2C) {
  const float PI = 3.14159265358979323846f;
  const int p = (int)(idx / K), k = (int)(idx - (int64_t)p * K);
  const float d = p < P ? pd[p] : 0.f;
  const float kc = PI / up;
  const bool in = d < up;
  const float c0 = in ? 0.5f * (cosf(d * kc) + 1.0f) : 0.f, c1 = in ? -0.5f * sinf(d * kc) * kc : 0.f,
              c2 = in ? -0.5f * cosf(d * kc) * kc * kc : 0.f;
  const float alpha = 5.0f / (up - lo), u = expf(-alpha * (d - lo)), u1 = -alpha * u, u2 = alpha * alpha * u;
  const float mu = means[k], beta = betas[k];
  const float g = expf(-beta * (u - mu) * (u - mu));
  const float h = -2.0f * beta * (u - mu) * u1, h1 = -2.0f * beta * (u1 * u1 + (u - mu) * u2);
  const float g1 = g * h, g2 = g * (h * h + h1);
  d2phi[idx] = c2 * g + 2.0f * c1 * g1 + c0 * g2;
  if (k == 0) cutoff_d2(d, lo, up, d2C[p]);
}
// out[p] (+)= sum_j x[p, j] y[p, j] ;  out_t[p] (+)= sum_j ( x_t y + x y2 d_t[p] )      (y = null: 1 ; y2 = null: no second term)
HVP_FN void pair_rowdot(int p, int W, const float* x, const float* x_t, const float* y, const float* y2, const float* d_t, int accumulate,
                        float* out, float* out_t) {
  const int64_t b = (int64_t)p * W;
  float s = 0.f, st = 0.f;
  for (int j = 0; j < W; ++j) {
    const float yy = y ? y[b + j] : 1.0f;
    s += x[b + j] * yy;
    st += x_t[b + j] * yy;
    if (y2) st += x[b + j] * y2[b + j] * d_t[p];
  }
  out[p] = (accumulate ? out[p] : 0.f) + s;
  out_t[p] = (accumulate ? out_t[p] : 0.f) + st;
}
// directed edge e = (i <- j) of the embedding: its part of g_C[pair] and the gradient wrt its unit vector, summed over the channels
//   ec[dir][p] = ( sum_{k,f} gW_k Zij Q_k ,  g_re[0..2] ) with  g_re = sum_f ( gv W_1 + dq(gT, r) W_2 ),  W_k = C Zij Q_k
// edge_geom_term: channel f's summand into acc = (gc, gre[3]) and acc_t; edge_geom_row: the row atom of edge e (binary search in rowptr)
HVP_FN int edge_geom_row(int e, int N, const int* rowptr) {
  int lo_ = 0, hi_ = N;  // the last i with rowptr[i] <= e
  while (hi_ - lo_ > 1) {
    const int mid = (lo_ + hi_) >> 1;
    if (rowptr[mid] <= e) lo_ = mid;
    else hi_ = mid;
  }
  return lo_;
}
// the arithmetic of one (edge, channel) on values already at hand: a / at = gA / gA_t of the row atom (10 components), zij = U[z_i] + V[z_j],
// q / qt = the pair's distance projections and their tangents
HVP_FN void edge_geom_core(const float a[10], const float at[10], float zij, const float q[3], const float qt[3], const float r[3],
                           const float rt[3], float c, float ct, float acc[4], float acc_t[4]) {
  float w[3], wt[3];
  edge_gw(a, at, r, rt, w, wt);
  for (int k = 0; k < 3; ++k) {
    acc[0] += w[k] * zij * q[k];
    acc_t[0] += (wt[k] * q[k] + w[k] * qt[k]) * zij;
  }
  const float W1 = c * zij * q[1], W1t = ct * zij * q[1] + c * zij * qt[1], W2 = c * zij * q[2], W2t = ct * zij * q[2] + c * zij * qt[2];
  const float* T = a + 4;
  const float* Tt = at + 4;
  const float dq[3] = {2.f * T[0] * r[0] + T[1] * r[1] + T[2] * r[2], T[1] * r[0] + 2.f * T[3] * r[1] + T[4] * r[2],
                       T[2] * r[0] + T[4] * r[1] + 2.f * T[5] * r[2]};
  const float dqt[3] = {2.f * Tt[0] * r[0] + Tt[1] * r[1] + Tt[2] * r[2] + 2.f * T[0] * rt[0] + T[1] * rt[1] + T[2] * rt[2],
                        Tt[1] * r[0] + 2.f * Tt[3] * r[1] + Tt[4] * r[2] + T[1] * rt[0] + 2.f * T[3] * rt[1] + T[4] * rt[2],
                        Tt[2] * r[0] + Tt[4] * r[1] + 2.f * Tt[5] * r[2] + T[2] * rt[0] + T[4] * rt[1] + 2.f * T[5] * rt[2]};
  for (int x = 0; x < 3; ++x) {
    acc[1 + x] += a[1 + x] * W1 + dq[x] * W2;
    acc_t[1 + x] += at[1 + x] * W1 + a[1 + x] * W1t + dqt[x] * W2 + dq[x] * W2t;
  }
}
HVP_FN void edge_geom_term(int i, int j, int p, int f, int F, const float r[3], const float rt[3], float c, float ct, const int64_t* z,
                           const float* Utab, const float* Vtab, const float* Q, const float* Q_t, const float* gA, const float* gA_t,
                           float acc[4], float acc_t[4]) {
  const int F3 = 3 * F, F10 = 10 * F;
  const float zij = Utab[z[i] * F + f] + Vtab[z[j] * F + f];
  float a[10], at[10], q[3], qt[3];
  for (int cc = 0; cc < 10; ++cc) {
    a[cc] = gA[(int64_t)i * F10 + cc * F + f];
    at[cc] = gA_t[(int64_t)i * F10 + cc * F + f];
  }
  for (int k = 0; k < 3; ++k) {
    q[k] = Q[(int64_t)p * F3 + k * F + f];
    qt[k] = Q_t[(int64_t)p * F3 + k * F + f];
  }
  edge_geom_core(a, at, zij, q, qt, r, rt, c, ct, acc, acc_t);
}
// the whole edge by one thread (host check; the kernel spreads the channels over a wave and reduces)
HVP_FN void edge_geom_dual(int e, int N, int F, int P, const int* rowptr, const int* col, const int* epair, const float* esign,
                           const int64_t* z, const float* Utab, const float* Vtab, const float* Q, const float* Q_t, const float* C,
                           const float* C_t, const float* prhat, const float* rhat_t, const float* gA, const float* gA_t, float* ec,
                           float* ec_t, int64_t dir_stride) {
  const float sg = esign[e];
  const int p = epair[e];
  if (sg == 0.f || p >= P) return;
  const int i = edge_geom_row(e, N, rowptr), j = col[e];
  float r[3], rt[3], acc[4] = {0.f, 0.f, 0.f, 0.f}, acc_t[4] = {0.f, 0.f, 0.f, 0.f};
  for (int a = 0; a < 3; ++a) {
    r[a] = sg * prhat[p * 3 + a];
    rt[a] = sg * rhat_t[p * 3 + a];
  }
  for (int f = 0; f < F; ++f) edge_geom_term(i, j, p, f, F, r, rt, C[p], C_t[p], z, Utab, Vtab, Q, Q_t, gA, gA_t, acc, acc_t);
  const int64_t o = (sg > 0.f ? 0 : dir_stride) + (int64_t)p * 4;
  for (int x = 0; x < 4; ++x) {
    ec[o + x] = acc[x];
    ec_t[o + x] = acc_t[x];
  }
}
// pair p < P: distance gradient g_d, unit-vector gradient g_rh and their tangents -> g_delta (value: minus the force contribution of
// the pair, for checking) and g_delta_t                                        (reference neighbor_utils.py:11-46, differentiated)
HVP_FN void geom_dual(int p, int P, const float* pd, const float* prhat, const float* d_t, const float* rhat_t, const float* dC,
                      const float* d2C, const float* gC, const float* gC_t, const float* gphid, const float* gphid_t, const float* ec,
                      const float* ec_t, int64_t dir_stride, float* gdel, float* gdel_t) {
  (void)P;
  const float d = pd[p], dt = d_t[p], inv = 1.0f / d;
  const float c_ = gC[p] + ec[(int64_t)p * 4] + ec[dir_stride + (int64_t)p * 4];
  const float ct_ = gC_t[p] + ec_t[(int64_t)p * 4] + ec_t[dir_stride + (int64_t)p * 4];
  const float gd = c_ * dC[p] + gphid[p];
  const float gdt = ct_ * dC[p] + c_ * d2C[p] * dt + gphid_t[p];
  float r[3], rt[3], h[3], ht[3];
  float a_ = 0.f, at_ = 0.f;
  for (int x = 0; x < 3; ++x) {
    r[x] = prhat[p * 3 + x];
    rt[x] = rhat_t[p * 3 + x];
    h[x] = ec[(int64_t)p * 4 + 1 + x] - ec[dir_stride + (int64_t)p * 4 + 1 + x];  // sum over the two directions of sign * g_re
    ht[x] = ec_t[(int64_t)p * 4 + 1 + x] - ec_t[dir_stride + (int64_t)p * 4 + 1 + x];
  }
  for (int x = 0; x < 3; ++x) {
    a_ += h[x] * r[x];
    at_ += ht[x] * r[x] + h[x] * rt[x];
  }
  for (int x = 0; x < 3; ++x) {
    const float t1 = (h[x] - a_ * r[x]) * inv;
    const float t1t = (ht[x] - at_ * r[x] - a_ * rt[x]) * inv - t1 * dt * inv;
    gdel[p * 3 + x] = t1 + gd * r[x];
    gdel_t[p * 3 + x] = t1t + gdt * r[x] + gd * rt[x];
  }
}
// atom i: out[i] = sum over its edges of sign(e) g[pair(e)]   (+g at the pair's first atom, -g at its second, nothing for the self edge)
HVP_FN void pair_to_atom(int i, int P, const int* rowptr, const int* epair, const float* esign, const float* g, float* out) {
  float s[3] = {0.f, 0.f, 0.f};
  for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) {
    const float sg = esign[e];
    const int p = epair[e];
    if (sg == 0.f || p >= P) continue;
    for (int x = 0; x < 3; ++x) s[x] += sg * g[p * 3 + x];
  }
  for (int x = 0; x < 3; ++x) out[i * 3 + x] = s[x];
}

}  // namespace hvp
}  // namespace tn
