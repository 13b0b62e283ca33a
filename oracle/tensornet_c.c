/* TEST INFRASTRUCTURE ONLY -- plain-C restatement (scalar arithmetic; OpenMP over independent rows only, so every output
 * element is still one sequential chain and the results do not depend on the thread count) of the TensorNet energy+force path.
 *
 * Role: checker for the HIP kernels (tests/, __graft_entry__.smoke(), bench.py cpu_baseline leg).
 * The product (torchmd-net_amd/) never links, imports or calls this file.
 *
 * Algorithm = reference torchmdnet/models/tensornet.py:308-402 (TensorNet.forward), :543-619
 * (TensorEmbedding.forward, :405-445), :729-814 (Interaction.forward, :622-679), models/utils.py:402-407
 * (ExpNormalSmearing), :506-528 (CosineCutoff), output_modules.py:43-117 (Scalar head + reduce),
 * models/model.py:591-628 (std/mean/atomref, forces = -dE/dpos), extensions/neighbor_utils.py:11-46
 * (edge -> position gradient, zero for d = 0), warp_kernels/neighbors_brute.py:98-197 (pair search,
 * triclinic minimum image z -> y -> x).  The reverse pass is written out by hand
 * instead of autograd.  It is a scalar transliteration of oracle/tensornet_adjoint.py and is pinned
 * (tests/test_oracle.py) against the reference's golden vector tests/expected.pkl and against outputs
 * of the unmodified reference (tests/golden).
 *
 * Build: make -C oracle   ->  oracle/_build/libtn_oracle_f64.so, libtn_oracle_f32.so  (-DREAL=double|float)
 * Storage of per-atom tensors: 9 irreducible components [N][9][F] (I, v0..v2, s0..s4), A = skew(v) with
 * A01=-v2, A02=+v1, A12=-v0 (tensornet.py:84-103), S22 = -(s0+s3).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef REAL
#define REAL double
#endif
#define MAXL 16

typedef struct {
  int32_t F, L, K, max_z, H, o3, has_atomref;
  REAL lo, up, mean, std;
  const REAL *means, *betas;             /* [K] */
  const REAL *dp_w[3], *dp_b[3];         /* distance_proj{1,2,3}: [F,K], [F] */
  const REAL *emb;                       /* [max_z,F] */
  const REAL *emb2_w, *emb2_b;           /* [F,2F], [F] */
  const REAL *Ue[3];                     /* tensor_embedding.linears_tensor: [F,F] */
  const REAL *L1_w, *L1_b, *L2_w, *L2_b; /* linears_scalar.0 [2F,F], .1 [3F,2F] */
  const REAL *ln0_w, *ln0_b;             /* init_norm [F] */
  const REAL *M_w[MAXL][3], *M_b[MAXL][3]; /* layers.l.linears_scalar.{0,1,2} */
  const REAL *V[MAXL][6];                /* layers.l.linears_tensor.{0..5} [F,F] */
  const REAL *lnr_w, *lnr_b;             /* out_norm [3F] */
  const REAL *Lin_w, *Lin_b;             /* linear [F,3F] */
  const REAL *O1_w, *O1_b, *O2_w, *O2_b; /* output MLP [H,F],[H],[1,H],[1] */
  const REAL *atomref;                   /* [max_z] or NULL */
} TnParams;

static const int TYPE_OF[9] = {0, 1, 1, 1, 2, 2, 2, 2, 2};

static REAL sigm(REAL x) { return (REAL)1 / ((REAL)1 + exp(-x)); }
static REAL silu(REAL x) { return x * sigm(x); }
static REAL dsilu(REAL x) { REAL s = sigm(x); return s * (1 + x * (1 - s)); }

/* CosineCutoff and derivative, models/utils.py:506-528 */
static void cutoff(REAL d, REAL lo, REAL up, REAL* c, REAL* dc) {
  const REAL PI = (REAL)3.14159265358979323846;
  if (lo > 0) {
    REAL arg = PI * (2 * (d - lo) / (up - lo) + 1);
    int in = d < up && d > lo;
    *c = in ? (REAL)0.5 * (cos(arg) + 1) : 0;
    *dc = in ? (REAL)-0.5 * sin(arg) * (2 * PI / (up - lo)) : 0;
  } else {
    int in = d < up;
    *c = in ? (REAL)0.5 * (cos(d * PI / up) + 1) : 0;
    *dc = in ? (REAL)-0.5 * sin(d * PI / up) * (PI / up) : 0;
  }
}

/* ---- 3x3 algebra on the 9-component storage (tensornet.py:18-51, 84-146) */
static void compose(const REAL u[9], REAL m[3][3]) {
  m[0][0] = u[0] + u[4]; m[0][1] = u[5] - u[3]; m[0][2] = u[6] + u[2];
  m[1][0] = u[5] + u[3]; m[1][1] = u[0] + u[7]; m[1][2] = u[8] - u[1];
  m[2][0] = u[6] - u[2]; m[2][1] = u[8] + u[1]; m[2][2] = u[0] - u[4] - u[7];
}
static void decomp(const REAL m[3][3], REAL u[9]) {
  REAL I = (m[0][0] + m[1][1] + m[2][2]) / 3;
  u[0] = I;
  u[1] = (REAL)0.5 * (m[2][1] - m[1][2]); u[2] = (REAL)0.5 * (m[0][2] - m[2][0]); u[3] = (REAL)0.5 * (m[1][0] - m[0][1]);
  u[4] = m[0][0] - I; u[5] = (REAL)0.5 * (m[0][1] + m[1][0]); u[6] = (REAL)0.5 * (m[0][2] + m[2][0]);
  u[7] = m[1][1] - I; u[8] = (REAL)0.5 * (m[1][2] + m[2][1]);
}
static void compose_T(const REAL g[3][3], REAL u[9]) { /* adjoint of compose */
  u[0] = g[0][0] + g[1][1] + g[2][2];
  u[1] = g[2][1] - g[1][2]; u[2] = g[0][2] - g[2][0]; u[3] = g[1][0] - g[0][1];
  u[4] = g[0][0] - g[2][2]; u[5] = g[0][1] + g[1][0]; u[6] = g[0][2] + g[2][0];
  u[7] = g[1][1] - g[2][2]; u[8] = g[1][2] + g[2][1];
}
static void decomp_T(const REAL g[9], REAL m[3][3]) { /* adjoint of decomp */
  REAL t = (g[0] - g[4] - g[7]) / 3;
  m[0][0] = t + g[4]; m[0][1] = (REAL)0.5 * (g[5] - g[3]); m[0][2] = (REAL)0.5 * (g[6] + g[2]);
  m[1][0] = (REAL)0.5 * (g[5] + g[3]); m[1][1] = t + g[7]; m[1][2] = (REAL)0.5 * (g[8] - g[1]);
  m[2][0] = (REAL)0.5 * (g[6] - g[2]); m[2][1] = (REAL)0.5 * (g[8] + g[1]); m[2][2] = t;
}
static REAL quad(const REAL u[9]) {
  REAL t = u[4] + u[7];
  return 3 * u[0] * u[0] + 2 * (u[1] * u[1] + u[2] * u[2] + u[3] * u[3]) + u[4] * u[4] + u[7] * u[7] + t * t +
         2 * (u[5] * u[5] + u[6] * u[6] + u[8] * u[8]);
}
static void dquad(const REAL u[9], REAL g[9]) {
  g[0] = 6 * u[0]; g[1] = 4 * u[1]; g[2] = 4 * u[2]; g[3] = 4 * u[3];
  g[4] = 4 * u[4] + 2 * u[7]; g[5] = 4 * u[5]; g[6] = 4 * u[6]; g[7] = 4 * u[7] + 2 * u[4]; g[8] = 4 * u[8];
}
static void mm3(const REAL a[3][3], const REAL b[3][3], REAL c[3][3]) {
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) c[i][j] = a[i][0] * b[0][j] + a[i][1] * b[1][j] + a[i][2] * b[2][j];
}
static void tr3(const REAL a[3][3], REAL t[3][3]) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) t[i][j] = a[j][i]; }

/* y[rows,out] = x[rows,in] W[out,in]^T (+b) ; transposed: y[rows,in] = x[rows,out] W[out,in] */
static void linear(const REAL* x, int rows, int in, int out, const REAL* W, const REAL* b, REAL* y) {
#pragma omp parallel for schedule(static)
  for (int r = 0; r < rows; ++r)
    for (int o = 0; o < out; ++o) {
      REAL s = b ? b[o] : 0;
      for (int k = 0; k < in; ++k) s += x[(size_t)r * in + k] * W[(size_t)o * in + k];
      y[(size_t)r * out + o] = s;
    }
}
static void linear_T(const REAL* g, int rows, int in, int out, const REAL* W, REAL* gx) {
#pragma omp parallel for schedule(static)
  for (int r = 0; r < rows; ++r)
    for (int k = 0; k < in; ++k) {
      REAL s = 0;
      for (int o = 0; o < out; ++o) s += g[(size_t)r * out + o] * W[(size_t)o * in + k];
      gx[(size_t)r * in + k] = s;
    }
}
/* channel mixing of the 9 components with 3 weight matrices (I, A, S)  (tensornet.py:595-617,752-754,808-810) */
static void tensor_linear(const REAL* u, int N, int F, const REAL* const W[3], int transpose, REAL* out) {
#pragma omp parallel for schedule(static)
  for (int n = 0; n < N; ++n)
    for (int c = 0; c < 9; ++c) {
      const REAL* x = u + ((size_t)n * 9 + c) * F;
      REAL* y = out + ((size_t)n * 9 + c) * F;
      const REAL* Wc = W[TYPE_OF[c]];
      for (int o = 0; o < F; ++o) {
        REAL s = 0;
        if (!transpose) for (int k = 0; k < F; ++k) s += x[k] * Wc[(size_t)o * F + k];
        else for (int k = 0; k < F; ++k) s += x[k] * Wc[(size_t)k * F + o];
        y[o] = s;
      }
    }
}
static void layernorm_fwd(const REAL* x, int rows, int R, const REAL* w, const REAL* b, REAL* y, REAL* xh, REAL* rstd) {
  for (int r = 0; r < rows; ++r) {
    REAL mu = 0, var = 0;
    for (int k = 0; k < R; ++k) mu += x[(size_t)r * R + k];
    mu /= R;
    for (int k = 0; k < R; ++k) { REAL d = x[(size_t)r * R + k] - mu; var += d * d; }
    var /= R;
    REAL rs = 1 / sqrt(var + (REAL)1e-5);
    rstd[r] = rs;
    for (int k = 0; k < R; ++k) {
      REAL h = (x[(size_t)r * R + k] - mu) * rs;
      xh[(size_t)r * R + k] = h;
      y[(size_t)r * R + k] = h * w[k] + b[k];
    }
  }
}
static void layernorm_bwd(const REAL* g, int rows, int R, const REAL* xh, const REAL* rstd, const REAL* w, REAL* gx) {
  for (int r = 0; r < rows; ++r) {
    REAL s1 = 0, s2 = 0;
    for (int k = 0; k < R; ++k) { REAL gw = g[(size_t)r * R + k] * w[k]; s1 += gw; s2 += gw * xh[(size_t)r * R + k]; }
    s1 /= R; s2 /= R;
    for (int k = 0; k < R; ++k) gx[(size_t)r * R + k] = (g[(size_t)r * R + k] * w[k] - s1 - xh[(size_t)r * R + k] * s2) * rstd[r];
  }
}

#define NEW(n) ((REAL*)calloc((size_t)(n) > 0 ? (size_t)(n) : 1, sizeof(REAL)))

/* out[i,c,:] = sum over directed edges (i <- j) of w[pair,type(c),:] * src[j,c,:]   (tensornet.py:622-679) */
static void gather_sum(int N, int F, int E, const int* er, const int* ec, const int* ep, const REAL* w, const REAL* src, REAL* out) {
  memset(out, 0, sizeof(REAL) * (size_t)N * 9 * F);
  /* threads own channel ranges; every output element still adds its edges in list order */
#pragma omp parallel for schedule(static)
  for (int f0 = 0; f0 < F; f0 += 8)
    for (int e = 0; e < E; ++e)
      for (int c = 0; c < 9; ++c)
        for (int f = f0; f < F && f < f0 + 8; ++f)
          out[((size_t)er[e] * 9 + c) * F + f] += w[((size_t)ep[e] * 3 + TYPE_OF[c]) * F + f] * src[((size_t)ec[e] * 9 + c) * F + f];
}

int tn_oracle_energy_forces(const TnParams* p, int N, int B, const int64_t* z, const REAL* pos, const int64_t* batch,
                            const REAL* box, int box_mode, const REAL* q, REAL* energy, REAL* forces) {
  const int F = p->F, L = p->L, K = p->K, H = p->H, F3 = 3 * F;
  if (L > MAXL) return -1;
  /* ---------------- pair list (i > j) + directed edge list incl. self loops */
  int P = 0, cap = 64;
  int *pi = malloc(sizeof(int) * cap), *pj = malloc(sizeof(int) * cap);
  REAL *pd = malloc(sizeof(REAL) * cap), *pr = malloc(sizeof(REAL) * 3 * cap);
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < i; ++j) {
      if (batch[i] != batch[j]) continue;
      REAL dx = pos[i * 3] - pos[j * 3], dy = pos[i * 3 + 1] - pos[j * 3 + 1], dz = pos[i * 3 + 2] - pos[j * 3 + 2];
      if (box_mode) {
        const REAL* b = box_mode == 1 ? box : box + batch[i] * 9;
        REAL s3 = round(dz / b[8]); dx -= s3 * b[6]; dy -= s3 * b[7]; dz -= s3 * b[8];
        REAL s2 = round(dy / b[4]); dx -= s2 * b[3]; dy -= s2 * b[4];
        REAL s1 = round(dx / b[0]); dx -= s1 * b[0];
      }
      REAL d2 = dx * dx + dy * dy + dz * dz;
      if (!(d2 < p->up * p->up && d2 >= p->lo * p->lo)) continue;
      if (P == cap) {
        cap *= 2;
        pi = realloc(pi, sizeof(int) * cap); pj = realloc(pj, sizeof(int) * cap);
        pd = realloc(pd, sizeof(REAL) * cap); pr = realloc(pr, sizeof(REAL) * 3 * cap);
      }
      REAL d = sqrt(d2);
      pi[P] = i; pj[P] = j; pd[P] = d; pr[P * 3] = dx / d; pr[P * 3 + 1] = dy / d; pr[P * 3 + 2] = dz / d;
      ++P;
    }
  const int P1 = P + 1, E = 2 * P + N; /* self pair index = P */
  int *er = malloc(sizeof(int) * E), *ec = malloc(sizeof(int) * E), *ep = malloc(sizeof(int) * E);
  REAL* re = NEW((size_t)E * 3);
  for (int k = 0; k < P; ++k) {
    er[2 * k] = pi[k]; ec[2 * k] = pj[k]; ep[2 * k] = k;
    er[2 * k + 1] = pj[k]; ec[2 * k + 1] = pi[k]; ep[2 * k + 1] = k;
    for (int a = 0; a < 3; ++a) { re[(2 * k) * 3 + a] = pr[k * 3 + a]; re[(2 * k + 1) * 3 + a] = -pr[k * 3 + a]; }
  }
  for (int i = 0; i < N; ++i) { er[2 * P + i] = i; ec[2 * P + i] = i; ep[2 * P + i] = P; }

  /* ---------------- radial functions per pair (self pair: d = 0) */
  REAL *phi = NEW((size_t)P1 * K), *dphi = NEW((size_t)P1 * K), *C = NEW(P1), *dC = NEW(P1);
  const REAL alpha = 5 / (p->up - p->lo);
  for (int k = 0; k < P1; ++k) {
    REAL d = k < P ? pd[k] : 0, c0, dc0;
    cutoff(d, 0, p->up, &c0, &dc0); /* ExpNormalSmearing's own cutoff has lower = 0: utils.py:371 */
    REAL u = exp(-alpha * (d - p->lo));
    for (int r = 0; r < K; ++r) {
      REAL g = exp(-p->betas[r] * (u - p->means[r]) * (u - p->means[r]));
      phi[(size_t)k * K + r] = c0 * g;
      dphi[(size_t)k * K + r] = dc0 * g + c0 * g * (-2 * p->betas[r] * (u - p->means[r])) * (-alpha * u);
    }
    cutoff(d, p->lo, p->up, &C[k], &dC[k]);
  }
  REAL* kap = NEW(N);
  for (int i = 0; i < N; ++i) kap[i] = q ? 1 + (REAL)0.1 * q[batch[i]] : 1;

  /* ================= forward: embedding */
  REAL* Q = NEW((size_t)P1 * F3);
  for (int c = 0; c < 3; ++c) {
    REAL* tmp = NEW((size_t)P1 * F);
    linear(phi, P1, K, F, p->dp_w[c], p->dp_b[c], tmp);
    for (int k = 0; k < P1; ++k) memcpy(Q + ((size_t)k * 3 + c) * F, tmp + (size_t)k * F, sizeof(REAL) * F);
    free(tmp);
  }
  REAL *Zij = NEW((size_t)E * F);
#pragma omp parallel for schedule(static)
  for (int e = 0; e < E; ++e) { /* emb2([emb(z_i), emb(z_j)]), tensornet.py:526-541 */
    REAL cat[2 * F];
    memcpy(cat, p->emb + (size_t)z[er[e]] * F, sizeof(REAL) * F);
    memcpy(cat + F, p->emb + (size_t)z[ec[e]] * F, sizeof(REAL) * F);
    linear(cat, 1, 2 * F, F, p->emb2_w, p->emb2_b, Zij + (size_t)e * F);
  }
  REAL *I0 = NEW((size_t)N * F), *vec = NEW((size_t)N * 3 * F), *T6 = NEW((size_t)N * 6 * F), *Wd = NEW((size_t)E * F3);
  for (int e = 0; e < E; ++e) {
    const REAL* r = re + (size_t)e * 3;
    const REAL r6[6] = {r[0] * r[0], r[0] * r[1], r[0] * r[2], r[1] * r[1], r[1] * r[2], r[2] * r[2]};
    for (int f = 0; f < F; ++f) {
      REAL cz = C[ep[e]] * Zij[(size_t)e * F + f];
      REAL W0 = cz * Q[((size_t)ep[e] * 3 + 0) * F + f], W1 = cz * Q[((size_t)ep[e] * 3 + 1) * F + f],
           W2 = cz * Q[((size_t)ep[e] * 3 + 2) * F + f];
      Wd[((size_t)e * 3 + 0) * F + f] = W0; Wd[((size_t)e * 3 + 1) * F + f] = W1; Wd[((size_t)e * 3 + 2) * F + f] = W2;
      I0[(size_t)er[e] * F + f] += W0;
      for (int a = 0; a < 3; ++a) vec[((size_t)er[e] * 3 + a) * F + f] += W1 * r[a];
      for (int a = 0; a < 6; ++a) T6[((size_t)er[e] * 6 + a) * F + f] += W2 * r6[a];
    }
  }
  const size_t N9 = (size_t)N * 9 * F;
  REAL *u0 = NEW(N9), *s0n = NEW((size_t)N * F);
  for (int n = 0; n < N; ++n)
    for (int f = 0; f < F; ++f) {
      const REAL* t = T6 + (size_t)n * 6 * F + f;
      REAL trT = (t[0] + t[3 * F] + t[5 * F]) / 3;
      REAL u[9] = {I0[(size_t)n * F + f], vec[((size_t)n * 3) * F + f], vec[((size_t)n * 3 + 1) * F + f], vec[((size_t)n * 3 + 2) * F + f],
                   t[0] - trT, t[F], t[2 * F], t[3 * F] - trT, t[4 * F]};
      for (int c = 0; c < 9; ++c) u0[((size_t)n * 9 + c) * F + f] = u[c];
      s0n[(size_t)n * F + f] = quad(u);
    }
  REAL *ln0 = NEW((size_t)N * F), *xh0 = NEW((size_t)N * F), *rstd0 = NEW(N);
  layernorm_fwd(s0n, N, F, p->ln0_w, p->ln0_b, ln0, xh0, rstd0);
  REAL *a1 = NEW((size_t)N * 2 * F), *h1 = NEW((size_t)N * 2 * F), *a2 = NEW((size_t)N * F3), *gates = NEW((size_t)N * F3);
  linear(ln0, N, F, 2 * F, p->L1_w, p->L1_b, a1);
  for (size_t k = 0; k < (size_t)N * 2 * F; ++k) h1[k] = silu(a1[k]);
  linear(h1, N, 2 * F, F3, p->L2_w, p->L2_b, a2);
  for (size_t k = 0; k < (size_t)N * F3; ++k) gates[k] = silu(a2[k]);
  REAL* UX = NEW(N9);
  tensor_linear(u0, N, F, p->Ue, 0, UX);
  REAL** X = malloc(sizeof(REAL*) * (L + 1));
  X[0] = NEW(N9);
  for (int n = 0; n < N; ++n)
    for (int c = 0; c < 9; ++c)
      for (int f = 0; f < F; ++f)
        X[0][((size_t)n * 9 + c) * F + f] = UX[((size_t)n * 9 + c) * F + f] * gates[((size_t)n * 3 + TYPE_OF[c]) * F + f];

  /* ================= forward: interaction layers */
  REAL *e1[MAXL], *e2[MAXL], *e3[MAXL], *w[MAXL], *Pn[MAXL], *Mi[MAXL], *D[MAXL];
  REAL *Xh = NEW(N9), *Ch = NEW(N9), *tA = NEW((size_t)P1 * 2 * F), *tB = NEW((size_t)P1 * 2 * F);
  for (int l = 0; l < L; ++l) {
    e1[l] = NEW((size_t)P1 * F); e2[l] = NEW((size_t)P1 * 2 * F); e3[l] = NEW((size_t)P1 * F3); w[l] = NEW((size_t)P1 * F3);
    Pn[l] = NEW(N9); Mi[l] = NEW(N9); D[l] = NEW(N9); X[l + 1] = NEW(N9);
    linear(phi, P1, K, F, p->M_w[l][0], p->M_b[l][0], e1[l]);
    for (size_t k = 0; k < (size_t)P1 * F; ++k) tA[k] = silu(e1[l][k]);
    linear(tA, P1, F, 2 * F, p->M_w[l][1], p->M_b[l][1], e2[l]);
    for (size_t k = 0; k < (size_t)P1 * 2 * F; ++k) tB[k] = silu(e2[l][k]);
    linear(tB, P1, 2 * F, F3, p->M_w[l][2], p->M_b[l][2], e3[l]);
    for (int k = 0; k < P1; ++k)
      for (int f = 0; f < F3; ++f) w[l][(size_t)k * F3 + f] = silu(e3[l][(size_t)k * F3 + f]) * C[k];
    for (int n = 0; n < N; ++n)
      for (int f = 0; f < F; ++f) {
        REAL u[9];
        for (int c = 0; c < 9; ++c) u[c] = X[l][((size_t)n * 9 + c) * F + f];
        REAL inv = 1 / (quad(u) + 1); /* tensornet.py:745 */
        for (int c = 0; c < 9; ++c) Xh[((size_t)n * 9 + c) * F + f] = u[c] * inv;
      }
    tensor_linear(Xh, N, F, (const REAL* const*)p->V[l], 0, Pn[l]);
    gather_sum(N, F, E, er, ec, ep, w[l], Pn[l], Mi[l]);
    for (int n = 0; n < N; ++n)
      for (int f = 0; f < F; ++f) {
        REAL y[9], m[9], Y[3][3], M[3][3], A1[3][3], A2[3][3], Cm[3][3], uc[9];
        for (int c = 0; c < 9; ++c) { y[c] = Pn[l][((size_t)n * 9 + c) * F + f]; m[c] = Mi[l][((size_t)n * 9 + c) * F + f]; }
        compose(y, Y); compose(m, M); mm3(Y, M, A1); mm3(M, Y, A2);
        REAL t = 0;
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) {
          Cm[a][b] = p->o3 ? kap[n] * (A1[a][b] + A2[a][b]) : 2 * A1[a][b]; /* tensornet.py:788-793 */
          t += Cm[a][b] * Cm[a][b];
        }
        decomp(Cm, uc);
        for (int c = 0; c < 9; ++c) Ch[((size_t)n * 9 + c) * F + f] = uc[c] / (t + 1);
      }
    tensor_linear(Ch, N, F, (const REAL* const*)(p->V[l] + 3), 0, D[l]);
    for (int n = 0; n < N; ++n)
      for (int f = 0; f < F; ++f) {
        REAL xh[9], d[9], Xf[3][3], dX[3][3], dd[3][3], o[9];
        for (int c = 0; c < 9; ++c) { xh[c] = Xh[((size_t)n * 9 + c) * F + f]; d[c] = D[l][((size_t)n * 9 + c) * F + f]; }
        compose(xh, Xf); compose(d, dX); mm3(dX, dX, dd);
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) Xf[a][b] += dX[a][b] + kap[n] * dd[a][b]; /* :812 */
        decomp(Xf, o);
        for (int c = 0; c < 9; ++c) X[l + 1][((size_t)n * 9 + c) * F + f] = o[c];
      }
  }
  /* ================= forward: readout, head, per-molecule sum */
  REAL *feat = NEW((size_t)N * F3), *lnr = NEW((size_t)N * F3), *xhr = NEW((size_t)N * F3), *rstdr = NEW(N);
  for (int n = 0; n < N; ++n)
    for (int f = 0; f < F; ++f) {
      REAL u[9];
      for (int c = 0; c < 9; ++c) u[c] = X[L][((size_t)n * 9 + c) * F + f];
      REAL t = u[4] + u[7];
      feat[(size_t)n * F3 + f] = 3 * u[0] * u[0];
      feat[(size_t)n * F3 + F + f] = 2 * (u[1] * u[1] + u[2] * u[2] + u[3] * u[3]);
      feat[(size_t)n * F3 + 2 * F + f] = u[4] * u[4] + u[7] * u[7] + t * t + 2 * (u[5] * u[5] + u[6] * u[6] + u[8] * u[8]);
    }
  layernorm_fwd(feat, N, F3, p->lnr_w, p->lnr_b, lnr, xhr, rstdr);
  REAL *al = NEW((size_t)N * F), *x = NEW((size_t)N * F), *ao = NEW((size_t)N * H);
  linear(lnr, N, F3, F, p->Lin_w, p->Lin_b, al);
  for (size_t k = 0; k < (size_t)N * F; ++k) x[k] = silu(al[k]);
  linear(x, N, F, H, p->O1_w, p->O1_b, ao);
  for (int m = 0; m < B; ++m) energy[m] = p->mean;
  for (int n = 0; n < N; ++n) {
    REAL s = p->O2_b[0];
    for (int k = 0; k < H; ++k) s += silu(ao[(size_t)n * H + k]) * p->O2_w[k];
    s *= p->std;
    if (p->has_atomref) s += p->atomref[z[n]];
    energy[batch[n]] += s;
  }
  if (!forces) goto done;

  /* ================= reverse pass, seed dL/dE_m = 1 */
  {
    REAL *g_ao = NEW((size_t)N * H), *g_x = NEW((size_t)N * F), *g_ln = NEW((size_t)N * F3), *g_feat = NEW((size_t)N * F3);
    for (int n = 0; n < N; ++n) for (int k = 0; k < H; ++k) g_ao[(size_t)n * H + k] = p->std * p->O2_w[k] * dsilu(ao[(size_t)n * H + k]);
    linear_T(g_ao, N, F, H, p->O1_w, g_x);
    for (size_t k = 0; k < (size_t)N * F; ++k) g_x[k] *= dsilu(al[k]);
    linear_T(g_x, N, F3, F, p->Lin_w, g_ln);
    layernorm_bwd(g_ln, N, F3, xhr, rstdr, p->lnr_w, g_feat);
    REAL* G = NEW(N9);
    for (int n = 0; n < N; ++n)
      for (int f = 0; f < F; ++f) {
        REAL u[9], dq[9];
        for (int c = 0; c < 9; ++c) u[c] = X[L][((size_t)n * 9 + c) * F + f];
        dquad(u, dq);
        for (int c = 0; c < 9; ++c) G[((size_t)n * 9 + c) * F + f] = dq[c] * g_feat[(size_t)n * F3 + TYPE_OF[c] * F + f];
      }
    REAL *g_phi = NEW((size_t)P1 * K), *gC = NEW(P1);
    REAL *gD = NEW(N9), *gCh = NEW(N9), *gMi = NEW(N9), *gPn = NEW(N9), *gXl = NEW(N9), *tmp9 = NEW(N9);
    REAL *g_e3 = NEW((size_t)P1 * F3), *g_e2 = NEW((size_t)P1 * 2 * F), *g_e1 = NEW((size_t)P1 * F), *g_k = NEW((size_t)P1 * K);
    for (int l = L - 1; l >= 0; --l) {
      const REAL* const* V = (const REAL* const*)p->V[l];
      for (int n = 0; n < N; ++n)
        for (int f = 0; f < F; ++f) {
          REAL gg[9], d[9], Gf[3][3], dX[3][3], dXt[3][3], A1[3][3], A2[3][3], o[9];
          for (int c = 0; c < 9; ++c) { gg[c] = G[((size_t)n * 9 + c) * F + f]; d[c] = D[l][((size_t)n * 9 + c) * F + f]; }
          decomp_T(gg, Gf); compose(d, dX); tr3(dX, dXt); mm3(Gf, dXt, A1); mm3(dXt, Gf, A2);
          for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) Gf[a][b] += kap[n] * (A1[a][b] + A2[a][b]);
          compose_T(Gf, o);
          for (int c = 0; c < 9; ++c) gD[((size_t)n * 9 + c) * F + f] = o[c];
        }
      tensor_linear(gD, N, F, V + 3, 1, gCh);
      for (int n = 0; n < N; ++n)
        for (int f = 0; f < F; ++f) {
          REAL gc[9], y[9], m[9], Y[3][3], M[3][3], Yt[3][3], Mt[3][3], A1[3][3], A2[3][3], Cm[3][3], uc[9], guc[9], gCm[3][3],
              gY[3][3], gM[3][3], o[9];
          for (int c = 0; c < 9; ++c) {
            gc[c] = gCh[((size_t)n * 9 + c) * F + f]; y[c] = Pn[l][((size_t)n * 9 + c) * F + f]; m[c] = Mi[l][((size_t)n * 9 + c) * F + f];
          }
          compose(y, Y); compose(m, M); mm3(Y, M, A1); mm3(M, Y, A2);
          REAL t = 0, dot = 0;
          for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) {
            Cm[a][b] = p->o3 ? kap[n] * (A1[a][b] + A2[a][b]) : 2 * A1[a][b];
            t += Cm[a][b] * Cm[a][b];
          }
          decomp(Cm, uc);
          for (int c = 0; c < 9; ++c) { dot += gc[c] * uc[c]; guc[c] = gc[c] / (t + 1); }
          REAL g_t = -dot / ((t + 1) * (t + 1));
          decomp_T(guc, gCm);
          for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) gCm[a][b] += 2 * Cm[a][b] * g_t;
          tr3(Y, Yt); tr3(M, Mt);
          if (p->o3) {
            mm3(gCm, Mt, A1); mm3(Mt, gCm, A2);
            for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) gY[a][b] = kap[n] * (A1[a][b] + A2[a][b]);
            mm3(Yt, gCm, A1); mm3(gCm, Yt, A2);
            for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) gM[a][b] = kap[n] * (A1[a][b] + A2[a][b]);
          } else {
            mm3(gCm, Mt, A1); mm3(Yt, gCm, A2);
            for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) { gY[a][b] = 2 * A1[a][b]; gM[a][b] = 2 * A2[a][b]; }
          }
          compose_T(gM, o);
          for (int c = 0; c < 9; ++c) gMi[((size_t)n * 9 + c) * F + f] = o[c];
          compose_T(gY, o);
          for (int c = 0; c < 9; ++c) gPn[((size_t)n * 9 + c) * F + f] = o[c];
        }
      gather_sum(N, F, E, er, ec, ep, w[l], gMi, tmp9); /* symmetric graph + pair-symmetric weights */
      for (size_t k = 0; k < N9; ++k) gPn[k] += tmp9[k];
#pragma omp parallel for schedule(static)
      for (int k = 0; k < P; ++k) { /* the self pair has no position dependence */
        for (int f = 0; f < F; ++f) {
          REAL gw[3] = {0, 0, 0};
          for (int c = 0; c < 9; ++c)
            gw[TYPE_OF[c]] += gMi[((size_t)pi[k] * 9 + c) * F + f] * Pn[l][((size_t)pj[k] * 9 + c) * F + f] +
                               gMi[((size_t)pj[k] * 9 + c) * F + f] * Pn[l][((size_t)pi[k] * 9 + c) * F + f];
          for (int t = 0; t < 3; ++t) {
            REAL e = e3[l][(size_t)k * F3 + t * F + f];
            gC[k] += gw[t] * silu(e);
            g_e3[(size_t)k * F3 + t * F + f] = gw[t] * C[k] * dsilu(e);
          }
        }
      }
      linear_T(g_e3, P, 2 * F, F3, p->M_w[l][2], g_e2);
      for (size_t k = 0; k < (size_t)P * 2 * F; ++k) g_e2[k] *= dsilu(e2[l][k]);
      linear_T(g_e2, P, F, 2 * F, p->M_w[l][1], g_e1);
      for (size_t k = 0; k < (size_t)P * F; ++k) g_e1[k] *= dsilu(e1[l][k]);
      linear_T(g_e1, P, K, F, p->M_w[l][0], g_k);
      for (size_t k = 0; k < (size_t)P * K; ++k) g_phi[k] += g_k[k];
      tensor_linear(gPn, N, F, V, 1, gXl);
      for (int n = 0; n < N; ++n)
        for (int f = 0; f < F; ++f) {
          REAL u[9], gx[9], dq[9];
          REAL dot = 0;
          for (int c = 0; c < 9; ++c) {
            u[c] = X[l][((size_t)n * 9 + c) * F + f];
            gx[c] = G[((size_t)n * 9 + c) * F + f] + gXl[((size_t)n * 9 + c) * F + f]; /* residual enters at X_hat */
            dot += gx[c] * u[c];
          }
          dquad(u, dq);
          REAL s = quad(u), g_s = -dot / ((s + 1) * (s + 1));
          for (int c = 0; c < 9; ++c) G[((size_t)n * 9 + c) * F + f] = gx[c] / (s + 1) + dq[c] * g_s;
        }
    }
    /* ---- embedding adjoint */
    REAL *gUX = NEW(N9), *g_a2 = NEW((size_t)N * F3), *g_h1 = NEW((size_t)N * 2 * F), *g_ln0 = NEW((size_t)N * F), *g_s0n = NEW((size_t)N * F);
    for (int n = 0; n < N; ++n)
      for (int f = 0; f < F; ++f) {
        REAL acc[3] = {0, 0, 0};
        for (int c = 0; c < 9; ++c) {
          REAL g = G[((size_t)n * 9 + c) * F + f];
          gUX[((size_t)n * 9 + c) * F + f] = g * gates[((size_t)n * 3 + TYPE_OF[c]) * F + f];
          acc[TYPE_OF[c]] += g * UX[((size_t)n * 9 + c) * F + f];
        }
        for (int t = 0; t < 3; ++t) g_a2[(size_t)n * F3 + t * F + f] = acc[t] * dsilu(a2[(size_t)n * F3 + t * F + f]);
      }
    linear_T(g_a2, N, 2 * F, F3, p->L2_w, g_h1);
    for (size_t k = 0; k < (size_t)N * 2 * F; ++k) g_h1[k] *= dsilu(a1[k]);
    linear_T(g_h1, N, F, 2 * F, p->L1_w, g_ln0);
    layernorm_bwd(g_ln0, N, F, xh0, rstd0, p->ln0_w, g_s0n);
    REAL *g_u0 = NEW(N9), *gA = NEW((size_t)N * 10 * F);
    tensor_linear(gUX, N, F, p->Ue, 1, g_u0);
    for (int n = 0; n < N; ++n)
      for (int f = 0; f < F; ++f) {
        REAL u[9], dq[9], g[9];
        for (int c = 0; c < 9; ++c) u[c] = u0[((size_t)n * 9 + c) * F + f];
        dquad(u, dq);
        for (int c = 0; c < 9; ++c) g[c] = g_u0[((size_t)n * 9 + c) * F + f] + dq[c] * g_s0n[(size_t)n * F + f];
        REAL third = (g[4] + g[7]) / 3;
        REAL o[10] = {g[0], g[1], g[2], g[3], g[4] - third, g[5], g[6], g[7] - third, g[8], -third};
        for (int c = 0; c < 10; ++c) gA[((size_t)n * 10 + c) * F + f] = o[c];
      }
    REAL *gQ = NEW((size_t)P1 * F3), *g_rh = NEW((size_t)P1 * 3);
    for (int e = 0; e < E; ++e) { /* per directed edge; contributions of the self pair are dropped below */
      const REAL* r = re + (size_t)e * 3;
      const REAL sg = e < 2 * P ? ((e & 1) ? -1 : 1) : 0;
      for (int f = 0; f < F; ++f) {
        const REAL* a = gA + (size_t)er[e] * 10 * F + f;
        REAL gW0 = a[0], gW1 = a[F] * r[0] + a[2 * F] * r[1] + a[3 * F] * r[2];
        REAL gW2 = a[4 * F] * r[0] * r[0] + a[5 * F] * r[0] * r[1] + a[6 * F] * r[0] * r[2] + a[7 * F] * r[1] * r[1] +
                   a[8 * F] * r[1] * r[2] + a[9 * F] * r[2] * r[2];
        REAL zc = Zij[(size_t)e * F + f];
        const REAL* qq = Q + (size_t)ep[e] * F3 + f;
        gC[ep[e]] += zc * (gW0 * qq[0] + gW1 * qq[F] + gW2 * qq[2 * F]);
        gQ[(size_t)ep[e] * F3 + f] += gW0 * C[ep[e]] * zc;
        gQ[(size_t)ep[e] * F3 + F + f] += gW1 * C[ep[e]] * zc;
        gQ[(size_t)ep[e] * F3 + 2 * F + f] += gW2 * C[ep[e]] * zc;
        REAL W1 = Wd[((size_t)e * 3 + 1) * F + f], W2 = Wd[((size_t)e * 3 + 2) * F + f];
        REAL d0 = 2 * a[4 * F] * r[0] + a[5 * F] * r[1] + a[6 * F] * r[2];
        REAL d1 = a[5 * F] * r[0] + 2 * a[7 * F] * r[1] + a[8 * F] * r[2];
        REAL d2 = a[6 * F] * r[0] + a[8 * F] * r[1] + 2 * a[9 * F] * r[2];
        g_rh[(size_t)ep[e] * 3 + 0] += sg * (a[F] * W1 + d0 * W2);
        g_rh[(size_t)ep[e] * 3 + 1] += sg * (a[2 * F] * W1 + d1 * W2);
        g_rh[(size_t)ep[e] * 3 + 2] += sg * (a[3 * F] * W1 + d2 * W2);
      }
    }
    for (int c = 0; c < 3; ++c) { /* g_phi += gQ_c P_c */
      REAL *tq = NEW((size_t)P1 * F), *tk = NEW((size_t)P1 * K);
      for (int k = 0; k < P1; ++k) memcpy(tq + (size_t)k * F, gQ + ((size_t)k * 3 + c) * F, sizeof(REAL) * F);
      linear_T(tq, P1, K, F, p->dp_w[c], tk);
      for (size_t k = 0; k < (size_t)P1 * K; ++k) g_phi[k] += tk[k];
      free(tq); free(tk);
    }
    /* ---- geometry: neighbor_utils.py:11-46 (self pair dropped) */
    memset(forces, 0, sizeof(REAL) * (size_t)N * 3);
    for (int k = 0; k < P; ++k) {
      REAL gd = gC[k] * dC[k];
      for (int r = 0; r < K; ++r) gd += g_phi[(size_t)k * K + r] * dphi[(size_t)k * K + r];
      const REAL* r = pr + (size_t)k * 3;
      REAL dot = g_rh[k * 3] * r[0] + g_rh[k * 3 + 1] * r[1] + g_rh[k * 3 + 2] * r[2];
      for (int a = 0; a < 3; ++a) {
        REAL gdel = (g_rh[k * 3 + a] - dot * r[a]) / pd[k] + gd * r[a];
        forces[pi[k] * 3 + a] -= gdel;
        forces[pj[k] * 3 + a] += gdel;
      }
    }
    free(g_ao); free(g_x); free(g_ln); free(g_feat); free(G); free(g_phi); free(gC); free(gD); free(gCh); free(gMi); free(gPn);
    free(gXl); free(tmp9); free(g_e3); free(g_e2); free(g_e1); free(g_k); free(gUX); free(g_a2); free(g_h1); free(g_ln0);
    free(g_s0n); free(g_u0); free(gA); free(gQ); free(g_rh);
  }
done:
  for (int l = 0; l < L; ++l) { free(e1[l]); free(e2[l]); free(e3[l]); free(w[l]); free(Pn[l]); free(Mi[l]); free(D[l]); }
  for (int l = 0; l <= L; ++l) free(X[l]);
  free(X); free(Xh); free(Ch); free(tA); free(tB); free(feat); free(lnr); free(xhr); free(rstdr); free(al); free(x); free(ao);
  free(pi); free(pj); free(pd); free(pr); free(er); free(ec); free(ep); free(re); free(phi); free(dphi); free(C); free(dC); free(kap);
  free(Q); free(Zij); free(I0); free(vec); free(T6); free(Wd); free(u0); free(s0n); free(ln0); free(xh0); free(rstd0);
  free(a1); free(h1); free(a2); free(gates); free(UX);
  return P;
}

int tn_oracle_sizeof_real(void) { return (int)sizeof(REAL); }
int tn_oracle_sizeof_params(void) { return (int)sizeof(TnParams); }
