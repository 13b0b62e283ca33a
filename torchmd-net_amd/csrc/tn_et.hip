// HIP kernels of the Equivariant Transformer energy+force path (reference
// torchmdnet/models/torchmd_et.py:188-426, models/utils.py:45-117, 583-655, output_modules.py:120-163) for gfx950.
//
// Layouts: x [N][F]; vec [N][3][F] (= [3N][F] row-major for the GEMMs); qkv [N][5F] = q | k | vx | v1 | v2 with the
// reference's per-head value layout [H][3][hd] permuted to thirds at parameter upload, so channel c = h*hd + c' is the
// same column in every tensor; per-pair dkv [P+1][Wd] = dk (F) | dvx | dv1 | dv2 (only the parts the model has) and
// its d/dd tangent tkv; the graph is the symmetric CSR of tn_kernels.h (row atom = message TARGET in the forward sweep).
// rhat(target <- source) = (pos_s - pos_t)/d = -esign * prhat for the row atom as target, +esign * prhat as source.
//
// Edge sweeps are row-per-block, one channel per lane, deterministic (no atomics).  The reverse pass runs two sweeps
// over the same CSR: "t" (row = target: g_q and the per-edge scalars g_d, g_rhat) and "s" (row = source: g_k, g_v,
// g_vec); both recompute the attention weights instead of storing per-edge activations.
#include <cstdlib>

#include "tn_common.h"
#include "tn_et.h"

namespace tn {

static inline int cdiv_(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline int bthreads(int F) { return ((F + 63) / 64) * 64; }

__device__ __forceinline__ float head_sum(float v, int hd) {  // sum over the hd lanes of one head (hd = 2^k <= 64)
  v = row_sum(v, hd < 16 ? hd : 16);
  if (hd >= 32) v += __shfl_xor(v, 16, 64);
  if (hd >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}

// ---------------------------------------------------------------------------------------------- embedding
__global__ void k_et_embed(const int64_t* __restrict__ z, const float* __restrict__ emb, int N, int F, float* __restrict__ x) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)N * F) return;
  const int n = (int)(idx / F), c = (int)(idx - (int64_t)n * F);
  x[idx] = emb[z[n] * F + c];
}
void launch_et_embed(const int64_t* z, const float* emb, int N, int F, float* x, hipStream_t s) {
  if (N <= 0) return;
  hipLaunchKernelGGL(k_et_embed, dim3(cdiv_((int64_t)N * F, 256)), dim3(256), 0, s, z, emb, N, F, x);
}

// xcat[i] = [ emb[z_i] | sum_{j != i} Wn[pair] * embN[z_j] ]      (models/utils.py:100-116; self loops removed)
__global__ void k_et_nbr_embed(Graph g, int N, int F, const int64_t* __restrict__ z, const float* __restrict__ emb,
                               const float* __restrict__ embN, const float* __restrict__ Wn, float* __restrict__ xcat) {
  const int i = xcd_chunk(blockIdx.x, gridDim.x);
  if (g.counts[2]) return;
  const int e0 = g.rowptr[i], e1 = g.rowptr[i + 1];
  for (int c = threadIdx.x; c < F; c += blockDim.x) {
    float acc = 0.f;
    int e = e0;
    for (; e + 4 <= e1; e += 4) {  // four edges' rows in flight (the same sum in the same order; self edges contribute zero)
      int j[4], p[4];
      float w[4], m[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        j[u] = g.col[e + u];
        p[u] = g.epair[e + u];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        w[u] = Wn[(int64_t)p[u] * F + c];
        m[u] = embN[z[j[u]] * F + c];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (j[u] != i) acc += w[u] * m[u];
    }
    for (; e < e1; ++e) {
      const int j = g.col[e];
      if (j == i) continue;
      acc += Wn[(int64_t)g.epair[e] * F + c] * embN[z[j] * F + c];
    }
    xcat[(int64_t)i * 2 * F + c] = emb[z[i] * F + c];
    xcat[(int64_t)i * 2 * F + F + c] = acc;
  }
}
void launch_et_nbr_embed(const Graph& g, int N, int F, const int64_t* z, const float* emb, const float* embN, const float* Wn,
                         float* xcat, hipStream_t s) {
  if (N <= 0) return;
  hipLaunchKernelGGL(k_et_nbr_embed, dim3(N), dim3(bthreads(F)), 0, s, g, N, F, z, emb, embN, Wn, xcat);
}

// g_d[pair] += sum_c (g_xn[i,c] embN[z_j,c] + g_xn[j,c] embN[z_i,c]) dWn[pair,c]   (one wave per pair)
__global__ __launch_bounds__(256) void k_et_nbr_embed_bwd(Graph g, int F, const int64_t* __restrict__ z, const float* __restrict__ embN,
                                                           const float* __restrict__ g_xcat, const float* __restrict__ dWn,
                                                           float* __restrict__ gd2) {
  if (g.counts[2]) return;
  const int item = xcd_chunk_act(blockIdx.x, (g.counts[0] + 3) >> 2);
  if (item < 0) return;
  const int p = item * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (p >= g.counts[0]) return;
  const int i = g.pair_i[p], j = g.pair_j[p];
  const float* gi = g_xcat + (int64_t)i * 2 * F + F;
  const float* gj = g_xcat + (int64_t)j * 2 * F + F;
  const float* ei = embN + z[i] * F;
  const float* ej = embN + z[j] * F;
  float acc = 0.f;
  for (int c = lane; c < F; c += 64) acc += (gi[c] * ej[c] + gj[c] * ei[c]) * dWn[(int64_t)p * F + c];
  acc = wave_sum(acc);
  if (lane == 0) gd2[2 * (int64_t)p] += acc;
}
void launch_et_nbr_embed_bwd(const Graph& g, int Pcap, int F, const int64_t* z, const float* embN, const float* g_xcat,
                             const float* dWn, float* gd2, hipStream_t s) {
  if (Pcap <= 0) return;
  hipLaunchKernelGGL(k_et_nbr_embed_bwd, dim3(cdiv_(Pcap, 4)), dim3(256), 0, s, g, F, z, embN, g_xcat, dWn, gd2);
}

// ---------------------------------------------------------------------------------------------- attention, forward
struct EtEdge {  // what one lane needs for one edge (channel c of the row atom r and the neighbour j)
  float A, sx, s1, s2, a, cv, ca;
};

// xagg[t,c] = sum_e sx * A_h ; vagg[t,a,c] = sum_e vec[s,a,c] * s1 + s2 * rhat_a      (torchmd_et.py:376-426)
__global__ void k_et_attn_fwd(Graph g, EtAttnArgs a, float* __restrict__ xagg, float* __restrict__ vagg) {
  const int t = xcd_chunk(blockIdx.x, gridDim.x);
  if (g.counts[2] || (a.tile_open && *a.tile_open == 0)) return;  // (closed tiles: the tile sweeps run instead)
  const int F = a.F, hd = a.hd, c = threadIdx.x;
  const bool live = c < F;
  const int cc = live ? c : 0;
  const int e0 = g.rowptr[t], e1 = g.rowptr[t + 1];
  const int64_t F5 = 5 * (int64_t)F;
  const float qt = a.qkv[(int64_t)t * F5 + cc];
  float xa = 0.f, va0 = 0.f, va1 = 0.f, va2 = 0.f;
  for (int e = e0; e < e1; ++e) {
    const int s = g.col[e], p = g.epair[e];
    const float sg = g.esign[e];
    const float* qs = a.qkv + (int64_t)s * F5 + cc;
    const int64_t dkv_b = (int64_t)p * a.Wd;
    const float C = a.C[p];
    const float cv = a.vector_cutoff ? C : 1.0f, ca = a.vector_cutoff ? 1.0f : C;
    const float dk = a.dk_off >= 0 ? ldpair(a.dkv, dkv_b + (a.dk_off + cc), a.pair_bf16) : 1.0f;
    float ak = live ? qt * qs[F] * dk : 0.f;
    ak = head_sum(ak, hd);
    const float A = silu(ak) * ca;
    float dvx = 1.f, dv1 = 1.f, dv2 = 1.f;
    if (a.dv_off >= 0) {
      dvx = ldpair(a.dkv, dkv_b + (a.dv_off + cc), a.pair_bf16);
      dv1 = ldpair(a.dkv, dkv_b + (a.dv_off + F + cc), a.pair_bf16);
      dv2 = ldpair(a.dkv, dkv_b + (a.dv_off + 2 * F + cc), a.pair_bf16);
    }
    const float sx = qs[2 * F] * cv * dvx, s1 = qs[3 * F] * cv * dv1, s2 = qs[4 * F] * cv * dv2;
    const float* vs = a.vec + (int64_t)s * 3 * F + cc;
    float r0 = 0.f, r1 = 0.f, r2 = 0.f;  // self edge: rhat = 0 (the self pair's geometry slot is not written)
    if (sg != 0.f) {
      r0 = -sg * g.prhat[(int64_t)p * 3];
      r1 = -sg * g.prhat[(int64_t)p * 3 + 1];
      r2 = -sg * g.prhat[(int64_t)p * 3 + 2];
    }
    xa += sx * A;
    va0 += vs[0] * s1 + s2 * r0;
    va1 += vs[F] * s1 + s2 * r1;
    va2 += vs[2 * F] * s1 + s2 * r2;
  }
  if (live) {
    xagg[(int64_t)t * F + c] = xa;
    float* o = vagg + (int64_t)t * 3 * F + c;
    o[0] = va0;
    o[F] = va1;
    o[2 * F] = va2;
  }
}

// ---- pipelined sweeps (the configurations that ship: every channel live, head width <= 16).  The generic kernels above
// walk an edge in three dependent memory phases (indices -> rows of j and of the pair -> second batch) with run-time
// branches on the model's options in between: ~3 us per edge, latency-bound at a third of the HBM rate.  Here the options
// are template parameters and ALL loads of edge e+1 are issued before edge e is computed (one struct of registers ahead).
struct EtFwdIn {
  float kj, vxj, v1j, v2j, dk, dvx, dv1, dv2, vs0, vs1, vs2, C, r0, r1, r2;
};
struct EtIdx {  // an edge's indices: loaded two edges ahead (scalar loads), its rows one edge ahead
  int j, p;
  float sg;
};
__device__ __forceinline__ EtIdx et_idx(const Graph& g, int e) { return EtIdx{g.col[e], g.epair[e], g.esign[e]}; }
// rows walk their edge lists in the order (row + column) mod 64 - symmetric in the two atoms of a pair, so the rows of a
// molecule (dispatched together on one XCD) reach a common pair at the same point of their loops
__device__ __forceinline__ int et_rot_start(const int* __restrict__ col, int e0, int e1, int i) {
  if (e1 <= e0) return e0;
  const int j0 = col[e0];
  const int target = j0 + ((64 - ((i + j0) & 63)) & 63);
  int lo = e0, hi = e1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (col[mid] < target) lo = mid + 1; else hi = mid;
  }
  return lo < e1 ? lo : e0;
}
__device__ __forceinline__ int et_rot(int it, int e0, int e1, int rot) {  // it-th edge of the rotated list
  const int e = it + rot - e0;
  return e < e1 ? e : e - (e1 - e0);
}
template <int HD>
__device__ __forceinline__ float head_sum_t(float v, int hd) {  // HD = 16 / 8 / 4: compile-time width (no branches)
  return HD > 0 ? row_sum(v, HD) : head_sum(v, hd);
}
// pair-row element with the storage mode known at compile time: the fp32 default is a plain load (the run-time select of ldpair
// costs four VALU instructions per value: a fifth of the sweeps' instruction count; ET-SPICE step 11.0 -> 10.6 ms).  The bf16
// instantiation keeps ldpair's run-time form ON PURPOSE: with the flag folded to a constant the same sweep was 0.5 ms per step
// SLOWER (A/B on one box: 9.85 vs 10.39 ms) - the compiler then schedules the unpacking differently.
template <bool BF>
__device__ __forceinline__ float ldpair_t(const float* base, int64_t idx, int rt_bf16) {
  return BF ? ldpair(base, idx, rt_bf16) : base[idx];
}
template <bool HAS_DK, bool HAS_DV, bool BF>
__device__ __forceinline__ void et_fwd_load(const Graph& g, const EtAttnArgs& a, const EtIdx& ix, int c, EtFwdIn& o) {
  const int F = a.F;
  const int s = ix.j, p = ix.p;
  const float sg = ix.sg;
  const float* qs = a.qkv + (int64_t)s * 5 * F + c;
  o.kj = qs[F];
  o.vxj = qs[2 * F];
  o.v1j = qs[3 * F];
  o.v2j = qs[4 * F];
  const int64_t dkv_b = (int64_t)p * a.Wd + c;
  o.dk = HAS_DK ? ldpair_t<BF>(a.dkv, dkv_b + (a.dk_off), a.pair_bf16) : 1.0f;
  o.dvx = HAS_DV ? ldpair_t<BF>(a.dkv, dkv_b + (a.dv_off), a.pair_bf16) : 1.0f;
  o.dv1 = HAS_DV ? ldpair_t<BF>(a.dkv, dkv_b + (a.dv_off + F), a.pair_bf16) : 1.0f;
  o.dv2 = HAS_DV ? ldpair_t<BF>(a.dkv, dkv_b + (a.dv_off + 2 * F), a.pair_bf16) : 1.0f;
  const float* vs = a.vec + (int64_t)s * 3 * F + c;
  o.vs0 = vs[0];
  o.vs1 = vs[F];
  o.vs2 = vs[2 * F];
  o.C = a.C[p];
  const float h0 = g.prhat[(int64_t)p * 3], h1 = g.prhat[(int64_t)p * 3 + 1], h2 = g.prhat[(int64_t)p * 3 + 2];
  o.r0 = sg != 0.f ? -sg * h0 : 0.f;  // self edge: rhat = 0 (the self pair's geometry slot is not written)
  o.r1 = sg != 0.f ? -sg * h1 : 0.f;
  o.r2 = sg != 0.f ? -sg * h2 : 0.f;
}
template <bool HAS_DK, bool HAS_DV, bool VCUT, int HD, bool BF>
__global__ void k_et_attn_fwd_p(Graph g, EtAttnArgs a, float* __restrict__ xagg, float* __restrict__ vagg) {
  const int t = xcd_chunk(blockIdx.x, gridDim.x);
  if (g.counts[2] || (a.tile_open && *a.tile_open == 0)) return;  // (closed tiles: the tile sweeps run instead)
  const int F = a.F, hd = a.hd, c = threadIdx.x;  // blockDim.x == F
  const int e0 = g.rowptr[t], e1 = g.rowptr[t + 1];
  const float qt = a.qkv[(int64_t)t * 5 * F + c];
  float xa = 0.f, va0 = 0.f, va1 = 0.f, va2 = 0.f;
  const int rot = et_rot_start(g.col, e0, e1, t);
  auto add = [&](const EtFwdIn& cur) {
    const float cv = VCUT ? cur.C : 1.0f, ca = VCUT ? 1.0f : cur.C;
    const float ak = head_sum_t<HD>(qt * cur.kj * cur.dk, hd);
    const float A = silu(ak) * ca;
    const float sx = cur.vxj * cv * cur.dvx, s1 = cur.v1j * cv * cur.dv1, s2 = cur.v2j * cv * cur.dv2;
    xa += sx * A;
    va0 += cur.vs0 * s1 + s2 * cur.r0;
    va1 += cur.vs1 * s1 + s2 * cur.r1;
    va2 += cur.vs2 * s1 + s2 * cur.r2;
  };
  // four edges per trip: their rows are requested together (independent loads), then accumulated in list order
  int e = e0;
  for (; e + 4 <= e1; e += 4) {
    EtFwdIn a0, a1, a2, a3;
    et_fwd_load<HAS_DK, HAS_DV, BF>(g, a, et_idx(g, et_rot(e, e0, e1, rot)), c, a0);
    et_fwd_load<HAS_DK, HAS_DV, BF>(g, a, et_idx(g, et_rot(e + 1, e0, e1, rot)), c, a1);
    et_fwd_load<HAS_DK, HAS_DV, BF>(g, a, et_idx(g, et_rot(e + 2, e0, e1, rot)), c, a2);
    et_fwd_load<HAS_DK, HAS_DV, BF>(g, a, et_idx(g, et_rot(e + 3, e0, e1, rot)), c, a3);
    add(a0);
    add(a1);
    add(a2);
    add(a3);
  }
  for (; e < e1; ++e) {
    EtFwdIn a0;
    et_fwd_load<HAS_DK, HAS_DV, BF>(g, a, et_idx(g, et_rot(e, e0, e1, rot)), c, a0);
    add(a0);
  }
  xagg[(int64_t)t * F + c] = xa;
  float* o = vagg + (int64_t)t * 3 * F + c;
  o[0] = va0;
  o[F] = va1;
  o[2 * F] = va2;
}
static bool et_pipelined_ok(const EtAttnArgs& a) {
  static const bool off = getenv("TMDNET_ET_GENERIC_SWEEPS") != nullptr;  // developer switch: the generic kernels
  return !off && a.F % 64 == 0 && a.F <= 1024 && a.hd <= 16;
}
void launch_et_attn_fwd(const Graph& g, int N, const EtAttnArgs& a_in, float* xagg, float* vagg, hipStream_t s) {
  if (N <= 0) return;
  // tile sweeps (16-lane groups, two channels per lane, the tile's node rows in LDS: tn_et_g16.hip) when every tile is closed,
  // the row sweeps below otherwise: the flag lives on the device (no read-back, same launches every step: capturable), the
  // kernels of the generation that is not wanted return at once
  EtAttnArgs a = a_in;
  if (et_g16_ok(N, a.slot_stride / 2, a)) launch_et_attn_fwd_g16(g, N, a, xagg, vagg, s);
  else a.tile_open = nullptr;
  if (et_pipelined_ok(a)) {
    const dim3 grid(N), block(a.F);
#define ET_FWD2(DK, DV, VC, BF)                                                                                    \
  if (a.hd == 16) hipLaunchKernelGGL((k_et_attn_fwd_p<DK, DV, VC, 16, BF>), grid, block, 0, s, g, a, xagg, vagg); \
  else hipLaunchKernelGGL((k_et_attn_fwd_p<DK, DV, VC, 0, BF>), grid, block, 0, s, g, a, xagg, vagg)
#define ET_FWD(DK, DV, VC)              \
  if (a.pair_bf16) {                    \
    ET_FWD2(DK, DV, VC, true);          \
  } else {                              \
    ET_FWD2(DK, DV, VC, false);         \
  }
    const int key = (a.dk_off >= 0 ? 4 : 0) | (a.dv_off >= 0 ? 2 : 0) | (a.vector_cutoff ? 1 : 0);
    switch (key) {
      case 0: ET_FWD(false, false, false); break;
      case 1: ET_FWD(false, false, true); break;
      case 2: ET_FWD(false, true, false); break;
      case 3: ET_FWD(false, true, true); break;
      case 4: ET_FWD(true, false, false); break;
      case 5: ET_FWD(true, false, true); break;
      case 6: ET_FWD(true, true, false); break;
      default: ET_FWD(true, true, true); break;
    }
#undef ET_FWD
#undef ET_FWD2
    return;
  }
  hipLaunchKernelGGL(k_et_attn_fwd, dim3(N), dim3(bthreads(a.F)), 0, s, g, a, xagg, vagg);
}

// x' = x + vdot o2 + o3 ; vec'[a] = vec[a] + vec3[a] o1 + vagg[a] ; vdot = sum_a vec1[a] vec2[a]   (torchmd_et.py:347-353, 215-217)
__global__ void k_et_update(const float* __restrict__ x, const float* __restrict__ vec, const float* __restrict__ vp,
                            const float* __restrict__ o, const float* __restrict__ vagg, int N, int F, float* __restrict__ xn,
                            float* __restrict__ vecn, float* __restrict__ vdot) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)N * F) return;
  const int n = (int)(idx / F), c = (int)(idx - (int64_t)n * F);
  const float* vpn = vp + (int64_t)n * 9 * F + c;  // rows n*3+a of [3N][3F]
  float vd = 0.f;
#pragma unroll
  for (int ax = 0; ax < 3; ++ax) vd += vpn[ax * 3 * F] * vpn[ax * 3 * F + F];
  const float o1 = o[(int64_t)n * 3 * F + c], o2 = o[(int64_t)n * 3 * F + F + c], o3 = o[(int64_t)n * 3 * F + 2 * F + c];
  vdot[idx] = vd;
  xn[idx] = x[idx] + vd * o2 + o3;
#pragma unroll
  for (int ax = 0; ax < 3; ++ax) {
    const int64_t k = (int64_t)n * 3 * F + ax * F + c;
    vecn[k] = vec[k] + vpn[ax * 3 * F + 2 * F] * o1 + vagg[k];
  }
}
void launch_et_update(const float* x, const float* vec, const float* vp, const float* o, const float* vagg, int N, int F, float* xn,
                      float* vecn, float* vdot, hipStream_t s) {
  if (N <= 0) return;
  hipLaunchKernelGGL(k_et_update, dim3(cdiv_((int64_t)N * F, 256)), dim3(256), 0, s, x, vec, vp, o, vagg, N, F, xn, vecn, vdot);
}

// g_o = (sum_a g_vec[a] vec3[a] | g_x vdot | g_x) ; g_vp[a] = (g_x o2 vec2[a] | g_x o2 vec1[a] | g_vec[a] o1)
__global__ void k_et_update_bwd(const float* __restrict__ g_x, const float* __restrict__ g_vec, const float* __restrict__ vp,
                                const float* __restrict__ o, const float* __restrict__ vdot, int N, int F, float* __restrict__ g_o,
                                float* __restrict__ g_vp) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)N * F) return;
  const int n = (int)(idx / F), c = (int)(idx - (int64_t)n * F);
  const float* vpn = vp + (int64_t)n * 9 * F + c;
  float* gvp = g_vp + (int64_t)n * 9 * F + c;
  const float gx = g_x[idx];
  const float o1 = o[(int64_t)n * 3 * F + c], o2 = o[(int64_t)n * 3 * F + F + c];
  float go1 = 0.f;
#pragma unroll
  for (int ax = 0; ax < 3; ++ax) {
    const float gv = g_vec[(int64_t)n * 3 * F + ax * F + c];
    go1 += gv * vpn[ax * 3 * F + 2 * F];
    gvp[ax * 3 * F] = gx * o2 * vpn[ax * 3 * F + F];
    gvp[ax * 3 * F + F] = gx * o2 * vpn[ax * 3 * F];
    gvp[ax * 3 * F + 2 * F] = gv * o1;
  }
  g_o[(int64_t)n * 3 * F + c] = go1;
  g_o[(int64_t)n * 3 * F + F + c] = gx * vdot[idx];
  g_o[(int64_t)n * 3 * F + 2 * F + c] = gx;
}
void launch_et_update_bwd(const float* g_x, const float* g_vec, const float* vp, const float* o, const float* vdot, int N, int F,
                          float* g_o, float* g_vp, hipStream_t s) {
  if (N <= 0) return;
  hipLaunchKernelGGL(k_et_update_bwd, dim3(cdiv_((int64_t)N * F, 256)), dim3(256), 0, s, g_x, g_vec, vp, o, vdot, N, F, g_o, g_vp);
}

// ---------------------------------------------------------------------------------------------- attention, reverse
// recompute the forward quantities of one edge for channel cc; `tq` = q of the TARGET, `sq` = qkv row of the SOURCE
__device__ __forceinline__ EtEdge et_edge(const EtAttnArgs& a, float tq, const float* __restrict__ sq, int p, int cc, bool live,
                                          float& dk, float& dvx, float& dv1, float& dv2) {
  const int F = a.F;
  const int64_t dkv_b = (int64_t)p * a.Wd;
  const float C = a.C[p];
  EtEdge r;
  r.cv = a.vector_cutoff ? C : 1.0f;
  r.ca = a.vector_cutoff ? 1.0f : C;
  dk = a.dk_off >= 0 ? ldpair(a.dkv, dkv_b + (a.dk_off + cc), a.pair_bf16) : 1.0f;
  float ak = live ? tq * sq[F] * dk : 0.f;
  r.a = head_sum(ak, a.hd);
  r.A = silu(r.a) * r.ca;
  dvx = dv1 = dv2 = 1.f;
  if (a.dv_off >= 0) {
    dvx = ldpair(a.dkv, dkv_b + (a.dv_off + cc), a.pair_bf16);
    dv1 = ldpair(a.dkv, dkv_b + (a.dv_off + F + cc), a.pair_bf16);
    dv2 = ldpair(a.dkv, dkv_b + (a.dv_off + 2 * F + cc), a.pair_bf16);
  }
  r.sx = sq[2 * F] * r.cv * dvx;
  r.s1 = sq[3 * F] * r.cv * dv1;
  r.s2 = sq[4 * F] * r.cv * dv2;
  return r;
}

// Both roles of the row atom r in ONE sweep (the per-pair rows dkv / tkv are then read once per directed edge instead of
// twice): as TARGET of the message j -> r (g_q[r], the per-edge scalars g_d, g_rhat) and as SOURCE of the message r -> j
// (g_k[r], g_v[r], g_vec[r]).  rhat(r <- j) = -esign * prhat, rhat(j <- r) = +esign * prhat.
__global__ void k_et_attn_bwd(Graph g, EtAttnArgs a, const float* __restrict__ g_xagg, const float* __restrict__ g_vagg,
                              float* __restrict__ g_qkv, float* __restrict__ g_vec, float* __restrict__ gd2, float* __restrict__ gr2) {
  const int r = xcd_chunk(blockIdx.x, gridDim.x);
  if (g.counts[2] || (a.tile_open && *a.tile_open == 0)) return;  // (closed tiles: the tile sweeps run instead)
  const int F = a.F, c = threadIdx.x, lane = c & 63, wave = c >> 6;
  const bool live = c < F;
  const int cc = live ? c : 0;
  const int e0 = g.rowptr[r], e1 = g.rowptr[r + 1];
  const int64_t F5 = 5 * (int64_t)F;
  const float* rq = a.qkv + (int64_t)r * F5 + cc;
  const float qr = rq[0], kr = rq[F], vxr = rq[2 * F], v1r = rq[3 * F], v2r = rq[4 * F];
  const float* vr = a.vec + (int64_t)r * 3 * F + cc;
  const float vr0 = vr[0], vr1 = vr[F], vr2 = vr[2 * F];
  const float gxr = live ? g_xagg[(int64_t)r * F + cc] : 0.f;
  const float* gvr = g_vagg + (int64_t)r * 3 * F + cc;
  const float gr0 = live ? gvr[0] : 0.f, gr1 = live ? gvr[F] : 0.f, gr2_ = live ? gvr[2 * F] : 0.f;
  float gq = 0.f, gk = 0.f, gvx = 0.f, gv1 = 0.f, gv2 = 0.f, gvec0 = 0.f, gvec1 = 0.f, gvec2 = 0.f;
  for (int e = e0; e < e1; ++e) {
    const int j = g.col[e], p = g.epair[e];
    const float sg = g.esign[e];
    const float* jq = a.qkv + (int64_t)j * F5 + cc;
    const float qj = jq[0], kj = jq[F], vxj = jq[2 * F], v1j = jq[3 * F], v2j = jq[4 * F];
    const int64_t dkv_b = (int64_t)p * a.Wd;
    const int64_t tkv_b = (int64_t)p * a.Wd;
    const float C = a.C[p];
    const float cv = a.vector_cutoff ? C : 1.0f, ca = a.vector_cutoff ? 1.0f : C;
    float dk = 1.f, tk = 0.f, dvx = 1.f, dv1 = 1.f, dv2 = 1.f, tvx = 0.f, tv1 = 0.f, tv2 = 0.f;
    if (a.dk_off >= 0) {
      dk = ldpair(a.dkv, dkv_b + (a.dk_off + cc), a.pair_bf16);
      tk = ldpair(a.tkv, tkv_b + (a.dk_off + cc), a.pair_bf16);
    }
    if (a.dv_off >= 0) {
      dvx = ldpair(a.dkv, dkv_b + (a.dv_off + cc), a.pair_bf16);
      dv1 = ldpair(a.dkv, dkv_b + (a.dv_off + F + cc), a.pair_bf16);
      dv2 = ldpair(a.dkv, dkv_b + (a.dv_off + 2 * F + cc), a.pair_bf16);
      tvx = ldpair(a.tkv, tkv_b + (a.dv_off + cc), a.pair_bf16);
      tv1 = ldpair(a.tkv, tkv_b + (a.dv_off + F + cc), a.pair_bf16);
      tv2 = ldpair(a.tkv, tkv_b + (a.dv_off + 2 * F + cc), a.pair_bf16);
    }
    float p0 = 0.f, p1 = 0.f, p2 = 0.f;  // prhat with the edge's sign: rhat(j <- r); rhat(r <- j) is its negative
    if (sg != 0.f) {
      p0 = sg * g.prhat[(int64_t)p * 3];
      p1 = sg * g.prhat[(int64_t)p * 3 + 1];
      p2 = sg * g.prhat[(int64_t)p * 3 + 2];
    }
    // ---- role TARGET: message j -> r
    {
      const float at = head_sum(live ? qr * kj * dk : 0.f, a.hd);
      const float A = silu(at) * ca;
      const float sx = vxj * cv * dvx, s2 = v2j * cv * dv2;
      const float* vj = a.vec + (int64_t)j * 3 * F + cc;
      const float g_sx = gxr * A;
      const float g_A = head_sum(gxr * sx, a.hd);
      const float g_s1 = gr0 * vj[0] + gr1 * vj[F] + gr2_ * vj[2 * F];
      const float g_s2 = -(gr0 * p0 + gr1 * p1 + gr2_ * p2);
      const float g_a = g_A * silu_grad(at) * ca;
      gq += g_a * kj * dk;
      float gd = cv * (g_sx * vxj * tvx + g_s1 * v1j * tv1 + g_s2 * v2j * tv2) + g_a * qr * kj * tk;
      const float gcv = g_sx * vxj * dvx + g_s1 * v1j * dv1 + g_s2 * v2j * dv2;
      const float gca = ((cc % a.hd) == 0) ? g_A * silu(at) : 0.f;
      gd += (a.vector_cutoff ? gcv : gca) * a.dC[p];
      if (!live) gd = 0.f;
      // g_rhat is taken with respect to rhat(r <- j) = -p: the slot convention of k_et_pair_combine
      const float tot = wave_sum4(gd, live ? gr0 * s2 : 0.f, live ? gr1 * s2 : 0.f, live ? gr2_ * s2 : 0.f, lane);
      if (sg != 0.f && (lane & 15) == 0) {
        const int64_t slot = (int64_t)wave * a.slot_stride + 2 * (int64_t)p + (sg > 0.f ? 0 : 1);
        const int comp = lane >> 4;
        if (comp == 0) gd2[slot] = tot;
        else gr2[slot * 3 + comp - 1] = tot;
        // the caller sums et_sweep_waves(F) slot arrays per pair direction (one per 32 channels: the tile sweeps' count); this
        // sweep fills one per 64 channels - the last wave of the row zeroes the others' entries of this (pair, direction)
        if (wave == (int)(blockDim.x >> 6) - 1) {
          const int nw_all = (F % 32 == 0) ? F / 32 : (int)(blockDim.x >> 6);
          for (int ws = (int)(blockDim.x >> 6); ws < nw_all; ++ws) {
            const int64_t sz = slot + (int64_t)(ws - wave) * a.slot_stride;
            if (comp == 0) gd2[sz] = 0.f;
            else gr2[sz * 3 + comp - 1] = 0.f;
          }
        }
      }
    }
    // ---- role SOURCE: message r -> j
    {
      const float as = head_sum(live ? qj * kr * dk : 0.f, a.hd);
      const float A = silu(as) * ca;
      const float sx = vxr * cv * dvx, s1 = v1r * cv * dv1;
      const float gxj = live ? g_xagg[(int64_t)j * F + cc] : 0.f;
      const float* gvj = g_vagg + (int64_t)j * 3 * F + cc;
      const float gj0 = gvj[0], gj1 = gvj[F], gj2 = gvj[2 * F];
      const float g_A = head_sum(gxj * sx, a.hd);
      const float g_a = g_A * silu_grad(as) * ca;
      gk += g_a * qj * dk;
      gvx += gxj * A * cv * dvx;
      gv1 += (gj0 * vr0 + gj1 * vr1 + gj2 * vr2) * cv * dv1;
      gv2 += (gj0 * p0 + gj1 * p1 + gj2 * p2) * cv * dv2;
      gvec0 += gj0 * s1;
      gvec1 += gj1 * s1;
      gvec2 += gj2 * s1;
    }
  }
  if (live) {
    float* o = g_qkv + (int64_t)r * F5 + c;
    o[0] = gq;
    o[F] = gk;
    o[2 * F] = gvx;
    o[3 * F] = gv1;
    o[4 * F] = gv2;
    float* gv = g_vec + (int64_t)r * 3 * F + c;
    gv[0] += gvec0;
    gv[F] += gvec1;
    gv[2 * F] += gvec2;
  }
}

struct EtBwdIn {
  float qj, kj, vxj, v1j, v2j, dk, tk, dvx, dv1, dv2, tvx, tv1, tv2, vj0, vj1, vj2, gxj, gj0, gj1, gj2, C, dC, p0, p1, p2, sg;
  int p;
};
template <bool HAS_DK, bool HAS_DV, bool BF>
__device__ __forceinline__ void et_bwd_load(const Graph& g, const EtAttnArgs& a, const float* __restrict__ g_xagg,
                                            const float* __restrict__ g_vagg, const EtIdx& ix, int c, EtBwdIn& o) {
  const int F = a.F;
  const int j = ix.j, p = ix.p;
  const float sg = ix.sg;
  o.p = p;
  o.sg = sg;
  const float* jq = a.qkv + (int64_t)j * 5 * F + c;
  o.qj = jq[0];
  o.kj = jq[F];
  o.vxj = jq[2 * F];
  o.v1j = jq[3 * F];
  o.v2j = jq[4 * F];
  const int64_t dkv_b = (int64_t)p * a.Wd + c;
  const int64_t tkv_b = (int64_t)p * a.Wd + c;
  o.dk = HAS_DK ? ldpair_t<BF>(a.dkv, dkv_b + (a.dk_off), a.pair_bf16) : 1.f;
  o.tk = HAS_DK ? ldpair_t<BF>(a.tkv, tkv_b + (a.dk_off), a.pair_bf16) : 0.f;
  o.dvx = HAS_DV ? ldpair_t<BF>(a.dkv, dkv_b + (a.dv_off), a.pair_bf16) : 1.f;
  o.dv1 = HAS_DV ? ldpair_t<BF>(a.dkv, dkv_b + (a.dv_off + F), a.pair_bf16) : 1.f;
  o.dv2 = HAS_DV ? ldpair_t<BF>(a.dkv, dkv_b + (a.dv_off + 2 * F), a.pair_bf16) : 1.f;
  o.tvx = HAS_DV ? ldpair_t<BF>(a.tkv, tkv_b + (a.dv_off), a.pair_bf16) : 0.f;
  o.tv1 = HAS_DV ? ldpair_t<BF>(a.tkv, tkv_b + (a.dv_off + F), a.pair_bf16) : 0.f;
  o.tv2 = HAS_DV ? ldpair_t<BF>(a.tkv, tkv_b + (a.dv_off + 2 * F), a.pair_bf16) : 0.f;
  const float* vj = a.vec + (int64_t)j * 3 * F + c;
  o.vj0 = vj[0];
  o.vj1 = vj[F];
  o.vj2 = vj[2 * F];
  o.gxj = g_xagg[(int64_t)j * F + c];
  const float* gvj = g_vagg + (int64_t)j * 3 * F + c;
  o.gj0 = gvj[0];
  o.gj1 = gvj[F];
  o.gj2 = gvj[2 * F];
  o.C = a.C[p];
  o.dC = a.dC[p];
  const float h0 = g.prhat[(int64_t)p * 3], h1 = g.prhat[(int64_t)p * 3 + 1], h2 = g.prhat[(int64_t)p * 3 + 2];
  o.p0 = sg != 0.f ? sg * h0 : 0.f;  // prhat with the edge's sign: rhat(j <- r); rhat(r <- j) is its negative
  o.p1 = sg != 0.f ? sg * h1 : 0.f;
  o.p2 = sg != 0.f ? sg * h2 : 0.f;
}
// k_et_attn_bwd (both roles of the row atom in one sweep), pipelined: same arithmetic in the same order
template <bool HAS_DK, bool HAS_DV, bool VCUT, int HD, bool BF>
__global__ void k_et_attn_bwd_p(Graph g, EtAttnArgs a, const float* __restrict__ g_xagg, const float* __restrict__ g_vagg,
                                float* __restrict__ g_qkv, float* __restrict__ g_vec, float* __restrict__ gd2,
                                float* __restrict__ gr2) {
  const int r = xcd_chunk(blockIdx.x, gridDim.x);
  if (g.counts[2] || (a.tile_open && *a.tile_open == 0)) return;  // (closed tiles: the tile sweeps run instead)
  const int F = a.F, c = threadIdx.x, lane = c & 63, wave = c >> 6, hd = a.hd;  // blockDim.x == F
  const int e0 = g.rowptr[r], e1 = g.rowptr[r + 1];
  const int64_t F5 = 5 * (int64_t)F;
  const float* rq = a.qkv + (int64_t)r * F5 + c;
  const float qr = rq[0], kr = rq[F], vxr = rq[2 * F], v1r = rq[3 * F], v2r = rq[4 * F];
  const float* vr = a.vec + (int64_t)r * 3 * F + c;
  const float vr0 = vr[0], vr1 = vr[F], vr2 = vr[2 * F];
  const float gxr = g_xagg[(int64_t)r * F + c];
  const float* gvr = g_vagg + (int64_t)r * 3 * F + c;
  const float gr0 = gvr[0], gr1 = gvr[F], gr2_ = gvr[2 * F];
  const bool head0 = (c % hd) == 0;
  float gq = 0.f, gk = 0.f, gvx = 0.f, gv1 = 0.f, gv2 = 0.f, gvec0 = 0.f, gvec1 = 0.f, gvec2 = 0.f;
  EtBwdIn u, nxt;
  EtIdx in = {0, 0, 0.f}, inn;
  const int rot = et_rot_start(g.col, e0, e1, r);
  if (e0 < e1) {
    et_bwd_load<HAS_DK, HAS_DV, BF>(g, a, g_xagg, g_vagg, et_idx(g, et_rot(e0, e0, e1, rot)), c, u);
    in = et_idx(g, et_rot(e0 + 1 < e1 ? e0 + 1 : e0, e0, e1, rot));
  }
  for (int e = e0; e < e1; ++e) {
    inn = et_idx(g, et_rot(e + 2 < e1 ? e + 2 : e1 - 1, e0, e1, rot));
    et_bwd_load<HAS_DK, HAS_DV, BF>(g, a, g_xagg, g_vagg, in, c, nxt);
    in = inn;
    const float cv = VCUT ? u.C : 1.0f, ca = VCUT ? 1.0f : u.C;
    // ---- role TARGET: message j -> r
    {
      const float at = head_sum_t<HD>(qr * u.kj * u.dk, hd);
      const float A = silu(at) * ca;
      const float sx = u.vxj * cv * u.dvx, s2 = u.v2j * cv * u.dv2;
      const float g_sx = gxr * A;
      const float g_A = head_sum_t<HD>(gxr * sx, hd);
      const float g_s1 = gr0 * u.vj0 + gr1 * u.vj1 + gr2_ * u.vj2;
      const float g_s2 = -(gr0 * u.p0 + gr1 * u.p1 + gr2_ * u.p2);
      const float g_a = g_A * silu_grad(at) * ca;
      gq += g_a * u.kj * u.dk;
      float gd = cv * (g_sx * u.vxj * u.tvx + g_s1 * u.v1j * u.tv1 + g_s2 * u.v2j * u.tv2) + g_a * qr * u.kj * u.tk;
      const float gcv = g_sx * u.vxj * u.dvx + g_s1 * u.v1j * u.dv1 + g_s2 * u.v2j * u.dv2;
      const float gca = head0 ? g_A * silu(at) : 0.f;
      gd += (VCUT ? gcv : gca) * u.dC;
      const float tot = wave_sum4(gd, gr0 * s2, gr1 * s2, gr2_ * s2, lane);
      if (u.sg != 0.f && (lane & 15) == 0) {
        const int64_t slot = (int64_t)wave * a.slot_stride + 2 * (int64_t)u.p + (u.sg > 0.f ? 0 : 1);
        const int comp = lane >> 4;
        if (comp == 0) gd2[slot] = tot;
        else gr2[slot * 3 + comp - 1] = tot;
        // the caller sums et_sweep_waves(F) slot arrays per pair direction (one per 32 channels: the tile sweeps' count); this
        // sweep fills one per 64 channels - the last wave of the row zeroes the others' entries of this (pair, direction)
        if (wave == (int)(blockDim.x >> 6) - 1) {
          const int nw_all = (F % 32 == 0) ? F / 32 : (int)(blockDim.x >> 6);
          for (int ws = (int)(blockDim.x >> 6); ws < nw_all; ++ws) {
            const int64_t sz = slot + (int64_t)(ws - wave) * a.slot_stride;
            if (comp == 0) gd2[sz] = 0.f;
            else gr2[sz * 3 + comp - 1] = 0.f;
          }
        }
      }
    }
    // ---- role SOURCE: message r -> j
    {
      const float as = head_sum_t<HD>(u.qj * kr * u.dk, hd);
      const float A = silu(as) * ca;
      const float sx = vxr * cv * u.dvx, s1 = v1r * cv * u.dv1;
      const float g_A = head_sum_t<HD>(u.gxj * sx, hd);
      const float g_a = g_A * silu_grad(as) * ca;
      gk += g_a * u.qj * u.dk;
      gvx += u.gxj * A * cv * u.dvx;
      gv1 += (u.gj0 * vr0 + u.gj1 * vr1 + u.gj2 * vr2) * cv * u.dv1;
      gv2 += (u.gj0 * u.p0 + u.gj1 * u.p1 + u.gj2 * u.p2) * cv * u.dv2;
      gvec0 += u.gj0 * s1;
      gvec1 += u.gj1 * s1;
      gvec2 += u.gj2 * s1;
    }
    u = nxt;
  }
  float* o = g_qkv + (int64_t)r * F5 + c;
  o[0] = gq;
  o[F] = gk;
  o[2 * F] = gvx;
  o[3 * F] = gv1;
  o[4 * F] = gv2;
  float* gv = g_vec + (int64_t)r * 3 * F + c;
  gv[0] += gvec0;
  gv[F] += gvec1;
  gv[2 * F] += gvec2;
}
void launch_et_attn_bwd(const Graph& g, int N, const EtAttnArgs& a_in, const float* g_xagg, const float* g_vagg, float* g_qkv,
                        float* g_vec, float* gd2, float* gr2, hipStream_t s) {
  if (N <= 0) return;
  EtAttnArgs a = a_in;
  const bool tiles = et_g16_ok(N, a.slot_stride / 2, a);
  if (!tiles) a.tile_open = nullptr;
  // (the row sweeps below write one slot array per 64 channels and zero the remaining et_sweep_waves(F) - F / 64 themselves)
  if (tiles) launch_et_attn_bwd_g16(g, N, a, g_xagg, g_vagg, g_qkv, g_vec, gd2, gr2, s);  // (see launch_et_attn_fwd)
  if (et_pipelined_ok(a)) {
    const dim3 grid(N), block(a.F);
#define ET_BWD2(DK, DV, VC, BF)                                                                                                  \
  if (a.hd == 16)                                                                                                               \
    hipLaunchKernelGGL((k_et_attn_bwd_p<DK, DV, VC, 16, BF>), grid, block, 0, s, g, a, g_xagg, g_vagg, g_qkv, g_vec, gd2, gr2); \
  else                                                                                                                          \
    hipLaunchKernelGGL((k_et_attn_bwd_p<DK, DV, VC, 0, BF>), grid, block, 0, s, g, a, g_xagg, g_vagg, g_qkv, g_vec, gd2, gr2)
#define ET_BWD(DK, DV, VC)      \
  if (a.pair_bf16) {            \
    ET_BWD2(DK, DV, VC, true);  \
  } else {                      \
    ET_BWD2(DK, DV, VC, false); \
  }
    const int key = (a.dk_off >= 0 ? 4 : 0) | (a.dv_off >= 0 ? 2 : 0) | (a.vector_cutoff ? 1 : 0);
    switch (key) {
      case 0: ET_BWD(false, false, false); break;
      case 1: ET_BWD(false, false, true); break;
      case 2: ET_BWD(false, true, false); break;
      case 3: ET_BWD(false, true, true); break;
      case 4: ET_BWD(true, false, false); break;
      case 5: ET_BWD(true, false, true); break;
      case 6: ET_BWD(true, true, false); break;
      default: ET_BWD(true, true, true); break;
    }
#undef ET_BWD
#undef ET_BWD2
    return;
  }
  hipLaunchKernelGGL(k_et_attn_bwd, dim3(N), dim3(bthreads(a.F)), 0, s, g, a, g_xagg, g_vagg, g_qkv, g_vec, gd2, gr2);
}

// gd[p] = gd2[2p] + gd2[2p+1] ; g_prhat[p] = -gr2[2p] + gr2[2p+1]   (rhat of the target sweep = -esign * prhat)
__global__ void k_et_pair_combine(Graph g, const float* __restrict__ gd2, const float* __restrict__ gr2, int nw, int64_t stride,
                                  const float* __restrict__ gd_extra, float* __restrict__ gd, float* __restrict__ g_rhat) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= g.counts[0] || g.counts[2]) return;
  float d = gd_extra[2 * (int64_t)p], r0 = 0.f, r1 = 0.f, r2 = 0.f;
  for (int w = 0; w < nw; ++w) {  // the per-(layer, wave) partial sums of the target sweeps, fixed order
    const int64_t s0 = w * stride + 2 * (int64_t)p, s1 = s0 + 1;
    d += gd2[s0] + gd2[s1];
    r0 += -gr2[s0 * 3] + gr2[s1 * 3];
    r1 += -gr2[s0 * 3 + 1] + gr2[s1 * 3 + 1];
    r2 += -gr2[s0 * 3 + 2] + gr2[s1 * 3 + 2];
  }
  gd[p] = d;
  g_rhat[(int64_t)p * 3] = r0;
  g_rhat[(int64_t)p * 3 + 1] = r1;
  g_rhat[(int64_t)p * 3 + 2] = r2;
}
void launch_et_pair_combine(const Graph& g, int Pcap, const float* gd2, const float* gr2, int nw, int64_t stride,
                            const float* gd_extra, float* gd, float* g_rhat, hipStream_t s) {
  if (Pcap <= 0) return;
  hipLaunchKernelGGL(k_et_pair_combine, dim3(cdiv_(Pcap, 256)), dim3(256), 0, s, g, gd2, gr2, nw, stride, gd_extra, gd, g_rhat);
}


// ---------------------------------------------------------------------------------------------- head (GatedEquivariantBlock)
// hcat[n] = [ xsrc[n] (Fx, optional) | norm_a u[3n+a] (Fn) ]      (models/utils.py:626-646)
__global__ void k_et_cat_norm(const float* __restrict__ xsrc, int Fx, const float* __restrict__ u, int ldu, int Fn, int N,
                              float* __restrict__ hcat) {
  const int W = Fx + Fn;
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)N * W) return;
  const int n = (int)(idx / W), c = (int)(idx - (int64_t)n * W);
  if (c < Fx) {
    if (xsrc) hcat[idx] = xsrc[(int64_t)n * Fx + c];
    return;
  }
  const float* up = u + (int64_t)n * 3 * ldu + (c - Fx);
  const float a0 = up[0], a1 = up[ldu], a2 = up[2 * (int64_t)ldu];
  hcat[idx] = sqrtf(a0 * a0 + a1 * a1 + a2 * a2);
}
void launch_et_cat_norm(const float* xsrc, int Fx, const float* u, int ldu, int Fn, int N, float* hcat, hipStream_t s) {
  if (N <= 0) return;
  hipLaunchKernelGGL(k_et_cat_norm, dim3(cdiv_((int64_t)N * (Fx + Fn), 256)), dim3(256), 0, s, xsrc, Fx, u, ldu, Fn, N, hcat);
}
// g_u[3n+a, c] = g_n[n, c] u[3n+a, c] / |u[., c]|   (0 where the norm is 0: the reference masks those rows)
__global__ void k_et_norm_bwd(const float* __restrict__ g_n, int ldg, const float* __restrict__ u, int ldu, int Fn, int N,
                              float* __restrict__ g_u, int ldgu) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)N * Fn) return;
  const int n = (int)(idx / Fn), c = (int)(idx - (int64_t)n * Fn);
  const float* up = u + (int64_t)n * 3 * ldu + c;
  const float a0 = up[0], a1 = up[ldu], a2 = up[2 * (int64_t)ldu];
  const float nn = sqrtf(a0 * a0 + a1 * a1 + a2 * a2);
  const float k = nn > 0.f ? g_n[(int64_t)n * ldg + c] / nn : 0.f;
  float* o = g_u + (int64_t)n * 3 * ldgu + c;
  o[0] = k * a0;
  o[ldgu] = k * a1;
  o[2 * (int64_t)ldgu] = k * a2;
}
void launch_et_norm_bwd(const float* g_n, int ldg, const float* u, int ldu, int Fn, int N, float* g_u, int ldgu, hipStream_t s) {
  if (N <= 0) return;
  hipLaunchKernelGGL(k_et_norm_bwd, dim3(cdiv_((int64_t)N * Fn, 256)), dim3(256), 0, s, g_n, ldg, u, ldu, Fn, N, g_u, ldgu);
}
// y [N][2 F2] -> hcat2[:, :F2] = silu(y[:, :F2]) ; vq[3n+a, c] = y[n, F2 + c] * u2[3n+a, c]
__global__ void k_et_head_mid(const float* __restrict__ y, const float* __restrict__ u2, int ldu, int F2, int N,
                              float* __restrict__ hcat2, float* __restrict__ vq) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)N * F2) return;
  const int n = (int)(idx / F2), c = (int)(idx - (int64_t)n * F2);
  hcat2[(int64_t)n * 2 * F2 + c] = silu(y[(int64_t)n * 2 * F2 + c]);
  const float gate = y[(int64_t)n * 2 * F2 + F2 + c];
#pragma unroll
  for (int ax = 0; ax < 3; ++ax) vq[((int64_t)n * 3 + ax) * F2 + c] = gate * u2[((int64_t)n * 3 + ax) * ldu + c];
}
void launch_et_head_mid(const float* y, const float* u2, int ldu, int F2, int N, float* hcat2, float* vq, hipStream_t s) {
  if (N <= 0) return;
  hipLaunchKernelGGL(k_et_head_mid, dim3(cdiv_((int64_t)N * F2, 256)), dim3(256), 0, s, y, u2, ldu, F2, N, hcat2, vq);
}
// g_y = [ g_xs silu'(y[:, :F2]) | sum_a g_vq u2 ] ; g_u2[3n+a, c] = g_vq[3n+a, c] * gate[n, c]
__global__ void k_et_head_mid_bwd(const float* __restrict__ y, const float* __restrict__ u2, int ldu, const float* __restrict__ g_h2,
                                  const float* __restrict__ g_vq, int F2, int N, float* __restrict__ g_y, float* __restrict__ g_u2,
                                  int ldgu) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)N * F2) return;
  const int n = (int)(idx / F2), c = (int)(idx - (int64_t)n * F2);
  const float gate = y[(int64_t)n * 2 * F2 + F2 + c];
  float gg = 0.f;
#pragma unroll
  for (int ax = 0; ax < 3; ++ax) {
    const float gv = g_vq[((int64_t)n * 3 + ax) * F2 + c];
    gg += gv * u2[((int64_t)n * 3 + ax) * ldu + c];
    g_u2[((int64_t)n * 3 + ax) * ldgu + c] = gv * gate;
  }
  g_y[(int64_t)n * 2 * F2 + c] = g_h2[(int64_t)n * 2 * F2 + c] * silu_grad(y[(int64_t)n * 2 * F2 + c]);
  g_y[(int64_t)n * 2 * F2 + F2 + c] = gg;
}
void launch_et_head_mid_bwd(const float* y, const float* u2, int ldu, const float* g_h2, const float* g_vq, int F2, int N, float* g_y,
                            float* g_u2, int ldgu, hipStream_t s) {
  if (N <= 0) return;
  hipLaunchKernelGGL(k_et_head_mid_bwd, dim3(cdiv_((int64_t)N * F2, 256)), dim3(256), 0, s, y, u2, ldu, g_h2, g_vq, F2, N, g_y, g_u2,
                     ldgu);
}

// dst[r, 0:cols] = src[r, 0:cols] (kernels, not memcpy nodes: HIP-graph replay, see profiles/r01_notes.md)
__global__ void k_et_copy2d(const float* __restrict__ src, int lds_, float* __restrict__ dst, int ldd, int rows, int cols) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)rows * cols) return;
  const int r = (int)(idx / cols), c = (int)(idx - (int64_t)r * cols);
  dst[(int64_t)r * ldd + c] = src[(int64_t)r * lds_ + c];
}
void launch_et_copy2d(const float* src, int lds_, float* dst, int ldd, int rows, int cols, hipStream_t s) {
  if (rows <= 0 || cols <= 0) return;
  hipLaunchKernelGGL(k_et_copy2d, dim3(cdiv_((int64_t)rows * cols, 256)), dim3(256), 0, s, src, lds_, dst, ldd, rows, cols);
}

// out[i] += in[i]
__global__ void k_et_axpy(const float* __restrict__ in, float* __restrict__ out, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] += in[i];
}
void launch_et_add(const float* in, float* out, int64_t n, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_et_axpy, dim3(cdiv_(n, 256)), dim3(256), 0, s, in, out, n);
}

// slot arrays per pair direction of the reverse sweep: one per 32-channel slice (tn_et_g16.hip) or per wave of 64 channels
int et_sweep_waves(int F) { return F % 32 == 0 ? F / 32 : bthreads(F) / 64; }


// ---------------------------------------------------------------------------------------------- parameter gradients
// Adjoint of the per-pair filter rows dkv[p] = silu(dk_proj phi) | silu(dv_proj phi) from the attention sweep: every directed
// edge t <- s contributes once, as the TARGET role of k_et_attn_bwd (same arithmetic):
//   g_dk = g_a q_t k_s ; g_dvx = cv g_sx vx_s ; g_dv1 = cv g_s1 v1_s ; g_dv2 = cv g_s2 v2_s
// The two directed edges of a pair share the row: slots[direction][p][Wd]; the self pair's row is shared by every atom's self
// edge: self_rows[t][Wd], summed over the atoms afterwards.  Block = target atom, thread = channel.
__global__ void k_et_train_filter(Graph g, EtAttnArgs a, const float* __restrict__ g_xagg, const float* __restrict__ g_vagg,
                                  float* __restrict__ slots, int64_t dir_stride, float* __restrict__ self_rows) {
  const int r = blockIdx.x;
  if (g.counts[2]) return;
  const int F = a.F, c = threadIdx.x;
  const bool live = c < F;
  const int cc = live ? c : 0;
  const int e0 = g.rowptr[r], e1 = g.rowptr[r + 1];
  const int64_t F5 = 5 * (int64_t)F;
  const float qr = a.qkv[(int64_t)r * F5 + cc];
  const float gxr = live ? g_xagg[(int64_t)r * F + cc] : 0.f;
  const float* gvr = g_vagg + (int64_t)r * 3 * F + cc;
  const float gr0 = live ? gvr[0] : 0.f, gr1 = live ? gvr[F] : 0.f, gr2_ = live ? gvr[2 * F] : 0.f;
  for (int e = e0; e < e1; ++e) {
    const int j = g.col[e], p = g.epair[e];
    const float sg = g.esign[e];
    const float* jq = a.qkv + (int64_t)j * F5 + cc;
    const float kj = jq[F], vxj = jq[2 * F], v1j = jq[3 * F], v2j = jq[4 * F];
    const int64_t dkv_b = (int64_t)p * a.Wd;
    const float C = a.C[p];
    const float cv = a.vector_cutoff ? C : 1.0f, ca = a.vector_cutoff ? 1.0f : C;
    const float dk = a.dk_off >= 0 ? a.dkv[dkv_b + a.dk_off + cc] : 1.f;
    const float dvx = a.dv_off >= 0 ? a.dkv[dkv_b + a.dv_off + cc] : 1.f;
    float p0 = 0.f, p1 = 0.f, p2 = 0.f;
    if (sg != 0.f) {
      p0 = sg * g.prhat[(int64_t)p * 3];
      p1 = sg * g.prhat[(int64_t)p * 3 + 1];
      p2 = sg * g.prhat[(int64_t)p * 3 + 2];
    }
    const float at = head_sum(live ? qr * kj * dk : 0.f, a.hd);
    const float A = silu(at) * ca;
    const float sx = vxj * cv * dvx;
    const float* vj = a.vec + (int64_t)j * 3 * F + cc;
    const float g_sx = gxr * A;
    const float g_A = head_sum(gxr * sx, a.hd);
    const float g_s1 = gr0 * vj[0] + gr1 * vj[F] + gr2_ * vj[2 * F];
    const float g_s2 = -(gr0 * p0 + gr1 * p1 + gr2_ * p2);
    const float g_a = g_A * silu_grad(at) * ca;
    if (!live) continue;
    float* row = sg != 0.f ? slots + (sg > 0.f ? 0 : dir_stride) + dkv_b : self_rows + (int64_t)r * a.Wd;
    if (a.dk_off >= 0) row[a.dk_off + c] = g_a * qr * kj;
    if (a.dv_off >= 0) {
      row[a.dv_off + c] = cv * g_sx * vxj;
      row[a.dv_off + F + c] = cv * g_s1 * v1j;
      row[a.dv_off + 2 * F + c] = cv * g_s2 * v2j;
    }
  }
}
void launch_et_train_filter(const Graph& g, int N, const EtAttnArgs& a, const float* g_xagg, const float* g_vagg, float* slots,
                            int64_t dir_stride, float* self_rows, hipStream_t s) {
  if (N <= 0) return;
  hipLaunchKernelGGL(k_et_train_filter, dim3(N), dim3(((a.F + 63) / 64) * 64), 0, s, g, a, g_xagg, g_vagg, slots, dir_stride, self_rows);
}

// g_pre[p] = (slot0[p] + slot1[p]) silu'(pre[p]) for the pairs, (sum of the atoms' self rows) silu'(pre[P]) for the self pair
__global__ void k_et_train_gpre(Graph g, int Pcap, int Wd, const float* __restrict__ slots, int64_t dir_stride,
                                const float* __restrict__ self_sum, const float* __restrict__ pre, float* __restrict__ g_pre) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int P = g.counts[0];
  if (idx >= (int64_t)(P + 1) * Wd || idx >= (int64_t)(Pcap + 1) * Wd) return;
  const int p = (int)(idx / Wd), c = (int)(idx - (int64_t)p * Wd);
  const float gsum = p == P ? self_sum[c] : slots[idx] + slots[dir_stride + idx];
  g_pre[idx] = gsum * silu_grad(pre[idx]);
}
void launch_et_train_gpre(const Graph& g, int P, int Wd, const float* slots, int64_t dir_stride, const float* self_sum, const float* pre,
                          float* g_pre, hipStream_t s) {
  const int64_t n = (int64_t)(P + 1) * Wd;
  if (n <= 0) return;
  hipLaunchKernelGGL(k_et_train_gpre, dim3((unsigned)cdiv_(n, 256)), dim3(256), 0, s, g, P, Wd, slots, dir_stride, self_sum, pre, g_pre);
}

// neighbour embedding nbr[i] = sum_{j != i} Wn[p] embN[z_j]: adjoint of the filter rows per direction (g_nbr[i] embN[z_j]) and of
// embN through the reverse edges (gEN[i] = sum_j g_nbr[j] Wn[p]: atom i is the source of the edge j <- i)
__global__ void k_et_train_nbr(Graph g, int N, int F, const int64_t* __restrict__ z, const float* __restrict__ embN,
                               const float* __restrict__ Wn, const float* __restrict__ g_xcat, float* __restrict__ slots,
                               int64_t dir_stride, float* __restrict__ gEN) {
  const int i = blockIdx.x;
  if (g.counts[2]) return;
  const int e0 = g.rowptr[i], e1 = g.rowptr[i + 1];
  for (int c = threadIdx.x; c < F; c += blockDim.x) {
    const float gi = g_xcat[(int64_t)i * 2 * F + F + c];
    float acc = 0.f;
    for (int e = e0; e < e1; ++e) {
      const int j = g.col[e], p = g.epair[e];
      if (j == i) continue;
      slots[(g.esign[e] > 0.f ? 0 : dir_stride) + (int64_t)p * F + c] = gi * embN[z[j] * F + c];
      acc += g_xcat[(int64_t)j * 2 * F + F + c] * Wn[(int64_t)p * F + c];
    }
    gEN[(int64_t)i * F + c] = acc;
  }
}
void launch_et_train_nbr(const Graph& g, int N, int F, const int64_t* z, const float* embN, const float* Wn, const float* g_xcat,
                         float* slots, int64_t dir_stride, float* gEN, hipStream_t s) {
  if (N <= 0) return;
  hipLaunchKernelGGL(k_et_train_nbr, dim3(N), dim3(bthreads(F)), 0, s, g, N, F, z, embN, Wn, g_xcat, slots, dir_stride, gEN);
}

}  // namespace tn
