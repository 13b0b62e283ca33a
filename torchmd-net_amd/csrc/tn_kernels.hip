// Non-GEMM HIP kernels of the TensorNet energy+force path for gfx950 (CDNA4, wave64).
// Thread mapping used by all per-atom kernels: the feature channel f is the fastest-varying thread
// index, so every global access is a contiguous 4*F-byte run (coalesced), and per-edge / per-atom
// scalars are wave-uniform (scalar loads).  Segmented sums over CSR rows are done in registers by
// the thread that owns (atom, channel): no atomics, deterministic results.
#include "tn_kernels.h"
#include "tn_interp.h"

#include <cstdlib>

#include "tn_common.h"

namespace tn {

static inline int fthreads(int F) {
  int t = ((F + 63) / 64) * 64;
  return t > 256 ? 256 : t;
}
static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// =====================================================================================
//                                   graph construction
// =====================================================================================
__global__ void k_mol_ranges(const int64_t* __restrict__ batch, int N, int B, int* mstart, int* mend, int* counts) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  int64_t b = batch[i];
  if (b < 0 || b >= B) {
    counts[3] = 1;
    counts[5] = 1;  // the host raises (the reference's scatter over `batch` raises as well); kernels skip via counts[2]
    return;
  }
  if (i > 0) {
    int64_t bp = batch[i - 1];
    if (bp > b) counts[3] = 1;
    if (bp != b) mstart[b] = i;
  } else {
    mstart[b] = 0;
  }
  if (i == N - 1 || batch[i + 1] != b) mend[b] = i + 1;
}

// canonical pair geometry: delta = pos[hi] - pos[lo] (+ triclinic minimum image, z -> y -> x;
// reference neighbors_brute.py:112-135).  Both atoms of a pair evaluate the identical expression.
__device__ __forceinline__ float pair_delta(const float* __restrict__ pos, int hi, int lo, const float* __restrict__ box,
                                            float& dx, float& dy, float& dz) {
  return pair_geometry(pos, hi, lo, box, dx, dy, dz);  // tn_common.h: the one definition
}

__device__ __forceinline__ void cand_range(const int* mstart, const int* mend, const int* counts, int64_t b, int N, int& j0,
                                           int& j1) {
  if (counts[3]) {
    j0 = 0;
    j1 = N;
  } else {
    j0 = mstart[b];
    j1 = mend[b];
  }
}

__global__ void k_nbr_count(Graph g, const float* __restrict__ pos, const int64_t* __restrict__ batch, const float* __restrict__ box,
                            int box_mode, int N, float lo2, float up2, int loop) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  int64_t b = batch[i];
  int j0, j1;
  cand_range(g.mstart, g.mend, g.counts, b, N, j0, j1);
  if (g.counts[5]) j1 = j0 = 0;
  const float* bx = box_mode == 0 ? nullptr : (box_mode == 1 ? box : box + b * 9);
  int nl = 0, nt = 0;
  for (int j = j0; j < j1; ++j) {
    if (j == i) {
      nt += loop ? 1 : 0;
      continue;
    }
    if (batch[j] != b) continue;
    float dx, dy, dz;
    float d2 = (j < i) ? pair_delta(pos, i, j, bx, dx, dy, dz) : pair_delta(pos, j, i, bx, dx, dy, dz);
    if (d2 < up2 && d2 >= lo2) {
      ++nt;
      if (j < i) ++nl;
    }
  }
  g.nlow[i] = nl;
  g.ntot[i] = nt;
}

// single-block exclusive scan of nlow -> pairptr and ntot -> rowptr; publishes P, E and the overflow flag.
// Tiles of 8192 elements go through LDS: coalesced loads (8 per thread and array), every thread scans 8 consecutive
// elements out of LDS, one block scan of the 1024 chunk sums, coalesced stores; a carry links the tiles.
__global__ __launch_bounds__(1024) void k_scan_counts(Graph g, int N) {
  constexpr int T = 8192, PER = 8;
  __shared__ int la[T], lb[T];
  __shared__ int wsum[2][16];
  __shared__ int carry[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry[0] = carry[1] = 0;
  for (int base = 0; base < N; base += T) {
#pragma unroll
    for (int r = 0; r < PER; ++r) {
      const int i = base + r * 1024 + tid;
      la[r * 1024 + tid] = i < N ? g.nlow[i] : 0;
      lb[r * 1024 + tid] = i < N ? g.ntot[i] : 0;
    }
    __syncthreads();
    int va[PER], vb[PER], a = 0, b = 0;
#pragma unroll
    for (int r = 0; r < PER; ++r) {
      va[r] = la[tid * PER + r];
      vb[r] = lb[tid * PER + r];
      a += va[r];
      b += vb[r];
    }
    int ia = a, ib = b;  // inclusive wave scans of the chunk sums
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      int ta = __shfl_up(ia, off, 64), tb = __shfl_up(ib, off, 64);
      if (lane >= off) {
        ia += ta;
        ib += tb;
      }
    }
    if (lane == 63) {
      wsum[0][wave] = ia;
      wsum[1][wave] = ib;
    }
    __syncthreads();
    int oa = carry[0] + ia - a, ob = carry[1] + ib - b;
    for (int w = 0; w < wave; ++w) {
      oa += wsum[0][w];
      ob += wsum[1][w];
    }
#pragma unroll
    for (int r = 0; r < PER; ++r) {
      la[tid * PER + r] = oa;
      lb[tid * PER + r] = ob;
      oa += va[r];
      ob += vb[r];
    }
    __syncthreads();
    if (tid == 1023) {
      carry[0] = oa;
      carry[1] = ob;
    }
#pragma unroll
    for (int r = 0; r < PER; ++r) {
      const int i = base + r * 1024 + tid;
      if (i < N) {
        g.pairptr[i] = la[r * 1024 + tid];
        g.rowptr[i] = lb[r * 1024 + tid];
      }
    }
    __syncthreads();
  }
  if (tid == 0) {
    const int P = carry[0], E = carry[1];
    g.pairptr[N] = P;
    g.rowptr[N] = E;
    g.counts[0] = P;
    g.counts[1] = E;
    g.counts[2] = (E > g.ecap || P > g.pcap || g.counts[5]) ? 1 : 0;
  }
}

__global__ void k_nbr_fill(Graph g, const float* __restrict__ pos, const int64_t* __restrict__ batch, const float* __restrict__ box,
                           int box_mode, int N, float lo2, float up2, int loop) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  if (g.counts[2]) return;  // overflow: the host raises (reference models/utils.py:297-300)
  const int P = g.counts[0];
  int64_t b = batch[i];
  int j0, j1;
  cand_range(g.mstart, g.mend, g.counts, b, N, j0, j1);
  const float* bx = box_mode == 0 ? nullptr : (box_mode == 1 ? box : box + b * 9);
  int e = g.rowptr[i];
  int p = g.pairptr[i];
  for (int j = j0; j < j1; ++j) {
    if (j == i) {
      if (loop) {
        g.col[e] = i;
        g.epair[e] = P;
        g.esign[e] = 0.f;
        ++e;
      }
      continue;
    }
    if (batch[j] != b) continue;
    float dx, dy, dz;
    float d2 = (j < i) ? pair_delta(pos, i, j, bx, dx, dy, dz) : pair_delta(pos, j, i, bx, dx, dy, dz);
    if (d2 < up2 && d2 >= lo2) {
      g.col[e] = j;
      if (j < i) {
        float d = sqrtf(d2);
        float inv = d > 0.f ? 1.0f / d : 0.f;
        g.pair_i[p] = i;
        g.pair_j[p] = j;
        g.pd[p] = d;
        g.pdelta[p * 3 + 0] = dx;
        g.pdelta[p * 3 + 1] = dy;
        g.pdelta[p * 3 + 2] = dz;
        g.prhat[p * 3 + 0] = dx * inv;
        g.prhat[p * 3 + 1] = dy * inv;
        g.prhat[p * 3 + 2] = dz * inv;
        g.epair[e] = p;
        g.esign[e] = 1.f;
        ++p;
      } else {
        g.epair[e] = -1;  // linked by k_nbr_link
        g.esign[e] = -1.f;
      }
      ++e;
    }
  }
  if (i == 0) g.pd[P] = 0.f;  // the self pair
}

// upper edges (i <- j, j > i) take the pair id of the lower edge (j <- i): binary search of i among the
// sorted lower neighbours of j
__global__ void k_nbr_link(Graph g, int N) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  if (g.counts[2]) return;
  int e0 = g.rowptr[i], e1 = g.rowptr[i + 1];
  for (int e = e0; e < e1; ++e) {
    int j = g.col[e];
    if (j <= i) continue;
    int lo = g.rowptr[j], hi = lo + g.nlow[j] - 1, base = lo;
    while (lo < hi) {
      int mid = (lo + hi) >> 1;
      if (g.col[mid] < i) lo = mid + 1; else hi = mid;
    }
    g.epair[e] = g.pairptr[j] + (lo - base);
  }
}

__global__ void k_graph_reset(Graph g, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) g.mstart[i] = g.mend[i] = 0;
  if (i < 8) g.counts[i] = 0;
}

void launch_graph_build_phase1(const Graph& g, const float* pos, const int64_t* batch, const float* box, int box_mode, int N,
                               int B, float lo, float up, bool loop, hipStream_t s) {
  // a kernel instead of hipMemsetAsync (memset nodes did not replay reliably from a captured HIP graph)
  hipLaunchKernelGGL(k_graph_reset, dim3(cdiv(B > 8 ? B : 8, 256)), dim3(256), 0, s, g, B);
  if (N <= 0) return;
  hipLaunchKernelGGL(k_mol_ranges, dim3(cdiv(N, 256)), dim3(256), 0, s, batch, N, B, g.mstart, g.mend, g.counts);
  if (getenv("TMDNET_SCALAR_GRAPH"))  // developer switch: thread-per-atom specification kernels
    hipLaunchKernelGGL(k_nbr_count, dim3(cdiv(N, 64)), dim3(64), 0, s, g, pos, batch, box, box_mode, N, lo * lo, up * up, (int)loop);
  else
    launch_nbr_count_wave(g, pos, batch, box, box_mode, N, B, lo, up, loop, s);
  hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, s, g, N);
}

void launch_scan_counts(const Graph& g, int N, hipStream_t s) {
  hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, s, g, N);
}

void launch_graph_build_phase2(const Graph& g, const float* pos, const int64_t* batch, const float* box, int box_mode, int N,
                               int B, float lo, float up, bool loop, hipStream_t s) {
  if (N <= 0) return;
  if (getenv("TMDNET_SCALAR_GRAPH")) {
    hipLaunchKernelGGL(k_nbr_fill, dim3(cdiv(N, 64)), dim3(64), 0, s, g, pos, batch, box, box_mode, N, lo * lo, up * up, (int)loop);
    hipLaunchKernelGGL(k_nbr_link, dim3(cdiv(N, 64)), dim3(64), 0, s, g, N);
  } else {
    launch_nbr_fill_link_wave(g, pos, batch, box, box_mode, N, B, lo, up, loop, s);
  }
}

// COO list in the reference operator's format: lower pairs (i>j) [+ transposes] [+ self loops], padded with -1/0.
// perm != null (cell strategy): pair ends are internal (cell-order) indices; they are mapped back to the caller's and
// oriented so that the first block again has i > j.
__global__ void k_export_pairs(Graph g, int N, int include_transpose, int loop, int64_t max_pairs, const int* __restrict__ perm,
                               int64_t* neighbors, float* deltas, float* distances, int* num_pairs) {
  const int P = g.counts[0];
  const int64_t total = (int64_t)P * (include_transpose ? 2 : 1) + (loop ? N : 0);
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx == 0) num_pairs[0] = (int)total;
  if (idx >= max_pairs) return;
  int64_t a = -1, b = -1;
  float dx = 0.f, dy = 0.f, dz = 0.f, d = 0.f;
  if (idx < total && !g.counts[2]) {
    int64_t k = idx;
    const bool tr = include_transpose && k >= P && k < 2 * (int64_t)P;
    if (k < P || tr) {
      if (tr) k -= P;
      a = g.pair_i[k]; b = g.pair_j[k];
      dx = g.pdelta[k * 3]; dy = g.pdelta[k * 3 + 1]; dz = g.pdelta[k * 3 + 2]; d = g.pd[k];
      if (perm) {
        a = perm[a]; b = perm[b];
      }
      if ((a < b) != tr) {  // orient: i > j in the first block, j < i ... transposed in the second
        const int64_t t = a; a = b; b = t;
        dx = -dx; dy = -dy; dz = -dz;
      }
    } else {
      a = b = k - (int64_t)P * (include_transpose ? 2 : 1);
    }
  }
  neighbors[idx] = a;
  neighbors[max_pairs + idx] = b;
  deltas[idx * 3] = dx; deltas[idx * 3 + 1] = dy; deltas[idx * 3 + 2] = dz;
  distances[idx] = d;
}

void launch_export_pairs(const Graph& g, int N, bool include_transpose, bool loop, int64_t max_pairs, const int* perm,
                         int64_t* neighbors, float* deltas, float* distances, int* num_pairs, hipStream_t s) {
  int64_t n = max_pairs > 0 ? max_pairs : 1;
  hipLaunchKernelGGL(k_export_pairs, dim3(cdiv(n, 256)), dim3(256), 0, s, g, N, (int)include_transpose, (int)loop, max_pairs, perm,
                     neighbors, deltas, distances, num_pairs);
}

// Gradient of the neighbour operator's outputs wrt the positions (reference extensions/neighbor_utils.py:11-46):
//   g = g_delta[p] + delta[p] / d[p] * g_dist[p]   (zero for padded entries and for d = 0) ; out[i] += g ; out[j] -= g
// The COO list carries no row structure, so the sum uses float atomics (this operator is off the model's hot path, whose
// own force reduction is the atomic-free CSR gather k_force_gather).
__global__ void k_neighbor_grad(const int64_t* __restrict__ nb, const float* __restrict__ deltas, const float* __restrict__ dist,
                                const float* __restrict__ g_delta, const float* __restrict__ g_dist, int64_t M, int N,
                                float* __restrict__ out) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= M) return;
  const int64_t i = nb[p], j = nb[M + p];
  const float d = dist[p];
  if (i < 0 || j < 0 || i >= N || j >= N || d == 0.f) return;
  const float gd = g_dist ? g_dist[p] / d : 0.f;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float g = (g_delta ? g_delta[p * 3 + a] : 0.f) + deltas[p * 3 + a] * gd;
    atomicAdd(out + i * 3 + a, g);
    atomicAdd(out + j * 3 + a, -g);
  }
}
void launch_neighbor_grad(const int64_t* nb, const float* deltas, const float* dist, const float* g_delta, const float* g_dist,
                          int64_t M, int N, float* out, hipStream_t s) {
  launch_fill(out, 0.f, (int64_t)N * 3, s);
  if (M <= 0) return;
  hipLaunchKernelGGL(k_neighbor_grad, dim3(cdiv(M, 256)), dim3(256), 0, s, nb, deltas, dist, g_delta, g_dist, M, N, out);
}

// =====================================================================================
//                         radial basis + cutoff (one thread per (pair, k))
// =====================================================================================
// NOTE (all per-pair kernels): the launch grid is sized by the host-side pair CAPACITY; the true pair count is
// read from device memory (g.counts[0]) so that the schedule can be replayed from a HIP graph without a host sync.
__global__ void k_radial(Graph g, int Pcap, RadialParams rp, float* __restrict__ phi, float* __restrict__ dphi, float* __restrict__ C,
                         float* __restrict__ dC) {
  const int P = g.counts[0];
  if (g.counts[2]) return;
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t total = (int64_t)(P + 1) * rp.K;
  if (idx >= total) return;
  int p = (int)(idx / rp.K), k = (int)(idx - (int64_t)p * rp.K);
  float d = p < P ? g.pd[p] : 0.f;
  float c0, dc0;
  cosine_cutoff(d, 0.f, rp.up, c0, dc0);  // ExpNormalSmearing uses CosineCutoff(0, upper): utils.py:371
  float alpha = 5.0f / (rp.up - rp.lo);
  float u = expf(-alpha * (d - rp.lo));
  float mu = rp.means[k], beta = rp.betas[k];
  float gk = expf(-beta * (u - mu) * (u - mu));
  phi[idx] = c0 * gk;
  dphi[idx] = dc0 * gk + c0 * gk * (-2.0f * beta * (u - mu)) * (-alpha * u);
  if (k == 0) {
    float c, dc;
    cosine_cutoff(d, rp.lo, rp.up, c, dc);
    C[p] = c;
    dC[p] = dc;
  }
}
void launch_radial(const Graph& g, int P, RadialParams rp, float* phi, float* dphi, float* C, float* dC, hipStream_t s) {
  int64_t total = (int64_t)(P + 1) * rp.K;
  hipLaunchKernelGGL(k_radial, dim3(cdiv(total, 256)), dim3(256), 0, s, g, P, rp, phi, dphi, C, dC);
}

// =====================================================================================
//                                     embedding scatter
// =====================================================================================
// U[z,f] = b2[f] + sum_k emb[z,k] Wa[f,k] ; V[z,f] = sum_k emb[z,k] Wb[f,k]   (Wa/Wb stored transposed)
__global__ void k_ztables(const float* __restrict__ emb, const float* __restrict__ WaT, const float* __restrict__ WbT,
                          const float* __restrict__ b2, int Z, int F, float* __restrict__ Utab, float* __restrict__ Vtab) {
  const int z = blockIdx.x;
  for (int f = threadIdx.x; f < F; f += blockDim.x) {
    float u = b2[f], v = 0.f;
    for (int k = 0; k < F; ++k) {
      const float e = emb[(int64_t)z * F + k];
      u += e * WaT[(int64_t)k * F + f];
      v += e * WbT[(int64_t)k * F + f];
    }
    Utab[(int64_t)z * F + f] = u;
    Vtab[(int64_t)z * F + f] = v;
  }
}
void launch_ztables(const float* emb, const float* WaT, const float* WbT, const float* b2, int Z, int F, float* Utab, float* Vtab,
                    hipStream_t s) {
  hipLaunchKernelGGL(k_ztables, dim3(Z), dim3(fthreads(F)), 0, s, emb, WaT, WbT, b2, Z, F, Utab, Vtab);
}

static int embed_split_rows() {
  static const int v = [] {
    const char* e = getenv("TMDNET_SPLIT_ROWS");
    return e ? atoi(e) : 1024;
  }();
  return v;
}
// block = one atom, thread = channel.  I0 = sum W0 ; v = sum W1 r ; T = sum W2 r r^T ; u0 = (I0, v, T - tr(T)/3)
__global__ void k_embed_scatter(Graph g, int N, int F, const int64_t* __restrict__ z, const float* __restrict__ Utab,
                                const float* __restrict__ Vtab, const float* __restrict__ Q, const float* __restrict__ C,
                                float* __restrict__ u0, float* __restrict__ s0n) {
  const int i = xcd_chunk(blockIdx.x, gridDim.x);
  if (g.counts[2]) return;  // pair overflow: the adjacency was not filled (the host reports the error)
  const int e0 = g.rowptr[i], e1 = g.rowptr[i + 1];
  const int64_t zi = z[i];
  const int F3 = 3 * F;
  for (int f = threadIdx.x; f < F; f += blockDim.x) {
    const float Ui = Utab[zi * F + f];
    float I0 = 0.f, v0 = 0.f, v1 = 0.f, v2 = 0.f, t00 = 0.f, t01 = 0.f, t02 = 0.f, t11 = 0.f, t12 = 0.f, t22 = 0.f;
    struct In {
      float rx, ry, rz, c, v, q0, q1, q2;
    };
    auto load = [&](int e, In& o) {
      const int j = g.col[e], p = g.epair[e];
      const float sg = g.esign[e];
      const float h0 = g.prhat[p * 3], h1 = g.prhat[p * 3 + 1], h2 = g.prhat[p * 3 + 2];
      o.rx = sg != 0.f ? sg * h0 : 0.f;
      o.ry = sg != 0.f ? sg * h1 : 0.f;
      o.rz = sg != 0.f ? sg * h2 : 0.f;
      o.c = C[p];
      o.v = Vtab[z[j] * F + f];
      const float* q = Q + (int64_t)p * F3 + f;
      o.q0 = q[0];
      o.q1 = q[F];
      o.q2 = q[2 * F];
    };
    auto add = [&](const In& c) {
      const float rx = c.rx, ry = c.ry, rz = c.rz;
      const float cz = c.c * (Ui + c.v);
      const float W0 = cz * c.q0, W1 = cz * c.q1, W2 = cz * c.q2;
      I0 += W0;
      v0 += W1 * rx; v1 += W1 * ry; v2 += W1 * rz;
      t00 += W2 * rx * rx; t01 += W2 * rx * ry; t02 += W2 * rx * rz;
      t11 += W2 * ry * ry; t12 += W2 * ry * rz; t22 += W2 * rz * rz;
    };
    int e = e0;
    for (; e + 4 <= e1; e += 4) {  // four edges' operands in flight together, accumulated in edge order
      In a0, a1, a2, a3;
      load(e, a0);
      load(e + 1, a1);
      load(e + 2, a2);
      load(e + 3, a3);
      add(a0);
      add(a1);
      add(a2);
      add(a3);
    }
    for (; e < e1; ++e) {
      In a0;
      load(e, a0);
      add(a0);
    }
    const float tr3 = (t00 + t11 + t22) * (1.0f / 3.0f);
    float u[9] = {I0, v0, v1, v2, t00 - tr3, t01, t02, t11 - tr3, t12};
    float* o = u0 + (int64_t)i * 9 * F + f;
#pragma unroll
    for (int c = 0; c < 9; ++c) o[c * F] = u[c];
    s0n[(int64_t)i * F + f] = quad(u);
  }
}
// small-system variant: kES thread groups take every kES-th edge of the row (see k_message_split)
constexpr int kES = 8;
__global__ __launch_bounds__(1024) void k_embed_scatter_split(Graph g, int N, int F, const int64_t* __restrict__ z,
                                                             const float* __restrict__ Utab, const float* __restrict__ Vtab,
                                                             const float* __restrict__ Q, const float* __restrict__ C,
                                                             float* __restrict__ u0, float* __restrict__ s0n) {
  __shared__ float part[kES - 1][10][128];
  const int i = blockIdx.x;
  if (g.counts[2]) return;
  const int grp = threadIdx.x / F, f = threadIdx.x - grp * F;
  const int e0 = g.rowptr[i], e1 = g.rowptr[i + 1];
  const int64_t zi = z[i];
  const int F3 = 3 * F;
  const float Ui = Utab[zi * F + f];
  float a[10] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // I0, v0..2, t00, t01, t02, t11, t12, t22
  for (int e = e0 + grp; e < e1; e += kES) {
    const int j = g.col[e], p = g.epair[e];
    const float sg = g.esign[e];
    float rx = 0.f, ry = 0.f, rz = 0.f;
    if (sg != 0.f) {
      rx = sg * g.prhat[p * 3];
      ry = sg * g.prhat[p * 3 + 1];
      rz = sg * g.prhat[p * 3 + 2];
    }
    const float cz = C[p] * (Ui + Vtab[z[j] * F + f]);
    const float* qq = Q + (int64_t)p * F3 + f;
    const float W0 = cz * qq[0], W1 = cz * qq[F], W2 = cz * qq[2 * F];
    a[0] += W0;
    a[1] += W1 * rx; a[2] += W1 * ry; a[3] += W1 * rz;
    a[4] += W2 * rx * rx; a[5] += W2 * rx * ry; a[6] += W2 * rx * rz;
    a[7] += W2 * ry * ry; a[8] += W2 * ry * rz; a[9] += W2 * rz * rz;
  }
  if (grp > 0) {
#pragma unroll
    for (int c = 0; c < 10; ++c) part[grp - 1][c][f] = a[c];
  }
  __syncthreads();
  if (grp > 0) return;
#pragma unroll
  for (int k = 0; k < kES - 1; ++k)
#pragma unroll
    for (int c = 0; c < 10; ++c) a[c] += part[k][c][f];
  const float tr3 = (a[4] + a[7] + a[9]) * (1.0f / 3.0f);
  float u[9] = {a[0], a[1], a[2], a[3], a[4] - tr3, a[5], a[6], a[7] - tr3, a[8]};
  float* o = u0 + (int64_t)i * 9 * F + f;
#pragma unroll
  for (int c = 0; c < 9; ++c) o[c * F] = u[c];
  s0n[(int64_t)i * F + f] = quad(u);
}
void launch_embed_scatter(const Graph& g, int N, int F, const int64_t* z, const float* Utab, const float* Vtab, const float* Q,
                          const float* C, float* u0, float* s0n, hipStream_t s) {
  if (N <= 0) return;
  // measured on MI355X (profiles/r01_notes.md): the 16-byte CSR sweep is slower than one-channel-per-lane here (scalar edge loads, 4x the waves)
  if (N <= embed_split_rows() && F <= 128 && F % 64 == 0) {
    hipLaunchKernelGGL(k_embed_scatter_split, dim3(N), dim3(kES * F), 0, s, g, N, F, z, Utab, Vtab, Q, C, u0, s0n);
    return;
  }
  hipLaunchKernelGGL(k_embed_scatter, dim3(N), dim3(fthreads(F)), 0, s, g, N, F, z, Utab, Vtab, Q, C, u0, s0n);
}

// =====================================================================================
//                                        LayerNorm
// =====================================================================================
__global__ __launch_bounds__(256) void k_layernorm_fwd(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ b, int rows, int R, float* __restrict__ y,
                                                       float* __restrict__ xhat, float* __restrict__ rstd) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (int64_t)row * R;
  float sum = 0.f;
  for (int k = lane; k < R; k += 64) sum += xr[k];
  const float mean = wave_sum(sum) / R;
  float var = 0.f;
  for (int k = lane; k < R; k += 64) {
    float d = xr[k] - mean;
    var += d * d;
  }
  var = wave_sum(var) / R;
  const float rs = 1.0f / sqrtf(var + 1e-5f);
  for (int k = lane; k < R; k += 64) {
    float xh = (xr[k] - mean) * rs;
    xhat[(int64_t)row * R + k] = xh;
    y[(int64_t)row * R + k] = xh * w[k] + b[k];
  }
  if (lane == 0) rstd[row] = rs;
}
__global__ __launch_bounds__(256) void k_layernorm_bwd(const float* __restrict__ g, const float* __restrict__ xhat,
                                                       const float* __restrict__ rstd, const float* __restrict__ w, int rows, int R,
                                                       float* __restrict__ gx) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* gr = g + (int64_t)row * R;
  const float* xh = xhat + (int64_t)row * R;
  float s1 = 0.f, s2 = 0.f;
  for (int k = lane; k < R; k += 64) {
    float gw = gr[k] * w[k];
    s1 += gw;
    s2 += gw * xh[k];
  }
  s1 = wave_sum(s1) / R;
  s2 = wave_sum(s2) / R;
  const float rs = rstd[row];
  for (int k = lane; k < R; k += 64) gx[(int64_t)row * R + k] = (gr[k] * w[k] - s1 - xh[k] * s2) * rs;
}
void launch_layernorm_fwd(const float* x, const float* w, const float* b, int rows, int R, float* y, float* xhat, float* rstd,
                          hipStream_t s) {
  if (rows <= 0) return;
  hipLaunchKernelGGL(k_layernorm_fwd, dim3(cdiv(rows, 4)), dim3(256), 0, s, x, w, b, rows, R, y, xhat, rstd);
}
void launch_layernorm_bwd(const float* g, const float* xhat, const float* rstd, const float* w, int rows, int R, float* gx,
                          hipStream_t s) {
  if (rows <= 0) return;
  hipLaunchKernelGGL(k_layernorm_bwd, dim3(cdiv(rows, 4)), dim3(256), 0, s, g, xhat, rstd, w, rows, R, gx);
}


// =====================================================================================
//                               interaction layer, node side
// =====================================================================================
__device__ __forceinline__ void load9(const float* __restrict__ p, int F, float u[9]) {
#pragma unroll
  for (int c = 0; c < 9; ++c) u[c] = p[c * F];
}
__device__ __forceinline__ void store9(float* __restrict__ p, int F, const float u[9]) {
#pragma unroll
  for (int c = 0; c < 9; ++c) p[c * F] = u[c];
}
__device__ __forceinline__ float kappa_of(const float* __restrict__ q, const int64_t* __restrict__ batch, int n) {
  // batch == null: q already is the per-atom factor (k_kappa, energy_forces passes it that way)
  return q ? (batch ? 1.0f + 0.1f * q[batch[n]] : q[n]) : 1.0f;
}

// X_hat = X / (||X||^2 + 1)   (reference tensornet.py:745)
__global__ void k_norm_x(const float* __restrict__ X, float* __restrict__ Xh, int N, int F) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)N * F) return;
  int n = (int)(idx / F), f = (int)(idx - (int64_t)n * F);
  float u[9];
  load9(X + (int64_t)n * 9 * F + f, F, u);
  const float inv = 1.0f / (quad(u) + 1.0f);
#pragma unroll
  for (int c = 0; c < 9; ++c) u[c] *= inv;
  store9(Xh + (int64_t)n * 9 * F + f, F, u);
}
void launch_norm_x(const float* X, float* Xh, int N, int F, hipStream_t s) {
  if (N <= 0) return;
  hipLaunchKernelGGL(k_norm_x, dim3(cdiv((int64_t)N * F, 256)), dim3(256), 0, s, X, Xh, N, F);
}

// CSR segmented gather-sum: acc[c] = sum_{e in row(i)} w[pair(e), type(c), f] * src[col(e), c, f]
// RC: the per-pair rows are not stored; each is interpolated from the layer's radial table here (tn_interp.h, option
// "recompute_pair_rows": 12 table loads per edge instead of 3 row loads, 6 KB of workspace per pair saved)
template <bool RC = false>
__device__ __forceinline__ void csr_gather(const Graph& g, int i, int F, int f, const float* __restrict__ w,
                                           const float* __restrict__ src, float acc[9], const PairRowTable* rt = nullptr) {
  const int e0 = g.rowptr[i], e1 = g.rowptr[i + 1];
  const int F3 = 3 * F, F9 = 9 * F;
#pragma unroll
  for (int c = 0; c < 9; ++c) acc[c] = 0.f;
  struct In {
    float w0, w1, w2, s[9];
  };
  auto load = [&](int e, In& o) {
    const int j = g.col[e], p = g.epair[e];
    const float* sp = src + (int64_t)j * F9 + f;
    if (RC) {
      float w3[3], unused[3];
      pair_row_eval<false>(*rt, p, F3, F, f, w3, unused);
      o.w0 = w3[0];
      o.w1 = w3[1];
      o.w2 = w3[2];
    } else {
      const float* wp = w + (int64_t)p * F3 + f;
      o.w0 = wp[0];
      o.w1 = wp[F];
      o.w2 = wp[2 * F];
    }
#pragma unroll
    for (int c = 0; c < 9; ++c) o.s[c] = sp[c * F];
  };
  auto add = [&](const In& o) {  // explicit fused multiply-adds: both instantiations (stored / interpolated rows) round alike
    acc[0] = __fmaf_rn(o.w0, o.s[0], acc[0]);
    acc[1] = __fmaf_rn(o.w1, o.s[1], acc[1]);
    acc[2] = __fmaf_rn(o.w1, o.s[2], acc[2]);
    acc[3] = __fmaf_rn(o.w1, o.s[3], acc[3]);
    acc[4] = __fmaf_rn(o.w2, o.s[4], acc[4]);
    acc[5] = __fmaf_rn(o.w2, o.s[5], acc[5]);
    acc[6] = __fmaf_rn(o.w2, o.s[6], acc[6]);
    acc[7] = __fmaf_rn(o.w2, o.s[7], acc[7]);
    acc[8] = __fmaf_rn(o.w2, o.s[8], acc[8]);
  };
  int e = e0;
  for (; e + 4 <= e1; e += 4) {  // four edges' rows requested together, accumulated in list order
    In a0, a1, a2, a3;
    load(e, a0);
    load(e + 1, a1);
    load(e + 2, a2);
    load(e + 3, a3);
    add(a0);
    add(a1);
    add(a2);
    add(a3);
  }
  for (; e < e1; ++e) {
    In a0;
    load(e, a0);
    add(a0);
  }
}

// message passing + group product + normalisation (reference tensornet.py:757-806)
template <bool RC>
__global__ void k_message(Graph g, int N, int F, const float* __restrict__ w, const float* __restrict__ Pn,
                          const float* __restrict__ q, const int64_t* __restrict__ batch, int o3, float* __restrict__ Mi,
                          float* __restrict__ Ch, PairRowTable rt, int row0) {
  const int i = row0 + xcd_chunk(blockIdx.x, gridDim.x);  // row0 > 0: the owned rows of a halo exchange (gridDim.x of them)
  if (g.counts[2]) return;  // pair overflow: the adjacency was not filled (the host reports the error)
  const float kap = kappa_of(q, batch, i);
  for (int f = threadIdx.x; f < F; f += blockDim.x) {
    float m[9], y[9];
    csr_gather<RC>(g, i, F, f, w, Pn, m, &rt);
    load9(Pn + (int64_t)i * 9 * F + f, F, y);
    store9(Mi + (int64_t)i * 9 * F + f, F, m);
    const M3 Y = compose(y), M = compose(m);
    M3 Cm = o3 ? scale(add(matmul(Y, M), matmul(M, Y)), kap) : scale(matmul(Y, M), 2.0f);
    float uc[9];
    decompose(Cm, uc);
    const float inv = 1.0f / (frob2(Cm) + 1.0f);
#pragma unroll
    for (int c = 0; c < 9; ++c) uc[c] *= inv;
    store9(Ch + (int64_t)i * 9 * F + f, F, uc);
  }
}
// Small systems (single-molecule MD): a row's ~25 edges are a serial chain of dependent loads, and the chip is mostly idle, so
// EG = 4 thread groups of a block take every 4th edge of the row and their partial sums are combined through LDS in a
// fixed order.  Used below `kSplitRows` rows; the summation order differs from the one-group kernels by rounding only.
constexpr int kEG = 8;
static int split_rows_max() {
  static const int v = [] {
    const char* e = getenv("TMDNET_SPLIT_ROWS");  // developer switch
    return e ? atoi(e) : 1024;
  }();
  return v;
}
// measured (profiles/r02_notes.md): 64 atoms 13 -> 7 us per sweep with 4 groups, another -10 us per step with 8; 1024 atoms
// 0.67 -> 0.64 ms per step; 2048 atoms slower with the split
#define kSplitRows split_rows_max()
// MODE 0: forward message + group product + normalisation ; 1: adjoint (out += gather) ; 2: adjoint + the per-pair distance
// gradient halves h(i <- j) = sum dw[p] * gMi[j] * Pn[i] (see k_message_adjoint_gd; Mi then carries Pn, q carries dw)
template <int MODE>
__global__ __launch_bounds__(1024) void k_message_split(Graph g, int N, int F, const float* __restrict__ w,
                                                       const float* __restrict__ src, const float* __restrict__ q,
                                                       const int64_t* __restrict__ batch, int o3, float* __restrict__ Mi,
                                                       float* __restrict__ out, float* __restrict__ slots = nullptr,
                                                       int64_t slot_stride = 0) {
  __shared__ float part[kEG - 1][9][128];
  const int i = blockIdx.x;
  if (g.counts[2]) return;
  const int grp = threadIdx.x / F, f = threadIdx.x - grp * F;
  const int e0 = g.rowptr[i], e1 = g.rowptr[i + 1];
  const int F3 = 3 * F, F9 = 9 * F;
  float acc[9], y[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) acc[c] = y[c] = 0.f;
  if (MODE == 2) load9(Mi + (int64_t)i * F9 + f, F, y);  // Pn[i]
  // an edge of this thread group: accumulate, and (MODE 2) the channel product whose wave sum is the g_d half
  auto edge = [&](int e, float& h, int& si) {
    const int j = g.col[e], p = g.epair[e];
    const float* wp = w + (int64_t)p * F3 + f;
    const float* sp = src + (int64_t)j * F9 + f;
    const float w0 = wp[0], w1 = wp[F], w2 = wp[2 * F];
    float s9[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) s9[c] = sp[c * F];
    acc[0] += w0 * s9[0];
    acc[1] += w1 * s9[1];
    acc[2] += w1 * s9[2];
    acc[3] += w1 * s9[3];
    acc[4] += w2 * s9[4];
    acc[5] += w2 * s9[5];
    acc[6] += w2 * s9[6];
    acc[7] += w2 * s9[7];
    acc[8] += w2 * s9[8];
    if (MODE == 2) {
      const float* dp = q + (int64_t)p * F3 + f;  // dw
      h = dp[0] * (s9[0] * y[0]) + dp[F] * (s9[1] * y[1] + s9[2] * y[2] + s9[3] * y[3]) +
          dp[2 * F] * (s9[4] * y[4] + s9[5] * y[5] + s9[6] * y[6] + s9[7] * y[7] + s9[8] * y[8]);
      const float sg = g.esign[e];
      si = sg != 0.f ? 2 * p + (sg > 0.f ? 0 : 1) : -1;
    }
  };
  // two edges of the group per trip: their loads are independent and issue together (a group walks 3-4 edges of a row)
  int e = e0 + grp;
  for (; e + kEG < e1; e += 2 * kEG) {
    float h0 = 0.f, h1 = 0.f;
    int s0 = -1, s1 = -1;
    edge(e, h0, s0);
    edge(e + kEG, h1, s1);
    if (MODE == 2) {
      h0 = wave_sum(h0);
      h1 = wave_sum(h1);
      if ((f & 63) == 0) {
        if (s0 >= 0) slots[(int64_t)(f >> 6) * slot_stride + s0] = h0;
        if (s1 >= 0) slots[(int64_t)(f >> 6) * slot_stride + s1] = h1;
      }
    }
  }
  if (e < e1) {
    float h0 = 0.f;
    int s0 = -1;
    edge(e, h0, s0);
    if (MODE == 2) {
      h0 = wave_sum(h0);
      if ((f & 63) == 0 && s0 >= 0) slots[(int64_t)(f >> 6) * slot_stride + s0] = h0;
    }
  }
  if (grp > 0) {
#pragma unroll
    for (int c = 0; c < 9; ++c) part[grp - 1][c][f] = acc[c];
  }
  __syncthreads();
  if (grp > 0) return;
#pragma unroll
  for (int k = 0; k < kEG - 1; ++k)
#pragma unroll
    for (int c = 0; c < 9; ++c) acc[c] += part[k][c][f];
  float* o = out + (int64_t)i * F9 + f;
  if (MODE >= 1) {
#pragma unroll
    for (int c = 0; c < 9; ++c) o[c * F] += acc[c];
  } else {
    load9(src + (int64_t)i * F9 + f, F, y);
    store9(Mi + (int64_t)i * F9 + f, F, acc);
    const float kap = kappa_of(q, batch, i);
    const M3 Y = compose(y), M = compose(acc);
    M3 Cm = o3 ? scale(add(matmul(Y, M), matmul(M, Y)), kap) : scale(matmul(Y, M), 2.0f);
    float uc[9];
    decompose(Cm, uc);
    const float inv = 1.0f / (frob2(Cm) + 1.0f);
#pragma unroll
    for (int c = 0; c < 9; ++c) uc[c] *= inv;
    store9(o, F, uc);
  }
}
static bool split_rows_ok(int N, int F) { return N <= kSplitRows && F <= 128 && F % 64 == 0; }

void launch_message(const Graph& g, int N, int F, const float* w, const float* src, const float* q, const int64_t* batch, int o3,
                    float* Mi, float* Ch, hipStream_t s, const PairRowTable* rt, int row0, int nrows) {
  if (N <= 0) return;
  if (nrows < 0) {
    row0 = 0;
    nrows = N;
  }
  if (rt) {  // rows evaluated from the table inside the sweep
    hipLaunchKernelGGL((k_message<true>), dim3(nrows), dim3(fthreads(F)), 0, s, g, N, F, w, src, q, batch, o3, Mi, Ch, *rt, row0);
    return;
  }
  if (nrows != N && !split_rows_ok(N, F) && !(g.small_mols && message_pair_ok(N, F))) {  // a row range: the row kernel (one large
                                                                                            // system; the other kernels take all rows)
    hipLaunchKernelGGL((k_message<false>), dim3(nrows), dim3(fthreads(F)), 0, s, g, N, F, w, src, q, batch, o3, Mi, Ch, PairRowTable{}, row0);
    return;
  }
  if (split_rows_ok(N, F)) {
    hipLaunchKernelGGL((k_message_split<0>), dim3(N), dim3(kEG * F), 0, s, g, N, F, w, src, q, batch, o3, Mi, Ch);
    return;
  }
  // the tile kernels pay off when a tile's column window fits LDS (batches of small molecules); one large system in cell
  // order has windows of hundreds of rows: there the row kernel with four edges in flight is faster (10 k-atom box:
  // 0.36 -> 0.26 ms per sweep)
  if (g.small_mols && message_pair_ok(N, F)) return launch_message_pair(g, N, F, w, src, q, batch, o3, Mi, Ch, s);
  hipLaunchKernelGGL((k_message<false>), dim3(N), dim3(fthreads(F)), 0, s, g, N, F, w, src, q, batch, o3, Mi, Ch, PairRowTable{}, 0);
}

// adjoint of the message sum: the graph and the edge weights are symmetric, so the transpose sweep is the
// same CSR sweep with the message gradient as source:  gPn[i] += sum_e w * gMi[col(e)]
__global__ void k_message_adjoint(Graph g, int N, int F, const float* __restrict__ w, const float* __restrict__ gMi,
                                  float* __restrict__ gPn) {
  const int i = xcd_chunk(blockIdx.x, gridDim.x);
  if (g.counts[2]) return;  // pair overflow: the adjacency was not filled (the host reports the error)
  for (int f = threadIdx.x; f < F; f += blockDim.x) {
    float a[9];
    csr_gather(g, i, F, f, w, gMi, a);
    float* o = gPn + (int64_t)i * 9 * F + f;
#pragma unroll
    for (int c = 0; c < 9; ++c) o[c * F] += a[c];
  }
}
// Adjoint sweep that also produces the layer's distance gradient: at row i, edge (i, j, pair p) contributes
//     h(i <- j) = sum_{k,f} dw[p,k,f] * sum_{c in k} gMi[j,c,f] * Pn[i,c,f]
// and g_d[p] = h(i <- j) + h(j <- i) (tn_pairgrad.hip); gMi[j] is loaded for the adjoint anyway and Pn[i] is the row's own,
// so the separate per-pair kernel (a 4-row gather per pair) disappears.  The per-edge channel sum of a wave goes to its
// own slot (wave, pair, direction): one writer per slot, summed in fixed order by k_geom_gd -> deterministic.
// TWO (halo exchange, the graph holds no pairs of two ghosts): the grid covers the owned rows [own0, own1) only, and an owned row
// also writes the GHOST's half of every pair it has with a ghost j, h(j <- i) = sum dw[p] * gMi[i] * Pn[j] - the pair's rows are in
// registers anyway; it costs the gather of Pn[j] and saves the ghost's whole row (its own gather of gMi[i], a second read of the
// pair's rows, a block per ghost).
template <bool RC, bool TWO = false>
__global__ void k_message_adjoint_gd(Graph g, int N, int F, const float* __restrict__ w, const float* __restrict__ dw,
                                     const float* __restrict__ gMi, const float* __restrict__ Pn, float* __restrict__ gPn,
                                     float* __restrict__ slots, int64_t slot_stride, PairRowTable rt, int own0, int own1, int narrow) {
  const int i = (TWO ? own0 : 0) + xcd_chunk(blockIdx.x, gridDim.x);
  if (g.counts[2]) return;
  const int f = threadIdx.x, lane = f & 63, wave = f >> 6;  // blockDim.x == F (multiple of 64)
  int e0 = g.rowptr[i], e1 = g.rowptr[i + 1];
  // halo exchange, owned rows [own0, own1): a ghost's row is here for its half of the distance gradient of the pairs it has
  // with owned atoms only - the columns of a row ascend (tn_cell.hip), so those edges are one sub-range of the row; its gPn
  // is not wanted
  const bool ghost = !TWO && (i < own0 || i >= own1);
  if (ghost && narrow) {  // (not needed when the graph was built without the pairs of two ghosts)
    int lo = e0, hi = e1;
    while (lo < hi) {  // first edge with col >= own0
      const int mid = (lo + hi) >> 1;
      if (g.col[mid] < own0) lo = mid + 1; else hi = mid;
    }
    e0 = lo;
    hi = e1;
    while (lo < hi) {  // first edge with col >= own1
      const int mid = (lo + hi) >> 1;
      if (g.col[mid] < own1) lo = mid + 1; else hi = mid;
    }
    e1 = lo;
  }
  const int F3 = 3 * F, F9 = 9 * F;
  float y[9], acc[9];
  load9(Pn + (int64_t)i * F9 + f, F, y);
  float gi[TWO ? 9 : 1];
  if (TWO) load9(gMi + (int64_t)i * F9 + f, F, gi);
#pragma unroll
  for (int c = 0; c < 9; ++c) acc[c] = 0.f;
  // the per-edge channel sums h are reduced FOUR edges at a time (wave_sum4: seven exchanges for four wave sums instead
  // of six each; the totals land in lanes 0 / 16 / 32 / 48, which write the four slots)
  auto edge = [&](int e, float& h, int& slot_idx) {
    const int j = g.col[e], p = g.epair[e];
    const float sg = g.esign[e];
    const float* sp = gMi + (int64_t)j * F9 + f;
    float w3[3], d3[3];
    if (RC) {
      pair_row_eval<true>(rt, p, F3, F, f, w3, d3);
    } else {
      const float* wp = w + (int64_t)p * F3 + f;
      const float* dp = dw + (int64_t)p * F3 + f;
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        w3[t] = wp[t * F];
        d3[t] = dp[t * F];
      }
    }
    const float w0 = w3[0], w1 = w3[1], w2 = w3[2];
    const float d0 = d3[0], d1 = d3[1], d2 = d3[2];
    float s9[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) s9[c] = sp[c * F];
    // explicit fused multiply-adds: both instantiations (stored / interpolated rows) round alike
    acc[0] = __fmaf_rn(w0, s9[0], acc[0]);
    acc[1] = __fmaf_rn(w1, s9[1], acc[1]);
    acc[2] = __fmaf_rn(w1, s9[2], acc[2]);
    acc[3] = __fmaf_rn(w1, s9[3], acc[3]);
    acc[4] = __fmaf_rn(w2, s9[4], acc[4]);
    acc[5] = __fmaf_rn(w2, s9[5], acc[5]);
    acc[6] = __fmaf_rn(w2, s9[6], acc[6]);
    acc[7] = __fmaf_rn(w2, s9[7], acc[7]);
    acc[8] = __fmaf_rn(w2, s9[8], acc[8]);
    const float tA = __fmaf_rn(s9[3], y[3], __fmaf_rn(s9[2], y[2], __fmul_rn(s9[1], y[1])));
    const float tS = __fmaf_rn(s9[8], y[8], __fmaf_rn(s9[7], y[7], __fmaf_rn(s9[6], y[6], __fmaf_rn(s9[5], y[5], __fmul_rn(s9[4], y[4])))));
    h = __fmaf_rn(d2, tS, __fmaf_rn(d1, tA, __fmul_rn(d0, __fmul_rn(s9[0], y[0]))));
    slot_idx = sg != 0.f ? 2 * p + (sg > 0.f ? 0 : 1) : -1;  // self edge: no slot
    if (TWO && (j < own0 || j >= own1)) {  // uniform over the block: the ghost's half of this pair
      float pj[9];
      load9(Pn + (int64_t)j * F9 + f, F, pj);
      const float uA = __fmaf_rn(gi[3], pj[3], __fmaf_rn(gi[2], pj[2], __fmul_rn(gi[1], pj[1])));
      const float uS = __fmaf_rn(gi[8], pj[8], __fmaf_rn(gi[7], pj[7], __fmaf_rn(gi[6], pj[6], __fmaf_rn(gi[5], pj[5], __fmul_rn(gi[4], pj[4])))));
      const float h2 = wave_sum(__fmaf_rn(d2, uS, __fmaf_rn(d1, uA, __fmul_rn(d0, __fmul_rn(gi[0], pj[0])))));
      if (lane == 0) slots[(int64_t)wave * slot_stride + 2 * p + (sg > 0.f ? 1 : 0)] = h2;
    }
  };
  int e = e0;
  for (; e + 4 <= e1; e += 4) {
    float h0, h1, h2, h3;
    int s0, s1, s2, s3;
    edge(e, h0, s0);
    edge(e + 1, h1, s1);
    edge(e + 2, h2, s2);
    edge(e + 3, h3, s3);
    const float tot = wave_sum4(h0, h1, h2, h3, lane);
    if ((lane & 15) == 0) {
      const int q4 = lane >> 4;
      const int si = q4 == 0 ? s0 : (q4 == 1 ? s1 : (q4 == 2 ? s2 : s3));
      if (si >= 0) slots[(int64_t)wave * slot_stride + si] = tot;
    }
  }
  for (; e < e1; ++e) {
    float h;
    int si;
    edge(e, h, si);
    h = wave_sum(h);
    if (lane == 0 && si >= 0) slots[(int64_t)wave * slot_stride + si] = h;
  }
  if (ghost) return;
  float* o = gPn + (int64_t)i * F9 + f;
#pragma unroll
  for (int c = 0; c < 9; ++c) o[c * F] += acc[c];
}
bool message_adjoint_gd_ok(int N, int F) { return F % 64 == 0 && (split_rows_ok(N, F) || (N > kSplitRows && F <= 1024)); }
// slot arrays per layer: one per wave of the row kernel's block (F / 64), or one per 32-channel chunk of the tile kernel
int message_adjoint_gd_waves(const Graph& g, int N, int F, bool rows_from_table) {
  return (!rows_from_table && !split_rows_ok(N, F) && message_adjoint_pair_ok(g, N, F)) ? F / 32 : F / 64;
}
void launch_message_adjoint_gd(const Graph& g, int N, int F, const float* w, const float* dw, const float* gMi, const float* Pn,
                               float* gPn, float* slots, int64_t slot_stride, hipStream_t s, const PairRowTable* rt, int own0, int own1, bool narrow) {
  const bool two = !narrow && own1 >= 0 && (own0 != 0 || own1 != N);  // no ghost-ghost pairs in the graph: the owned rows serve both halves
  if (N <= 0) return;
  if (own1 < 0) {
    own0 = 0;
    own1 = N;
  }
  if (rt) {
    hipLaunchKernelGGL((k_message_adjoint_gd<true>), dim3(N), dim3(F), 0, s, g, N, F, w, dw, gMi, Pn, gPn, slots, slot_stride, *rt, own0, own1, narrow ? 1 : 0);
    return;
  }
  if ((own0 != 0 || own1 != N) && !split_rows_ok(N, F) && !message_adjoint_pair_ok(g, N, F)) {  // an owned range: the row kernel
                                                                                                   // (the others treat every row alike)
    if (two)
      hipLaunchKernelGGL((k_message_adjoint_gd<false, true>), dim3(own1 - own0), dim3(F), 0, s, g, N, F, w, dw, gMi, Pn, gPn, slots,
                         slot_stride, PairRowTable{}, own0, own1, 0);
    else
      hipLaunchKernelGGL((k_message_adjoint_gd<false>), dim3(N), dim3(F), 0, s, g, N, F, w, dw, gMi, Pn, gPn, slots, slot_stride,
                         PairRowTable{}, own0, own1, narrow ? 1 : 0);
    return;
  }
  if (split_rows_ok(N, F)) {
    hipLaunchKernelGGL((k_message_split<2>), dim3(N), dim3(kEG * F), 0, s, g, N, F, w, gMi, dw, nullptr, 0, const_cast<float*>(Pn), gPn,
                       slots, slot_stride);
    return;
  }
  // batches of small molecules: the tile kernel (tn_message_pair.hip: gMi window and adjacency slice in LDS, balanced rows)
  if (message_adjoint_pair_ok(g, N, F)) return launch_message_adjoint_pair(g, N, F, w, dw, gMi, Pn, gPn, slots, slot_stride, s);
  hipLaunchKernelGGL((k_message_adjoint_gd<false>), dim3(N), dim3(F), 0, s, g, N, F, w, dw, gMi, Pn, gPn, slots, slot_stride,
                     PairRowTable{}, 0, N, 0);
}

// Value + tangent of a neighbour sum in ONE sweep (second-order pass, tn_hvp_api.hip):
//     out[i] += sum_e w[p(e)] src[col(e)] ;  out_t[i] += sum_e w[p(e)] src_t[col(e)]  (+)  sum_e w_t[p(e)] src[col(e)]
// Three plain sweeps read w twice and w_t once (and the adjacency three times); here every pair row is read once per row atom.
// The three sums are accumulated separately in list order and added in the order the three launches did: bit-identical to them.
template <bool ACC>  // ACC: out += ... (the reverse site adds to what the group-product adjoint wrote); else out = ... (no zero fill, no read)
__global__ void k_message_dual(Graph g, int N, int F, const float* __restrict__ w, const float* __restrict__ w_t,
                               const float* __restrict__ src, const float* __restrict__ src_t, float* __restrict__ out,
                               float* __restrict__ out_t) {
  const int i = xcd_chunk(blockIdx.x, gridDim.x);
  if (g.counts[2]) return;  // pair overflow: the adjacency was not filled (the host reports the error)
  const int F3 = 3 * F, F9 = 9 * F;
  const int e0 = g.rowptr[i], e1 = g.rowptr[i + 1];
  for (int f = threadIdx.x; f < F; f += blockDim.x) {
    float a[9], b1[9], b2[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) a[c] = b1[c] = b2[c] = 0.f;
    struct In {
      float w[3], wt[3], s[9], st[9];
    };
    auto load = [&](int e, In& o) {
      const int j = g.col[e], p = g.epair[e];
      const float* wp = w + (int64_t)p * F3 + f;
      const float* wq = w_t + (int64_t)p * F3 + f;
      const float* sp = src + (int64_t)j * F9 + f;
      const float* sq = src_t + (int64_t)j * F9 + f;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        o.w[k] = wp[k * F];
        o.wt[k] = wq[k * F];
      }
#pragma unroll
      for (int c = 0; c < 9; ++c) {
        o.s[c] = sp[c * F];
        o.st[c] = sq[c * F];
      }
    };
    auto add = [&](const In& o) {
#pragma unroll
      for (int c = 0; c < 9; ++c) {
        const int k = type_of(c);
        a[c] = __fmaf_rn(o.w[k], o.s[c], a[c]);
        b1[c] = __fmaf_rn(o.w[k], o.st[c], b1[c]);
        b2[c] = __fmaf_rn(o.wt[k], o.s[c], b2[c]);
      }
    };
    int e = e0;
    for (; e + 2 <= e1; e += 2) {  // two edges' rows requested together (48 loads in flight), accumulated in list order
      In x0, x1;
      load(e, x0);
      load(e + 1, x1);
      add(x0);
      add(x1);
    }
    if (e < e1) {
      In x0;
      load(e, x0);
      add(x0);
    }
    float* o = out + (int64_t)i * F9 + f;
    float* ot = out_t + (int64_t)i * F9 + f;
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      if (ACC) {
        o[c * F] += a[c];
        ot[c * F] = (ot[c * F] + b1[c]) + b2[c];
      } else {  // (= 0 + a, (0 + b1) + b2: what the three launches over a zero-filled buffer gave)
        o[c * F] = a[c];
        ot[c * F] = b1[c] + b2[c];
      }
    }
  }
}
void launch_message_dual(const Graph& g, int N, int F, const float* w, const float* w_t, const float* src, const float* src_t,
                         float* out, float* out_t, bool accumulate, hipStream_t s) {
  if (N <= 0) return;
  if (split_rows_ok(N, F)) {  // small systems: the split-row sweeps (the chip is mostly idle there: three launches are fine)
    if (!accumulate) {
      launch_fill(out, 0.f, (int64_t)N * 9 * F, s);
      launch_fill(out_t, 0.f, (int64_t)N * 9 * F, s);
    }
    launch_message_adjoint(g, N, F, w, src, out, s);
    launch_message_adjoint(g, N, F, w, src_t, out_t, s);
    launch_message_adjoint(g, N, F, w_t, src, out_t, s);
    return;
  }
  if (accumulate) hipLaunchKernelGGL((k_message_dual<true>), dim3(N), dim3(fthreads(F)), 0, s, g, N, F, w, w_t, src, src_t, out, out_t);
  else hipLaunchKernelGGL((k_message_dual<false>), dim3(N), dim3(fthreads(F)), 0, s, g, N, F, w, w_t, src, src_t, out, out_t);
}

void launch_message_adjoint(const Graph& g, int N, int F, const float* w, const float* gMi, float* gPn, hipStream_t s) {
  if (N <= 0) return;
  if (split_rows_ok(N, F)) {
    hipLaunchKernelGGL((k_message_split<1>), dim3(N), dim3(kEG * F), 0, s, g, N, F, w, gMi, nullptr, nullptr, 0, nullptr, gPn);
    return;
  }
  hipLaunchKernelGGL(k_message_adjoint, dim3(N), dim3(fthreads(F)), 0, s, g, N, F, w, gMi, gPn);
}

// X_new = X_hat + dX + kappa * dX.dX    (reference tensornet.py:811-812; residual on the normalised X)
// NEXT 0: plain ; 1: also the next layer's X_hat = X_new / (||X_new||^2 + 1) (k_norm_x) ; 2: also the readout invariants
// [3 I^2 ; ||A||^2 ; ||S||^2] of X_new (k_readout_feat) -- the consumer's read of X_new and one launch are saved
template <int NEXT>
__global__ void k_layer_update(const float* __restrict__ Xh, const float* __restrict__ D, const float* __restrict__ q,
                               const int64_t* __restrict__ batch, int N, int F, float* __restrict__ Xn, float* __restrict__ nxt) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)N * F) return;
  int n = (int)(idx / F), f = (int)(idx - (int64_t)n * F);
  float xh[9], d[9];
  load9(Xh + (int64_t)n * 9 * F + f, F, xh);
  load9(D + (int64_t)n * 9 * F + f, F, d);
  const float kap = kappa_of(q, batch, n);
  const M3 dX = compose(d);
  const M3 Xf = add(add(compose(xh), dX), scale(matmul(dX, dX), kap));
  float o[9];
  decompose(Xf, o);
  store9(Xn + (int64_t)n * 9 * F + f, F, o);
  if (NEXT == 1) {
    const float inv = 1.0f / (quad(o) + 1.0f);
#pragma unroll
    for (int c = 0; c < 9; ++c) o[c] *= inv;
    store9(nxt + (int64_t)n * 9 * F + f, F, o);
  } else if (NEXT == 2) {
    float* ft = nxt + (int64_t)n * 3 * F + f;
    const float t = o[4] + o[7];
    ft[0] = 3.0f * o[0] * o[0];
    ft[F] = 2.0f * (o[1] * o[1] + o[2] * o[2] + o[3] * o[3]);
    ft[2 * F] = o[4] * o[4] + o[7] * o[7] + t * t + 2.0f * (o[5] * o[5] + o[6] * o[6] + o[8] * o[8]);
  }
}
void launch_layer_update(const float* Xh, const float* D, const float* q, const int64_t* batch, int N, int F, float* Xn, int next,
                         float* nxt, hipStream_t s) {
  if (N <= 0) return;
  const dim3 grid(cdiv((int64_t)N * F, 256)), block(256);
  if (next == 1) hipLaunchKernelGGL((k_layer_update<1>), grid, block, 0, s, Xh, D, q, batch, N, F, Xn, nxt);
  else if (next == 2) hipLaunchKernelGGL((k_layer_update<2>), grid, block, 0, s, Xh, D, q, batch, N, F, Xn, nxt);
  else hipLaunchKernelGGL((k_layer_update<0>), grid, block, 0, s, Xh, D, q, batch, N, F, Xn, nxt);
}

// =====================================================================================
//                                         readout
// =====================================================================================
// feat = [3 I^2 ; ||A||^2 ; ||S||^2]   (reference tensornet.py:385-386)
__global__ void k_readout_feat(const float* __restrict__ X, int N, int F, float* __restrict__ feat) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)N * F) return;
  int n = (int)(idx / F), f = (int)(idx - (int64_t)n * F);
  float u[9];
  load9(X + (int64_t)n * 9 * F + f, F, u);
  float* o = feat + (int64_t)n * 3 * F + f;
  const float t = u[4] + u[7];
  o[0] = 3.0f * u[0] * u[0];
  o[F] = 2.0f * (u[1] * u[1] + u[2] * u[2] + u[3] * u[3]);
  o[2 * F] = u[4] * u[4] + u[7] * u[7] + t * t + 2.0f * (u[5] * u[5] + u[6] * u[6] + u[8] * u[8]);
}
void launch_readout_feat(const float* X, int N, int F, float* feat, hipStream_t s) {
  if (N <= 0) return;
  hipLaunchKernelGGL(k_readout_feat, dim3(cdiv((int64_t)N * F, 256)), dim3(256), 0, s, X, N, F, feat);
}

// per-atom energy: e = silu(ao) . O2 + b ; e*std + atomref[z]   (one wave per atom)
__global__ __launch_bounds__(256) void k_head_energy(const float* __restrict__ ao, const float* __restrict__ O2,
                                                     const float* __restrict__ bO2, int N, int H, float std,
                                                     const float* __restrict__ atomref, const int64_t* __restrict__ z,
                                                     float* __restrict__ ea, float* __restrict__ g_ao,
                                                     const float* __restrict__ aw, const int* __restrict__ perm) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (n >= N) return;
  const float wgt = aw ? aw[perm ? perm[n] : n] : 1.0f;  // weight of this atom in the energy sum (tmdnet_set_atom_weights)
  float s = 0.f;
  for (int k = lane; k < H; k += 64) {
    const float a = ao[(int64_t)n * H + k], o = O2[k];
    s += silu(a) * o;
    if (g_ao) g_ao[(int64_t)n * H + k] = wgt * std * o * silu_grad(a);  // d energy / d ao (k_head_bwd), while the row is in registers
  }
  s = wave_sum(s);
  if (lane == 0) {
    float e = (s + bO2[0]) * std;
    if (atomref) e += atomref[z[n]];
    ea[n] = wgt * e;
  }
}
void launch_head_energy(const float* ao, const float* O2, const float* bO2, int N, int H, float std, const float* atomref,
                        const int64_t* z, float* ea, hipStream_t s, float* g_ao, const float* aw, const int* perm) {
  if (N <= 0) return;
  hipLaunchKernelGGL(k_head_energy, dim3(cdiv(N, 4)), dim3(256), 0, s, ao, O2, bO2, N, H, std, atomref, z, ea, g_ao, aw, perm);
}

// per-molecule sum (reference output_modules.py:43-73): one block per molecule, fixed order (deterministic).
// Sorted batch: the molecule's atom range; unsorted batch (rare): the block scans all atoms for its members.
__global__ __launch_bounds__(256) void k_mol_sum(Graph g, const float* __restrict__ ea, const int64_t* __restrict__ batch, int N, int B,
                                                 float mean, float* __restrict__ energy) {
  __shared__ float part[4];
  const int m = blockIdx.x;
  float s = 0.f;
  if (!g.counts[3]) {
    const int i0 = g.mstart[m], i1 = g.mend[m];
    for (int i = i0 + threadIdx.x; i < i1; i += 256) s += ea[i];
  } else {
    for (int i = threadIdx.x; i < N; i += 256)
      if (batch[i] == m) s += ea[i];
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) energy[m] = part[0] + part[1] + part[2] + part[3] + mean;
}
void launch_mol_sum(const Graph& g, const float* ea, const int64_t* batch, int N, int B, float mean, float* energy, hipStream_t s) {
  if (B <= 0) return;
  hipLaunchKernelGGL(k_mol_sum, dim3(B), dim3(256), 0, s, g, ea, batch, N, B, mean, energy);
}

// head energy + per-molecule sum in one launch: block per molecule, its 16 waves take the molecule's atoms in turn
// (e = silu(ao) . O2 + b, * std, + atomref[z]); unsorted batch: the block scans all atoms for its members
__global__ __launch_bounds__(1024) void k_head_mol_sum(Graph g, const float* __restrict__ ao, const float* __restrict__ O2,
                                                      const float* __restrict__ bO2, int N, int H, float std,
                                                      const float* __restrict__ atomref, const int64_t* __restrict__ z,
                                                      const int64_t* __restrict__ batch, float mean, float* __restrict__ energy,
                                                      float* __restrict__ g_ao, const float* __restrict__ aw,
                                                      const int* __restrict__ perm) {
  __shared__ float part[16];
  const int m = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool sorted = !g.counts[3];
  const int i0 = sorted ? g.mstart[m] : 0, i1 = sorted ? g.mend[m] : N;
  float acc = 0.f;  // lane 0 of each wave: sum over its atoms, fixed order
  for (int n = i0 + wave; n < i1; n += 16) {
    if (!sorted && batch[n] != m) continue;
    const float wgt = aw ? aw[perm ? perm[n] : n] : 1.0f;
    float s = 0.f;
    for (int k = lane; k < H; k += 64) {
      const float a = ao[(int64_t)n * H + k], o = O2[k];
      s += silu(a) * o;
      if (g_ao) g_ao[(int64_t)n * H + k] = wgt * std * o * silu_grad(a);
    }
    s = wave_sum(s);
    float e = (s + bO2[0]) * std;
    if (atomref) e += atomref[z[n]];
    acc += wgt * e;
  }
  if (lane == 0) part[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int k = 0; k < 16; ++k) tot += part[k];
    energy[m] = tot + mean;
  }
}
void launch_head_mol_sum(const Graph& g, const float* ao, const float* O2, const float* bO2, int N, int B, int H, float std,
                         const float* atomref, const int64_t* z, const int64_t* batch, float mean, float* energy, hipStream_t s,
                         float* g_ao, const float* aw, const int* perm) {
  if (B <= 0) return;
  hipLaunchKernelGGL(k_head_mol_sum, dim3(B), dim3(1024), 0, s, g, ao, O2, bO2, N, H, std, atomref, z, batch, mean, energy, g_ao, aw,
                     perm);
}

// LayerNorm adjoint of the readout row [3F] followed by the adjoint of the invariants (k_readout_bwd): with F % 64 == 0 the
// lane that holds g_feat[f] also holds g_feat[F + f] and g_feat[2F + f], so G[n, :, f] is written straight away
__global__ __launch_bounds__(256) void k_lnbwd_readout_bwd(const float* __restrict__ g, const float* __restrict__ xhat,
                                                           const float* __restrict__ rstd, const float* __restrict__ w, int rows,
                                                           int F, const float* __restrict__ X, float* __restrict__ G) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const int R = 3 * F;
  const float* gr = g + (int64_t)row * R;
  const float* xh = xhat + (int64_t)row * R;
  float s1 = 0.f, s2 = 0.f;
  for (int k = lane; k < R; k += 64) {
    float gw = gr[k] * w[k];
    s1 += gw;
    s2 += gw * xh[k];
  }
  s1 = wave_sum(s1) / R;
  s2 = wave_sum(s2) / R;
  const float rs = rstd[row];
  for (int f = lane; f < F; f += 64) {
    const float gI = (gr[f] * w[f] - s1 - xh[f] * s2) * rs;
    const float gA = (gr[F + f] * w[F + f] - s1 - xh[F + f] * s2) * rs;
    const float gS = (gr[2 * F + f] * w[2 * F + f] - s1 - xh[2 * F + f] * s2) * rs;
    float u[9], dq[9];
    load9(X + (int64_t)row * 9 * F + f, F, u);
    dquad(u, dq);
    float o[9] = {dq[0] * gI, dq[1] * gA, dq[2] * gA, dq[3] * gA, dq[4] * gS, dq[5] * gS, dq[6] * gS, dq[7] * gS, dq[8] * gS};
    store9(G + (int64_t)row * 9 * F + f, F, o);
  }
}
void launch_lnbwd_readout_bwd(const float* g, const float* xhat, const float* rstd, const float* w, int N, int F, const float* X,
                              float* G, hipStream_t s) {
  if (N <= 0) return;
  hipLaunchKernelGGL(k_lnbwd_readout_bwd, dim3(cdiv(N, 4)), dim3(256), 0, s, g, xhat, rstd, w, N, F, X, G);
}

// =====================================================================================
//                                       reverse pass
// =====================================================================================
__global__ void k_head_bwd(const float* __restrict__ ao, const float* __restrict__ O2, int N, int H, float std,
                           float* __restrict__ g_ao) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)N * H) return;
  int k = (int)(idx % H);
  g_ao[idx] = std * O2[k] * silu_grad(ao[idx]);
}
void launch_head_bwd(const float* ao, const float* O2, int N, int H, float std, float* g_ao, hipStream_t s) {
  if (N <= 0) return;
  hipLaunchKernelGGL(k_head_bwd, dim3(cdiv((int64_t)N * H, 256)), dim3(256), 0, s, ao, O2, N, H, std, g_ao);
}

__global__ void k_readout_bwd(const float* __restrict__ X, const float* __restrict__ g_feat, int N, int F, float* __restrict__ G) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)N * F) return;
  int n = (int)(idx / F), f = (int)(idx - (int64_t)n * F);
  float u[9], dq[9];
  load9(X + (int64_t)n * 9 * F + f, F, u);
  dquad(u, dq);
  const float* gf = g_feat + (int64_t)n * 3 * F + f;
  const float gI = gf[0], gA = gf[F], gS = gf[2 * F];
  // feat_I = 3 I^2 = quad restricted to I, feat_A = 2|v|^2, feat_S = ||S||^2: same partials as dquad
  float o[9] = {dq[0] * gI, dq[1] * gA, dq[2] * gA, dq[3] * gA, dq[4] * gS, dq[5] * gS, dq[6] * gS, dq[7] * gS, dq[8] * gS};
  store9(G + (int64_t)n * 9 * F + f, F, o);
}
void launch_readout_bwd(const float* X, const float* g_feat, int N, int F, float* G, hipStream_t s) {
  if (N <= 0) return;
  hipLaunchKernelGGL(k_readout_bwd, dim3(cdiv((int64_t)N * F, 256)), dim3(256), 0, s, X, g_feat, N, F, G);
}

// g_D = compose^T( Gf + kappa (Gf dX^T + dX^T Gf) ),  Gf = dec^T(G)
__global__ void k_update_bwd(const float* __restrict__ G, const float* __restrict__ D, const float* __restrict__ q,
                             const int64_t* __restrict__ batch, int N, int F, float* __restrict__ gD) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)N * F) return;
  int n = (int)(idx / F), f = (int)(idx - (int64_t)n * F);
  float gg[9], d[9];
  load9(G + (int64_t)n * 9 * F + f, F, gg);
  load9(D + (int64_t)n * 9 * F + f, F, d);
  const float kap = kappa_of(q, batch, n);
  const M3 Gf = decompose_T(gg);
  const M3 dXt = transpose(compose(d));
  const M3 gdx = add(Gf, scale(add(matmul(Gf, dXt), matmul(dXt, Gf)), kap));
  float o[9];
  compose_T(gdx, o);
  store9(gD + (int64_t)n * 9 * F + f, F, o);
}
void launch_update_bwd(const float* G, const float* D, const float* q, const int64_t* batch, int N, int F, float* gD, hipStream_t s) {
  if (N <= 0) return;
  hipLaunchKernelGGL(k_update_bwd, dim3(cdiv((int64_t)N * F, 256)), dim3(256), 0, s, G, D, q, batch, N, F, gD);
}

// adjoint of (Y, M) -> C_hat ; writes gMi (wrt the message components) and the direct part of gPn
__global__ void k_message_bwd_node(const float* __restrict__ gCh, const float* __restrict__ Pn, const float* __restrict__ Mi,
                                   const float* __restrict__ q, const int64_t* __restrict__ batch, int o3, int N, int F,
                                   float* __restrict__ gMi, float* __restrict__ gPn) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)N * F) return;
  int n = (int)(idx / F), f = (int)(idx - (int64_t)n * F);
  float gc[9], y[9], m[9];
  load9(gCh + (int64_t)n * 9 * F + f, F, gc);
  load9(Pn + (int64_t)n * 9 * F + f, F, y);
  load9(Mi + (int64_t)n * 9 * F + f, F, m);
  const float kap = kappa_of(q, batch, n);
  const M3 Y = compose(y), M = compose(m);
  const M3 Cm = o3 ? scale(add(matmul(Y, M), matmul(M, Y)), kap) : scale(matmul(Y, M), 2.0f);
  float uc[9];
  decompose(Cm, uc);
  const float inv = 1.0f / (frob2(Cm) + 1.0f);
  float dot = 0.f, guc[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) {
    dot += gc[c] * uc[c];
    guc[c] = gc[c] * inv;
  }
  const float g_t = -dot * inv * inv;
  const M3 gCm = add(decompose_T(guc), scale(Cm, 2.0f * g_t));
  const M3 Yt = transpose(Y), Mt = transpose(M);
  M3 gY, gM;
  if (o3) {
    gY = scale(add(matmul(gCm, Mt), matmul(Mt, gCm)), kap);
    gM = scale(add(matmul(Yt, gCm), matmul(gCm, Yt)), kap);
  } else {
    gY = scale(matmul(gCm, Mt), 2.0f);
    gM = scale(matmul(Yt, gCm), 2.0f);
  }
  float o[9];
  compose_T(gM, o);
  store9(gMi + (int64_t)n * 9 * F + f, F, o);
  compose_T(gY, o);
  store9(gPn + (int64_t)n * 9 * F + f, F, o);
}
void launch_message_bwd_node(const float* gCh, const float* Pn, const float* Mi, const float* q, const int64_t* batch, int o3, int N,
                             int F, float* gMi, float* gPn, hipStream_t s) {
  if (N <= 0) return;
  hipLaunchKernelGGL(k_message_bwd_node, dim3(cdiv((int64_t)N * F, 256)), dim3(256), 0, s, gCh, Pn, Mi, q, batch, o3, N, F, gMi,
                     gPn);
}

// block-level sum of up to 4 values over blockDim threads (blockDim multiple of 64, <= 256)
template <int NV>
__device__ __forceinline__ void block_sum(float (&v)[NV], float* red /* [4*NV] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = wave_sum(v[k]);
  if (nw == 1) return;
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < NV; ++k) red[wave * NV + k] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    float s = 0.f;
    for (int w = 0; w < nw; ++w) s += red[w * NV + k];
    v[k] = s;
  }
}

// G <- (G + gXh_lin)/(s+1) + dquad(X) * g_s ,  g_s = -sum (G + gXh_lin).X / (s+1)^2
// NEXT 0: plain ; 1: also the update adjoint of the layer below, gD = G + kappa (G D^T + D^T G) (k_update_bwd) ;
// 2: also the embedding gate adjoint (k_embed_gate_bwd) -- then the new G itself is not stored at all
template <int NEXT>
__global__ void k_norm_bwd(const float* __restrict__ X, const float* __restrict__ gl, int N, int F, float* __restrict__ G,
                           const float* __restrict__ Dn, const float* __restrict__ q, const int64_t* __restrict__ batch,
                           float* __restrict__ gD, const float* __restrict__ UX, const float* __restrict__ gates,
                           const float* __restrict__ a2, float* __restrict__ gUX, float* __restrict__ g_a2) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)N * F) return;
  int n = (int)(idx / F), f = (int)(idx - (int64_t)n * F);
  float u[9], gx[9], l[9], dq[9];
  load9(X + (int64_t)n * 9 * F + f, F, u);
  load9(G + (int64_t)n * 9 * F + f, F, gx);
  load9(gl + (int64_t)n * 9 * F + f, F, l);
  dquad(u, dq);
  const float inv = 1.0f / (quad(u) + 1.0f);
  float dot = 0.f;
#pragma unroll
  for (int c = 0; c < 9; ++c) {
    gx[c] += l[c];
    dot += gx[c] * u[c];
  }
  const float g_s = -dot * inv * inv;
#pragma unroll
  for (int c = 0; c < 9; ++c) gx[c] = gx[c] * inv + dq[c] * g_s;
  if (NEXT != 2) store9(G + (int64_t)n * 9 * F + f, F, gx);
  if (NEXT == 1) {
    float d[9];
    load9(Dn + (int64_t)n * 9 * F + f, F, d);
    const float kap = kappa_of(q, batch, n);
    const M3 Gf = decompose_T(gx);
    const M3 dXt = transpose(compose(d));
    const M3 gdx = add(Gf, scale(add(matmul(Gf, dXt), matmul(dXt, Gf)), kap));
    float o[9];
    compose_T(gdx, o);
    store9(gD + (int64_t)n * 9 * F + f, F, o);
  } else if (NEXT == 2) {
    float ux[9], o[9];
    load9(UX + (int64_t)n * 9 * F + f, F, ux);
    const float* gt = gates + (int64_t)n * 3 * F + f;
    const float g3[3] = {gt[0], gt[F], gt[2 * F]};
    float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      o[c] = gx[c] * g3[type_of(c)];
      acc[type_of(c)] += gx[c] * ux[c];
    }
    store9(gUX + (int64_t)n * 9 * F + f, F, o);
#pragma unroll
    for (int k = 0; k < 3; ++k) g_a2[(int64_t)n * 3 * F + k * F + f] = acc[k] * silu_grad(a2[(int64_t)n * 3 * F + k * F + f]);
  }
}
void launch_norm_bwd(const float* X, const float* gXh_lin, int N, int F, float* G, hipStream_t s) {
  if (N <= 0) return;
  hipLaunchKernelGGL((k_norm_bwd<0>), dim3(cdiv((int64_t)N * F, 256)), dim3(256), 0, s, X, gXh_lin, N, F, G, nullptr, nullptr, nullptr,
                     nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
}
void launch_norm_bwd_update_bwd(const float* X, const float* gXh_lin, int N, int F, float* G, const float* Dn, const float* q,
                                const int64_t* batch, float* gD, hipStream_t s) {
  if (N <= 0) return;
  hipLaunchKernelGGL((k_norm_bwd<1>), dim3(cdiv((int64_t)N * F, 256)), dim3(256), 0, s, X, gXh_lin, N, F, G, Dn, q, batch, gD,
                     nullptr, nullptr, nullptr, nullptr, nullptr);
}
void launch_norm_bwd_gate_bwd(const float* X, const float* gXh_lin, int N, int F, float* G, const float* UX, const float* gates,
                              const float* a2, float* gUX, float* g_a2, hipStream_t s) {
  if (N <= 0) return;
  hipLaunchKernelGGL((k_norm_bwd<2>), dim3(cdiv((int64_t)N * F, 256)), dim3(256), 0, s, X, gXh_lin, N, F, G, nullptr, nullptr, nullptr,
                     nullptr, UX, gates, a2, gUX, g_a2);
}

// X1[c] = UX[c] * gate[type(c)]: gUX = G * gate ; g_gate = sum_c G*UX ; g_a2 = g_gate * silu'(a2)
__global__ void k_embed_gate_bwd(const float* __restrict__ G, const float* __restrict__ UX, const float* __restrict__ gates,
                                 const float* __restrict__ a2, int N, int F, float* __restrict__ gUX, float* __restrict__ g_a2) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)N * F) return;
  int n = (int)(idx / F), f = (int)(idx - (int64_t)n * F);
  float gg[9], ux[9], o[9];
  load9(G + (int64_t)n * 9 * F + f, F, gg);
  load9(UX + (int64_t)n * 9 * F + f, F, ux);
  const float* gt = gates + (int64_t)n * 3 * F + f;
  const float g3[3] = {gt[0], gt[F], gt[2 * F]};
  float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 9; ++c) {
    o[c] = gg[c] * g3[type_of(c)];
    acc[type_of(c)] += gg[c] * ux[c];
  }
  store9(gUX + (int64_t)n * 9 * F + f, F, o);
#pragma unroll
  for (int k = 0; k < 3; ++k) g_a2[(int64_t)n * 3 * F + k * F + f] = acc[k] * silu_grad(a2[(int64_t)n * 3 * F + k * F + f]);
}
void launch_embed_gate_bwd(const float* G, const float* UX, const float* gates, const float* a2, int N, int F, float* gUX, float* g_a2,
                           hipStream_t s) {
  if (N <= 0) return;
  hipLaunchKernelGGL(k_embed_gate_bwd, dim3(cdiv((int64_t)N * F, 256)), dim3(256), 0, s, G, UX, gates, a2, N, F, gUX, g_a2);
}

// g_u0 = g_u0_lin + dquad(u0) g_s0n  ->  gradients wrt (I0, v[3], T00,T01,T02,T11,T12,T22)   [N,10,F]
__global__ void k_embed_bwd_atom(const float* __restrict__ gl, const float* __restrict__ u0, const float* __restrict__ g_s0n, int N,
                                 int F, float* __restrict__ gA) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)N * F) return;
  int n = (int)(idx / F), f = (int)(idx - (int64_t)n * F);
  float g[9], u[9], dq[9];
  load9(gl + (int64_t)n * 9 * F + f, F, g);
  load9(u0 + (int64_t)n * 9 * F + f, F, u);
  dquad(u, dq);
  const float gs = g_s0n[(int64_t)n * F + f];
#pragma unroll
  for (int c = 0; c < 9; ++c) g[c] += dq[c] * gs;
  const float third = (g[4] + g[7]) * (1.0f / 3.0f);
  float* o = gA + (int64_t)n * 10 * F + f;
  o[0] = g[0];
  o[F] = g[1];
  o[2 * F] = g[2];
  o[3 * F] = g[3];
  o[4 * F] = g[4] - third;  // T00
  o[5 * F] = g[5];          // T01
  o[6 * F] = g[6];          // T02
  o[7 * F] = g[7] - third;  // T11
  o[8 * F] = g[8];          // T12
  o[9 * F] = -third;        // T22
}
void launch_embed_bwd_atom(const float* g_u0_lin, const float* u0, const float* g_s0n, int N, int F, float* gA, hipStream_t s) {
  if (N <= 0) return;
  hipLaunchKernelGGL(k_embed_bwd_atom, dim3(cdiv((int64_t)N * F, 256)), dim3(256), 0, s, g_u0_lin, u0, g_s0n, N, F, gA);
}

// F_i = - sum_{e in row(i)} sign(e) * g_delta[pair(e)]     (no atomics: CSR gather)
// 16 lanes per atom stride over the row's edges; fixed-order shuffle tree (deterministic)
__global__ __launch_bounds__(256) void k_force_gather(Graph g, int N, const float* __restrict__ g_delta, const int* __restrict__ perm,
                                                      float* __restrict__ forces) {
  const int i = blockIdx.x * 16 + (threadIdx.x >> 4), sub = threadIdx.x & 15;
  const bool live = i < N && !g.counts[2];
  float fx = 0.f, fy = 0.f, fz = 0.f;
  if (live) {
    const int e1 = g.rowptr[i + 1];
    for (int e = g.rowptr[i] + sub; e < e1; e += 16) {
      const float sg = g.esign[e];
      if (sg == 0.f) continue;
      const int p = g.epair[e];
      fx -= sg * g_delta[p * 3];
      fy -= sg * g_delta[p * 3 + 1];
      fz -= sg * g_delta[p * 3 + 2];
    }
  }
#pragma unroll
  for (int off = 8; off >= 1; off >>= 1) {
    fx += __shfl_down(fx, off, 16);
    fy += __shfl_down(fy, off, 16);
    fz += __shfl_down(fz, off, 16);
  }
  if (live && sub == 0) {
    const int o = perm ? perm[i] : i;  // cell-list path: back to the caller's atom order
    forces[o * 3] = fx;
    forces[o * 3 + 1] = fy;
    forces[o * 3 + 2] = fz;
  }
}
void launch_force_gather(const Graph& g, int N, const float* g_delta, const int* perm, float* forces, hipStream_t s) {
  if (N <= 0) return;
  hipLaunchKernelGGL(k_force_gather, dim3(cdiv(N, 16)), dim3(256), 0, s, g, N, g_delta, perm, forces);
}

// per-atom charge factor kappa_n = 1 + 0.1 q[batch_n] (reference tensornet.py:789, 812); an out-of-range molecule index
// (reported through counts[5]) reads q[0]
__global__ void k_kappa(const float* __restrict__ q, const int64_t* __restrict__ batch, int N, int B, float* __restrict__ kap) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  int64_t b = batch[i];
  if (b < 0 || b >= B) b = 0;
  kap[i] = 1.0f + 0.1f * q[b];
}
void launch_kappa(const float* q, const int64_t* batch, int N, int B, float* kap, hipStream_t s) {
  if (N <= 0 || B <= 0) return;
  hipLaunchKernelGGL(k_kappa, dim3(cdiv(N, 256)), dim3(256), 0, s, q, batch, N, B, kap);
}

__global__ void k_fill(float* p, float v, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
void launch_fill(float* p, float v, int64_t n, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_fill, dim3(cdiv(n, 256)), dim3(256), 0, s, p, v, n);
}

// ---- halo exchange: range of the owned rows (one block; a few tens of thousands of atoms)
__global__ __launch_bounds__(1024) void k_owned_range(const float* __restrict__ aw, const int* __restrict__ perm, int N,
                                                      int* __restrict__ out) {
  __shared__ int red[3][16];
  int lo = 0x7fffffff, hi = -1, cnt = 0;
  for (int n = threadIdx.x; n < N; n += 1024)
    if (aw[perm ? perm[n] : n] != 0.f) {
      lo = min(lo, n);
      hi = max(hi, n);
      ++cnt;
    }
  for (int off = 32; off >= 1; off >>= 1) {
    lo = min(lo, __shfl_xor(lo, off, 64));
    hi = max(hi, __shfl_xor(hi, off, 64));
    cnt += __shfl_xor(cnt, off, 64);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    red[0][wave] = lo;
    red[1][wave] = hi;
    red[2][wave] = cnt;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w) {
      lo = min(lo, red[0][w]);
      hi = max(hi, red[1][w]);
      cnt += red[2][w];
    }
    out[0] = lo;
    out[1] = hi;
    out[2] = cnt;
    out[3] = 0;
  }
}
void launch_owned_range(const float* aw, const int* perm, int N, int* out, hipStream_t s) {
  hipLaunchKernelGGL(k_owned_range, dim3(1), dim3(1024), 0, s, aw, perm, N, out);
}

}  // namespace tn
