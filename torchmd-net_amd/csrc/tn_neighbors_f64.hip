// Double-precision neighbour operator (gfx950): the pair list, displacement vectors and distances of fp64 positions, and the
// position gradient of those outputs.
//
// The reference's neighbour kernels are generic over the position dtype (extensions/warp_kernels/neighbors_brute.py:27-175,
// neighbors_cell.py:17-153 are instantiated for float32 and float64; tests/test_neighbors.py:83,157,281 run both), while the
// model path of this library is fp32 only (tn_graph_wave.hip, tn_cell.hip build its CSR).  This file is the fp64 instantiation of
// the OPERATOR alone (torch.ops.tmdnet.neighbor_pairs / neighbor_grad with double tensors): exact reference semantics
//     pair (i > j, same molecule) kept iff  |delta|^2 < cutoff_upper^2  and  |delta|^2 >= cutoff_lower^2
//     delta = pos[i] - pos[j] - round(dz / b22) b2 - round(dy' / b11) b1 - round(dx'' / b00) b0      (triclinic, z -> y -> x)
// with one box for all molecules or one per molecule, [+ transposed entries] [+ self loops], padded with -1 / 0, and the TOTAL
// count reported even when it exceeds max_num_pairs (the host raises then, models/utils.py:297-300).
//
// Layout and order are deterministic (the reference's are not: it appends through an atomic counter): a wave per atom i walks
// j < i with its 64 lanes, the count pass and the fill pass evaluate the SAME expression (contraction off, like
// pair_geometry in tn_common.h), a one-block scan turns the per-atom counts into offsets, and a lane's slot is its rank among
// the wave's accepted lanes (ballot + mbcnt): no atomics anywhere, ascending (i, j).  The molecule search is O(N^2 / 64) wave
// trips over ALL j < i with a batch comparison, as the reference's brute-force kernel does (batch need not be sorted); the
// "cell" strategy of an fp64 call takes this kernel too - same pair set, and the operator's fp64 users are the reference's
// tests and gradient checks at <= 12 800 atoms, not the MD hot path.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tmdnet_amd.h"

namespace tn {

struct NbrF64Args {
  const double* pos;
  const int64_t* batch;
  const double* box;   // nullptr, [3,3] or [n_mol,3,3]
  int box_stride;      // 0: one box, 9: one per molecule
  double lo2, up2;
  int N;
};

// round half away from zero (wp.round / std::round), spelled out: rint() would round half to even
__device__ __forceinline__ double nbr64_round(double x) { return round(x); }

__device__ __forceinline__ bool nbr64_geometry(const NbrF64Args& a, int i, int j, double& dx, double& dy, double& dz, double& d2) {
#pragma clang fp contract(off)
  const int64_t bi = a.batch[i];
  if (bi != a.batch[j]) return false;
  dx = a.pos[3 * (int64_t)i + 0] - a.pos[3 * (int64_t)j + 0];
  dy = a.pos[3 * (int64_t)i + 1] - a.pos[3 * (int64_t)j + 1];
  dz = a.pos[3 * (int64_t)i + 2] - a.pos[3 * (int64_t)j + 2];
  if (a.box) {
    const double* b = a.box + bi * a.box_stride;
    const double s3 = nbr64_round(dz / b[8]);
    dx = dx - s3 * b[6];
    dy = dy - s3 * b[7];
    dz = dz - s3 * b[8];
    const double s2 = nbr64_round(dy / b[4]);
    dx = dx - s2 * b[3];
    dy = dy - s2 * b[4];
    const double s1 = nbr64_round(dx / b[0]);
    dx = dx - s1 * b[0];
  }
  d2 = dx * dx + dy * dy + dz * dz;
  return d2 < a.up2 && d2 >= a.lo2;
}

// pass 1: counts[i] = accepted j < i.  One wave per atom.
__global__ __launch_bounds__(256) void k_nbr64_count(NbrF64Args a, int* __restrict__ counts) {
  const int wave = (int)((blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6), lane = threadIdx.x & 63;
  if (wave >= a.N) return;
  const int i = wave;
  int c = 0;
  for (int j0 = 0; j0 < i; j0 += 64) {
    const int j = j0 + lane;
    double dx, dy, dz, d2;
    const bool ok = j < i && nbr64_geometry(a, i, j, dx, dy, dz, d2);
    c += __popcll(__ballot(ok));
  }
  if (lane == 0) counts[i] = c;
}

// exclusive scan of counts[0..N) into offsets[0..N], one block (N is the operator's, not the model's: <= ~10^5 in every use)
__global__ __launch_bounds__(1024) void k_nbr64_scan(const int* __restrict__ counts, int N, int* __restrict__ offsets) {
  __shared__ int part[1024];
  const int t = threadIdx.x, nt = blockDim.x;
  const int per = (N + nt - 1) / nt, b = t * per, e = min(N, b + per);
  int s = 0;
  for (int k = b; k < e; ++k) s += counts[k];
  part[t] = s;
  __syncthreads();
  for (int off = 1; off < nt; off <<= 1) {
    const int v = t >= off ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run = t ? part[t - 1] : 0;
  for (int k = b; k < e; ++k) {
    offsets[k] = run;
    run += counts[k];
  }
  if (t == nt - 1) offsets[N] = part[nt - 1];
}

__global__ void k_nbr64_pad(int64_t max_pairs, int64_t* __restrict__ nb, double* __restrict__ deltas, double* __restrict__ dist) {
  const int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (k >= max_pairs) return;
  nb[k] = -1;
  nb[max_pairs + k] = -1;
  deltas[3 * k] = deltas[3 * k + 1] = deltas[3 * k + 2] = 0.0;
  dist[k] = 0.0;
}

// pass 2: same walk, a lane's slot = offsets[i] + (accepted lanes before it in this and earlier trips).  Entry k of the lower block
// is (i, j, delta, d); its transpose sits at P + k; the self loops follow.
__global__ __launch_bounds__(256) void k_nbr64_fill(NbrF64Args a, const int* __restrict__ offsets, int include_transpose, int loop,
                                                    int64_t max_pairs, int64_t* __restrict__ nb, double* __restrict__ deltas,
                                                    double* __restrict__ dist, int* __restrict__ num_pairs) {
  const int wave = (int)((blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6), lane = threadIdx.x & 63;
  if (wave >= a.N) return;
  const int i = wave;
  const int64_t P = offsets[a.N];
  const int64_t total = P * (include_transpose ? 2 : 1) + (loop ? a.N : 0);
  if (i == 0 && lane == 0) num_pairs[0] = (int)total;
  int64_t base = offsets[i];
  for (int j0 = 0; j0 < i; j0 += 64) {
    const int j = j0 + lane;
    double dx = 0, dy = 0, dz = 0, d2 = 0;
    const bool ok = j < i && nbr64_geometry(a, i, j, dx, dy, dz, d2);
    const unsigned long long m = __ballot(ok);
    if (ok) {
      const int64_t k = base + __popcll(m & ((1ull << lane) - 1ull));
      const double d = sqrt(d2);
      if (k < max_pairs) {
        nb[k] = i;
        nb[max_pairs + k] = j;
        deltas[3 * k] = dx; deltas[3 * k + 1] = dy; deltas[3 * k + 2] = dz;
        dist[k] = d;
      }
      const int64_t kt = P + k;
      if (include_transpose && kt < max_pairs) {
        nb[kt] = j;
        nb[max_pairs + kt] = i;
        deltas[3 * kt] = -dx; deltas[3 * kt + 1] = -dy; deltas[3 * kt + 2] = -dz;
        dist[kt] = d;
      }
    }
    base += __popcll(m);
  }
  if (loop && lane == 0) {
    const int64_t k = P * (include_transpose ? 2 : 1) + i;
    if (k < max_pairs) {
      nb[k] = i;
      nb[max_pairs + k] = i;  // delta = 0, d = 0: already written by the padding kernel
    }
  }
}

// Gradient of the operator's outputs wrt the positions (reference extensions/neighbor_utils.py:11-46), fp64:
//   g = g_delta[p] + delta[p] / d[p] * g_dist[p]   (zero for padded entries and for d = 0) ; out[i] += g ; out[j] -= g
// The COO list has no row structure: double atomics (off the hot path, like the fp32 operator's k_neighbor_grad).
__global__ void k_nbr64_grad(const int64_t* __restrict__ nb, const double* __restrict__ deltas, const double* __restrict__ dist,
                             const double* __restrict__ g_delta, const double* __restrict__ g_dist, int64_t M, int N,
                             double* __restrict__ out) {
  const int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (p >= M) return;
  const int64_t i = nb[p], j = nb[M + p];
  const double d = dist[p];
  if (i < 0 || j < 0 || i >= N || j >= N || d == 0.0) return;
  const double gd = g_dist ? g_dist[p] / d : 0.0;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const double g = (g_delta ? g_delta[3 * p + c] : 0.0) + deltas[3 * p + c] * gd;
    atomicAdd(out + 3 * i + c, g);
    atomicAdd(out + 3 * j + c, -g);
  }
}

__global__ void k_nbr64_zero(double* __restrict__ p, int64_t n) {
  const int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (k < n) p[k] = 0.0;
}

}  // namespace tn

extern "C" {

int tmdnet_neighbor_pairs_f64(void* stream, void* ws, size_t ws_bytes, int64_t n_atoms, int64_t n_mol, const double* pos,
                              const int64_t* batch, const double* box, int32_t box_mode, double cutoff_lower, double cutoff_upper,
                              int64_t max_num_pairs, int32_t loop, int32_t include_transpose, int64_t* neighbors, double* deltas,
                              double* distances, int32_t* num_pairs) {
  using namespace tn;
  if (!ws || !neighbors || !deltas || !distances || !num_pairs || n_atoms < 0 || n_mol < 0 || max_num_pairs < 0) return TMDNET_ERR_INVALID;
  if (n_atoms > 0 && (!pos || !batch)) return TMDNET_ERR_INVALID;
  if (box_mode != 0 && !box) return TMDNET_ERR_INVALID;
  if (n_atoms >= (int64_t)1 << 30) return TMDNET_ERR_INVALID;
  const size_t need = (size_t)(2 * n_atoms + 4) * sizeof(int);
  if (need > ws_bytes) return TMDNET_ERR_WORKSPACE;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int* counts = reinterpret_cast<int*>(ws);
  int* offsets = counts + n_atoms + 1;
  const int N = (int)n_atoms;
  NbrF64Args a{pos, batch, box_mode ? box : nullptr, box_mode == 2 ? 9 : 0, cutoff_lower * cutoff_lower, cutoff_upper * cutoff_upper, N};
  if (max_num_pairs > 0)
    hipLaunchKernelGGL(k_nbr64_pad, dim3((unsigned)((max_num_pairs + 255) / 256)), dim3(256), 0, s, max_num_pairs, neighbors, deltas, distances);
  const unsigned wblocks = (unsigned)((n_atoms * 64 + 255) / 256);
  if (N > 0) hipLaunchKernelGGL(k_nbr64_count, dim3(wblocks), dim3(256), 0, s, a, counts);
  hipLaunchKernelGGL(k_nbr64_scan, dim3(1), dim3(1024), 0, s, counts, N, offsets);
  if (N > 0) {
    hipLaunchKernelGGL(k_nbr64_fill, dim3(wblocks), dim3(256), 0, s, a, offsets, (int)include_transpose, (int)loop, max_num_pairs, neighbors,
                       deltas, distances, num_pairs);
  } else {
    (void)hipMemsetAsync(num_pairs, 0, sizeof(int32_t), s);
  }
  return hipGetLastError() == hipSuccess ? TMDNET_OK : TMDNET_ERR_HIP;
}

int tmdnet_neighbor_grad_f64(void* stream, const int64_t* neighbors, const double* deltas, const double* distances,
                             const double* grad_deltas, const double* grad_distances, int64_t num_entries, int64_t n_atoms,
                             double* grad_positions) {
  using namespace tn;
  if (!grad_positions || num_entries < 0 || n_atoms < 0) return TMDNET_ERR_INVALID;
  if (num_entries > 0 && (!neighbors || !deltas || !distances)) return TMDNET_ERR_INVALID;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int64_t n3 = 3 * n_atoms;
  if (n3 > 0) hipLaunchKernelGGL(k_nbr64_zero, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0, s, grad_positions, n3);
  if (num_entries > 0)
    hipLaunchKernelGGL(k_nbr64_grad, dim3((unsigned)((num_entries + 255) / 256)), dim3(256), 0, s, neighbors, deltas, distances, grad_deltas,
                       grad_distances, num_entries, (int)n_atoms, grad_positions);
  return hipGetLastError() == hipSuccess ? TMDNET_OK : TMDNET_ERR_HIP;
}

}  // extern "C"
