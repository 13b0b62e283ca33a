// O(N) cell-list neighbour graph for one large periodic system (orthorhombic or reduced triclinic box), gfx950.
//
// Reference: torchmdnet/extensions/neighbor_utils.py:89-150 (build_cell_list: wrap, bin, sort, cell bounds) and
// warp_kernels/neighbors_cell.py:17-153 (27-cell sweep; the reference is orthorhombic-only, fractional cells extend it to triclinic).  Same pair set as the
// brute-force search (the reference's tests assert that, tests/test_neighbors.py:74-148).
//
// MI355X design: atoms are RENUMBERED in cell order (stable radix sort of (cell id, atom index), rocPRIM) and the
// whole model then runs on the spatially sorted atoms - every neighbour of an atom lives in at most 27
// contiguous index ranges, so (a) the wave-per-atom sweep reads candidate positions coalesced, (b) visiting the
// ranges in ascending cell id yields rows already sorted by neighbour index (no per-row sort, no atomics,
// deterministic), and (c) all later feature gathers of the path hit nearby memory.  Forces are scattered back
// through the permutation by the final force kernel.
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "tn_common.h"
#include "tn_kernels.h"

namespace tn {

static inline int cdivc(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

size_t cell_sort_temp_bytes(int64_t n) {
  // the size query walks rocPRIM's tuning tables (milliseconds on the host): ask once per problem size
  static thread_local int64_t last_n = -1;
  static thread_local size_t last_bytes = 0;
  if (n == last_n) return last_bytes;
  last_n = n;
  size_t& bytes = last_bytes;
  bytes = 0;
  int* k = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, k, k, k, k, (size_t)(n > 0 ? n : 1), 0, 32, (hipStream_t)0);
  return bytes;
}

// cell id of every atom from its wrapped FRACTIONAL position; iota for the sort values.  Box rows a = (ax,0,0),
// b = (bx,by,0), c = (cx,cy,cz) (the reference's reduced form, torchmdnet/models/utils.py:206-229; orthorhombic when
// the off-diagonals vanish): r = sa a + sb b + sc c is solved back to front.
__global__ void k_cell_assign(const float* __restrict__ pos, const float* __restrict__ box, int N, int ncx, int ncy, int ncz,
                              int* __restrict__ key, int* __restrict__ iota) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const float x = pos[i * 3], y = pos[i * 3 + 1], zc = pos[i * 3 + 2];
  float fz = zc / box[8];
  float fy = (y - fz * box[7]) / box[4];
  float fx = (x - fy * box[3] - fz * box[6]) / box[0];
  fx -= floorf(fx);
  fy -= floorf(fy);
  fz -= floorf(fz);
  int cx = min((int)(fx * ncx), ncx - 1), cy = min((int)(fy * ncy), ncy - 1), cz = min((int)(fz * ncz), ncz - 1);
  key[i] = (cx * ncy + cy) * ncz + cz;
  iota[i] = i;
}

// cell_start[c] = first sorted position whose cell id >= c (lower bound); c in [0, ncells]
__global__ void k_cell_bounds(const int* __restrict__ sorted_key, int N, int ncells, int* __restrict__ cell_start) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c > ncells) return;
  int lo = 0, hi = N;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (sorted_key[mid] < c) lo = mid + 1; else hi = mid;
  }
  cell_start[c] = lo;
}

__global__ void k_permute_pos(const float* __restrict__ pos, const int* __restrict__ perm, int N, float* __restrict__ pos_s) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int o = perm[i];
  pos_s[i * 3] = pos[o * 3];
  pos_s[i * 3 + 1] = pos[o * 3 + 1];
  pos_s[i * 3 + 2] = pos[o * 3 + 2];
}
__global__ void k_permute_z(const int64_t* __restrict__ z, const int* __restrict__ perm, int N, int64_t* __restrict__ z_s) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) z_s[i] = z[perm[i]];
}
void launch_permute_z(const Graph& g, const int64_t* z, int N, hipStream_t s) {
  hipLaunchKernelGGL(k_permute_z, dim3(cdivc(N, 256)), dim3(256), 0, s, z, g.perm, N, g.z_s);
}

// minimum image, z -> y -> x (reference neighbors_brute.py / models/utils.py:206-229); same operation order as the brute-force
// kernels (tn_graph_wave.hip) so that both strategies produce bit-identical deltas
__device__ __forceinline__ float cell_d2(const float* __restrict__ pos, int hi, int lo, const float* __restrict__ box, float& dx,
                                         float& dy, float& dz) {
  dx = pos[hi * 3 + 0] - pos[lo * 3 + 0];
  dy = pos[hi * 3 + 1] - pos[lo * 3 + 1];
  dz = pos[hi * 3 + 2] - pos[lo * 3 + 2];
  const float s3 = roundf(dz / box[8]);
  dx -= s3 * box[6]; dy -= s3 * box[7]; dz -= s3 * box[8];
  const float s2 = roundf(dy / box[4]);
  dx -= s2 * box[3]; dy -= s2 * box[4];
  const float s1 = roundf(dx / box[0]);
  dx -= s1 * box[0];
  return dx * dx + dy * dy + dz * dz;
}

// wave per (cell-sorted) atom: sort the 27 neighbour cell ids, sweep their atom ranges in ascending order
template <bool FILL>
__global__ __launch_bounds__(256) void k_nbr_cell(Graph g, int N, float lo2, float up2) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= N) return;
  if (FILL && g.counts[2]) return;
  const float* __restrict__ pos = g.pos_s;
  const float* __restrict__ box = g.boxd;
  const int ncx = g.ncx, ncy = g.ncy, ncz = g.ncz;
  const int ci = g.cell_key_sorted[i];
  const int cz = ci % ncz, cy = (ci / ncz) % ncy, cx = ci / (ncz * ncy);
  // lanes 0..26 -> neighbour cell id (periodic wrap); others -> +inf; 32-lane bitonic sort ascending
  int nid = 0x7fffffff;
  if (lane < 27) {
    const int dx = lane / 9 - 1, dy = (lane / 3) % 3 - 1, dz = lane % 3 - 1;
    const int x = (cx + dx + ncx) % ncx, y = (cy + dy + ncy) % ncy, zc = (cz + dz + ncz) % ncz;
    nid = (x * ncy + y) * ncz + zc;
  }
#pragma unroll
  for (int k = 2; k <= 32; k <<= 1)
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      const int other = __shfl_xor(nid, j, 64);
      const bool up = ((lane & k) == 0);
      const bool lower = ((lane & j) == 0);
      nid = (lower == up) ? min(nid, other) : max(nid, other);
    }
  const int P = FILL ? g.counts[0] : 0;
  int e = FILL ? g.rowptr[i] : 0, p = FILL ? g.pairptr[i] : 0;
  int nl = 0, nt = 0;
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  int t = 0;
  while (t < 27) {
    // merge consecutive cell ids into one contiguous atom range
    int c0 = __shfl(nid, t, 64), c1 = c0;
    ++t;
    while (t < 27) {
      const int cn = __shfl(nid, t, 64);
      if (cn != c1 + 1) break;
      c1 = cn;
      ++t;
    }
    const int j0 = g.cell_start[c0], j1 = g.cell_start[c1 + 1];
    for (int base = j0; base < j1; base += 64) {
      const int j = base + lane;
      bool hit = false, self = false;
      float dx = 0.f, dy = 0.f, dz = 0.f, d2 = 0.f;
      if (j < j1) {
        if (j == i) {
          hit = true;
          self = true;
        } else {
          d2 = (j < i) ? cell_d2(pos, i, j, box, dx, dy, dz) : cell_d2(pos, j, i, box, dx, dy, dz);
          hit = d2 < up2 && d2 >= lo2;
        }
      }
      const unsigned long long mall = __ballot(hit);
      const unsigned long long mlow = __ballot(hit && j < i);
      if (FILL) {
        if (hit) {
          const int ee = e + __popcll(mall & lt);
          g.col[ee] = j;
          if (self) {
            g.epair[ee] = P;
            g.esign[ee] = 0.f;
          } else if (j < i) {
            const int pp = p + __popcll(mlow & lt);
            const float d = sqrtf(d2);
            const float inv = d > 0.f ? 1.0f / d : 0.f;
            g.pair_i[pp] = i;
            g.pair_j[pp] = j;
            g.pd[pp] = d;
            g.pdelta[pp * 3 + 0] = dx; g.pdelta[pp * 3 + 1] = dy; g.pdelta[pp * 3 + 2] = dz;
            g.prhat[pp * 3 + 0] = dx * inv; g.prhat[pp * 3 + 1] = dy * inv; g.prhat[pp * 3 + 2] = dz * inv;
            g.epair[ee] = pp;
            g.esign[ee] = 1.f;
          } else {
            g.epair[ee] = -1;  // linked by k_nbr_link_wave
            g.esign[ee] = -1.f;
          }
        }
        e += __popcll(mall);
        p += __popcll(mlow);
      } else {
        nt += __popcll(mall);
        nl += __popcll(mlow);
      }
    }
  }
  if (!FILL && lane == 0) {
    g.nlow[i] = nl;
    g.ntot[i] = nt;
  }
  if (FILL && i == 0 && lane == 0) g.pd[P] = 0.f;
}

__global__ void k_set_boxd(const float* __restrict__ box, float* boxd, int* mstart, int* mend, int N) {
  if (threadIdx.x == 0) {
    for (int k = 0; k < 9; ++k) boxd[k] = box[k];
    mstart[0] = 0;  // one molecule: the per-molecule energy sum runs over all atoms
    mend[0] = N;
  }
}

// phase 1: bin + stable sort + permute positions + count ; phase 2: fill (link: launch_nbr_link_wave)
void launch_cell_phase1(const Graph& g, const float* pos, const float* box, int N, float lo, float up, hipStream_t s) {
  const int ncells = g.ncx * g.ncy * g.ncz;
  hipLaunchKernelGGL(k_set_boxd, dim3(1), dim3(64), 0, s, box, g.boxd, g.mstart, g.mend, N);
  hipLaunchKernelGGL(k_cell_assign, dim3(cdivc(N, 256)), dim3(256), 0, s, pos, box, N, g.ncx, g.ncy, g.ncz, g.cell_key, g.iota);
  int bits = 1;
  while ((1 << bits) < ncells && bits < 31) ++bits;
  size_t tmp = g.sort_tmp_bytes;
  (void)rocprim::radix_sort_pairs(g.sort_tmp, tmp, g.cell_key, g.cell_key_sorted, g.iota, g.perm, (size_t)N, 0, (unsigned)bits, s);
  hipLaunchKernelGGL(k_cell_bounds, dim3(cdivc(ncells + 1, 256)), dim3(256), 0, s, g.cell_key_sorted, N, ncells, g.cell_start);
  hipLaunchKernelGGL(k_permute_pos, dim3(cdivc(N, 256)), dim3(256), 0, s, pos, g.perm, N, g.pos_s);
  hipLaunchKernelGGL(k_nbr_cell<false>, dim3(cdivc(N, 4)), dim3(256), 0, s, g, N, lo * lo, up * up);
}
void launch_cell_phase2(const Graph& g, int N, float lo, float up, hipStream_t s) {
  hipLaunchKernelGGL(k_nbr_cell<true>, dim3(cdivc(N, 4)), dim3(256), 0, s, g, N, lo * lo, up * up);
}

}  // namespace tn
