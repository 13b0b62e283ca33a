// O(N) cell-list neighbour graph for one large periodic system (orthorhombic or reduced triclinic box), gfx950.
//
// Reference: torchmdnet/extensions/neighbor_utils.py:89-150 (build_cell_list: wrap, bin, sort, cell bounds) and
// warp_kernels/neighbors_cell.py:17-153 (27-cell sweep; the reference is orthorhombic-only, fractional cells extend it to triclinic).  Same pair set as the
// brute-force search (the reference's tests assert that, tests/test_neighbors.py:74-148).
//
// MI355X design: atoms are RENUMBERED in cell order (counting sort by cell id, atoms of a cell in ascending original index:
// histogram -> single-block scan -> placement -> per-cell rank sort, all in this file; no vendor primitive) and the
// whole model then runs on the spatially sorted atoms - every neighbour of an atom lives in at most 27
// contiguous index ranges, so (a) the wave-per-atom sweep reads candidate positions coalesced, (b) visiting the
// ranges in ascending cell id yields rows already sorted by neighbour index (no per-row sort, no atomics,
// deterministic), and (c) all later feature gathers of the path hit nearby memory.  Forces are scattered back
// through the permutation by the final force kernel.
#include <cstring>

#include <hip/hip_runtime.h>

#include "tn_common.h"
#include "tn_kernels.h"

namespace tn {

static inline int cdivc(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// scratch of the counting sort: the per-cell cursors (cells <= 8 N + 1 by construction of the grid, launch_cell_phase1)
size_t cell_sort_temp_bytes(int64_t n) { return (size_t)(8 * (n > 0 ? n : 1) + 2) * sizeof(int); }

// cell id of every atom from its wrapped FRACTIONAL position; iota for the sort values.  Box rows a = (ax,0,0),
// b = (bx,by,0), c = (cx,cy,cz) (the reference's reduced form, torchmdnet/models/utils.py:206-229; orthorhombic when
// the off-diagonals vanish): r = sa a + sb b + sc c is solved back to front.
__global__ void k_cell_assign(const float* __restrict__ pos, const float* __restrict__ box, int N, const int* __restrict__ cgrid,
                              int* __restrict__ key, int* __restrict__ hist) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int ncx = cgrid[0], ncy = cgrid[1], ncz = cgrid[2];
  const float x = pos[i * 3], y = pos[i * 3 + 1], zc = pos[i * 3 + 2];
  float fz = zc / box[8];
  float fy = (y - fz * box[7]) / box[4];
  float fx = (x - fy * box[3] - fz * box[6]) / box[0];
  fx -= floorf(fx);
  fy -= floorf(fy);
  fz -= floorf(fz);
  int cx = min((int)(fx * ncx), ncx - 1), cy = min((int)(fy * ncy), ncy - 1), cz = min((int)(fz * ncz), ncz - 1);
  const int c = (cx * ncy + cy) * ncz + cz;
  key[i] = c;
  atomicAdd(&hist[c], 1);  // integer counts: the result does not depend on the order of the adds
}

// (1) zero the histogram of the live cells (grid sized by the 8 N capacity; the cell count is a device value)
__global__ void k_fill_cells(const int* __restrict__ cgrid, int* __restrict__ hist) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c <= cgrid[3]) hist[c] = 0;
}
// Counting sort, steps 2-4.  (2) exclusive scan of the cell histogram by ONE block (cells ~ N / 12 at liquid density, far
// below the 8 N capacity): cell_start[c] = first sorted position of cell c, cell_start[ncells] = N; the cursors start there.
__global__ __launch_bounds__(1024) void k_cell_scan(const int* __restrict__ cgrid, int* __restrict__ hist_cursor,
                                                   int* __restrict__ cell_start, int N) {
  __shared__ int wsum[16];
  __shared__ int carry_s;
  const int ncells = cgrid[3];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < ncells; base += 1024) {
    const int c = base + tid;
    const int v = c < ncells ? hist_cursor[c] : 0;
    int x = v;  // inclusive scan inside the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int y = __shfl_up(x, off, 64);
      if (lane >= off) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    const int excl = carry_s + woff + x - v;
    if (c < ncells) {
      cell_start[c] = excl;
      hist_cursor[c] = excl;
    }
    __syncthreads();
    if (tid == 1023) carry_s = excl + v;
    __syncthreads();
  }
  if (tid == 0) cell_start[ncells] = N;
}
// (3) placement through the cursors: the order inside a cell is whatever the atomics give ...
__global__ void k_cell_place(const int* __restrict__ key, int N, int* __restrict__ cursor, int* __restrict__ slot_atom) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  slot_atom[atomicAdd(&cursor[key[i]], 1)] = i;
}
// (4) ... and a wave per cell restores ascending original index (rank of each member among the cell's members: cells hold
// ~12 atoms, so the O(n^2 / 64) ranking is a handful of steps).  The permutation therefore equals a stable sort by cell id:
// deterministic, independent of the atomics' order.
__global__ __launch_bounds__(256) void k_cell_rank(const int* __restrict__ cgrid, const int* __restrict__ cell_start,
                                                   const int* __restrict__ slot_atom, int* __restrict__ perm,
                                                   int* __restrict__ key_sorted) {
  const int ncells = cgrid[3];
  const int lane = threadIdx.x & 63;
  for (int c = blockIdx.x * 4 + (threadIdx.x >> 6); c < ncells; c += gridDim.x * 4) {
    const int s0 = cell_start[c], n = cell_start[c + 1] - s0;
    for (int a0 = 0; a0 < n; a0 += 64) {
      const int a = a0 + lane;
      const int mine = a < n ? slot_atom[s0 + a] : 0x7fffffff;
      int rank = 0;
      for (int b0 = 0; b0 < n; b0 += 64) {
        const int other = b0 + lane < n ? slot_atom[s0 + b0 + lane] : 0x7fffffff;
        const int m = n - b0 < 64 ? n - b0 : 64;
        for (int k = 0; k < m; ++k) rank += __shfl(other, k, 64) < mine ? 1 : 0;
      }
      if (a < n) {
        perm[s0 + rank] = mine;
        key_sorted[s0 + rank] = c;
      }
    }
  }
}

__global__ void k_permute_pos(const float* __restrict__ pos, const int64_t* __restrict__ batch, const int* __restrict__ perm, int N,
                              int B, float* __restrict__ pos_s, int* __restrict__ bat_s, int64_t* __restrict__ bat_c,
                              int* __restrict__ counts) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int o = perm[i];
  pos_s[i * 3] = pos[o * 3];
  pos_s[i * 3 + 1] = pos[o * 3 + 1];
  pos_s[i * 3 + 2] = pos[o * 3 + 2];
  if (batch) {  // several molecules in one box: pairs stay inside a molecule; the molecules are interleaved in cell order, so
    int64_t b = batch[o];  // the per-molecule kernels take their "unsorted batch" path (counts[3])
    if (B > 0 && (b < 0 || b >= B)) {
      counts[5] = 1;
      b = b < 0 ? 0 : B - 1;
    }
    bat_s[i] = (int)b;
    bat_c[i] = b;
    if (i == 0) counts[3] = 1;
  }
}
__global__ void k_permute_z(const int64_t* __restrict__ z, const int* __restrict__ perm, int N, int64_t* __restrict__ z_s) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) z_s[i] = z[perm[i]];
}
void launch_permute_z(const Graph& g, const int64_t* z, int N, hipStream_t s) {
  hipLaunchKernelGGL(k_permute_z, dim3(cdivc(N, 256)), dim3(256), 0, s, z, g.perm, N, g.z_s);
}
__global__ void k_prepare_z(const int64_t* __restrict__ z, const int* __restrict__ perm, int N, int max_z, int64_t* __restrict__ z_c,
                            int* __restrict__ counts) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  int64_t v = z[perm ? perm[i] : i];
  if (v < 0 || v >= max_z) {
    counts[4] = 1;
    v = v < 0 ? 0 : max_z - 1;
  }
  z_c[i] = v;
}
void launch_prepare_z(const Graph& g, const int64_t* z, const int* perm, int N, int max_z, hipStream_t s) {
  if (N <= 0) return;
  hipLaunchKernelGGL(k_prepare_z, dim3(cdivc(N, 256)), dim3(256), 0, s, z, perm, N, max_z, g.z_c, g.counts);
}

// minimum image, z -> y -> x (reference neighbors_brute.py / models/utils.py:206-229); same operation order as the brute-force
// kernels (tn_graph_wave.hip) so that both strategies produce bit-identical deltas
__device__ __forceinline__ float cell_d2(const float* __restrict__ pos, int hi, int lo, const float* __restrict__ box, float& dx,
                                         float& dy, float& dz) {
  return pair_geometry(pos, hi, lo, box, dx, dy, dz);  // tn_common.h: the one definition
}

// wave per (cell-sorted) atom: sort the 27 neighbour cell ids, sweep their atom ranges in ascending order.
// With fewer than 3 cells along an axis several of the 27 offsets name the same cell: duplicates are adjacent after
// the sort and are skipped, so any grid (down to 1 x 1 x 1 = brute force) gives the same pair set.
template <bool FILL>
__global__ __launch_bounds__(256) void k_nbr_cell(Graph g, int N, float lo2, float up2, int loop) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= N) return;
  if (FILL && g.counts[2]) return;
  const float* __restrict__ pos = g.pos_s;
  const float* __restrict__ box = g.boxd;
  const int* __restrict__ bat = g.use_cell > 1 ? g.bat_s : nullptr;  // > 1: several molecules share the box
  const int ncx = g.cgrid[0], ncy = g.cgrid[1], ncz = g.cgrid[2];
  const int ci = g.cell_key_sorted[i];
  const int bi = bat ? bat[i] : 0;
  // halo exchange: nothing of a ghost's own neighbourhood is used (its rows come from its owner), so a ghost keeps its pairs with
  // owned atoms only
  const float* __restrict__ gw = g.ghost_w;
  const bool ghost_i = gw && gw[g.perm[i]] == 0.f;
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
    // merge consecutive (and repeated) cell ids into one contiguous atom range
    int c0 = __shfl(nid, t, 64), c1 = c0;
    ++t;
    while (t < 27) {
      const int cn = __shfl(nid, t, 64);
      if (cn != c1 && cn != c1 + 1) break;
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
          hit = loop != 0;
          self = true;
        } else if ((!bat || bat[j] == bi) && !(ghost_i && gw[g.perm[j]] == 0.f)) {
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

// One block: device copy of the box and the cell grid.
//   periodic (box != null): grid n_axis = floor(w_axis / rc), w = perpendicular width of the reduced lower-triangular box
//     a = (ax,0,0), b = (bx,by,0), c = (cx,cy,cz) (reference get_cell_dimensions, extensions/neighbor_utils.py:76-86, for the
//     orthorhombic case), recomputed from the CURRENT box on every call: no host copy, valid under graph replay when the
//     box changes; an explicit grid (tmdnet_set_cell_grid) overrides it.
//   non-periodic (box == null; neighbour operator only): a fictitious orthorhombic box around the bounding box of the
//     positions, edge = extent + 1.01 rc (>= 3 rc), so that no periodic image is ever within the cutoff (the reference
//     uses a fixed 3 rc box and clamps, models/utils.py:206-212).
// Every axis is clamped to [1, ncap] with ncap^3 <= 8 N (the capacity of cell_start).
__global__ __launch_bounds__(1024) void k_cell_setup(const float* __restrict__ box, const float* __restrict__ pos, int N, float rc,
                                                    int ncap, int ex, int ey, int ez, float* __restrict__ boxd,
                                                    int* __restrict__ cgrid, int* mstart, int* mend, int one_mol) {
  __shared__ float red[6][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float w[3];
  if (box) {
    if (tid == 0) {
      for (int k = 0; k < 9; ++k) boxd[k] = box[k];
      const float ax = box[0], bx = box[3], by = box[4], cx = box[6], cy = box[7], cz = box[8];
      const float vol = fabsf(ax * by * cz);
      const float bcx = by * cz, bcy = -bx * cz, bcz = bx * cy - by * cx;  // b x c
      w[0] = vol / sqrtf(bcx * bcx + bcy * bcy + bcz * bcz);
      w[1] = vol / (fabsf(ax) * sqrtf(cz * cz + cy * cy));                // |c x a| = |ax| sqrt(cz^2 + cy^2)
      w[2] = fabsf(cz);                                                  // |a x b| = |ax by|
    }
  } else {
    float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int i = tid; i < N; i += 1024)
      for (int a = 0; a < 3; ++a) {
        const float v = pos[i * 3 + a];
        mn[a] = fminf(mn[a], v);
        mx[a] = fmaxf(mx[a], v);
      }
    for (int a = 0; a < 3; ++a) {
      for (int off = 32; off >= 1; off >>= 1) {
        mn[a] = fminf(mn[a], __shfl_xor(mn[a], off, 64));
        mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off, 64));
      }
      if (lane == 0) {
        red[a][wave] = mn[a];
        red[3 + a][wave] = mx[a];
      }
    }
    __syncthreads();
    if (tid == 0) {
      for (int k = 0; k < 9; ++k) boxd[k] = 0.f;
      for (int a = 0; a < 3; ++a) {
        float lo = red[a][0], hi = red[3 + a][0];
        for (int k = 1; k < 16; ++k) {
          lo = fminf(lo, red[a][k]);
          hi = fmaxf(hi, red[3 + a][k]);
        }
        const float ext = N > 0 ? hi - lo : 0.f;
        w[a] = fmaxf(ext + 1.01f * rc + 1e-3f * ext, 3.0f * rc);
        boxd[a * 4] = w[a];
      }
    }
  }
  if (tid == 0) {
    const int ex3[3] = {ex, ey, ez};
    int n[3];
    for (int a = 0; a < 3; ++a) {
      int v = ex3[a] > 0 ? ex3[a] : (int)floorf(w[a] / rc);
      n[a] = v < 1 ? 1 : (v > ncap ? ncap : v);
      cgrid[a] = n[a];
    }
    cgrid[3] = n[0] * n[1] * n[2];
    if (one_mol) {
      mstart[0] = 0;  // one molecule: the per-molecule energy sum runs over all atoms
      mend[0] = N;
    }
  }
}

// phase 1: bin + stable sort + permute positions + count ; phase 2: fill (link: launch_nbr_link_wave)
void launch_cell_phase1(const Graph& g, const float* pos, const int64_t* batch, const float* box, int N, float lo, float up, bool loop,
                        hipStream_t s, int B) {
  int ncap = 1;
  while ((int64_t)(ncap + 1) * (ncap + 1) * (ncap + 1) <= 8 * (int64_t)N) ++ncap;  // cell_start holds 8 N + 2 entries
  hipLaunchKernelGGL(k_cell_setup, dim3(1), dim3(1024), 0, s, box, pos, N, up, ncap, g.ncx, g.ncy, g.ncz, g.boxd, g.cgrid, g.mstart,
                     g.mend, g.use_cell == 1 ? 1 : 0);
  int* cursor = reinterpret_cast<int*>(g.sort_tmp);  // [8 N + 2]: histogram, then the placement cursors
  hipLaunchKernelGGL(k_fill_cells, dim3(cdivc(8 * (int64_t)N + 2, 1024)), dim3(1024), 0, s, g.cgrid, cursor);
  hipLaunchKernelGGL(k_cell_assign, dim3(cdivc(N, 256)), dim3(256), 0, s, pos, g.boxd, N, g.cgrid, g.cell_key, cursor);
  hipLaunchKernelGGL(k_cell_scan, dim3(1), dim3(1024), 0, s, g.cgrid, cursor, g.cell_start, N);
  hipLaunchKernelGGL(k_cell_place, dim3(cdivc(N, 256)), dim3(256), 0, s, g.cell_key, N, cursor, g.iota);
  hipLaunchKernelGGL(k_cell_rank, dim3(cdivc(N > 4096 ? N / 8 : 512, 4)), dim3(256), 0, s, g.cgrid, g.cell_start, g.iota, g.perm,
                     g.cell_key_sorted);
  hipLaunchKernelGGL(k_permute_pos, dim3(cdivc(N, 256)), dim3(256), 0, s, pos, g.use_cell > 1 ? batch : nullptr, g.perm, N, B, g.pos_s,
                     g.bat_s, g.bat_c, g.counts);
  hipLaunchKernelGGL(k_nbr_cell<false>, dim3(cdivc(N, 4)), dim3(256), 0, s, g, N, lo * lo, up * up, (int)loop);
}
void launch_cell_phase2(const Graph& g, int N, float lo, float up, bool loop, hipStream_t s) {
  hipLaunchKernelGGL(k_nbr_cell<true>, dim3(cdivc(N, 4)), dim3(256), 0, s, g, N, lo * lo, up * up, (int)loop);
}

}  // namespace tn
