// Wave-per-atom neighbour-graph kernels (gfx950, wave64).
//
// One 64-lane wave owns one atom; lanes sweep the candidate range 64 atoms at a time (coalesced position
// loads), `__ballot` + popcount give the counts and the ordered slot of every hit, so rows come out sorted
// by neighbour index without atomics.  Replaces the thread-per-atom versions of tn_kernels.hip (kept as the
// readable specification): same pair set, same order, ~10x lower latency for single molecules and coalesced
// O(N^2/64) wave-passes for one large system.  Semantics: reference warp_kernels/neighbors_brute.py:98-197.
#include <cstdlib>

#include "tn_common.h"
#include "tn_kernels.h"

namespace tn {

static inline int cdivw(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

__device__ __forceinline__ float pair_d2(const float* __restrict__ pos, int hi, int lo, const float* __restrict__ box, float& dx,
                                         float& dy, float& dz) {
  return pair_geometry(pos, hi, lo, box, dx, dy, dz);  // tn_common.h: the one definition
}

// one atom's row, by one wave: count pass (FILL = false: nlow, ntot) or fill pass (FILL = true: col / epair / esign and the
// pair records of its lower neighbours)
template <bool FILL>
__device__ __forceinline__ void nbr_wave_row(const Graph& g, const float* __restrict__ pos, const int64_t* __restrict__ batch,
                                             const float* __restrict__ box, int box_mode, int N, int B, float lo2, float up2, int loop,
                                             int i, int lane) {
  int64_t b = batch[i];
  int j0 = 0, j1 = N;
  if (b < 0 || b >= B) {  // invalid molecule index (counts[5] is set, the host raises): no candidates, no box read
    b = 0;
    j1 = 0;
  } else if (!g.counts[3]) {
    j0 = g.mstart[b];
    j1 = g.mend[b];
  }
  const float* bx = box_mode == 0 ? nullptr : (box_mode == 1 ? box : box + b * 9);
  const int P = FILL ? g.counts[0] : 0;
  int e = FILL ? g.rowptr[i] : 0, p = FILL ? g.pairptr[i] : 0;
  int nl = 0, nt = 0;
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  for (int base = j0; base < j1; base += 64) {
    const int j = base + lane;
    bool hit = false, self = false;
    float dx = 0.f, dy = 0.f, dz = 0.f, d2 = 0.f;
    if (j < j1) {
      if (j == i) {
        hit = loop != 0;
        self = true;
      } else if (batch[j] == b) {
        d2 = (j < i) ? pair_d2(pos, i, j, bx, dx, dy, dz) : pair_d2(pos, j, i, bx, dx, dy, dz);
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
  if (!FILL && lane == 0) {
    g.nlow[i] = nl;
    g.ntot[i] = nt;
  }
  if (FILL && i == 0 && lane == 0) g.pd[P] = 0.f;  // the self pair
}

template <bool FILL>
__global__ __launch_bounds__(256) void k_nbr_wave(Graph g, const float* __restrict__ pos, const int64_t* __restrict__ batch,
                                                  const float* __restrict__ box, int box_mode, int N, int B, float lo2, float up2,
                                                  int loop) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= N) return;
  if (FILL && g.counts[2]) return;
  nbr_wave_row<FILL>(g, pos, batch, box, box_mode, N, B, lo2, up2, loop, i, lane);
}

// upper edges (i <- j, j > i) take the pair id of the lower edge (j <- i): lanes over the row's edges, binary
// search of i among the sorted lower neighbours of j
__device__ __forceinline__ void nbr_link_row(const Graph& g, int i, int lane) {
  const int e0 = g.rowptr[i], e1 = g.rowptr[i + 1];
  for (int e = e0 + lane; e < e1; e += 64) {
    const int j = g.col[e];
    if (j <= i) continue;
    int lo = g.rowptr[j], hi = lo + g.nlow[j] - 1;
    const int base = lo;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (g.col[mid] < i) lo = mid + 1; else hi = mid;
    }
    g.epair[e] = g.pairptr[j] + (lo - base);
  }
}
__global__ __launch_bounds__(256) void k_nbr_link_wave(Graph g, int N) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= N || g.counts[2]) return;
  nbr_link_row(g, i, lane);
}

// ---- small systems (single molecules, MD stepping): the whole graph phase in ONE launch.  The seven kernels of the
// general path (reset, molecule ranges, count, scan, fill, link, z check) each take microseconds of work but ~5 us of
// launch-to-launch latency inside a replayed HIP graph; one block of 16 waves runs the same phases separated by block
// barriers (the arrays are written and read through global memory by the same CU: a barrier orders them).  Same row
// functions as the general kernels: identical graph.
constexpr int GS_MAX_ATOMS = 256;
constexpr int GS_ECAP = 16384;  // adjacency entries kept in LDS (64 KB): 256 atoms x 64 neighbours
__global__ __launch_bounds__(1024) void k_graph_small(Graph g, const float* __restrict__ pos, const int64_t* __restrict__ batch,
                                                      const float* __restrict__ box, int box_mode, int N, int B, float lo2, float up2,
                                                      int loop, const int64_t* __restrict__ z, int max_z) {
  // Everything the phases hand to each other lives in LDS (positions, molecule indices and ranges, per-row counts, the two
  // prefix sums, the flags): the row functions reach it through the same Graph fields, re-pointed at the LDS copies (flat
  // addressing), so no phase starts with a round trip to L2.  The global copies are written once at the end.
  __shared__ int s_mstart[GS_MAX_ATOMS], s_mend[GS_MAX_ATOMS], s_nlow[GS_MAX_ATOMS], s_ntot[GS_MAX_ATOMS];
  __shared__ int s_rowptr[GS_MAX_ATOMS + 1], s_pairptr[GS_MAX_ATOMS + 1], s_counts[8], c_low[4], c_tot[4];
  __shared__ float s_pos[3 * GS_MAX_ATOMS];
  __shared__ int64_t s_batch[GS_MAX_ATOMS];
  // the neighbour columns too (when the capacity fits): the link phase finds the pair id of every upper neighbour by a binary
  // search in the partner's row - five to six DEPENDENT reads per edge, which were round trips to L2 for data this block had
  // written a moment ago (in-kernel share: most of the kernel's 14 us at 64 atoms)
  __shared__ int s_col[GS_ECAP];
  const bool col_lds = g.ecap <= GS_ECAP;  // uniform
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool mol_lds = B <= GS_MAX_ATOMS;  // more molecule slots than atoms (empty molecules): ranges stay in global memory
  Graph gl = g;
  gl.nlow = s_nlow;
  gl.ntot = s_ntot;
  gl.rowptr = s_rowptr;
  gl.pairptr = s_pairptr;
  gl.counts = s_counts;
  if (mol_lds) {
    gl.mstart = s_mstart;
    gl.mend = s_mend;
  }
  if (col_lds) gl.col = s_col;
  for (int k = tid; k < 3 * N; k += 1024) s_pos[k] = pos[k];
  if (tid < N) s_batch[tid] = batch[tid];
  for (int i = tid; i < B; i += 1024) gl.mstart[i] = gl.mend[i] = 0;  // k_graph_reset
  if (tid < 8) s_counts[tid] = 0;
  __syncthreads();
  // molecule ranges (k_mol_ranges) and the atomic-number check (k_prepare_z)
  if (tid < N) {
    const int i = tid;
    const int64_t b = s_batch[i];
    if (b < 0 || b >= B) {
      s_counts[3] = 1;
      s_counts[5] = 1;
    } else {
      if (i > 0) {
        const int64_t bp = s_batch[i - 1];
        if (bp > b) s_counts[3] = 1;
        if (bp != b) gl.mstart[b] = i;
      } else {
        gl.mstart[b] = 0;
      }
      if (i == N - 1 || s_batch[i + 1] != b) gl.mend[b] = i + 1;
    }
    if (z) {
      int64_t v = z[i];
      if (v < 0 || v >= max_z) {
        s_counts[4] = 1;
        v = v < 0 ? 0 : max_z - 1;
      }
      g.z_c[i] = v;
    }
  }
  __syncthreads();
  // count
  for (int i = wave; i < N; i += 16) nbr_wave_row<false>(gl, s_pos, s_batch, box, box_mode, N, B, lo2, up2, loop, i, lane);
  __syncthreads();
  // exclusive scans nlow -> pairptr, ntot -> rowptr (k_scan_counts): four waves of 64 elements each, then the three carries
  {
    int a = 0, b = 0;
    if (tid < GS_MAX_ATOMS) {
      a = tid < N ? s_nlow[tid] : 0;
      b = tid < N ? s_ntot[tid] : 0;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int ta = __shfl_up(a, off, 64), tb = __shfl_up(b, off, 64);
        if (lane >= off) {
          a += ta;
          b += tb;
        }
      }
      if (lane == 63) {
        c_low[wave] = a;
        c_tot[wave] = b;
      }
    }
    __syncthreads();
    if (tid < GS_MAX_ATOMS) {
      for (int w = 0; w < wave; ++w) {
        a += c_low[w];
        b += c_tot[w];
      }
      if (tid < N) {  // inclusive -> exclusive: element tid + 1
        s_pairptr[tid + 1] = a;
        s_rowptr[tid + 1] = b;
      }
      if (tid == 0) s_pairptr[0] = s_rowptr[0] = 0;
    }
    __syncthreads();
  }
  if (tid == 0) {
    const int P = s_pairptr[N], E = s_rowptr[N];
    s_counts[0] = P;
    s_counts[1] = E;
    s_counts[2] = (E > g.ecap || P > g.pcap || s_counts[5]) ? 1 : 0;
  }
  __syncthreads();
  if (!s_counts[2]) {  // block-uniform
    // fill, then link
    for (int i = wave; i < N; i += 16) nbr_wave_row<true>(gl, s_pos, s_batch, box, box_mode, N, B, lo2, up2, loop, i, lane);
    __syncthreads();
    for (int i = wave; i < N; i += 16) nbr_link_row(gl, i, lane);
    if (col_lds) {
      __syncthreads();
      const int E = s_counts[1];
      for (int e = tid; e < E; e += 1024) g.col[e] = s_col[e];
    }
  }
  // the global copies the later kernels (and the host, for the flags) read
  if (tid <= N) {
    g.rowptr[tid] = s_rowptr[tid];
    g.pairptr[tid] = s_pairptr[tid];
  }
  if (tid < N) {
    g.nlow[tid] = s_nlow[tid];
    g.ntot[tid] = s_ntot[tid];
  }
  if (mol_lds && tid < B) {
    g.mstart[tid] = s_mstart[tid];
    g.mend[tid] = s_mend[tid];
  }
  if (tid < 8) g.counts[tid] = s_counts[tid];
}
bool graph_small_ok(int N) {
  static const bool off = getenv("TMDNET_NO_GRAPH_SMALL") != nullptr;  // developer switch: the general kernels
  return !off && N > 0 && N <= GS_MAX_ATOMS;
}
void launch_graph_small(const Graph& g, const float* pos, const int64_t* batch, const float* box, int box_mode, int N, int B, float lo,
                        float up, bool loop, const int64_t* z, int max_z, hipStream_t s) {
  hipLaunchKernelGGL(k_graph_small, dim3(1), dim3(1024), 0, s, g, pos, batch, box, box_mode, N, B, lo * lo, up * up, (int)loop, z, max_z);
}

void launch_nbr_count_wave(const Graph& g, const float* pos, const int64_t* batch, const float* box, int box_mode, int N, int B,
                           float lo, float up, bool loop, hipStream_t s) {
  hipLaunchKernelGGL(k_nbr_wave<false>, dim3(cdivw(N, 4)), dim3(256), 0, s, g, pos, batch, box, box_mode, N, B, lo * lo, up * up,
                     (int)loop);
}
void launch_nbr_fill_link_wave(const Graph& g, const float* pos, const int64_t* batch, const float* box, int box_mode, int N, int B,
                               float lo, float up, bool loop, hipStream_t s) {
  hipLaunchKernelGGL(k_nbr_wave<true>, dim3(cdivw(N, 4)), dim3(256), 0, s, g, pos, batch, box, box_mode, N, B, lo * lo, up * up,
                     (int)loop);
  hipLaunchKernelGGL(k_nbr_link_wave, dim3(cdivw(N, 4)), dim3(256), 0, s, g, N);
}
void launch_nbr_link_wave(const Graph& g, int N, hipStream_t s) {
  hipLaunchKernelGGL(k_nbr_link_wave, dim3(cdivw(N, 4)), dim3(256), 0, s, g, N);
}

}  // namespace tn
