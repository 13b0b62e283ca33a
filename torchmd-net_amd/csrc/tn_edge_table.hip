// Radial tables of the per-pair functions of TensorNet (gfx950).
//
// Every per-pair quantity of the model is a function of the pair DISTANCE alone:
//     Q_c(d)   = P_c phi(d) + b_c                      (distance projections of the embedding, reference tensornet.py:558-560)
//     w^l(d)   = silu(M3 silu(M2 silu(M1 phi(d)))) C(d)  (edge MLP of interaction layer l, reference tensornet.py:738-743)
// (phi = ExpNormal basis, C = cosine cutoff; the atom types enter later, in the scatter).  They are smooth maps
// [r_lo, r_c) -> R^{3F}, so instead of pushing P ~ 2e5 pair rows through K -> F -> 2F -> 3F GEMMs every step
// (83 % of the model's FLOPs), the functions and their d/dd are evaluated ONCE, when the parameters are
// uploaded, on a uniform grid of T + 1 distances - in DOUBLE precision (plain fp64 kernels below; it happens once) - and
// each step evaluates the cubic Hermite interpolant (value AND slope are tabulated: the interpolant is C1) per pair,
// written in the divided-difference form that has no cancellation,  D_k = (f_{k+1} - f_k) / h  (formed in fp64):
//     f(d)  ~ f_k + h [ D_k (3t^2 - 2t^3) + s_k (t^3 - 2t^2 + t) + s_{k+1} (t^3 - t^2) ],      d = r_lo + (k + t) h
//     f'(d) ~ D_k (6t - 6t^2) + s_k (3t^2 - 4t + 1) + s_{k+1} (3t^2 - 2t)                     (s = tabulated slope)
// (the second line is the exact derivative of the first: energies and forces stay consistent).  The truncation error is
// of fourth order in h: ~1e-9 relative at h = (r_c - r_lo) / 8192; what remains is the fp32 rounding of the stored
// entries, i.e. the tables are CLOSER to the exact functions than the direct fp32 GEMM chain (accumulated rounding
// ~4e-7).  The build VERIFIES it: every table is compared with the fp64 evaluation at all interval midpoints, value and
// slope, and the grid is refined or the tables are left off if the bounds are not met.  Weights-only precomputation,
// like the per-type tables U[z], V[z] (k_ztables) and the split-bf16 weight images: valid for any input, rebuilt with
// the weights.
//
// Per step: pair ids are bucketed by grid interval (counting sort: histogram, scan, scatter), so that consecutive
// pairs read the same or neighbouring table rows (38 MB per table: the gathers hit L2), and one streaming kernel writes
// Q, dQ/dd, w^l, dw^l/dd for all tables.  It is bound by those writes (HBM).  The self pair (d = 0, below r_lo when a
// lower cutoff is set) has its own exactly evaluated row.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "tn_common.h"
#include "tn_kernels.h"
#include "tn_interp.h"

namespace tn {

static inline int cdive(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---- build (fp64) ------------------------------------------------------------------------------------------------
// rows 0..T: d_k = lo + k h ; row T + 1: the self pair (d = 0) ; mid = 1: the T interval midpoints
__device__ __forceinline__ double grid_d(int k, double lo, double h, int T, int mid) {
  if (mid) return lo + ((double)k + 0.5) * h;
  return k <= T ? lo + (double)k * h : 0.0;
}
__device__ __forceinline__ void cosine_cutoff_f64(double d, double lo, double up, double& c, double& dc) {
  const double PI = 3.14159265358979323846;
  if (lo > 0.0) {
    const double k = 2.0 * PI / (up - lo), arg = PI * (2.0 * (d - lo) / (up - lo) + 1.0);
    const bool in = (d < up) && (d > lo);
    c = in ? 0.5 * (cos(arg) + 1.0) : 0.0;
    dc = in ? -0.5 * sin(arg) * k : 0.0;
  } else {
    const double k = PI / up;
    const bool in = d < up;
    c = in ? 0.5 * (cos(d * k) + 1.0) : 0.0;
    dc = in ? -0.5 * sin(d * k) * k : 0.0;
  }
}
// ExpNormal basis + cutoffs (reference models/utils.py:402-407, 506-528) in double
__global__ void k_radial_f64(int rows, double lo, double up, double h, int T, int mid, const float* __restrict__ means,
                             const float* __restrict__ betas, int K, double* __restrict__ dist, double* __restrict__ phi,
                             double* __restrict__ dphi, double* __restrict__ C, double* __restrict__ dC) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)rows * K) return;
  const int p = (int)(idx / K), k = (int)(idx - (int64_t)p * K);
  const double d = grid_d(p, lo, h, T, mid);
  double c0, dc0;
  cosine_cutoff_f64(d, 0.0, up, c0, dc0);
  const double alpha = 5.0 / (up - lo);
  const double u = exp(-alpha * (d - lo));
  const double mu = means[k], beta = betas[k];
  const double gk = exp(-beta * (u - mu) * (u - mu));
  phi[idx] = c0 * gk;
  dphi[idx] = dc0 * gk + c0 * gk * (-2.0 * beta * (u - mu)) * (-alpha * u);
  if (k == 0) {
    double c, dc;
    cosine_cutoff_f64(d, lo, up, c, dc);
    C[p] = c;
    dC[p] = dc;
    dist[p] = d;
  }
}
// one output per thread: e = b + A W^T, r = A2 W^T ; kind 0: (e, r) ; 1: (silu e, silu' e r) ; 2: (silu e rs, silu' e r rs + silu e rs2) ;
// 3: (e rs, r rs + e rs2)
__global__ void k_dense_f64(const double* __restrict__ A, const double* __restrict__ A2, int lda, const float* __restrict__ W,
                            const float* __restrict__ bias, int M, int N, int K, int kind, const double* __restrict__ rs,
                            const double* __restrict__ rs2, double* __restrict__ C, double* __restrict__ C2, int ldc) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)M * N) return;
  const int m = (int)(idx / N), n = (int)(idx - (int64_t)m * N);
  const double* a = A + (int64_t)m * lda;
  const double* a2 = A2 + (int64_t)m * lda;
  const float* w = W + (int64_t)n * K;
  double e = bias ? (double)bias[n] : 0.0, r = 0.0;
  for (int k = 0; k < K; ++k) {
    const double wk = w[k];
    e += a[k] * wk;
    r += a2[k] * wk;
  }
  if (kind == 3) {  // plain * row scale: (e rs, r rs + e rs2)
    const double e0 = e;
    e = e0 * rs[m];
    r = r * rs[m] + e0 * rs2[m];
  } else if (kind != 0) {
    const double sg = 1.0 / (1.0 + exp(-e));
    const double f = e * sg, df = sg * (1.0 + e * (1.0 - sg));
    if (kind == 1) {
      e = f;
      r = df * r;
    } else {
      e = f * rs[m];
      r = df * r * rs[m] + f * rs2[m];
    }
  }
  C[(int64_t)m * ldc + n] = e;
  C2[(int64_t)m * ldc + n] = r;
}
// table rows [T + 2][3][R] (fp32): value | slope | divided difference to the next grid point (formed in fp64)
__global__ void k_table_pack(const double* __restrict__ f, const double* __restrict__ sl, int T, int R, double h, float* __restrict__ tab) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)(T + 2) * R) return;
  const int k = (int)(idx / R), c = (int)(idx - (int64_t)k * R);
  const double v = f[idx];
  float* row = tab + (int64_t)k * 3 * R + c;
  row[0] = (float)v;
  row[R] = (float)sl[idx];
  // true divided difference (fp64), rounded once: the interpolant's slope is then accurate to fp32 rounding, and its value
  // is continuous across grid points to one ulp of the stored values
  row[2 * R] = k < T ? (float)((f[idx + R] - v) / h) : 0.f;
}
void launch_radial_f64(int rows, double lo, double up, int T, bool mid, const float* means, const float* betas, int K, double* dist,
                       double* phi, double* dphi, double* C, double* dC, hipStream_t s) {
  hipLaunchKernelGGL(k_radial_f64, dim3(cdive((int64_t)rows * K, 256)), dim3(256), 0, s, rows, lo, up, (up - lo) / (double)T, T,
                     mid ? 1 : 0, means, betas, K, dist, phi, dphi, C, dC);
}
void launch_dense_f64(const double* A, const double* A2, int lda, const float* W, const float* bias, int M, int N, int K, int kind,
                      const double* rs, const double* rs2, double* C, double* C2, int ldc, hipStream_t s) {
  hipLaunchKernelGGL(k_dense_f64, dim3(cdive((int64_t)M * N, 256)), dim3(256), 0, s, A, A2, lda, W, bias, M, N, K, kind, rs, rs2, C, C2,
                     ldc);
}
void launch_table_pack(const double* f, const double* sl, int T, int R, double h, float* tab, hipStream_t s) {
  hipLaunchKernelGGL(k_table_pack, dim3(cdive((int64_t)(T + 2) * R, 256)), dim3(256), 0, s, f, sl, T, R, h, tab);
}

// ---- per step ---------------------------------------------------------------------------------------------------
// Counting sort of the pairs by grid interval (T + 2 buckets: intervals 0..T-1, the self pair, "beyond the pair list"):
// histogram -> single-block scan -> scatter.  The order inside a bucket is whatever the atomics give; it only decides
// which thread writes which output row, never a value, so the results stay bit-reproducible.
__device__ __forceinline__ int interval_of(float d, float lo, float inv_h, int T) {
  const int k = (int)((d - lo) * inv_h);
  return k < 0 ? 0 : (k > T - 1 ? T - 1 : k);
}
// cutoff function per pair (the only radial quantity the rest of the step still needs) + bucket histogram
// Block-level aggregation of the bucket atomics (round 6).  Pairs that share a bucket are common - a water box has 20 000 O-H pairs at
// one bond length - and one global atomic per pair on the same counter serialises (10k-atom box: 80 + 83 us for the two kernels
// against 17 + 19 us for the 194k pairs of the molecule batch).  A block first counts its pairs per bucket in a direct-mapped LDS
// table (slot = bucket mod BK_SLOTS, claimed by compare-and-swap; a pair whose slot belongs to another bucket takes the global
// counter directly), then issues ONE global atomic per occupied slot.
constexpr int BK_SLOTS = 2048, BK_THREADS = 1024;
struct BucketAgg {
  int key[BK_SLOTS], cnt[BK_SLOTS], base[BK_SLOTS];
};
__device__ __forceinline__ void bucket_agg_init(BucketAgg& A) {
  for (int i = threadIdx.x; i < BK_SLOTS; i += blockDim.x) {
    A.key[i] = -1;
    A.cnt[i] = 0;
  }
  __syncthreads();
}
// rank of this thread's pair inside its block's share of `bucket` (>= 0), or -1: the slot is another bucket's (take the global counter)
__device__ __forceinline__ int bucket_agg_add(BucketAgg& A, int bucket) {
  const int sl = bucket & (BK_SLOTS - 1);
  const int prev = atomicCAS(&A.key[sl], -1, bucket);
  if (prev != -1 && prev != bucket) return -1;
  return atomicAdd(&A.cnt[sl], 1);
}

__global__ __launch_bounds__(BK_THREADS) void k_pair_cutoff_hist(Graph g, int Pcap, float lo, float up, float inv_h, int T,
                                                                float* __restrict__ C, float* __restrict__ dC, int* __restrict__ hist) {
  __shared__ BucketAgg A;
  bucket_agg_init(A);
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int P = g.counts[0];
  if (p <= Pcap && p <= P && !g.counts[2]) {  // (beyond the pair list: static shapes size the grid by the capacity)
    const float d = p < P ? g.pd[p] : 0.f;  // p == P: the self pair
    float c, dc;
    cosine_cutoff(d, lo, up, c, dc);
    C[p] = c;
    dC[p] = dc;
    const int b = p < P ? interval_of(d, lo, inv_h, T) : T;
    if (bucket_agg_add(A, b) < 0) atomicAdd(hist + b, 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < BK_SLOTS; i += blockDim.x)
    if (A.cnt[i] > 0) atomicAdd(hist + A.key[i], A.cnt[i]);
}
// exclusive scan of the T + 1 bucket counts in place (one block), total -> hist[T + 1]
__global__ __launch_bounds__(1024) void k_bucket_scan(int* __restrict__ hist, int nb) {
  // tiles of 8192 counts: a thread scans 8 consecutive ones, one block scan of the 1024 thread sums, a carry links the tiles
  // (one trip for the 8193 buckets of the default tables instead of nine trips of 1024: 11.5 -> about 5 us)
  constexpr int PER = 8, TILE = 1024 * PER;
  __shared__ int wsum[16];
  __shared__ int carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nb; base += TILE) {
    int v[PER], sum = 0;
#pragma unroll
    for (int r = 0; r < PER; ++r) {
      const int i = base + tid * PER + r;
      v[r] = i < nb ? hist[i] : 0;
      sum += v[r];
    }
    int inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int t = __shfl_up(inc, off, 64);
      if (lane >= off) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int pre = carry + inc - sum;
    for (int w = 0; w < wave; ++w) pre += wsum[w];
#pragma unroll
    for (int r = 0; r < PER; ++r) {
      const int i = base + tid * PER + r;
      if (i < nb) hist[i] = pre;
      pre += v[r];
    }
    __syncthreads();
    if (tid == 1023) carry = pre;
    __syncthreads();
  }
  if (tid == 0) hist[nb] = carry;
}
__global__ __launch_bounds__(BK_THREADS) void k_bucket_scatter(Graph g, int Pcap, float lo, float inv_h, int T, int* __restrict__ cursor,
                                                              unsigned* __restrict__ keys_s, int* __restrict__ vals_s) {
  __shared__ BucketAgg A;
  bucket_agg_init(A);
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int P = g.counts[0];
  const bool in_cap = p <= Pcap, valid = in_cap && p <= P && !g.counts[2];
  float d = 0.f;
  int b = 0, rank = -1, slot = -1;
  if (valid) {
    d = p < P ? g.pd[p] : 0.f;
    b = p < P ? interval_of(d, lo, inv_h, T) : T;
    rank = bucket_agg_add(A, b);
    if (rank < 0) slot = atomicAdd(cursor + b, 1);
  } else if (in_cap) {
    // slots past the valid pairs: marked so that the interpolation stops there (the P + 1 valid pairs fill the slots 0..P, so a
    // slot p > P is never one of theirs)
    keys_s[p] = 0xFFFFFFFFu;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < BK_SLOTS; i += blockDim.x)
    if (A.cnt[i] > 0) A.base[i] = atomicAdd(cursor + A.key[i], A.cnt[i]);
  __syncthreads();
  if (valid) {
    if (rank >= 0) slot = A.base[b & (BK_SLOTS - 1)] + rank;
    keys_s[slot] = __float_as_uint(d);
    vals_s[slot] = p;
  }
}

__global__ void k_fill_int(int* p, int v, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

struct InterpArgs {
  const float* tab[8];  // [T + 2][3][R]: value | slope | divided difference per grid row, row T + 1 = self pair
  float* out[8];        // [P + 1][R]
  float* dout[8];       // [P + 1][R] or null
  int ntab;
  int out_bf16;         // 1: out / dout are [P + 1][R] bf16 rows (round to nearest even), not fp32
};

// one 4-channel piece of an output row: 16 bytes of fp32, or 8 bytes of bf16 (the ET's pair rows in reduced-precision storage)
__device__ __forceinline__ void store_piece(float* base, int64_t elem, float x, float y, float z, float w, int bf16) {
  if (bf16) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    union { b2 v; uint32_t u; } lo, hi;
    lo.v = __builtin_convertvector((f2){x, y}, b2);
    hi.v = __builtin_convertvector((f2){z, w}, b2);
    *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(base) + elem) = make_uint2(lo.u, hi.u);
  } else {
    *reinterpret_cast<float4*>(base + elem) = make_float4(x, y, z, w);
  }
}

// Pairs in distance order.  A group of R4 = 3F/4 threads (one float4 column each) walks a RUN of consecutive sorted pairs:
// consecutive pairs fall into the same or the next grid interval (2e5 pairs over 8192 intervals), so the four table
// vectors of a column stay in registers until the interval changes - the table traffic drops from 4 rows per pair to 4
// rows per visited interval and the kernel is left with its output writes.
constexpr int EI_RUN = 8;  // sorted pairs per group (measured at C2: 1 -> 0.56, 2 -> 0.47, 4 -> 0.41, 8 -> 0.39, 16 -> 0.41, 32 -> 0.42, 64 -> 0.44 ms)
template <int NT>
__global__ __launch_bounds__(256) void k_edge_interp(Graph g, int Pcap, const unsigned* __restrict__ keys_sorted,
                                                    const int* __restrict__ vals_sorted, InterpArgs a, int R4, int T, float lo,
                                                    float h, float inv_h, int run) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int groups = blockDim.x / R4;
  const int grp = threadIdx.x / R4, c4 = threadIdx.x - grp * R4;
  if (grp >= groups) return;
  const int R = 4 * R4;
  const int P = g.counts[0];
  const int s0 = (blockIdx.x * groups + grp) * run;
  int kcur = -1;
  f4 f0[NT], sl0[NT], D[NT], sl1[NT];
  for (int s = s0; s < s0 + run && s <= Pcap; ++s) {
    const unsigned key = keys_sorted[s];
    if (key == 0xFFFFFFFFu) break;  // sorted: nothing valid follows
    const int p = vals_sorted[s];
    if (p == P) {  // self pair: its own exact row
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float* row = a.tab[t] + (int64_t)(T + 1) * 3 * R + 4 * c4;
        const f4 v0 = *reinterpret_cast<const f4*>(row);
        store_piece(a.out[t], (int64_t)p * R + 4 * c4, v0.x, v0.y, v0.z, v0.w, a.out_bf16);
        if (a.dout[t]) {
          const f4 v1 = *reinterpret_cast<const f4*>(row + R);
          store_piece(a.dout[t], (int64_t)p * R + 4 * c4, v1.x, v1.y, v1.z, v1.w, a.out_bf16);
        }
      }
      continue;
    }
    const float d = __uint_as_float(key);  // the key holds all 32 bits of the distance
    const InterpCoef ic = interp_coef(d, lo, h, inv_h, T);  // tn_interp.h: the one definition of this arithmetic
    const int k = ic.k;
    if (k != kcur) {
      kcur = k;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float* row = a.tab[t] + (int64_t)k * 3 * R + 4 * c4;
        f0[t] = *reinterpret_cast<const f4*>(row);
        sl0[t] = *reinterpret_cast<const f4*>(row + R);
        D[t] = *reinterpret_cast<const f4*>(row + 2 * R);
        sl1[t] = *reinterpret_cast<const f4*>(row + 4 * R);
      }
    }
#pragma unroll
    for (int tb = 0; tb < NT; ++tb) {
      f4 v0, v1;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v0[j] = interp_value(f0[tb][j], sl0[tb][j], D[tb][j], sl1[tb][j], ic);
        v1[j] = interp_slope(sl0[tb][j], D[tb][j], sl1[tb][j], ic);
      }
      store_piece(a.out[tb], (int64_t)p * R + 4 * c4, v0.x, v0.y, v0.z, v0.w, a.out_bf16);
      if (a.dout[tb]) store_piece(a.dout[tb], (int64_t)p * R + 4 * c4, v1.x, v1.y, v1.z, v1.w, a.out_bf16);
    }
  }
}

// Few pairs (single molecules, MD stepping): the pair-order machinery above costs five launches and a serial walk of 32
// pairs per thread group for nothing - the whole pair list is smaller than one table.  One launch instead: a thread group
// per pair computes the cutoff, finds its grid interval and reads its four table rows (L2).  Same arithmetic on the same
// distance bits as k_edge_interp.
template <int NT>
__global__ __launch_bounds__(256) void k_edge_interp_direct(Graph g, int Pcap, InterpArgs a, int R4, int T, float lo, float up, float h,
                                                           float inv_h, float* __restrict__ C, float* __restrict__ dC) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int groups = blockDim.x / R4;
  const int grp = threadIdx.x / R4, c4 = threadIdx.x - grp * R4;
  if (grp >= groups) return;
  const int p = blockIdx.x * groups + grp;
  if (p > Pcap) return;
  const int P = g.counts[0];
  if (p > P || g.counts[2]) return;
  const int R = 4 * R4;
  const float d = p < P ? g.pd[p] : 0.f;  // p == P: the self pair
  if (C && c4 == 0) {
    float c, dc;
    cosine_cutoff(d, lo, up, c, dc);
    C[p] = c;
    dC[p] = dc;
  }
  if (p == P) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float* row = a.tab[t] + (int64_t)(T + 1) * 3 * R + 4 * c4;
      const f4 v0 = *reinterpret_cast<const f4*>(row);
      store_piece(a.out[t], (int64_t)p * R + 4 * c4, v0.x, v0.y, v0.z, v0.w, a.out_bf16);
      if (a.dout[t]) {
        const f4 v1 = *reinterpret_cast<const f4*>(row + R);
        store_piece(a.dout[t], (int64_t)p * R + 4 * c4, v1.x, v1.y, v1.z, v1.w, a.out_bf16);
      }
    }
    return;
  }
  const InterpCoef ic = interp_coef(d, lo, h, inv_h, T);
#pragma unroll
  for (int tb = 0; tb < NT; ++tb) {
    const float* row = a.tab[tb] + (int64_t)ic.k * 3 * R + 4 * c4;
    const f4 f0 = *reinterpret_cast<const f4*>(row), sl0 = *reinterpret_cast<const f4*>(row + R);
    const f4 D = *reinterpret_cast<const f4*>(row + 2 * R), sl1 = *reinterpret_cast<const f4*>(row + 4 * R);
    f4 v0, v1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v0[j] = interp_value(f0[j], sl0[j], D[j], sl1[j], ic);
      v1[j] = interp_slope(sl0[j], D[j], sl1[j], ic);
    }
    store_piece(a.out[tb], (int64_t)p * R + 4 * c4, v0.x, v0.y, v0.z, v0.w, a.out_bf16);
    if (a.dout[tb]) store_piece(a.dout[tb], (int64_t)p * R + 4 * c4, v1.x, v1.y, v1.z, v1.w, a.out_bf16);
  }
}

// pair capacities up to this take the one-launch path (no bucketing)
bool edge_interp_direct(int Pcap) {
  static const int lim = [] {
    const char* e = getenv("TMDNET_EDGE_DIRECT_MAX");
    return e ? atoi(e) : 16384;
  }();
  return Pcap + 1 <= lim;
}

// C, dC per pair and the pairs in grid-interval order.  hist: [T + 2] ints; keys_s / vals_s: [Pcap + 1] each
void launch_pair_buckets(const Graph& g, int Pcap, float lo, float up, int T, float* C, float* dC, int* hist, unsigned* keys_s,
                         int* vals_s, hipStream_t s) {
  const int n = Pcap + 1;
  if (edge_interp_direct(Pcap)) return;  // C, dC come from the first launch_edge_interp of the step
  const float h0 = (up - lo) / (float)T, inv_h0 = 1.0f / h0;
  hipLaunchKernelGGL(k_fill_int, dim3(cdive(T + 2, 256)), dim3(256), 0, s, hist, 0, T + 2);
  hipLaunchKernelGGL(k_pair_cutoff_hist, dim3(cdive(n, BK_THREADS)), dim3(BK_THREADS), 0, s, g, Pcap, lo, up, inv_h0, T, C, dC, hist);
  hipLaunchKernelGGL(k_bucket_scan, dim3(1), dim3(1024), 0, s, hist, T + 1);
  hipLaunchKernelGGL(k_bucket_scatter, dim3(cdive(n, BK_THREADS)), dim3(BK_THREADS), 0, s, g, Pcap, lo, inv_h0, T, hist, keys_s, vals_s);
}

// the tables' outputs for all pairs (tables of one row length R per call)
void launch_edge_interp(const Graph& g, int Pcap, float lo, float up, int T, int R, int ntab, const float* const* tabs,
                        float* const* outs, float* const* douts, const unsigned* keys_s, const int* vals_s, hipStream_t s, float* C,
                        float* dC, int out_bf16) {
  const int n = Pcap + 1;
  InterpArgs a{};  // filled per chunk of four tables below: `ntab` is not bounded by the struct's eight slots
  const float h = (up - lo) / (float)T;
  const int R4 = R / 4;
  const int groups = R4 >= 256 ? 1 : 256 / R4;
  const bool direct = edge_interp_direct(Pcap);
  // pairs per thread group: EI_RUN at batch scale (consecutive sorted pairs share their interval's rows, but long runs leave
  // too few groups in flight); a mid-size pair list (one system of ~1000 atoms) has ~1 pair per interval: shorter runs
  static const int run_env = getenv("TMDNET_EI_RUN") ? atoi(getenv("TMDNET_EI_RUN")) : 0;  // developer switch
  const int run = run_env > 0 ? run_env : (n / 4096 < 4 ? 4 : (n / 4096 > EI_RUN ? EI_RUN : n / 4096));
  const dim3 grid(direct ? cdive(n, groups) : cdive(n, groups * run)), block(groups * R4);
#define EI_LAUNCH(NT)                                                                                                          \
  if (direct) {                                                                                                                \
    hipLaunchKernelGGL((k_edge_interp_direct<NT>), grid, block, 0, s, g, Pcap, a, R4, T, lo, up, h, 1.0f / h, C, dC);          \
    C = dC = nullptr; /* written once */                                                                                        \
  } else                                                                                                                       \
    hipLaunchKernelGGL((k_edge_interp<NT>), grid, block, 0, s, g, Pcap, keys_s, vals_s, a, R4, T, lo, h, 1.0f / h, run)
  for (int t0 = 0; t0 < ntab; t0 += 4) {  // four tables per launch (one launch for TensorNet's L + 1 <= 4)
    a = InterpArgs{};
    a.ntab = ntab - t0 < 4 ? ntab - t0 : 4;
    a.out_bf16 = out_bf16;
    for (int t = 0; t < a.ntab; ++t) {
      a.tab[t] = tabs[t0 + t];
      a.out[t] = outs[t0 + t];
      a.dout[t] = douts ? douts[t0 + t] : nullptr;
    }
    switch (a.ntab) {
      case 1: EI_LAUNCH(1); break;
      case 2: EI_LAUNCH(2); break;
      case 3: EI_LAUNCH(3); break;
      default: EI_LAUNCH(4); break;
    }
  }
#undef EI_LAUNCH
}

// interpolate table `tab` at M listed distances with the step's fp32 arithmetic (build-time verification)
__global__ void k_interp_list(const float* __restrict__ tab, const double* __restrict__ dist, int M, int R, int T, float lo, float h,
                              float inv_h, float* __restrict__ out, float* __restrict__ dout) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)M * R) return;
  const int m = (int)(idx / R), c = (int)(idx - (int64_t)m * R);
  const InterpCoef ic = interp_coef((float)dist[m], lo, h, inv_h, T);
  const float* row = tab + (int64_t)ic.k * 3 * R + c;
  const float f0 = row[0], s0 = row[R], D = row[2 * R], s1 = row[4 * R];
  out[idx] = interp_value(f0, s0, D, s1, ic);
  dout[idx] = interp_slope(s0, D, s1, ic);
}
void launch_interp_list(const float* tab, const double* dist, int M, int R, int T, float lo, float up, float* out, float* dout,
                        hipStream_t s) {
  const float h = (up - lo) / (float)T;
  hipLaunchKernelGGL(k_interp_list, dim3(cdive((int64_t)M * R, 256)), dim3(256), 0, s, tab, dist, M, R, T, lo, h, 1.0f / h, out, dout);
}

}  // namespace tn
