// First-order parameter gradients of the TensorNet + Scalar path (gfx950): the kernels the reverse pass of tn_api.hip calls
// when a training context is attached (tmdnet_energy_param_grads).  Reference: the reference gets these from autograd over
// tensornet.py:543-619 (embedding), :729-814 (interaction), :384-398 (readout), output_modules.py:108-117, models/utils.py:552-580.
//
// Every dense layer y = x W^T + b contributes  dW = g_y^T x  and  db = colsum(g_y): one generic "transposed" product on the fp32
// matrix pipe (v_mfma_f32_32x32x2_f32: both operands are read row by row, 32 consecutive floats per half-wave, which IS the
// operand layout of that instruction - no LDS, no transposition), split over row slices with a fixed-order reduction of the
// partial tiles (deterministic, no atomics).  Rows may be the (atom, component) rows of one irreducible type of a [N, 9, F]
// tensor (RowMap).  These are not on the inference path; they are sized to be correct and reasonably fast, not tuned.
#include <algorithm>
#include <cstdlib>

#include "tn_common.h"
#include "tn_kernels.h"
#include "tn_train.h"

namespace tn {

typedef float floatx16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void load9(const float* __restrict__ p, int F, float u[9]) {
#pragma unroll
  for (int c = 0; c < 9; ++c) u[c] = p[c * F];
}

__device__ __forceinline__ int64_t row_off(const RowMap& m, int r) {
  return m.reps == 1 ? (int64_t)r * m.ld : (int64_t)(r / m.reps) * m.ld + (int64_t)(r % m.reps) * m.sub;
}

// out_part[slice][n][k] = sum_{r in slice} A[r][n] * rs[r] * B[r][k]     (64 x 64 tile per block, four waves share the slice)
// A2 / B2 (optional, same row maps): a second product whose slices follow the first one's (blockIdx.y >= seg_slices) - see k_tn_gemm128
__global__ __launch_bounds__(256) void k_tn_gemm(const float* __restrict__ A, RowMap ma, const float* __restrict__ B, RowMap mb,
                                                 const float* __restrict__ A2, const float* __restrict__ B2, int seg_slices,
                                                 const float* __restrict__ rs, const int* __restrict__ r_dev, int R, int Nout, int Kin,
                                                 int tiles_k, int rows_per_slice, float* __restrict__ part) {
  __shared__ float red[3][64 * 64];
  if (r_dev) R = min(R, *r_dev);
  const bool second = (int)blockIdx.y >= seg_slices;  // block-uniform
  if (second) {
    A = A2;
    B = B2;
  }
  const int tile = blockIdx.x, slice = (int)blockIdx.y - (second ? seg_slices : 0);
  const int tn_ = tile / tiles_k, tk = tile - tn_ * tiles_k;
  const int n0 = tn_ * 64, k0 = tk * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int kk = lane >> 5, cl = lane & 31;
  const int r_lo = slice * rows_per_slice, r_hi = min(R, r_lo + rows_per_slice);
  floatx16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const bool na0 = n0 + cl < Nout, na1 = n0 + 32 + cl < Nout, kb0 = k0 + cl < Kin, kb1 = k0 + 32 + cl < Kin;
  // a wave takes every fourth pair of rows; four pairs per trip so that 16 loads are in flight
  for (int r = r_lo + 2 * wave; r < r_hi; r += 32) {
    float a[4][2], b[4][2];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int rr = r + 8 * u + kk;
      const bool ok = rr < r_hi;
      const int rc = ok ? rr : r_lo;
      const float* pa = A + row_off(ma, rc) + n0 + cl;
      const float* pb = B + row_off(mb, rc) + k0 + cl;
      const float sc = ok ? (rs ? rs[rc] : 1.f) : 0.f;
      a[u][0] = na0 ? pa[0] * sc : 0.f;
      a[u][1] = na1 ? pa[32] * sc : 0.f;
      b[u][0] = (kb0 && ok) ? pb[0] : 0.f;
      b[u][1] = (kb1 && ok) ? pb[32] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][i], b[u][j], acc[i][j], 0, 0, 0);
  }
  // fixed-order sum of the four waves' tiles (wave 0 + 1 + 2 + 3), then the partial tile of this slice
  auto at = [&](int i, int j, int e) { return (32 * i + (e & 3) + 8 * (e >> 2) + 4 * kk) * 64 + 32 * j + cl; };
  if (wave > 0) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) red[wave - 1][at(i, j, e)] = acc[i][j][e];
  }
  __syncthreads();
  if (wave > 0) return;
  float* o = part + (int64_t)blockIdx.y * Nout * Kin;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int idx = at(i, j, e);
        const float v = ((acc[i][j][e] + red[0][idx]) + red[1][idx]) + red[2][idx];
        const int n = n0 + idx / 64, k = k0 + (idx & 63);
        if (n < Nout && k < Kin) o[(int64_t)n * Kin + k] = v;
      }
}

// out[i] (+)= sum_s part[s][i], slices in order
__global__ void k_reduce_slices(const float* __restrict__ part, int slices, int64_t n, int accumulate, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = accumulate ? out[i] : 0.f;
  for (int k = 0; k < slices; ++k) s += part[(int64_t)k * n + i];
  out[i] = s;
}

// few outputs, many slices (a column sum over the pair rows: 384 outputs x 757 slices; one thread per output would walk them one
// after the other, ~95 us of load latency): 16 threads share an output, fixed order inside the thread and across the 16
__global__ __launch_bounds__(256) void k_reduce_slices_wide(const float* __restrict__ part, int slices, int64_t n, int accumulate,
                                                            float* __restrict__ out) {
  __shared__ float red[16][17];
  const int e = threadIdx.x & 15, lane = threadIdx.x >> 4;
  const int64_t i = (int64_t)blockIdx.x * 16 + e;
  float s = 0.f;
  if (i < n)
    for (int k = lane; k < slices; k += 16) s += part[(int64_t)k * n + i];
  red[lane][e] = s;
  __syncthreads();
  if (lane == 0 && i < n) {
    float t = accumulate ? out[i] : 0.f;
#pragma unroll
    for (int l = 0; l < 16; ++l) t += red[l][e];
    out[i] = t;
  }
}
// many outputs (a 3F x 2F weight: 98 304) x ~100 slices: one thread per output is 1.5 waves per SIMD walking its slices one
// dependent-latency load at a time (73 us, 0.7 TB/s): four threads (one per wave of the block) share an output - consecutive
// lanes = consecutive outputs, so every load instruction is a full line - eight slices requested per trip, fixed order inside
// the thread and across the four
__global__ __launch_bounds__(256) void k_reduce_slices4(const float* __restrict__ part, int slices, int64_t n, int accumulate,
                                                        float* __restrict__ out) {
  __shared__ float red[4][64];
  const int e = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + e;
  float s = 0.f;
  if (i < n) {
    int k = grp;
    for (; k + 28 < slices; k += 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(int64_t)(k + 4 * u) * n + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < slices; k += 4) s += part[(int64_t)k * n + i];
  }
  red[grp][e] = s;
  __syncthreads();
  if (grp == 0 && i < n) {
    float t = accumulate ? out[i] : 0.f;
#pragma unroll
    for (int l = 0; l < 4; ++l) t += red[l][e];
    out[i] = t;
  }
}
static void reduce_slices(hipStream_t s, const float* part, int slices, int64_t n, bool accumulate, float* out) {
  if (slices >= 16 && n > 16384)
    hipLaunchKernelGGL(k_reduce_slices4, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, s, part, slices, n, accumulate ? 1 : 0, out);
  else if (slices >= 32 && n <= 16384)
    hipLaunchKernelGGL(k_reduce_slices_wide, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, s, part, slices, n, accumulate ? 1 : 0, out);
  else
    hipLaunchKernelGGL(k_reduce_slices, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, part, slices, n, accumulate ? 1 : 0, out);
}

// 128 x 128 output tile per WAVE (sixteen 32 x 32 accumulators), every wave its own row range and its own partial output: per pair
// of rows a wave loads 4 + 4 operand values for 16 matrix instructions, where the 64 x 64 kernel loads 2 + 2 for 4 - half the
// operand traffic per flop, no LDS, no barrier.  For every product whose output is whole 128 x 128 tiles (F = 128: the edge MLP's
// 3F x 2F and 2F x F, the per-atom MLPs, the F x F tensor linears by irreducible type).
// Nout, Kin multiples of 128, no row scale.
// A2 / B2 (optional): a second product with the same row maps and shapes whose partial outputs follow the first one's - the two
// terms g_y_t^T x + g_y^T x_t of a dual weight gradient are then ONE launch and ONE slice reduction (blockIdx.y >= seg_slices).
__global__ __launch_bounds__(256) void k_tn_gemm128(const float* __restrict__ A, RowMap ma, const float* __restrict__ B, RowMap mb,
                                                    const float* __restrict__ A2, const float* __restrict__ B2, int seg_slices,
                                                    const int* __restrict__ r_dev, int R, int Nout, int Kin,
                                                    int tiles_k, int rows_per_wave, float* __restrict__ part) {
  if (r_dev) R = min(R, *r_dev);
  const int tile = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool second = (int)blockIdx.y >= seg_slices;  // block-uniform
  if (second) {
    A = A2;
    B = B2;
  }
  const int sub = ((int)blockIdx.y - (second ? seg_slices : 0)) * 4 + wave;
  const int tn_ = tile / tiles_k, tk = tile - tn_ * tiles_k;
  const int n0 = tn_ * 128, k0 = tk * 128;
  const int kk = lane >> 5, cl = lane & 31;
  const int r_lo = sub * rows_per_wave, r_hi = min(R, r_lo + rows_per_wave);
  floatx16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  // two pairs of rows per trip: 16 loads, 32 matrix instructions; the next trip's operands are requested before this trip's matrix
  // instructions issue (one wave per SIMD: nobody else hides the load latency)
  auto load = [&](int r, float (&a)[2][4], float (&b)[2][4]) {  // no branch around a load: rows past the end are clamped and masked
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int rr = r + 2 * u + kk;
      const int rc = min(rr, r_hi - 1);
      const float* pa = A + row_off(ma, rc) + n0 + cl;
      const float* pb = B + row_off(mb, rc) + k0 + cl;
      const float sc = rr < r_hi ? 1.f : 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a[u][i] = pa[32 * i] * sc;
        b[u][i] = pb[32 * i];
      }
    }
  };
  float a[2][4], b[2][4], an[2][4], bn[2][4];
  if (r_lo < r_hi) load(r_lo, a, b);
  for (int r = r_lo; r < r_hi; r += 4) {
    load(r + 4, an, bn);
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][i], b[u][j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a[u][i] = an[u][i];
        b[u][i] = bn[u][i];
      }
  }
  float* o = part + (int64_t)((int)blockIdx.y * 4 + wave) * Nout * Kin;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int n = n0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * kk, k = k0 + 32 * j + cl;
        o[(int64_t)n * Kin + k] = acc[i][j][e];
      }
}

// partial outputs one product may write: 128, or 512 for outputs of at most 32 768 elements (launch_tn_gemm and the sizing below agree)
static int max_partials(int64_t out_elems) { return out_elems <= 32768 ? 512 : 128; }

size_t train_part_floats(int R, int64_t out_elems) {
  const int64_t tn = std::max<int64_t>(128 * out_elems, 512 * std::min<int64_t>(out_elems, 32768));  // transposed products: max_partials()
  const int64_t cs = ((int64_t)R / 256 + 2) * std::min<int64_t>(out_elems, 4096);  // column sums: one partial row per 256 rows
  return (size_t)std::max(tn, cs);
}

void launch_tn_gemm(hipStream_t s, const float* A, RowMap ma, const float* B, RowMap mb, const float* rowscale, const int* r_dev, int R,
                    int Nout, int Kin, float* out, bool accumulate, float* part) {
  if (Nout <= 0 || Kin <= 0) return;
  const int64_t n = (int64_t)Nout * Kin;
  if (Nout % 128 == 0 && Kin % 128 == 0 && R >= 4096 && !rowscale) {
    // one 128 x 128 tile per wave: about 1 500 waves when the partial-output budget and the row count (>= 64 rows per wave) allow it
    const int tiles = (Nout / 128) * (Kin / 128);
    int slices = std::max(1, std::min({max_partials(n) / 4, (1536 + 4 * tiles - 1) / (4 * tiles), R / 256}));
    int rpw = (R + 4 * slices - 1) / (4 * slices);
    rpw = (rpw + 1) & ~1;  // pairs of rows
    hipLaunchKernelGGL(k_tn_gemm128, dim3(tiles, slices), dim3(256), 0, s, A, ma, B, mb, nullptr, nullptr, slices, r_dev, R, Nout, Kin,
                       Kin / 128, rpw, part);
    reduce_slices(s, part, 4 * slices, n, accumulate, out);
    return;
  }
  const int tiles_n = (Nout + 63) / 64, tiles_k = (Kin + 63) / 64;
  // row slices of >= 256 rows: as many as it takes to put about four blocks on every CU, at most 128 (512 for an output of one or
  // two tiles: the 128 x 32 product of the first edge layer) and within the partial-output budget
  const int tiles = tiles_n * tiles_k;
  int slices = R > 0 ? std::max(1, std::min({max_partials(n), tiles <= 2 ? 512 : 128, (R + 255) / 256})) : 1;
  slices = std::max(1, std::min(slices, std::max(8, 1024 / tiles)));
  int rps = R > 0 ? (R + slices - 1) / slices : 1;
  rps = (rps + 1) & ~1;  // pairs of rows
  hipLaunchKernelGGL(k_tn_gemm, dim3(tiles_n * tiles_k, slices), dim3(256), 0, s, A, ma, B, mb, nullptr, nullptr, slices, rowscale, r_dev, R,
                     Nout, Kin, tiles_k, rps, part);
  reduce_slices(s, part, slices, n, accumulate, out);
}

// out (+)= A1^T B1 + A2^T B2 (same row maps, shapes and row count): one launch and one slice reduction where the 128 x 128 kernel
// applies, two plain products otherwise
void launch_tn_gemm_pair(hipStream_t s, const float* A1, const float* B1, const float* A2, const float* B2, RowMap ma, RowMap mb,
                         const int* r_dev, int R, int Nout, int Kin, float* out, bool accumulate, float* part) {
  if (Nout <= 0 || Kin <= 0) return;
  const int64_t n = (int64_t)Nout * Kin;
  if (Nout % 128 == 0 && Kin % 128 == 0 && R >= 4096 && max_partials(n) >= 8) {
    const int tiles = (Nout / 128) * (Kin / 128);
    int slices = std::max(1, std::min({max_partials(n) / 8, (1536 + 4 * tiles - 1) / (4 * tiles), R / 256}));
    int rpw = (R + 4 * slices - 1) / (4 * slices);
    rpw = (rpw + 1) & ~1;
    hipLaunchKernelGGL(k_tn_gemm128, dim3(tiles, 2 * slices), dim3(256), 0, s, A1, ma, B1, mb, A2, B2, slices, r_dev, R, Nout, Kin, Kin / 128,
                       rpw, part);
    reduce_slices(s, part, 8 * slices, n, accumulate, out);
    return;
  }
  // the 64 x 64 kernel (narrow outputs: F x K, 3F x K): half the slices per product, both products in one launch
  const int tiles_n = (Nout + 63) / 64, tiles_k = (Kin + 63) / 64, tiles = tiles_n * tiles_k;
  int slices = R > 0 ? std::max(1, std::min({max_partials(n) / 2, tiles <= 2 ? 256 : 64, (R + 255) / 256})) : 1;
  slices = std::max(1, std::min(slices, std::max(4, 512 / tiles)));
  int rps = R > 0 ? (R + slices - 1) / slices : 1;
  rps = (rps + 1) & ~1;
  hipLaunchKernelGGL(k_tn_gemm, dim3(tiles, 2 * slices), dim3(256), 0, s, A1, ma, B1, mb, A2, B2, slices, nullptr, r_dev, R, Nout, Kin, tiles_k,
                     rps, part);
  reduce_slices(s, part, 2 * slices, n, accumulate, out);
}

// part[slice][c] = sum_{r in slice} A[r][c] * (B ? B[r][c] : 1) * (rs ? rs[r] : 1)
// block = 64 columns x 4 row lanes over a slice of 256 rows (a thread adds every fourth row, the four lanes are summed in order)
constexpr int CS_ROWS = 256;
__global__ __launch_bounds__(256) void k_colsum(const float* __restrict__ A, RowMap ma, const float* __restrict__ B, RowMap mb,
                                                const float* __restrict__ rs, const int* __restrict__ r_dev, int R, int ncol,
                                                float* __restrict__ part) {
  __shared__ float red[3][64];
  if (r_dev) R = min(R, *r_dev);
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int slice = blockIdx.y;
  const int r_lo = slice * CS_ROWS, r_hi = min(R, r_lo + CS_ROWS);
  float acc = 0.f;
  if (c < ncol) {
    for (int r = r_lo + rl; r < r_hi; r += 16) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int rr = r + 4 * u;
        const bool ok = rr < r_hi;
        const int rc = ok ? rr : r_lo;
        float x = A[row_off(ma, rc) + c];
        if (B) x *= B[row_off(mb, rc) + c];
        if (rs) x *= rs[rc];
        v[u] = ok ? x : 0.f;
      }
      acc += (v[0] + v[1]) + (v[2] + v[3]);
    }
  }
  if (rl > 0) red[rl - 1][cl] = acc;
  __syncthreads();
  if (rl == 0 && c < ncol) part[(int64_t)slice * ncol + c] = ((acc + red[0][cl]) + red[1][cl]) + red[2][cl];
}
void launch_colsum(hipStream_t s, const float* A, RowMap ma, const float* B, RowMap mb, const float* rowscale, const int* r_dev, int R,
                   int ncol, float* out, bool accumulate, float* part) {
  if (ncol <= 0) return;
  const int slices = R > 0 ? (R + CS_ROWS - 1) / CS_ROWS : 1;
  hipLaunchKernelGGL(k_colsum, dim3((ncol + 63) / 64, slices), dim3(256), 0, s, A, ma, B, mb, rowscale, r_dev, R, ncol, part);
  reduce_slices(s, part, slices, (int64_t)ncol, accumulate, out);
}

// g_ao[i, :] *= gE[mol(i)]  (the seed of molecule m's energy in the loss); head[i, 0..H) = silu(ao) s_i, head[i, H] = s_i with
// s_i = gE[mol(i)] std: the column sums of `head` are dO2 and dbO2 (k_head_energy: e_i = (silu(ao_i) . O2 + bO2) std + atomref)
__global__ void k_train_seed(const float* __restrict__ ao, const float* __restrict__ gE, const int64_t* __restrict__ batch, int N, int H,
                             float std, float* __restrict__ g_ao, float* __restrict__ head) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)N * (H + 1)) return;
  const int n = (int)(idx / (H + 1)), h = (int)(idx - (int64_t)n * (H + 1));
  const float ge = gE[batch ? batch[n] : 0];
  if (h < H) {
    g_ao[(int64_t)n * H + h] *= ge;
    head[idx] = silu(ao[(int64_t)n * H + h]) * ge * std;
  } else {
    head[idx] = ge * std;
  }
}
void launch_train_seed(const float* ao, const float* gE, const int64_t* batch, int N, int H, float std, float* g_ao, float* head,
                       hipStream_t s) {
  if (N <= 0) return;
  const int64_t n = (int64_t)N * (H + 1);
  hipLaunchKernelGGL(k_train_seed, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, ao, gE, batch, N, H, std, g_ao, head);
}

// one-hot of the atomic numbers [N, Zp] (the per-species sums of per-atom rows become a transposed product)
__global__ void k_onehot(const int64_t* __restrict__ z, int N, int Zp, float* __restrict__ oh) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)N * Zp) return;
  const int n = (int)(idx / Zp), t = (int)(idx - (int64_t)n * Zp);
  oh[idx] = z[n] == t ? 1.f : 0.f;
}
void launch_onehot(const int64_t* z, int N, int Zp, float* oh, hipStream_t s) {
  if (N <= 0) return;
  const int64_t n = (int64_t)N * Zp;
  hipLaunchKernelGGL(k_onehot, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, z, N, Zp, oh);
}

// out[c][r] = in[r][c]  (small weights: the data-gradient GEMMs of the edge MLP take W^T as their [N][K] operand)
__global__ void k_transpose(const float* __restrict__ in, int rows, int cols, float* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)rows * cols) return;
  const int r = (int)(idx / cols), c = (int)(idx - (int64_t)r * cols);
  out[(int64_t)c * rows + r] = in[idx];
}
void launch_transpose(const float* in, int rows, int cols, float* out, hipStream_t s) {
  const int64_t n = (int64_t)rows * cols;
  if (n <= 0) return;
  hipLaunchKernelGGL(k_transpose, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, rows, cols, out);
}

// Adjoint of the message weights (reference tensornet.py:757-806):  g_w[p, k, f] = sum_{c in k} gMi[i,c,f] Pn[j,c,f] + gMi[j,c,f] Pn[i,c,f]
// for the pair p = (i, j) (both of its directed edges read the same row w[p]);  w = silu(pre3) C(d)  ->
// g_pre3 = g_w C silu'(pre3).  The self pair (row P, shared by every atom's self edge) takes its sum over the atoms from `self_gw`.
__global__ __launch_bounds__(256) void k_train_gw(Graph g, int Pcap, int F, const float* __restrict__ gMi, const float* __restrict__ Pn,
                                                  const float* __restrict__ pre3, const float* __restrict__ C,
                                                  const float* __restrict__ self_gw, float* __restrict__ g_pre3) {
  const int P = g.counts[0];
  const int p = blockIdx.x;
  if (p > P || p > Pcap || g.counts[2]) return;
  const int F3 = 3 * F, F9 = 9 * F;
  const float c = C[p];
  for (int f = threadIdx.x; f < F; f += blockDim.x) {
    float gw[3];
    if (p == P) {
      gw[0] = self_gw[f];
      gw[1] = self_gw[F + f];
      gw[2] = self_gw[2 * F + f];
    } else {
      const int i = g.pair_i[p], j = g.pair_j[p];
      float a[9], b[9], x[9], y[9];
      load9(gMi + (int64_t)i * F9 + f, F, a);
      load9(Pn + (int64_t)j * F9 + f, F, b);
      load9(gMi + (int64_t)j * F9 + f, F, x);
      load9(Pn + (int64_t)i * F9 + f, F, y);
      gw[0] = a[0] * b[0] + x[0] * y[0];
      gw[1] = a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3];
      gw[2] = 0.f;
#pragma unroll
      for (int q = 4; q < 9; ++q) gw[2] += a[q] * b[q] + x[q] * y[q];
    }
    const float* pp = pre3 + (int64_t)p * F3 + f;
    float* o = g_pre3 + (int64_t)p * F3 + f;
#pragma unroll
    for (int k = 0; k < 3; ++k) o[k * F] = gw[k] * c * silu_grad(pp[k * F]);
  }
}
void launch_train_gw(const Graph& g, int P, int F, const float* gMi, const float* Pn, const float* pre3, const float* C,
                     const float* self_gw, float* g_pre3, hipStream_t s) {
  hipLaunchKernelGGL(k_train_gw, dim3(P + 1), dim3(F > 256 ? 256 : ((F + 63) / 64) * 64), 0, s, g, P, F, gMi, Pn, pre3, C, self_gw, g_pre3);
}

// Adjoint of the embedding's edge weights W_k(e) = C(d) (U[z_i] + V[z_j]) Q[p, k]   (k_embed_scatter; reference tensornet.py:526-541,
// 405-445) from gA = d/d(I0, v, T) per atom.  Block = atom i, thread = channel.  For every edge e = (i <- j):
//   own direction  : gW = gA[i] . geometry(r)      -> gq[dir][p, k] = gW_k cz   (self edge: selfq[i] = gW_0 cz) ; gZu[i] += C sum_k gW_k Q_k
//   reverse (j <- i): gW' = gA[j] . geometry(-r)   -> gZv[i] += C sum_k gW'_k Q_k   (atom i is the V-side atom of that edge)
__global__ void k_train_embed(Graph g, int N, int F, const int64_t* __restrict__ z, const float* __restrict__ Utab,
                              const float* __restrict__ Vtab, const float* __restrict__ Q, const float* __restrict__ C,
                              const float* __restrict__ gA, float* __restrict__ gq, int64_t dir_stride, float* __restrict__ selfq,
                              float* __restrict__ gZu, float* __restrict__ gZv) {
  const int i = blockIdx.x;
  if (g.counts[2]) return;
  const int e0 = g.rowptr[i], e1 = g.rowptr[i + 1];
  const int64_t zi = z[i];
  const int F3 = 3 * F, F10 = 10 * F;
  for (int f = threadIdx.x; f < F; f += blockDim.x) {
    const float Ui = Utab[zi * F + f];
    float ai[10];
#pragma unroll
    for (int c = 0; c < 10; ++c) ai[c] = gA[(int64_t)i * F10 + c * F + f];
    float zu = 0.f, zv = 0.f, sq = 0.f;
    for (int e = e0; e < e1; ++e) {
      const int j = g.col[e], p = g.epair[e];
      const float sg = g.esign[e];
      float rx = 0.f, ry = 0.f, rz = 0.f;
      if (sg != 0.f) {
        rx = sg * g.prhat[p * 3];
        ry = sg * g.prhat[p * 3 + 1];
        rz = sg * g.prhat[p * 3 + 2];
      }
      const float c = C[p];
      const float* q = Q + (int64_t)p * F3 + f;
      const float q0 = q[0], q1 = q[F], q2 = q[2 * F];
      const float cz = c * (Ui + Vtab[z[j] * F + f]);
      const float xx = rx * rx, xy = rx * ry, xz = rx * rz, yy = ry * ry, yz = ry * rz, zz = rz * rz;
      const float w0 = ai[0], w1 = ai[1] * rx + ai[2] * ry + ai[3] * rz,
                  w2 = ai[4] * xx + ai[5] * xy + ai[6] * xz + ai[7] * yy + ai[8] * yz + ai[9] * zz;
      zu += c * (w0 * q0 + w1 * q1 + w2 * q2);
      if (sg == 0.f) {
        sq = w0 * cz;
        zv += c * (w0 * q0);  // the self edge is its own reverse
      } else {
        float* o = gq + (sg > 0.f ? 0 : dir_stride) + (int64_t)p * F3 + f;
        o[0] = w0 * cz;
        o[F] = w1 * cz;
        o[2 * F] = w2 * cz;
        float aj[10];
#pragma unroll
        for (int cc = 0; cc < 10; ++cc) aj[cc] = gA[(int64_t)j * F10 + cc * F + f];
        const float v1 = -(aj[1] * rx + aj[2] * ry + aj[3] * rz),
                    v2 = aj[4] * xx + aj[5] * xy + aj[6] * xz + aj[7] * yy + aj[8] * yz + aj[9] * zz;
        zv += c * (aj[0] * q0 + v1 * q1 + v2 * q2);
      }
    }
    selfq[(int64_t)i * F + f] = sq;
    gZu[(int64_t)i * F + f] = zu;
    gZv[(int64_t)i * F + f] = zv;
  }
}
void launch_train_embed(const Graph& g, int N, int F, const int64_t* z, const float* Utab, const float* Vtab, const float* Q, const float* C,
                        const float* gA, float* gq, int64_t dir_stride, float* selfq, float* gZu, float* gZv, hipStream_t s) {
  if (N <= 0) return;
  hipLaunchKernelGGL(k_train_embed, dim3(N), dim3(F > 256 ? 256 : ((F + 63) / 64) * 64), 0, s, g, N, F, z, Utab, Vtab, Q, C, gA, gq,
                     dir_stride, selfq, gZu, gZv);
}

}  // namespace tn
