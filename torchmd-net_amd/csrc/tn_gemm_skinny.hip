// Latency-oriented fp32 MFMA GEMM for small row counts (single molecules / small MD systems), gfx950.
//
// The throughput kernel (tn_gemm.hip) walks K serially inside one 128-row tile: with a handful of tiles the
// chip is empty and each launch costs K/32 dependent iterations (20-70 us).  Here one block owns a 32x32
// output tile and its 4 waves split K (intra-block split-K); every wave streams its operands straight from
// global/L2 into MFMA registers (lane l holds row l&31, k-quad l>>5: one 16-byte load feeds four
// v_mfma_f32_32x32x2_f32), so the main loop has no LDS traffic and no barrier.  The four partial tiles are
// summed through LDS in a FIXED order (deterministic) and written as coalesced float4 rows with the fused
// epilogue.  ceil(M/32)*ceil(N/32) blocks: 252 blocks for the 671-pair edge MLP of one 64-atom molecule.
#include "tn_common.h"
#include "tn_gemm_epi.h"

namespace tn {

typedef float floatx16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float4 ld_row4(const float* __restrict__ base, int64_t ld, int row, int nrows, int k, int K, bool vec) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (row < nrows) {
    const float* p = base + (int64_t)row * ld + k;
    if (vec) {
      if (k < K) v = *reinterpret_cast<const float4*>(p);
    } else {
      if (k + 0 < K) v.x = p[0];
      if (k + 1 < K) v.y = p[1];
      if (k + 2 < K) v.z = p[2];
      if (k + 3 < K) v.w = p[3];
    }
  }
  return v;
}

// NW waves split K: 4 (K <= 192: 128 threads of reduction work per 32 columns is already the launch floor) or 8 (longer
// contractions: the readout / gate GEMMs with K = 256..640 were 8-10 us of dependent slabs in the single-molecule step)
template <int EPI, int NW>
__global__ __launch_bounds__(64 * NW) void k_gemm_skinny(GemmArgs a, int tiles_m, int tiles_n) {
  __shared__ __attribute__((aligned(16))) float part[NW][32][33];
  const int per_group = tiles_m * tiles_n;
  const int g = blockIdx.x / per_group;
  const int rem = blockIdx.x - g * per_group;
  const int tm = rem / tiles_n, tn_ = rem - tm * tiles_n;
  const int m0 = tm * 32, n0 = tn_ * 32;
  int M = a.M;
  if (a.m_dev) {
    const int md = *a.m_dev + a.m_add;
    M = md < M ? md : M;
    if (m0 >= M) return;
  }
  const int N = a.N, K = a.K;
  const float* __restrict__ A = a.A + a.a_off[g];
  const float* __restrict__ W = a.W[g];
  const bool a_vec = ((a.lda & 3) == 0) && ((K & 3) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
  const bool w_vec = ((a.ldw & 3) == 0) && ((K & 3) == 0) && ((reinterpret_cast<uintptr_t>(W) & 15) == 0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, kq = (lane >> 5) << 2;

  // K range of this wave: multiples of 8 (one 16-byte load per lane covers k = k0 + kq + {0..3})
  const int chunk = ((K + 8 * NW - 1) / (8 * NW)) * 8;
  const int kbeg = wave * chunk, kend = (kbeg + chunk < K) ? kbeg + chunk : K;

  floatx16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  float4 av, bv;
  if (kbeg < kend) {
    av = ld_row4(A, a.lda, m0 + r, M, kbeg + kq, K, a_vec);
    bv = ld_row4(W, a.ldw, n0 + r, N, kbeg + kq, K, w_vec);
  }
  for (int k = kbeg; k < kend; k += 8) {
    float4 an = make_float4(0.f, 0.f, 0.f, 0.f), bn = an;
    if (k + 8 < kend) {  // prefetch the next 8-wide slab while the MFMAs of this one issue
      an = ld_row4(A, a.lda, m0 + r, M, k + 8 + kq, K, a_vec);
      bn = ld_row4(W, a.ldw, n0 + r, N, k + 8 + kq, K, w_vec);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
    av = an;
    bv = bn;
  }
  // partial tile -> LDS.  C/D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
  for (int e = 0; e < 16; ++e) part[wave][(e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)][lane & 31] = acc[e];
  __syncthreads();
  // fixed-order reduction over the NW waves; thread -> (row = tid/8, 4 consecutive columns)
  if (tid >= 256) return;
  const int row = tid >> 3, c0 = (tid & 7) << 2;
  const int grow = m0 + row;
  if (grow >= M) return;
  const float* __restrict__ bias = a.bias[g];
  float* __restrict__ C = a.C + a.c_off[g];
  float* __restrict__ pre = a.pre ? a.pre + a.pre_off[g] : nullptr;
  const float* __restrict__ aux = a.aux ? a.aux + a.aux_off[g] : nullptr;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = n0 + c0 + j;
    if (col >= N) continue;
    float v = ((part[0][row][c0 + j] + part[1][row][c0 + j]) + part[2][row][c0 + j]) + part[3][row][c0 + j];
#pragma unroll
    for (int wv = 4; wv < NW; ++wv) v += part[wv][row][c0 + j];
    if (bias) v += bias[col];
    epilogue_store<EPI>(a, C, pre, aux, grow, col, v);
  }
}

template <int EPI>
static int launch_skinny_one(const GemmArgs& a, hipStream_t stream) {
  const int tiles_m = (a.M + 31) / 32, tiles_n = (a.N + 31) / 32;
  const int total = tiles_m * tiles_n * a.groups;
  if (total <= 0) return 0;
  if (a.K >= 256)
    hipLaunchKernelGGL((k_gemm_skinny<EPI, 8>), dim3(total), dim3(512), 0, stream, a, tiles_m, tiles_n);
  else
    hipLaunchKernelGGL((k_gemm_skinny<EPI, 4>), dim3(total), dim3(256), 0, stream, a, tiles_m, tiles_n);
  return (int)hipGetLastError();
}

int launch_gemm_skinny(const GemmArgs& a, hipStream_t stream) {
  switch (epi_kind(a)) {
    case EPI_PLAIN: return launch_skinny_one<EPI_PLAIN>(a, stream);
    case EPI_SILU_PRE: return launch_skinny_one<EPI_SILU_PRE>(a, stream);
    case EPI_SILU_PRE_ROWSCALE: return launch_skinny_one<EPI_SILU_PRE_ROWSCALE>(a, stream);
    case EPI_MULAUX_PRE: return launch_skinny_one<EPI_MULAUX_PRE>(a, stream);
    case EPI_MULDSILU: return launch_skinny_one<EPI_MULDSILU>(a, stream);
    case EPI_ACCUM: return launch_skinny_one<EPI_ACCUM>(a, stream);
    default: return launch_skinny_one<EPI_GENERIC>(a, stream);
  }
}

}  // namespace tn
