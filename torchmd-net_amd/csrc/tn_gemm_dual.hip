// "Dual" fp32 MFMA GEMM for the edge MLPs: value and d/dd tangent in one pass (gfx950).
//
// The per-pair edge weights of TensorNet are a function of the distance only:
//     w(d) = C(d) * silu(M3 silu(M2 silu(M1 phi(d) + b1) + b2) + b3)          (reference tensornet.py:738-743)
// so the force path only needs  g_d = sum_f g_w[f] * dw[f]/dd .  Instead of the reverse-mode chain (3 more
// GEMMs with transposed weights + all pre-activations kept in HBM) the tangent dw/dd is propagated FORWARD
// next to the value, sharing every weight tile:
//     e = A W^T + b ,  r = A' W^T            (A' = tangent of the layer input, same W tile in LDS)
//     out = silu(e) ,  out' = silu'(e) * r   (last layer: * C(d), + silu(e) * C'(d))
// Same FLOPs as forward + reverse, but ~40 % less HBM traffic (no e1/e2/e3, no g_e3/g_e2/g_e1 arrays), the
// weight tile is read once for both operands, and the reverse pass needs no edge GEMM at all.
//
// Kernel shape: block = 4 waves side by side in N (wave tile 128 x 32): LDS rows 0..63 hold the A tile, rows
// 64..127 the tangent tile of the SAME 64 global rows, so accumulators mi and mi+2 of a lane are e and r of one
// output element.  A 32x32 split-K variant (4 waves split K, two accumulators) serves small row counts.
#include <cstdio>
#include <cstdlib>

#include "tn_common.h"
#include "tn_gemm_epi.h"

namespace tn {

typedef float floatx16 __attribute__((ext_vector_type(16)));


enum DualKind : int {
  DUAL_PLAIN = 0,  // C = e ; C2 = r
  DUAL_SILU,       // C = silu(e) ; C2 = silu'(e) r
  DUAL_FINAL,      // C = silu(e) rs[m] ; C2 = silu'(e) r rs[m] + silu(e) rs2[m]
  DUAL_SCALE       // C = e rs[m] ; C2 = r rs[m] + e rs2[m]     (ET neighbour-embedding filter, models/utils.py:99-100)
};

template <int KIND>
__device__ __forceinline__ void dual_store(const GemmArgs& a, int row, int col, float e, float r) {
  float* c1 = a.C + (int64_t)row * a.ldc + col;
  float* c2 = a.C2 + (int64_t)row * a.ldc + col;
  if (KIND == DUAL_PLAIN) {
    *c1 = e;
    *c2 = r;
  } else if (KIND == DUAL_SCALE) {
    const float c = a.rowscale[row], dc = a.rowscale2[row];
    *c1 = e * c;
    *c2 = r * c + e * dc;
  } else {
    const float s = fast_sigmoid(e);
    const float f = e * s, df = s * (1.0f + e * (1.0f - s));
    if (KIND == DUAL_SILU) {
      *c1 = f;
      *c2 = df * r;
    } else {
      const float c = a.rowscale[row], dc = a.rowscale2[row];
      *c1 = f * c;
      *c2 = df * r * c + f * dc;
    }
  }
}

__device__ __forceinline__ float4 dload4(const float* __restrict__ base, int64_t ld, int row, int nrows, int k, int K, bool vec) {
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

template <int KIND, int DBK, int OCC, int PRIO>
__global__ __launch_bounds__(256, OCC) void k_gemm_dual(GemmArgs a, int tiles_m, int tiles_n) {
  constexpr int DLD = DBK + 4;          // row stride in floats: 36 / 68 -> conflict-free ds_read_b128
  constexpr int NV = 128 * DBK / 4 / 256;  // float4 per thread per panel
  constexpr int CPR = DBK / 4;          // float4 chunks per row
  constexpr int NBUF = (PRIO & 8) ? 2 : 1;  // PRIO&8: LDS double buffer, one barrier per k-step
  constexpr int TILE = (128 + 128) * DLD;
  __shared__ __attribute__((aligned(16))) float smem[NBUF * TILE];
  float* As = smem;               // rows 0..63: A tile, rows 64..127: tangent tile
  float* Bs = smem + 128 * DLD;   // 128 weight rows
  // XCD-aware bijective remap, n-tiles of one row panel consecutive
  const int total = tiles_m * tiles_n;
  const int L = blockIdx.x;
  const int q = total >> 3, r8 = total & 7;
  const int xcd = L & 7, idx8 = L >> 3;
  const int t = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + idx8;
  const int tm = t / tiles_n, tn_ = t - tm * tiles_n;
  const int m0 = tm * 64, n0 = tn_ * 128;
  int M = a.M;
  if (a.m_dev) {
    const int md = *a.m_dev + a.m_add;
    M = md < M ? md : M;
    if (m0 >= M) return;
  }
  const int N = a.N, K = a.K;
  const float* __restrict__ A = a.A;
  const float* __restrict__ A2 = a.A2;
  const float* __restrict__ W = a.W[0];
  const bool a_vec = ((a.lda & 3) == 0) && ((K & 3) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(A2) & 15) == 0);
  const bool w_vec = ((a.ldw & 3) == 0) && ((K & 3) == 0) && ((reinterpret_cast<uintptr_t>(W) & 15) == 0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  floatx16 acc[4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[mi][e] = 0.f;

  float4 ra[NV], rb[NV];
  const int nk = (K + DBK - 1) / DBK;
  // block-uniform fast path: whole tile in range and 16-byte aligned -> unguarded float4 loads with
  // per-thread base pointers hoisted out of the k loop
  const bool full = a_vec && w_vec && (m0 + 64 <= M) && (n0 + 128 <= N) && (K % DBK == 0);
  const float* pa[NV];
  const float* pb[NV];
#pragma unroll
  for (int r = 0; r < NV; ++r) {
    const int idx = tid + r * 256;
    const int row = idx / CPR, kc = (idx % CPR) << 2;
    const int ar = (m0 + (row & 63) < M) ? m0 + (row & 63) : 0, br = (n0 + row < N) ? n0 + row : 0;
    pa[r] = (row < 64 ? A : A2) + (int64_t)ar * a.lda + kc;
    pb[r] = W + (int64_t)br * a.ldw + kc;
  }
  auto fetch = [&](int kt) {
    const int k0 = kt * DBK;
    if (full) {
#pragma unroll
      for (int r = 0; r < NV; ++r) {
        ra[r] = *reinterpret_cast<const float4*>(pa[r] + k0);
        rb[r] = *reinterpret_cast<const float4*>(pb[r] + k0);
      }
      return;
    }
#pragma unroll
    for (int r = 0; r < NV; ++r) {
      const int idx = tid + r * 256;
      const int row = idx / CPR, k = k0 + ((idx % CPR) << 2);
      ra[r] = dload4(row < 64 ? A : A2, a.lda, m0 + (row & 63), M, k, K, a_vec);
      rb[r] = dload4(W, a.ldw, n0 + row, N, k, K, w_vec);
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int r = 0; r < NV; ++r) {
      const int idx = tid + r * 256;
      const int off = buf * TILE + (idx / CPR) * DLD + ((idx % CPR) << 2);
      *reinterpret_cast<float4*>(&As[off]) = ra[r];
      *reinterpret_cast<float4*>(&Bs[off]) = rb[r];
    }
  };
  fetch(0);
  stash(0);
  __syncthreads();
  const int arow = lane & 31, brow = wave * 32 + (lane & 31), koff = (lane >> 5) << 2;
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = (PRIO & 2) ? false : (kt + 1 < nk);  // PRIO&2: ablation, no loads / LDS writes in the loop
    if (more) fetch(kt + 1);
#pragma unroll
    for (int kk = 0; kk < DBK / 8; ++kk) {
      float4 af[4];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) af[mi] = *reinterpret_cast<const float4*>(&As[cur * TILE + (arow + mi * 32) * DLD + kk * 8 + koff]);
      const float4 bf = *reinterpret_cast<const float4*>(&Bs[cur * TILE + brow * DLD + kk * 8 + koff]);
      const float* afp = reinterpret_cast<const float*>(af);
      const float* bfp = reinterpret_cast<const float*>(&bf);
      if (PRIO & 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(afp[mi * 4 + tt], bfp[tt], acc[mi], 0, 0, 0);
      if (PRIO & 1) __builtin_amdgcn_s_setprio(0);
    }
    if (NBUF == 2) {
      if (more) stash(cur ^ 1);  // the other buffer was last read one k-step ago, before the previous barrier
      __syncthreads();
      cur ^= 1;
    } else {
      if (!(PRIO & 4)) __syncthreads();
      if (more) {
        stash(0);
        __syncthreads();
      }
    }
  }
  const int col = n0 + wave * 32 + (lane & 31);
  if (col >= N) return;
  const float bv = a.bias[0] ? a.bias[0][col] : 0.f;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = m0 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
      if (row >= M) continue;
      dual_store<KIND>(a, row, col, acc[mi][e] + bv, acc[mi + 2][e]);
    }
}

// small row counts: 32x32 output tile, 4 waves split K, two accumulators share the W fragment
template <int KIND>
__global__ __launch_bounds__(256) void k_gemm_dual_skinny(GemmArgs a, int tiles_m, int tiles_n) {
  __shared__ __attribute__((aligned(16))) float part[2][4][32][33];
  const int tm = blockIdx.x / tiles_n, tn_ = blockIdx.x - tm * tiles_n;
  const int m0 = tm * 32, n0 = tn_ * 32;
  int M = a.M;
  if (a.m_dev) {
    const int md = *a.m_dev + a.m_add;
    M = md < M ? md : M;
    if (m0 >= M) return;
  }
  const int N = a.N, K = a.K;
  const float* __restrict__ A = a.A;
  const float* __restrict__ A2 = a.A2;
  const float* __restrict__ W = a.W[0];
  const bool a_vec = ((a.lda & 3) == 0) && ((K & 3) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(A2) & 15) == 0);
  const bool w_vec = ((a.ldw & 3) == 0) && ((K & 3) == 0) && ((reinterpret_cast<uintptr_t>(W) & 15) == 0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, kq = (lane >> 5) << 2;
  const int chunk = ((K + 31) / 32) * 8;
  const int kbeg = wave * chunk, kend = (kbeg + chunk < K) ? kbeg + chunk : K;
  floatx16 acc, acc2;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = acc2[e] = 0.f;
  float4 av = make_float4(0.f, 0.f, 0.f, 0.f), a2v = av, bv = av;
  if (kbeg < kend) {
    av = dload4(A, a.lda, m0 + r, M, kbeg + kq, K, a_vec);
    a2v = dload4(A2, a.lda, m0 + r, M, kbeg + kq, K, a_vec);
    bv = dload4(W, a.ldw, n0 + r, N, kbeg + kq, K, w_vec);
  }
  for (int k = kbeg; k < kend; k += 8) {
    float4 an = make_float4(0.f, 0.f, 0.f, 0.f), a2n = an, bn = an;
    if (k + 8 < kend) {
      an = dload4(A, a.lda, m0 + r, M, k + 8 + kq, K, a_vec);
      a2n = dload4(A2, a.lda, m0 + r, M, k + 8 + kq, K, a_vec);
      bn = dload4(W, a.ldw, n0 + r, N, k + 8 + kq, K, w_vec);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2v.x, bv.x, acc2, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2v.y, bv.y, acc2, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2v.z, bv.z, acc2, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2v.w, bv.w, acc2, 0, 0, 0);
    av = an;
    a2v = a2n;
    bv = bn;
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int rr = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
    part[0][wave][rr][lane & 31] = acc[e];
    part[1][wave][rr][lane & 31] = acc2[e];
  }
  __syncthreads();
  const int row = tid >> 3, c0 = (tid & 7) << 2;
  const int grow = m0 + row;
  if (grow >= M) return;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = n0 + c0 + j;
    if (col >= N) continue;
    float e = ((part[0][0][row][c0 + j] + part[0][1][row][c0 + j]) + part[0][2][row][c0 + j]) + part[0][3][row][c0 + j];
    float rr = ((part[1][0][row][c0 + j] + part[1][1][row][c0 + j]) + part[1][2][row][c0 + j]) + part[1][3][row][c0 + j];
    if (a.bias[0]) e += a.bias[0][col];
    dual_store<KIND>(a, grow, col, e, rr);
  }
}

template <int KIND>
static int launch_dual_kind(const GemmArgs& a, hipStream_t stream) {
  static const bool no_skinny = getenv("TMDNET_NO_SKINNY") != nullptr;
  const int64_t big_tiles = (int64_t)((a.M + 63) / 64) * ((a.N + 127) / 128);
  if (!no_skinny && big_tiles < 256) {
    const int tiles_m = (a.M + 31) / 32, tiles_n = (a.N + 31) / 32;
    hipLaunchKernelGGL((k_gemm_dual_skinny<KIND>), dim3(tiles_m * tiles_n), dim3(256), 0, stream, a, tiles_m, tiles_n);
  } else {
    const int tiles_m = (a.M + 63) / 64, tiles_n = (a.N + 127) / 128;
    // <BK 32, 3 blocks/CU> measured best on MI355X (profiles/r01_notes.md: BK 64, LDS double buffering, s_setprio and
    // 2 / 4 blocks per CU were all slower)
    hipLaunchKernelGGL((k_gemm_dual<KIND, 32, 3, 0>), dim3(tiles_m * tiles_n), dim3(256), 0, stream, a, tiles_m, tiles_n);
  }
  return (int)hipGetLastError();
}

// kind: 0 plain, 1 silu, 2 final (rowscale = C(d), rowscale2 = C'(d)), 3 plain * rowscale
int launch_gemm_dual(const GemmArgs& a, int kind, hipStream_t stream) {
  if (a.M <= 0 || a.N <= 0) return 0;
  if (gemm_dual_sb_ok(a)) return launch_gemm_dual_sb(a, kind, stream);
  switch (kind) {
    case DUAL_PLAIN: return launch_dual_kind<DUAL_PLAIN>(a, stream);
    case DUAL_SILU: return launch_dual_kind<DUAL_SILU>(a, stream);
    case DUAL_SCALE: return launch_dual_kind<DUAL_SCALE>(a, stream);
    default: return launch_dual_kind<DUAL_FINAL>(a, stream);
  }
}

}  // namespace tn
