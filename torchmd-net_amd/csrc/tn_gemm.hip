// fp32 MFMA GEMM for gfx950, see tn_gemm.h.  CDNA4-only code: 64-wide waves, v_mfma_f32_32x32x2_f32.
//
// Tiling: 256 threads = 4 waves; block tile BM x BN, K step 32; each wave owns a
// (BM/WAVES_M) x (BN/WAVES_N) sub-tile made of 32x32 MFMA blocks (16 accumulator registers each).
// LDS image: [rows][32 + 4 pad] floats, k contiguous, so that both MFMA operands are fetched with
// conflict-free ds_read_b128 (row stride 36 dwords spreads a 16-lane group over all 64 banks).
// A lane with k-half h = lane>>5 reads k = kk*8 + h*4 + {0..3}; MFMA t of that chunk multiplies the
// k pair {kk*8+t, kk*8+4+t} - any k order is allowed as long as A and B agree.
// Block -> tile mapping is XCD-aware: the n-tiles of one m-tile (they share the A panel) get
// consecutive tile ids, and tile ids are dealt to XCDs in contiguous chunks (block b runs on XCD b%8).
// The epilogue kind is a compile-time parameter (no per-element branching); the tile loader takes an
// unguarded float4 path whenever the whole tile is in range and 16-byte aligned (block-uniform test).
#include "tn_gemm.h"

#include <cstdlib>

#include "tn_common.h"
#include "tn_gemm_epi.h"

namespace tn {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;
constexpr int LDS_LD = BK + 4;

template <int ROWS, bool FAST>
__device__ __forceinline__ void load_panel(float4 (&reg)[ROWS * 8 / 256], const float* __restrict__ base, int64_t ld, int row0,
                                           int nrows, int k0, int K, bool vec_ok, int tid) {
  constexpr int NV = ROWS * 8 / 256;
#pragma unroll
  for (int r = 0; r < NV; ++r) {
    const int idx = tid + r * 256;
    const int row = idx >> 3;
    const int k = k0 + ((idx & 7) << 2);
    const int grow = row0 + row;
    if (FAST) {
      reg[r] = *reinterpret_cast<const float4*>(base + (int64_t)grow * ld + k);
    } else {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (grow < nrows) {
        const float* p = base + (int64_t)grow * ld + k;
        if (vec_ok) {
          if (k < K) v = *reinterpret_cast<const float4*>(p);
        } else {
          if (k + 0 < K) v.x = p[0];
          if (k + 1 < K) v.y = p[1];
          if (k + 2 < K) v.z = p[2];
          if (k + 3 < K) v.w = p[3];
        }
      }
      reg[r] = v;
    }
  }
}

template <int ROWS>
__device__ __forceinline__ void store_panel(const float4 (&reg)[ROWS * 8 / 256], float* lds, int tid) {
  constexpr int NV = ROWS * 8 / 256;
#pragma unroll
  for (int r = 0; r < NV; ++r) {
    const int idx = tid + r * 256;
    *reinterpret_cast<float4*>(&lds[(idx >> 3) * LDS_LD + ((idx & 7) << 2)]) = reg[r];
  }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int EPI>
__global__ __launch_bounds__(256) void k_gemm_nt(GemmArgs a, int tiles_m, int tiles_n) {
  static_assert(WAVES_M * WAVES_N == 4, "4 waves per block");
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int MI = WTM / 32, NI = WTN / 32;
  static_assert(MI >= 1 && NI >= 1, "wave tile must hold at least one 32x32 MFMA block");
  __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LDS_LD];
  float* As = smem;
  float* Bs = smem + BM * LDS_LD;

  // ---- XCD-aware bijective block -> tile remap
  const int total = tiles_m * tiles_n * a.groups;
  const int L = blockIdx.x;
  const int q = total >> 3, r8 = total & 7;
  const int xcd = L & 7, idx8 = L >> 3;
  const int t = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + idx8;
  const int per_group = tiles_m * tiles_n;
  const int g = t / per_group;
  const int rem = t - g * per_group;
  const int tm = rem / tiles_n, tn_ = rem - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn_ * BN;

  const float* __restrict__ A = a.A + a.a_off[g];
  const float* __restrict__ W = a.W[g];
  int M = a.M;
  if (a.m_dev) {
    const int md = *a.m_dev + a.m_add;
    M = md < M ? md : M;
    if (m0 >= M) return;  // whole block: before any barrier
  }
  const int N = a.N, K = a.K;
  const bool a_vec = ((a.lda & 3) == 0) && ((K & 3) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
  const bool w_vec = ((a.ldw & 3) == 0) && ((K & 3) == 0) && ((reinterpret_cast<uintptr_t>(W) & 15) == 0);
  const bool a_full = a_vec && (m0 + BM <= M);
  const bool w_full = w_vec && (n0 + BN <= N);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  floatx16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  constexpr int NVA = BM * 8 / 256, NVB = BN * 8 / 256;
  float4 ra[NVA], rb[NVB];
  const int nk = (K + BK - 1) / BK;

  auto fetch = [&](int kt) {
    const int k0 = kt * BK;
    const bool kfull = k0 + BK <= K;
    if (a_full && kfull) load_panel<BM, true>(ra, A, a.lda, m0, M, k0, K, true, tid);
    else load_panel<BM, false>(ra, A, a.lda, m0, M, k0, K, a_vec, tid);
    if (w_full && kfull) load_panel<BN, true>(rb, W, a.ldw, n0, N, k0, K, true, tid);
    else load_panel<BN, false>(rb, W, a.ldw, n0, N, k0, K, w_vec, tid);
  };

  fetch(0);
  store_panel<BM>(ra, As, tid);
  store_panel<BN>(rb, Bs, tid);
  __syncthreads();

  const int arow = wm * WTM + (lane & 31);
  const int brow = wn * WTN + (lane & 31);
  const int koff = (lane >> 5) << 2;

  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more) fetch(kt + 1);
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      float4 af[MI], bf[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) af[mi] = *reinterpret_cast<const float4*>(&As[(arow + mi * 32) * LDS_LD + kk * 8 + koff]);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) bf[ni] = *reinterpret_cast<const float4*>(&Bs[(brow + ni * 32) * LDS_LD + kk * 8 + koff]);
      const float* afp = reinterpret_cast<const float*>(af);
      const float* bfp = reinterpret_cast<const float*>(bf);
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(afp[mi * 4 + tt], bfp[ni * 4 + tt], acc[mi][ni], 0, 0, 0);
          }
    }
    __syncthreads();
    if (more) {
      store_panel<BM>(ra, As, tid);
      store_panel<BN>(rb, Bs, tid);
      __syncthreads();
    }
  }

  // ---- fused epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const float* __restrict__ bias = a.bias[g];
  float* __restrict__ C = a.C + a.c_off[g];
  float* __restrict__ pre = a.pre ? a.pre + a.pre_off[g] : nullptr;
  const float* __restrict__ aux = a.aux ? a.aux + a.aux_off[g] : nullptr;
  const int flags = a.flags;
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int col = n0 + wn * WTN + ni * 32 + (lane & 31);
    if (col >= N) continue;
    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * WTM + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (row >= M) continue;
        float v = acc[mi][ni][e] + bv;
        float* cp = C + (int64_t)row * a.ldc + col;
        if (EPI == EPI_PLAIN) {
          *cp = v;
        } else if (EPI == EPI_SILU_PRE) {
          pre[(int64_t)row * a.ldpre + col] = v;
          *cp = fast_silu(v);
        } else if (EPI == EPI_SILU_PRE_ROWSCALE) {
          pre[(int64_t)row * a.ldpre + col] = v;
          *cp = fast_silu(v) * a.rowscale[row];
        } else if (EPI == EPI_MULAUX_PRE) {
          pre[(int64_t)row * a.ldpre + col] = v;
          *cp = v * aux[(int64_t)row * a.ldaux + col];
        } else if (EPI == EPI_MULDSILU) {
          *cp = v * fast_silu_grad(aux[(int64_t)row * a.ldaux + col]);
        } else if (EPI == EPI_ACCUM) {
          *cp = v + *cp;
        } else {
          if (pre) pre[(int64_t)row * a.ldpre + col] = v;
          if (flags & GEMM_ACT_SILU) v = fast_silu(v);
          if (flags & GEMM_ROWSCALE) v *= a.rowscale[row];
          if (flags & GEMM_MUL_AUX) v *= aux[(int64_t)row * a.ldaux + col];
          if (flags & GEMM_MUL_DSILU_AUX) v *= fast_silu_grad(aux[(int64_t)row * a.ldaux + col]);
          if (flags & GEMM_ACCUM) v += *cp;
          *cp = v;
        }
      }
    }
  }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int EPI>
static int launch_one(const GemmArgs& a, hipStream_t stream) {
  int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
  int total = tiles_m * tiles_n * a.groups;
  if (total <= 0) return 0;
  hipLaunchKernelGGL((k_gemm_nt<BM, BN, WAVES_M, WAVES_N, EPI>), dim3(total), dim3(256), 0, stream, a, tiles_m, tiles_n);
  return (int)hipGetLastError();
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
static int launch_variant(const GemmArgs& a, hipStream_t stream) {
  switch (epi_kind(a)) {
    case EPI_PLAIN: return launch_one<BM, BN, WAVES_M, WAVES_N, EPI_PLAIN>(a, stream);
    case EPI_SILU_PRE: return launch_one<BM, BN, WAVES_M, WAVES_N, EPI_SILU_PRE>(a, stream);
    case EPI_SILU_PRE_ROWSCALE: return launch_one<BM, BN, WAVES_M, WAVES_N, EPI_SILU_PRE_ROWSCALE>(a, stream);
    case EPI_MULAUX_PRE: return launch_one<BM, BN, WAVES_M, WAVES_N, EPI_MULAUX_PRE>(a, stream);
    case EPI_MULDSILU: return launch_one<BM, BN, WAVES_M, WAVES_N, EPI_MULDSILU>(a, stream);
    case EPI_ACCUM: return launch_one<BM, BN, WAVES_M, WAVES_N, EPI_ACCUM>(a, stream);
    default: return launch_one<BM, BN, WAVES_M, WAVES_N, EPI_GENERIC>(a, stream);
  }
}

int launch_gemm(const GemmArgs& a, hipStream_t stream) {
  if (a.M <= 0 || a.N <= 0) return 0;
  if (gemm_sb1_ok(a)) return launch_gemm_sb1(a, stream);  // bf16 matrix pipe, exact 3-way split (tn_gemm_sb1.hip)
  // few tiles: the chip would be mostly idle and every launch would cost K/32 dependent iterations ->
  // latency-oriented split-K kernel (single molecules, small MD systems)
  {
    static const bool no_skinny = getenv("TMDNET_NO_SKINNY") != nullptr;
    const int bn = (a.N % 128 == 0 || a.N > 192) ? 128 : (a.N > 32 ? 64 : 32);
    const int64_t tiles = (int64_t)((a.M + 127) / 128) * ((a.N + bn - 1) / bn) * a.groups;
    if (!no_skinny && tiles < 256) return launch_gemm_skinny(a, stream);
  }
  if (a.N % 128 == 0 || a.N > 192) return launch_variant<128, 128, 2, 2>(a, stream);
  if (a.N > 32) return launch_variant<128, 64, 2, 2>(a, stream);
  return launch_variant<128, 32, 4, 1>(a, stream);
}

}  // namespace tn
