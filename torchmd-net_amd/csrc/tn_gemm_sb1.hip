// Split-bf16 MFMA GEMM with the node-side epilogues (gfx950): C[g] = epilogue(A[g] W[g]^T), fp32 in / fp32 out, every
// fp32 product evaluated as six bf16 MFMA products of an exact 3-way split (arithmetic: tn_gemm_sb.hip).
//
// Serves the per-atom contractions of the path -- the 9-component tensor linears (grouped: 9 groups, 3 weight sets;
// reference tensornet.py:595-617, 752-754, 808-810), the gate MLPs (:590-593), readout linear (:398), output MLP
// (models/utils.py:552-580) and their transposed-weight counterparts in the reverse pass -- plus the plain pair-row
// GEMMs when forces are not requested.  With K = 128..384 these are short contractions: on the fp32 matrix pipe they
// were MFMA-bound (4.8 GF tensor linear = 31 us at the fp32 MFMA peak), on the bf16 pipe they are streaming kernels
// (150 MB in + out), so the kernel is built around the memory path:
//   * persistent blocks, 128 x 128 tile, K-step 16, LDS double buffer, one barrier per step; the split of the next
//     chunk and its LDS stores ride between the MFMAs of the current one (same pipeline as k_gemm_dual_sb2);
//   * epilogue transposed through wave-private LDS so every global access is 16 bytes per lane / 128 contiguous
//     bytes per row (C, saved pre-activation, aux operand, accumulate-into-C all as float4).
// Shapes it does not take (K % 16, N % 4, unaligned leading dimensions, < 128 tiles) stay on k_gemm_nt / k_gemm_skinny.
#include <cstdlib>

#include "tn_common.h"
#include "tn_gemm_epi.h"
#include "tn_gemm_sb.h"

namespace tn {

constexpr int SB1_STAGE = 6 * SB2_PLANE;  // A h/m/l + W h/m/l = 24 KB

template <int EPI>
__device__ __forceinline__ float4 epi4(const GemmArgs& a, float* __restrict__ C, float* __restrict__ pre,
                                       const float* __restrict__ aux, int row, int col, float4 v) {
  // v = acc + bias for 4 consecutive columns of one row; returns the value to store in C
  auto ld4 = [](const float* p) { return *reinterpret_cast<const float4*>(p); };
  auto silu4 = [](float4 x) { return make_float4(fast_silu(x.x), fast_silu(x.y), fast_silu(x.z), fast_silu(x.w)); };
  auto mul4 = [](float4 x, float4 y) { return make_float4(x.x * y.x, x.y * y.y, x.z * y.z, x.w * y.w); };
  auto dsilu4 = [](float4 x) {
    return make_float4(fast_silu_grad(x.x), fast_silu_grad(x.y), fast_silu_grad(x.z), fast_silu_grad(x.w));
  };
  if (EPI == EPI_PLAIN) return v;
  if (EPI == EPI_SILU_PRE) {
    *reinterpret_cast<float4*>(pre + (int64_t)row * a.ldpre + col) = v;
    return silu4(v);
  }
  if (EPI == EPI_MULAUX_PRE) {
    *reinterpret_cast<float4*>(pre + (int64_t)row * a.ldpre + col) = v;
    return mul4(v, ld4(aux + (int64_t)row * a.ldaux + col));
  }
  if (EPI == EPI_MULDSILU) return mul4(v, dsilu4(ld4(aux + (int64_t)row * a.ldaux + col)));
  // generic: run-time flags
  const int flags = a.flags;
  if (pre) *reinterpret_cast<float4*>(pre + (int64_t)row * a.ldpre + col) = v;
  if (flags & GEMM_ACT_SILU) v = silu4(v);
  if (flags & GEMM_ROWSCALE) {
    const float r = a.rowscale[row];
    v = make_float4(v.x * r, v.y * r, v.z * r, v.w * r);
  }
  if (flags & GEMM_MUL_AUX) v = mul4(v, ld4(aux + (int64_t)row * a.ldaux + col));
  if (flags & GEMM_MUL_DSILU_AUX) v = mul4(v, dsilu4(ld4(aux + (int64_t)row * a.ldaux + col)));
  if (flags & GEMM_ACCUM) {
    const float4 o = ld4(C + (int64_t)row * a.ldc + col);
    v = make_float4(v.x + o.x, v.y + o.y, v.z + o.z, v.w + o.w);
  }
  return v;
}

template <int EPI>
__global__ __launch_bounds__(256, 3) void k_gemm_sb1(GemmArgs a, int tiles_m, int tiles_n) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * SB1_STAGE];
  const int total = tiles_m * tiles_n * a.groups;
  int M = a.M;
  if (a.m_dev) {
    const int md = *a.m_dev + a.m_add;
    M = md < M ? md : M;
  }
  const int N = a.N, K = a.K;
  const int nk = K >> 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int srow = tid >> 1, skh = tid & 1;  // staging role: row of the tile, k-half (8 consecutive k)
  const int soff = sb_piece(srow, skh);
  const int ra_ = wm * 64 + (lane & 31), rb_ = wn * 64 + (lane & 31), kh = lane >> 5;
  const int foa = sb_piece(ra_, kh);
  const int fob = 3 * SB2_PLANE + sb_piece(rb_, kh);

  for (int vt = blockIdx.x; vt < total; vt += gridDim.x) {
    // tile order: the groups of one row panel are consecutive (they read neighbouring columns of the same atom rows)
    const int t = xcd_chunk(vt, total);
    const int tn_ = t % tiles_n, tg = t / tiles_n;
    const int g = tg % a.groups, tm = tg / a.groups;
    const int m0 = tm * 128, n0 = tn_ * 128;
    if (m0 >= M) continue;
    __syncthreads();  // the previous tile's epilogue region / last chunk may still be in use by a slower wave
    const int grow = (m0 + srow < M) ? m0 + srow : 0;  // rows past M: any valid row (their outputs are not stored)
    const float* pa = a.A + a.a_off[g] + (int64_t)grow * a.lda + skh * 8;
    const uint4* pw = reinterpret_cast<const uint4*>(a.Wsbg[g]) + (int64_t)tn_ * nk * 768 + tid;  // 768 uint4 per chunk

    float4 a0, a1;
    uint4 w0, w1, w2;
#define SB1_FETCH(kt)                                          \
  a0 = *reinterpret_cast<const float4*>(pa + (kt) * 16);      \
  a1 = *reinterpret_cast<const float4*>(pa + (kt) * 16 + 4);  \
  w0 = pw[(int64_t)(kt) * 768];                                \
  w1 = pw[(int64_t)(kt) * 768 + 256];                          \
  w2 = pw[(int64_t)(kt) * 768 + 512];

    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    {  // prologue: chunk 0 -> buffer 0, chunk 1 -> registers
      SB1_FETCH(0)
      uint4 h, m, l;
      split8(a0, a1, h, m, l);
      *reinterpret_cast<uint4*>(smem + 0 * SB2_PLANE + soff) = h;
      *reinterpret_cast<uint4*>(smem + 1 * SB2_PLANE + soff) = m;
      *reinterpret_cast<uint4*>(smem + 2 * SB2_PLANE + soff) = l;
      *reinterpret_cast<uint4*>(smem + 3 * SB2_PLANE + soff) = w0;
      *reinterpret_cast<uint4*>(smem + 4 * SB2_PLANE + soff) = w1;
      *reinterpret_cast<uint4*>(smem + 5 * SB2_PLANE + soff) = w2;
      if (nk > 1) { SB1_FETCH(1) }
    }
    __syncthreads();

#define SB1_LOAD_B()                                                                                       \
  bf16x8 bf[2][3];                                                                                         \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) _Pragma("unroll") for (int p = 0; p < 3; ++p) bf[j][p] =   \
      *reinterpret_cast<const bf16x8*>(cur + fob + p * SB2_PLANE + j * 1024);                              \
  bf16x8 af[2][3];
#define SB1_AF(i) \
  _Pragma("unroll") for (int p = 0; p < 3; ++p) af[i][p] = *reinterpret_cast<const bf16x8*>(cur + foa + p * SB2_PLANE + (i) * 1024);
// one product of the split on all four accumulators of the wave: consecutive MFMAs never touch the same accumulator, and an
// accumulator is revisited after four issues (the six products of a k-step in the order of k_gemm_sb: 02 20 11 01 10 00)
#define SB1_MMA4(pa_, pb_)                                                                          \
  acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][pa_], bf[0][pb_], acc[0][0], 0, 0, 0); \
  acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][pa_], bf[1][pb_], acc[0][1], 0, 0, 0); \
  acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][pa_], bf[0][pb_], acc[1][0], 0, 0, 0); \
  acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][pa_], bf[1][pb_], acc[1][1], 0, 0, 0);

    // steady state: chunk kt+1 is in the registers; branch-free body (the chunk index of the next loads is clamped)
    for (int kt = 0; kt + 1 < nk; ++kt) {
      const unsigned char* cur = smem + (kt & 1) * SB1_STAGE;
      unsigned char* nxt = smem + ((kt + 1) & 1) * SB1_STAGE;
      const int kf = (kt + 2 < nk) ? kt + 2 : nk - 1;
      SB1_LOAD_B()
      uint4 h, m, l;
      SB1_AF(0)
      SB1_AF(1)
      SB1_MMA4(0, 2)
      split2(a0.x, a0.y, h.x, m.x, l.x);
      split2(a0.z, a0.w, h.y, m.y, l.y);
      SB1_MMA4(2, 0)
      split2(a1.x, a1.y, h.z, m.z, l.z);
      split2(a1.z, a1.w, h.w, m.w, l.w);
      SB1_MMA4(1, 1)
      *reinterpret_cast<uint4*>(nxt + 0 * SB2_PLANE + soff) = h;
      *reinterpret_cast<uint4*>(nxt + 1 * SB2_PLANE + soff) = m;
      *reinterpret_cast<uint4*>(nxt + 2 * SB2_PLANE + soff) = l;
      SB1_MMA4(0, 1)
      *reinterpret_cast<uint4*>(nxt + 3 * SB2_PLANE + soff) = w0;
      *reinterpret_cast<uint4*>(nxt + 4 * SB2_PLANE + soff) = w1;
      *reinterpret_cast<uint4*>(nxt + 5 * SB2_PLANE + soff) = w2;
      SB1_MMA4(1, 0)
      SB1_FETCH(kf)
      SB1_MMA4(0, 0)
      __syncthreads();
    }
    {  // last chunk: multiply only
      const unsigned char* cur = smem + ((nk - 1) & 1) * SB1_STAGE;
      SB1_LOAD_B()
      SB1_AF(0)
      SB1_AF(1)
      SB1_MMA4(0, 2) SB1_MMA4(2, 0) SB1_MMA4(1, 1) SB1_MMA4(0, 1) SB1_MMA4(1, 0) SB1_MMA4(0, 0)
    }
#undef SB1_LOAD_B
#undef SB1_AF
#undef SB1_MMA4
#undef SB1_FETCH

    // epilogue through wave-private LDS (4 KB per wave): 16 bytes per lane, 8 rows x 128 contiguous bytes per access
    __syncthreads();  // every wave is done reading the last chunk
    float* xv = reinterpret_cast<float*>(smem) + wave * 1024;
    float* Cg = a.C + a.c_off[g];
    float* preg = a.pre ? a.pre + a.pre_off[g] : nullptr;
    const float* auxg = a.aux ? a.aux + a.aux_off[g] : nullptr;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int cb = n0 + wn * 64 + j * 32;
      const int col = cb + (lane & 31);
      const float bv = (a.bias[g] && col < N) ? a.bias[g][col] : 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int rb = m0 + wm * 64 + i * 32;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int rl = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
          xv[rl * 32 + (lane & 31)] = acc[i][j][e] + bv;
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int rl = rr * 8 + (lane >> 3), c4 = (lane & 7) * 4;
          const float4 v = *reinterpret_cast<const float4*>(xv + rl * 32 + c4);
          const int row = rb + rl;
          if (row < M && cb + c4 < N)  // N % 4 == 0 (gemm_sb1_ok)
            *reinterpret_cast<float4*>(Cg + (int64_t)row * a.ldc + cb + c4) = epi4<EPI>(a, Cg, preg, auxg, row, cb + c4, v);
        }
      }
    }
  }  // tile loop
}


// ---- half tiles (round 6): 64 x 64 outputs per block for launches with fewer 128 x 128 tiles than CUs (N <= 128 at M = 16 384:
// 128 tiles on 256 CUs), so that every CU works.
// Waves 0, 1 stage A (row, k-half: two 16-byte loads, split), waves 2, 3 stage W (three 16-byte loads of the fragment image,
// copied): one instruction stream, the roles differ in pointers and in a branch around arithmetic only.
constexpr int SBH_PLANE = 64 * 32;        // one bf16 plane of a [64 rows][16 k] chunk
constexpr int SBH_STAGE = 6 * SBH_PLANE;  // 12 KB

template <int EPI>
__global__ __launch_bounds__(256, 6) void k_gemm_sb1h(GemmArgs a, int tiles_m, int tiles_n) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * SBH_STAGE];  // 24 KB (the epilogue's 4 x 4 KB alias it)
  const int total = tiles_m * tiles_n * a.groups;
  int M = a.M;
  if (a.m_dev) {
    const int md = *a.m_dev + a.m_add;
    M = md < M ? md : M;
  }
  const int N = a.N, K = a.K;
  const int nk = K >> 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const bool role_a = wave_u < 2;
  const int wm = wave & 1, wn = wave >> 1;
  const int st = tid & 127, srow = st >> 1, skh = st & 1;  // staging role: row (A) / column (W) of the tile, k-half
  const int soff = (role_a ? 0 : 3 * SBH_PLANE) + sb_piece(srow, skh);
  const int ra_ = wm * 32 + (lane & 31), rb_ = wn * 32 + (lane & 31), kh = lane >> 5;
  const int foa = sb_piece(ra_, kh);
  const int fob = 3 * SBH_PLANE + sb_piece(rb_, kh);

  for (int vt = blockIdx.x; vt < total; vt += gridDim.x) {
    const int t = xcd_chunk(vt, total);
    const int tn_ = t % tiles_n, tg = t / tiles_n;
    const int g = tg % a.groups, tm = tg / a.groups;
    const int m0 = tm * 64, n0 = tn_ * 64;
    if (m0 >= M || n0 >= N) continue;
    __syncthreads();  // the previous tile's epilogue region / last chunk may still be in use by a slower wave
    const int grow = (m0 + srow < M) ? m0 + srow : 0;  // rows past M: any valid row (their outputs are not stored)
    // three 16-byte pieces per thread and chunk: A: 8 consecutive k of a row (two pieces, the third a repeat); W: the h / m / l
    // pieces of (column, k-half) in the fragment image (768 uint4 per 128-column tile and chunk, plane stride 256)
    const float4* p0;
    int64_t step0, d1, d2;  // distance between chunks, and of pieces 1, 2 from piece 0 (in float4)
    if (role_a) {
      p0 = reinterpret_cast<const float4*>(a.A + a.a_off[g] + (int64_t)grow * a.lda + skh * 8);
      step0 = 4; d1 = 1; d2 = 0;
    } else {
      p0 = reinterpret_cast<const float4*>(a.Wsbg[g]) + (int64_t)(n0 >> 7) * nk * 768 + ((n0 & 64) + srow) * 2 + skh;
      step0 = 768; d1 = 256; d2 = 512;
    }
    float4 r0, r1, r2;
#define SBH_FETCH(kt)           \
  r0 = p0[(kt) * step0];        \
  r1 = p0[(kt) * step0 + d1];   \
  r2 = p0[(kt) * step0 + d2];
#define SBH_PUT(buf)                                                                       \
  {                                                                                        \
    uint4 h, m, l;                                                                         \
    if (role_a) split8(r0, r1, h, m, l);                                                   \
    else {                                                                                 \
      h = *reinterpret_cast<uint4*>(&r0); m = *reinterpret_cast<uint4*>(&r1); l = *reinterpret_cast<uint4*>(&r2); \
    }                                                                                      \
    *reinterpret_cast<uint4*>((buf) + 0 * SBH_PLANE + soff) = h;                           \
    *reinterpret_cast<uint4*>((buf) + 1 * SBH_PLANE + soff) = m;                           \
    *reinterpret_cast<uint4*>((buf) + 2 * SBH_PLANE + soff) = l;                           \
  }
    floatx16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    SBH_FETCH(0)
    SBH_PUT(smem)
    if (nk > 1) { SBH_FETCH(1) }
    __syncthreads();
#define SBH_MMA(cur)                                                                                       \
  {                                                                                                        \
    bf16x8 af[3], bf[3];                                                                                   \
    _Pragma("unroll") for (int p = 0; p < 3; ++p) {                                                        \
      af[p] = *reinterpret_cast<const bf16x8*>((cur) + foa + p * SBH_PLANE);                               \
      bf[p] = *reinterpret_cast<const bf16x8*>((cur) + fob + p * SBH_PLANE);                               \
    }                                                                                                      \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[2], acc, 0, 0, 0);                             \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], bf[0], acc, 0, 0, 0);                             \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[1], acc, 0, 0, 0);                             \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[1], acc, 0, 0, 0);                             \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[0], acc, 0, 0, 0);                             \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[0], acc, 0, 0, 0);                             \
  }
    for (int kt = 0; kt + 1 < nk; ++kt) {
      const unsigned char* cur = smem + (kt & 1) * SBH_STAGE;
      unsigned char* nxt = smem + ((kt + 1) & 1) * SBH_STAGE;
      const int kf = (kt + 2 < nk) ? kt + 2 : nk - 1;
      SBH_PUT(nxt)       // chunk kt + 1 (registers) -> the other buffer
      SBH_FETCH(kf)      // chunk kt + 2 on its way
      SBH_MMA(cur)
      __syncthreads();
    }
    SBH_MMA(smem + ((nk - 1) & 1) * SBH_STAGE)
#undef SBH_MMA
#undef SBH_PUT
#undef SBH_FETCH

    // epilogue through wave-private LDS (4 KB per wave): 16 bytes per lane, 8 rows x 128 contiguous bytes per access
    __syncthreads();  // every wave is done reading the last chunk
    float* xv = reinterpret_cast<float*>(smem) + wave * 1024;
    float* Cg = a.C + a.c_off[g];
    float* preg = a.pre ? a.pre + a.pre_off[g] : nullptr;
    const float* auxg = a.aux ? a.aux + a.aux_off[g] : nullptr;
    const int cb = n0 + wn * 32, rb = m0 + wm * 32;
    const int col = cb + (lane & 31);
    const float bv = (a.bias[g] && col < N) ? a.bias[g][col] : 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int rl = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
      xv[rl * 32 + (lane & 31)] = acc[e] + bv;
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int rl = rr * 8 + (lane >> 3), c4 = (lane & 7) * 4;
      const float4 v = *reinterpret_cast<const float4*>(xv + rl * 32 + c4);
      const int row = rb + rl;
      if (row < M && cb + c4 < N)  // N % 4 == 0 (gemm_sb1_ok)
        *reinterpret_cast<float4*>(Cg + (int64_t)row * a.ldc + cb + c4) = epi4<EPI>(a, Cg, preg, auxg, row, cb + c4, v);
    }
  }  // tile loop
}

static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

bool gemm_sb1_ok(const GemmArgs& a) {
  static const bool off = getenv("TMDNET_NO_SPLIT_BF16") != nullptr;  // developer switch: fp32 MFMA everywhere
  if (off || a.K <= 0 || (a.K & 15) || (a.N & 3) || (a.lda & 3) || (a.ldc & 3) || !al16(a.A) || !al16(a.C)) return false;
  if (a.pre && ((a.ldpre & 3) || !al16(a.pre))) return false;
  if (a.aux && ((a.ldaux & 3) || !al16(a.aux))) return false;
  for (int g = 0; g < a.groups; ++g) {
    if (!a.Wsbg[g] || (a.a_off[g] & 3) || (a.c_off[g] & 3)) return false;
    if (a.pre && (a.pre_off[g] & 3)) return false;
    if (a.aux && (a.aux_off[g] & 3)) return false;
  }
  return (int64_t)((a.M + 127) / 128) * ((a.N + 127) / 128) * a.groups >= 128;  // small launches: split-K fp32 kernel
}

int launch_gemm_sb1(const GemmArgs& a, hipStream_t stream) {
  const int tiles_m = (a.M + 127) / 128, tiles_n = (a.N + 127) / 128;
  static const int n_cu = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n;
  }();
  const int total = tiles_m * tiles_n * a.groups;
  static const int bpc = getenv("TMDNET_GEMM_BPC") ? atoi(getenv("TMDNET_GEMM_BPC")) : 3;  // developer switch
  // fewer 128 x 128 tiles than CUs (N <= 128 at M = 16 384): 64 x 64 tiles, every CU busy (22.9 -> 19.1 us for 16384 x 128 x 384;
  // from one tile per CU on the two kernels measure the same: these launches are at their streaming floor, ~3.8 TB/s of operands
  // and both outputs behind a 3 - 4 us launch).  Developer switch TMDNET_GEMM_HALF_BELOW: tiles per CU below which it is taken, 0 = never.
  static const int half_below = getenv("TMDNET_GEMM_HALF_BELOW") ? atoi(getenv("TMDNET_GEMM_HALF_BELOW")) : 1;
  if (total < half_below * n_cu) {
    const int hm = (a.M + 63) / 64, hn = (a.N + 63) / 64;
    const int htotal = hm * hn * a.groups;
    const dim3 hgrid(htotal < 6 * n_cu ? htotal : 6 * n_cu), block(256);
    switch (epi_kind(a)) {
      case EPI_PLAIN: hipLaunchKernelGGL((k_gemm_sb1h<EPI_PLAIN>), hgrid, block, 0, stream, a, hm, hn); break;
      case EPI_SILU_PRE: hipLaunchKernelGGL((k_gemm_sb1h<EPI_SILU_PRE>), hgrid, block, 0, stream, a, hm, hn); break;
      case EPI_MULAUX_PRE: hipLaunchKernelGGL((k_gemm_sb1h<EPI_MULAUX_PRE>), hgrid, block, 0, stream, a, hm, hn); break;
      case EPI_MULDSILU: hipLaunchKernelGGL((k_gemm_sb1h<EPI_MULDSILU>), hgrid, block, 0, stream, a, hm, hn); break;
      default: hipLaunchKernelGGL((k_gemm_sb1h<EPI_GENERIC>), hgrid, block, 0, stream, a, hm, hn); break;
    }
    return (int)hipGetLastError();
  }
  const dim3 grid(total < bpc * n_cu ? total : bpc * n_cu), block(256);  // persistent: 3 blocks per CU
  switch (epi_kind(a)) {
    case EPI_PLAIN: hipLaunchKernelGGL((k_gemm_sb1<EPI_PLAIN>), grid, block, 0, stream, a, tiles_m, tiles_n); break;
    case EPI_SILU_PRE: hipLaunchKernelGGL((k_gemm_sb1<EPI_SILU_PRE>), grid, block, 0, stream, a, tiles_m, tiles_n); break;
    case EPI_MULAUX_PRE: hipLaunchKernelGGL((k_gemm_sb1<EPI_MULAUX_PRE>), grid, block, 0, stream, a, tiles_m, tiles_n); break;
    case EPI_MULDSILU: hipLaunchKernelGGL((k_gemm_sb1<EPI_MULDSILU>), grid, block, 0, stream, a, tiles_m, tiles_n); break;
    default: hipLaunchKernelGGL((k_gemm_sb1<EPI_GENERIC>), grid, block, 0, stream, a, tiles_m, tiles_n); break;
  }
  return (int)hipGetLastError();
}

}  // namespace tn
