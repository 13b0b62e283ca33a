// Split-bf16 ("3 x bf16 = fp32") MFMA GEMM for the pair-row MLPs: value and d/dd tangent in one pass (gfx950).
//
// gfx950's bf16 matrix pipe is 16x faster than its fp32 one (v_mfma_f32_32x32x16_bf16: 32x32x16 MACs in 32
// cycles; v_mfma_f32_32x32x2_f32: 32x32x2 in 64).  An fp32 number is EXACTLY the sum of three bf16 numbers
// (8 + 8 + 8 significant bits), x = h + m + l, so
//     a * b = ah bh + (ah bm + am bh) + (am bm + ah bl + al bh) + O(2^-24 |a b|)
// and six bf16 MFMAs with fp32 accumulation reproduce the fp32 product to fp32 rounding (measured: error vs an
// fp64 contraction 0.4x that of the fp32 fmaf chain, profiles/r01_notes.md) at 6/16 of the fp32 MFMA cost.
//
// Operands: activations stay fp32 in HBM and are split while they are staged into LDS (each element once per
// block); weights are split on the host at tmdnet_finalize_params into the LDS tile image
//     Wsb[n_tile][k_chunk][plane 0..2][128 rows][16 k]   bf16
// so a block's weight chunk is one contiguous 12 KB read.
//
// Block = 4 waves (2 x 2), tile 128 pair rows x 128 outputs, K-step 16; wave tile 64 x 64 for BOTH the value and
// the tangent (8 accumulator blocks = 128 registers), 2 persistent blocks per CU.  See k_gemm_dual_sb2 for the
// pipeline.  Same interface and epilogues as k_gemm_dual (tn_gemm_dual.hip), which remains the path for small row
// counts and for shapes this kernel does not take (K % 16 != 0, N % 4 != 0, unaligned rows).
#include <cstdlib>
#include <cstring>

#include "tn_common.h"
#include "tn_gemm_epi.h"
#include "tn_gemm_sb.h"

namespace tn {

// ---- software-pipelined variant: LDS double buffer (unpadded planes, XOR-swizzled 16-byte pieces), ONE barrier per
// K-step.  Within a step a wave multiplies chunk kt out of buffer kt&1 and, between the MFMAs of the second half,
// splits chunk kt+1 (loaded during the previous step) into the other buffer, then issues the loads of chunk kt+2
// into the same registers.  The VALU / LDS-write work rides in the issue slots the 32-cycle MFMAs leave free.
constexpr int SB2_STAGE = 9 * SB2_PLANE;   // 36864 bytes

template <int KIND>
__global__ __launch_bounds__(256, 2) void k_gemm_dual_sb2(GemmArgs a, int tiles_m, int tiles_n) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * SB2_STAGE];
  const int total = tiles_m * tiles_n;
  int M = a.M;
  if (a.m_dev) {
    const int md = *a.m_dev + a.m_add;
    M = md < M ? md : M;
  }
  const int N = a.N, K = a.K;
  const int nk = K >> 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int srow = tid >> 1, skh = tid & 1;
  // persistent blocks (grid = 2 per CU); starting the second block of a CU half a tile late changed nothing
  for (int vt = blockIdx.x; vt < total; vt += gridDim.x) {
  const int t = xcd_chunk(vt, total);
  const int tm = t / tiles_n, tn_ = t - tm * tiles_n;
  const int m0 = tm * 128, n0 = tn_ * 128;
  if (m0 >= M) continue;
  __syncthreads();  // the previous tile's last chunk may still be read by a slower wave
  const int grow = (m0 + srow < M) ? m0 + srow : 0;
  const float* pa = a.A + (int64_t)grow * a.lda + skh * 8;
  const float* pa2 = a.A2 + (int64_t)grow * a.lda + skh * 8;
  const uint4* pw = reinterpret_cast<const uint4*>(a.Wsb) + (int64_t)tn_ * nk * 768 + tid;
  // piece (row r, k-half h) lives at 16 * (2 r + (h ^ f(r))), f(r) = ((r + 4) >> 3) & 1.  ds_read_b128 serves the lane
  // groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32) in one LDS cycle each over 64 banks: with this f every group
  // touches 16 distinct 16-byte slots; ds_write_b128 (8 contiguous lanes, 32 banks) is conflict-free as well.
  const int soff = (2 * srow + (skh ^ (((srow + 4) >> 3) & 1))) * 16;

  float4 a0, a1, t0, t1;
  uint4 w0, w1, w2;
#define SB2_FETCH(kt)                                                  \
  a0 = *reinterpret_cast<const float4*>(pa + (kt) * 16);              \
  a1 = *reinterpret_cast<const float4*>(pa + (kt) * 16 + 4);          \
  t0 = *reinterpret_cast<const float4*>(pa2 + (kt) * 16);             \
  t1 = *reinterpret_cast<const float4*>(pa2 + (kt) * 16 + 4);         \
  w0 = pw[(int64_t)(kt) * 768];                                        \
  w1 = pw[(int64_t)(kt) * 768 + 256];                                  \
  w2 = pw[(int64_t)(kt) * 768 + 512];

  floatx16 acc[2][2][2];
#pragma unroll
  for (int v = 0; v < 2; ++v)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[v][i][j][e] = 0.f;

  {  // prologue: chunk 0 -> buffer 0, chunk 1 -> registers
    SB2_FETCH(0)
    uint4 h, m, l;
    split8(a0, a1, h, m, l);
    *reinterpret_cast<uint4*>(smem + 0 * SB2_PLANE + soff) = h;
    *reinterpret_cast<uint4*>(smem + 1 * SB2_PLANE + soff) = m;
    *reinterpret_cast<uint4*>(smem + 2 * SB2_PLANE + soff) = l;
    split8(t0, t1, h, m, l);
    *reinterpret_cast<uint4*>(smem + 3 * SB2_PLANE + soff) = h;
    *reinterpret_cast<uint4*>(smem + 4 * SB2_PLANE + soff) = m;
    *reinterpret_cast<uint4*>(smem + 5 * SB2_PLANE + soff) = l;
    *reinterpret_cast<uint4*>(smem + 6 * SB2_PLANE + soff) = w0;
    *reinterpret_cast<uint4*>(smem + 7 * SB2_PLANE + soff) = w1;
    *reinterpret_cast<uint4*>(smem + 8 * SB2_PLANE + soff) = w2;
    if (nk > 1) { SB2_FETCH(1) }
  }
  __syncthreads();

  const int ra_ = wm * 64 + (lane & 31), rb_ = wn * 64 + (lane & 31), kh = lane >> 5;
  // rows r and r + 32 have the same swizzle bit
  const int foa = (2 * ra_ + (kh ^ (((ra_ + 4) >> 3) & 1))) * 16;
  const int fob = 6 * SB2_PLANE + (2 * rb_ + (kh ^ (((rb_ + 4) >> 3) & 1))) * 16;

#define SB2_AF(v, i)                                                                                     \
  _Pragma("unroll") for (int p = 0; p < 3; ++p) af[i][p] =                                               \
      *reinterpret_cast<const bf16x8*>(cur + foa + ((v) * 3 + p) * SB2_PLANE + (i) * 1024);
// one split product on the four accumulators of a half (value or tangent): an accumulator is revisited after four MFMAs
// (products per accumulator in the order 02 20 11 01 10 00, as before: same bits)
#define SB2_MMA4(v, pa_, pb_)                                                                                  \
  acc[v][0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][pa_], bf[0][pb_], acc[v][0][0], 0, 0, 0);     \
  acc[v][0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][pa_], bf[1][pb_], acc[v][0][1], 0, 0, 0);     \
  acc[v][1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][pa_], bf[0][pb_], acc[v][1][0], 0, 0, 0);     \
  acc[v][1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][pa_], bf[1][pb_], acc[v][1][1], 0, 0, 0);
#define SB2_HALF(v) SB2_MMA4(v, 0, 2) SB2_MMA4(v, 2, 0) SB2_MMA4(v, 1, 1) SB2_MMA4(v, 0, 1) SB2_MMA4(v, 1, 0) SB2_MMA4(v, 0, 0)
#define SB2_LOAD_B()                                                                                               \
  bf16x8 bf[2][3];                                                                                                 \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) _Pragma("unroll") for (int p = 0; p < 3; ++p) bf[j][p] =           \
      *reinterpret_cast<const bf16x8*>(cur + fob + p * SB2_PLANE + j * 1024);                                      \
  bf16x8 af[2][3];

  // steady state: chunk kt+1 is in the registers, branch-free body (the chunk index of the next loads is clamped)
  for (int kt = 0; kt + 1 < nk; ++kt) {
    const unsigned char* cur = smem + (kt & 1) * SB2_STAGE;
    unsigned char* nxt = smem + ((kt + 1) & 1) * SB2_STAGE;
    const int kf = (kt + 2 < nk) ? kt + 2 : nk - 1;
    SB2_LOAD_B()
    // ---- first half: value rows (the loads of chunk kt+1 land meanwhile)
    SB2_AF(0, 0)
    SB2_AF(0, 1)
    SB2_HALF(0)
    __builtin_amdgcn_sched_barrier(0);
    // ---- second half: tangent rows, interleaved with the split of chunk kt+1 and the loads of chunk kt+2
    uint4 h, m, l;
    SB2_AF(1, 0)
    SB2_AF(1, 1)
    SB2_MMA4(1, 0, 2)
    split2(a0.x, a0.y, h.x, m.x, l.x);
    split2(a0.z, a0.w, h.y, m.y, l.y);
    SB2_MMA4(1, 2, 0)
    split2(a1.x, a1.y, h.z, m.z, l.z);
    split2(a1.z, a1.w, h.w, m.w, l.w);
    *reinterpret_cast<uint4*>(nxt + 0 * SB2_PLANE + soff) = h;
    *reinterpret_cast<uint4*>(nxt + 1 * SB2_PLANE + soff) = m;
    *reinterpret_cast<uint4*>(nxt + 2 * SB2_PLANE + soff) = l;
    SB2_MMA4(1, 1, 1)
    split2(t0.x, t0.y, h.x, m.x, l.x);
    split2(t0.z, t0.w, h.y, m.y, l.y);
    SB2_MMA4(1, 0, 1)
    split2(t1.x, t1.y, h.z, m.z, l.z);
    split2(t1.z, t1.w, h.w, m.w, l.w);
    *reinterpret_cast<uint4*>(nxt + 3 * SB2_PLANE + soff) = h;
    *reinterpret_cast<uint4*>(nxt + 4 * SB2_PLANE + soff) = m;
    *reinterpret_cast<uint4*>(nxt + 5 * SB2_PLANE + soff) = l;
    SB2_MMA4(1, 1, 0)
    *reinterpret_cast<uint4*>(nxt + 6 * SB2_PLANE + soff) = w0;
    *reinterpret_cast<uint4*>(nxt + 7 * SB2_PLANE + soff) = w1;
    *reinterpret_cast<uint4*>(nxt + 8 * SB2_PLANE + soff) = w2;
    SB2_FETCH(kf)
    SB2_MMA4(1, 0, 0)
    __syncthreads();
  }
  {  // last chunk: multiply only
    const unsigned char* cur = smem + ((nk - 1) & 1) * SB2_STAGE;
    SB2_LOAD_B()
    SB2_AF(0, 0)
    SB2_AF(0, 1)
    SB2_HALF(0)
    SB2_AF(1, 0)
    SB2_AF(1, 1)
    SB2_HALF(1)
  }
#undef SB2_LOAD_B
#undef SB2_AF
#undef SB2_MMA4
#undef SB2_HALF
#undef SB2_FETCH
  // Epilogue.  A lane owns one column of each 32x32 accumulator block, so direct stores are 4 bytes per lane and the
  // 128 store instructions per lane clog the CU's vector-memory queue (the co-resident block's loads wait behind
  // them).  Each wave transposes its blocks through a private 8 KB of the (now idle) LDS instead: 16 B per lane, 8
  // rows x 128 contiguous bytes per store instruction, a quarter of the instructions.
  __syncthreads();  // every wave is done reading the last chunk
  float* xv = reinterpret_cast<float*>(smem) + wave * 2048;  // [2][32][32] value | tangent
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int cb = n0 + wn * 64 + j * 32;
    const int col = cb + (lane & 31);
    const float bv = (a.bias[0] && col < N) ? a.bias[0][col] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rb = m0 + wm * 64 + i * 32;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int rl = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        const float ev = acc[0][i][j][e] + bv, rv = acc[1][i][j][e];
        float o1, o2;
        if (KIND == 0 || KIND == 3) {
          o1 = ev;
          o2 = rv;
        } else {
          const float sg = fast_sigmoid(ev);
          const float f = ev * sg, df = sg * (1.0f + ev * (1.0f - sg));
          o1 = f;  // KIND 2: the per-row cutoff factors are applied after the transposition (one row per lane there)
          o2 = df * rv;
        }
        xv[rl * 32 + (lane & 31)] = o1;
        xv[1024 + rl * 32 + (lane & 31)] = o2;
      }
      // wave-private region: no barrier, the LDS queue is in order per wave
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int rl = rr * 8 + (lane >> 3), c4 = (lane & 7) * 4;
        float4 v1 = *reinterpret_cast<const float4*>(xv + rl * 32 + c4);
        float4 v2 = *reinterpret_cast<const float4*>(xv + 1024 + rl * 32 + c4);
        const int row = rb + rl;
        if ((KIND == 2 || KIND == 3) && row < M) {
          const float c = a.rowscale[row], dc = a.rowscale2[row];
          v2 = make_float4(v2.x * c + v1.x * dc, v2.y * c + v1.y * dc, v2.z * c + v1.z * dc, v2.w * c + v1.w * dc);
          v1 = make_float4(v1.x * c, v1.y * c, v1.z * c, v1.w * c);
        }
        if (row < M && cb + c4 < N) {  // N % 4 == 0 (gemm_dual_sb_ok)
          *reinterpret_cast<float4*>(a.C + (int64_t)row * a.ldc + cb + c4) = v1;
          *reinterpret_cast<float4*>(a.C2 + (int64_t)row * a.ldc + cb + c4) = v2;
        }
      }
    }
  }
  }  // tile loop
}

bool gemm_dual_sb_ok(const GemmArgs& a) {
  static const bool off = getenv("TMDNET_NO_SPLIT_BF16") != nullptr;  // developer switch: fp32 MFMA everywhere
  if (off || !a.Wsb || a.K <= 0 || (a.K & 15) || (a.lda & 3) || (a.N & 3) || (a.ldc & 3)) return false;
  if ((reinterpret_cast<uintptr_t>(a.A) & 15) || (reinterpret_cast<uintptr_t>(a.A2) & 15)) return false;
  if ((reinterpret_cast<uintptr_t>(a.C) & 15) || (reinterpret_cast<uintptr_t>(a.C2) & 15)) return false;
  return (int64_t)((a.M + 127) / 128) * ((a.N + 127) / 128) >= 256;  // small launches: split-K fp32 kernel
}

int launch_gemm_dual_sb(const GemmArgs& a, int kind, hipStream_t stream) {
  const int tiles_m = (a.M + 127) / 128, tiles_n = (a.N + 127) / 128;
  static const int n_cu = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n;
  }();
  const int total = tiles_m * tiles_n;
  const dim3 grid(total < 2 * n_cu ? total : 2 * n_cu), block(256);  // persistent: 2 blocks per CU
  switch (kind) {
    case 0: hipLaunchKernelGGL((k_gemm_dual_sb2<0>), grid, block, 0, stream, a, tiles_m, tiles_n); break;
    case 1: hipLaunchKernelGGL((k_gemm_dual_sb2<1>), grid, block, 0, stream, a, tiles_m, tiles_n); break;
    case 3: hipLaunchKernelGGL((k_gemm_dual_sb2<3>), grid, block, 0, stream, a, tiles_m, tiles_n); break;
    default: hipLaunchKernelGGL((k_gemm_dual_sb2<2>), grid, block, 0, stream, a, tiles_m, tiles_n); break;
  }
  return (int)hipGetLastError();
}

// ---- host side: split a [N][K] fp32 weight into the tile image described at the top (zero padded)
static inline uint16_t bf16_rne(float x) {
  uint32_t u;
  memcpy(&u, &x, 4);
  if ((u & 0x7f800000u) == 0x7f800000u) return (uint16_t)(u >> 16);  // inf / nan: truncate
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline float bf16_f32(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float x;
  memcpy(&x, &u, 4);
  return x;
}
size_t split_weight_elems(int64_t N, int64_t K) { return (size_t)((N + 127) / 128) * ((K + 15) / 16) * 3 * 128 * 16; }
void split_weight_tiles(const float* W, int64_t N, int64_t K, uint16_t* out) {
  const int64_t tn = (N + 127) / 128, nk = (K + 15) / 16;
  for (int64_t t = 0; t < tn; ++t)
    for (int64_t c = 0; c < nk; ++c) {
      uint16_t* blk = out + (t * nk + c) * 3 * 128 * 16;
      for (int r = 0; r < 128; ++r)
        for (int k = 0; k < 16; ++k) {
          const int64_t n = t * 128 + r, kk = c * 16 + k;
          const float x = (n < N && kk < K) ? W[n * K + kk] : 0.f;
          const uint16_t h = bf16_rne(x);
          const float r1 = x - bf16_f32(h);
          const uint16_t m = bf16_rne(r1);
          const float r2 = r1 - bf16_f32(m);
          const uint16_t l = bf16_rne(r2);
          blk[(0 * 128 + r) * 16 + k] = h;
          blk[(1 * 128 + r) * 16 + k] = m;
          blk[(2 * 128 + r) * 16 + k] = l;
        }
    }
}


// the same tile image from a weight that already sits in device memory (parameter upload: tn_api.hip / tn_et_api.hip); one thread
// per pair of consecutive k, rounding as split2 = bf16_rne at every level
__global__ void k_split_weight_tiles(const float* __restrict__ W, int N, int K, int nk, uint16_t* __restrict__ out, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int kp = (int)(idx & 7), r = (int)((idx >> 3) & 127);
  const int64_t tc = idx >> 10;  // tile * nk + chunk
  const int c = (int)(tc % nk);
  const int64_t t = tc / nk;
  const int64_t n = t * 128 + r;
  const int kk = c * 16 + 2 * kp;
  const float x0 = (n < N && kk < K) ? W[n * K + kk] : 0.f, x1 = (n < N && kk + 1 < K) ? W[n * K + kk + 1] : 0.f;
  uint32_t h, m, l;
  split2(x0, x1, h, m, l);
  uint32_t* blk = reinterpret_cast<uint32_t*>(out + tc * 3 * 128 * 16);
  blk[(0 * 128 + r) * 8 + kp] = h;
  blk[(1 * 128 + r) * 8 + kp] = m;
  blk[(2 * 128 + r) * 8 + kp] = l;
}
void launch_split_weight_tiles(const float* W_dev, int64_t N, int64_t K, uint16_t* out_dev, hipStream_t s) {
  const int64_t tn = (N + 127) / 128, nk = (K + 15) / 16, total = tn * nk * 128 * 8;
  if (total <= 0) return;
  hipLaunchKernelGGL(k_split_weight_tiles, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, W_dev, (int)N, (int)K, (int)nk, out_dev, total);
}

}  // namespace tn
