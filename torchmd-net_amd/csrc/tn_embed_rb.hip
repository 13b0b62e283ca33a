// TensorNet embedding in the RADIAL BASIS ("moment form"), gfx950.
//
// Reference: TensorEmbedding.forward, torchmdnet/models/tensornet.py:543-619 (edge weights W_c = C(d) Z_ij (P_c phi(d) + b_c),
// scatter of I / A / S :405-445, Z_ij = emb2([emb(z_i), emb(z_j)]) :526-541).  The reference materialises the per-edge
// weights [E, 3, F]; rounds 1-2 of this library materialised Q_c(d) = P_c phi(d) + b_c per PAIR ([P, 3F] plus its tangent,
// 0.6 GB at 256 x 64 atoms) and read it back in the scatter and in the reverse pair kernel.  Here neither exists:
//
//   W_c[e, f] = C Z_ij[f] sum_k P_c[f, k] phi_k(d) (+ bias)   and   Z_ij[f] = U[z_i][f] + V[z_j][f]
//   =>  S_i[g, f] = sum_t (U[z_i][f] + V[t][f]) * sum_k P_c(g)[f, k] * m_i[t, g, k] ,
//       m_i[t, g, k] = sum_{edges (i <- j), z_j = t} C(d) phi_k(d) * geom_g(rhat)          ("moments": K + 1 numbers, not F)
//
// with g over the 10 geometric components (1 | rhat | rhat rhat^T) and t over the atomic numbers PRESENT in the batch
// (compacted on the device: k_type_map).  The edge sweep therefore moves (K + 1) floats per (atom, type, component) instead
// of 3F per edge, and the channel dimension appears only in two small per-atom contractions that run on the matrix pipe
// (exact 3-way bf16 split, as every GEMM of the library).  Reverse pass: the same contraction transposed gives
// g_m[i, t, g, k]; the per-pair kernel (tn_pairgrad.hip: k_embed_pair_rb) gathers 2 x 10 x (K + 1) numbers per pair.
// Summation order differs from the per-edge form by rounding only.  Taken when the batch holds at most 8 species, K is 32 or
// 64 and F is 64 or 128 (everything else keeps the per-pair tables of tn_edge_table.hip).
#include <cstdlib>
#include <cstring>

#include "tn_common.h"
#include "tn_gemm_sb.h"
#include "tn_kernels.h"

namespace tn {

static inline int cdivr(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Layout of the moments m[R = i NTP + t][g][k] (k = K: bias moment).  The forward contraction reads them as MFMA A fragments
// (lane (r, h) of a 32-row tile takes k = 16 s + 8 h .. + 7 of row r), so they are stored fragment-major:
//   [tile = R / 32][g][k-step s (K / 16 + 1, the last one = bias)][h][r = R % 32][8]
// and a wave's fragment load is 2 KB of contiguous memory (with plain rows [R][g][K + 4] the same load touched 32 rows x 2
// pieces: 128 us of the kernel's 190, profiles/r03_notes.md).
__host__ __device__ __forceinline__ int64_t mom_off(int64_t R, int g, int k, int nksb) {
  return (R >> 5) * (10 * (int64_t)nksb * 512) + ((int64_t)(g * nksb + (k >> 4)) * 2 + ((k >> 3) & 1)) * 256 + (R & 31) * 8 + (k & 7);
}

// ------------------------------------------------------------------------------------------------ species of the batch
// type index of every atom = rank of its atomic number among the atomic numbers present (ascending: deterministic);
// tz[t] = atomic number of type t; counts[6] = number of types (read back with the pair counts).  One block.
__global__ __launch_bounds__(1024) void k_type_map(const int64_t* __restrict__ z_c, int N, int max_z, int* __restrict__ tix,
                                                  int* __restrict__ tz, int* __restrict__ counts) {
  __shared__ int flag[2048];
  const int tid = threadIdx.x;
  if (max_z > 2048) {
    if (tid == 0) counts[6] = 1 << 20;  // no table: the radial-basis embedding is not taken
    return;
  }
  for (int k = tid; k < max_z; k += 1024) flag[k] = 0;
  if (tid < 64) tz[tid] = 0;  // padded types (the kernels round the type count up to 4 / 8) read a valid table row
  __syncthreads();
  for (int i = tid; i < N; i += 1024) flag[(int)z_c[i]] = 1;
  __syncthreads();
  if (tid == 0) {
    int nt = 0;
    for (int k = 0; k < max_z; ++k)
      if (flag[k]) {
        if (nt < 64) tz[nt] = k;
        flag[k] = nt + 1;
        ++nt;
      }
    counts[6] = nt;
  }
  __syncthreads();
  for (int i = tid; i < N; i += 1024) tix[i] = flag[(int)z_c[i]] - 1;
}
void launch_type_map(const Graph& g, int N, int max_z, hipStream_t s) {
  if (N <= 0) return;
  hipLaunchKernelGGL(k_type_map, dim3(1), dim3(1024), 0, s, g.z_c, N, max_z, g.tix, g.tz, g.counts);
}

// ------------------------------------------------------------------------------------------------ per-pair scalars
// ps[p] = (C, C', C0, C0' | u = exp(-alpha (d - lo)), 0, 0, 0) once per pair (self pair p = P: d = 0): the precise sin / cos of
// the two cutoffs are the expensive part of psi_k and do not depend on k.
__global__ void k_pair_scalars(Graph g, int Pcap, float lo, float up, float* __restrict__ ps) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p > Pcap || g.counts[2]) return;
  const int P = g.counts[0];
  if (p > P) return;
  const float d = p < P ? g.pd[p] : 0.f;
  float c, dc, c0, dc0;
  cosine_cutoff(d, lo, up, c, dc);
  cosine_cutoff(d, 0.f, up, c0, dc0);
  const float alpha = 5.0f / (up - lo);
  *reinterpret_cast<float4*>(ps + (int64_t)p * 8) = make_float4(c, dc, c0, dc0);
  *reinterpret_cast<float4*>(ps + (int64_t)p * 8 + 4) = make_float4(expf(-alpha * (d - lo)), 0.f, 0.f, 0.f);
}
void launch_pair_scalars(const Graph& g, int Pcap, float lo, float up, float* ps, hipStream_t s) {
  hipLaunchKernelGGL(k_pair_scalars, dim3(cdivr(Pcap + 1, 256)), dim3(256), 0, s, g, Pcap, lo, up, ps);
}

// ------------------------------------------------------------------------------------------------ moments (edge sweep)
// wave per atom; lane = (k mod 32, half): half 0 owns g = 0..4, half 1 owns g = 5..9; KS = K / 32 radial functions per lane.
// A row's edge records (neighbour species, pair scalars, signed unit vector) are fetched 64 at a time, one edge per lane - one
// round of dependent loads for the whole row instead of one per edge - and handed round with v_readlane.
// Accumulators live in registers, selected by the neighbour's species with compare-and-add (NTP x 5 FMAs per edge and k).
// m[((i NTP + t) 10 + g) KP + k], KP = K + 4 (bias moment at k = K, row 16-byte aligned).
__device__ __forceinline__ float rlane(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
template <int NTP, int KS>
__global__ __launch_bounds__(256) void k_embed_moments(Graph g, int N, RadialParams rp, const float* __restrict__ ps,
                                                       float* __restrict__ m) {
  const int i = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  const int lane = threadIdx.x & 63, kk = lane & 31, half = lane >> 5;
  if (i >= N || g.counts[2]) return;
  const int K = rp.K;
  const int e0 = __builtin_amdgcn_readfirstlane(g.rowptr[i]), e1 = __builtin_amdgcn_readfirstlane(g.rowptr[i + 1]);
  const int P = __builtin_amdgcn_readfirstlane(g.counts[0]);
  float acc[KS][NTP][5], accb[NTP][5];
#pragma unroll
  for (int t = 0; t < NTP; ++t)
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      accb[t][q] = 0.f;
#pragma unroll
      for (int s = 0; s < KS; ++s) acc[s][t][q] = 0.f;
    }
  float mu[KS], beta[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    mu[s] = rp.means[kk + 32 * s];
    beta[s] = rp.betas[kk + 32 * s];
  }
  for (int base = e0; base < e1; base += 64) {
    const int n = e1 - base < 64 ? e1 - base : 64;
    // lane-parallel fetch of this chunk's edge records
    int lt = 0;
    float lc = 0.f, lc0 = 0.f, lu = 0.f, lx = 0.f, ly = 0.f, lz = 0.f;
    if (lane < n) {
      const int e = base + lane;
      const int j = g.col[e], p = g.epair[e];
      const float sg = g.esign[e];
      lt = g.tix[j];
      const int pp = sg != 0.f ? p : P;  // self edge: the self pair's record (d = 0)
      const float4 a = *reinterpret_cast<const float4*>(ps + (int64_t)pp * 8);
      lc = a.x;
      lc0 = a.z;
      lu = ps[(int64_t)pp * 8 + 4];
      if (sg != 0.f) {
        lx = sg * g.prhat[p * 3];
        ly = sg * g.prhat[p * 3 + 1];
        lz = sg * g.prhat[p * 3 + 2];
      }
    }
    for (int q_ = 0; q_ < n; ++q_) {
      const int t = __builtin_amdgcn_readlane(lt, q_);
      const float c = rlane(lc, q_), c0 = rlane(lc0, q_), u = rlane(lu, q_);
      const float rx = rlane(lx, q_), ry = rlane(ly, q_), rz = rlane(lz, q_);
      float ge[5];
      if (half == 0) {
        ge[0] = 1.f; ge[1] = rx; ge[2] = ry; ge[3] = rz; ge[4] = rx * rx;
      } else {
        ge[0] = rx * ry; ge[1] = rx * rz; ge[2] = ry * ry; ge[3] = ry * rz; ge[4] = rz * rz;
      }
      const float cc0 = c * c0;
      float psi[KS];
#pragma unroll
      for (int s = 0; s < KS; ++s) psi[s] = cc0 * __expf(-beta[s] * (u - mu[s]) * (u - mu[s]));
      // the neighbour's species is wave-uniform (an SGPR): a scalar branch per species instead of NTP compare-and-adds
#pragma unroll
      for (int tt = 0; tt < NTP; ++tt)
        if (tt == t) {
#pragma unroll
          for (int q = 0; q < 5; ++q) {
#pragma unroll
            for (int s = 0; s < KS; ++s) acc[s][tt][q] += psi[s] * ge[q];
            accb[tt][q] += c * ge[q];
          }
        }
    }
  }
  const int nksb = K / 16 + 1;
#pragma unroll
  for (int t = 0; t < NTP; ++t)
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      const int64_t R = (int64_t)i * NTP + t;
      const int gq = half * 5 + q;
#pragma unroll
      for (int s = 0; s < KS; ++s) m[mom_off(R, gq, kk + 32 * s, nksb)] = acc[s][t][q];
      if (kk == 0) m[mom_off(R, gq, K, nksb)] = accb[t][q];
    }
}

// ------------------------------------------------------------------------------------------------ split-bf16 fragments
// A / B operand of v_mfma_f32_32x32x16_bf16: lane l holds row (column) l & 31, k = 8 (l >> 5) .. + 7 -> 8 fp32 = three bf16x8
struct Frag3 {
  bf16x8 p[3];
};
__device__ __forceinline__ Frag3 frag_split(const float4& a, const float4& b) {
  uint4 h, m, l;
  split8(a, b, h, m, l);
  Frag3 f;
  f.p[0] = *reinterpret_cast<bf16x8*>(&h);
  f.p[1] = *reinterpret_cast<bf16x8*>(&m);
  f.p[2] = *reinterpret_cast<bf16x8*>(&l);
  return f;
}
// six products of the exact 3-way split, small terms first (as SB1_GROUP in tn_gemm_sb1.hip)
__device__ __forceinline__ void mma6(const Frag3& a, const Frag3& b, floatx16& acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[0], b.p[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[2], b.p[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[1], b.p[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[0], b.p[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[1], b.p[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[0], b.p[0], acc, 0, 0, 0);
}
__device__ __forceinline__ Frag3 frag_load(const uint4* img) {  // three planes, 64 lanes x 16 bytes each, consecutive
  Frag3 f;
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    const uint4 v = img[p * 64];
    f.p[p] = *reinterpret_cast<const bf16x8*>(&v);
  }
  return f;
}
__device__ __forceinline__ int rb_type_of(int g) { return g == 0 ? 0 : (g < 4 ? 1 : 2); }

// ------------------------------------------------------------------------------------------------ forward contraction
// S[i, g, f] = sum_t Z[i, t][f] * ( sum_k P_c[f, k] m[i, t, g, k] + b_c[f] m[i, t, g, K] ),  then u0 / s0n as k_embed_scatter.
// Block = 32 rows (row = atom_local * NTP + t) x 128 channels; wave w owns channels 32 w .. 32 w + 31 (one 32 x 32 MFMA tile).
// B fragments (weights; image built at finalize: [c][column tile][k-step][plane][lane]) stay in registers while the component
// type c does not change; the bias is one more k-step ([m_K, 0 ...] x [b_c, 0 ...]).  The A rows of component g + 1 are
// requested before component g is multiplied (the kernel is a chain of short dependent phases otherwise).
// Accumulator layout: lane holds column l & 31 and rows (e & 3) + 8 (e >> 2) + 4 (l >> 5), i.e. for NTP = 4 the four species of
// an atom are four consecutive accumulator registers (the species sum needs no exchange); NTP = 8 adds one 32-lane exchange.
template <int NKS>
struct ARow {
  float4 x[NKS][2];
  float b;
};
template <int NKS>
__device__ __forceinline__ void arow_load(const float* __restrict__ base, bool val, int half, ARow<NKS>& a) {
  // base: this lane's 8 floats of (tile, g, k-step 0, h = lane's half, r); k-steps are 512 floats apart, the bias step last
  // unconditional 16-byte loads: every row of a tile is allocated, and rows of atoms past N (never written: any bits) only
  // reach accumulator rows that are not stored.  (`val ? load : 0` compiled to eight 4-byte loads behind eight branches per
  // k-step: 120 of the kernel's 180 us.)
  (void)val;
#pragma unroll
  for (int s = 0; s < NKS; ++s) {
    a.x[s][0] = *reinterpret_cast<const float4*>(base + 512 * s);
    a.x[s][1] = *reinterpret_cast<const float4*>(base + 512 * s + 4);
  }
  const float bv = base[512 * NKS - 256 * half];  // the bias sits in the h = 0 half of its step: both halves read it there
  a.b = half == 0 ? bv : 0.f;
}
template <int NTP, int NKS>  // NKS = K / 16
__global__ __launch_bounds__(256) void k_embed_combine(int N, int F, const int64_t* __restrict__ z, const int* __restrict__ tz,
                                                       const float* __restrict__ Utab, const float* __restrict__ Vtab,
                                                       const uint4* __restrict__ Bimg, const float* __restrict__ m,
                                                       float* __restrict__ u0, float* __restrict__ s0n) {
  constexpr int TA = 32 / NTP;  // atoms per block
  constexpr int NS = 4;         // accumulator slots (atoms) per lane
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5;
  const int ct = blockIdx.y * 4 + wave, CT = F >> 5;
  if (ct >= CT) return;
  const int f = ct * 32 + (lane & 31);
  const int a0 = blockIdx.x * TA;
  constexpr int NKSB = NKS + 1, GS = NKSB * 512;  // floats per component of a tile
  const int r = lane & 31;
  const int ratom = a0 + r / NTP;
  const bool aval = ratom < N;
  const float* arow = m + (int64_t)blockIdx.x * (10 * GS) + half * 256 + r * 8;  // tile = blockIdx.x (TA NTP = 32 rows)
  // Z factors of the accumulator rows this lane holds: slot sl = e >> 2 -> atom_local (NTP 4: 2 sl + half; NTP 8: sl),
  // species t = (e & 3) (+ 4 half for NTP 8)
  float Uv[NS], Vv[4];
#pragma unroll
  for (int sl = 0; sl < NS; ++sl) {
    const int atom = a0 + (NTP == 4 ? 2 * sl + half : sl);
    Uv[sl] = atom < N ? Utab[z[atom] * F + f] : 0.f;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) Vv[q] = Vtab[(int64_t)tz[NTP == 4 ? q : q + 4 * half] * F + f];
  // per-component sums wait in wave-private LDS ([g][slot][lane]: conflict-free) - with the loop over g unrolled and the sums in
  // registers the compiler kept ten accumulator tiles alive (160 AGPRs, one wave per SIMD)
  __shared__ float Sl[4][10][NS][64];
  Frag3 bfr[NKS + 1];
  ARow<NKS> cur, nxt;
  arow_load<NKS>(arow, aval, half, cur);
  int c_cur = -1;
#pragma unroll 1
  for (int g = 0; g < 10; ++g) {
    const int c = rb_type_of(g);
    if (c != c_cur) {  // wave-uniform
      c_cur = c;
#pragma unroll
      for (int s = 0; s <= NKS; ++s) bfr[s] = frag_load(Bimg + ((int64_t)(c * CT + ct) * (NKS + 1) + s) * 192 + lane);
    }
    arow_load<NKS>(arow + (g < 9 ? g + 1 : g) * GS, aval, half, nxt);
    floatx16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int s = 0; s < NKS; ++s) mma6(frag_split(cur.x[s][0], cur.x[s][1]), bfr[s], acc);
    {  // bias step: A = [m_K, 0 ...] in the lanes of k-half 0
      const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
      mma6(frag_split(make_float4(cur.b, 0.f, 0.f, 0.f), zero), bfr[NKS], acc);
    }
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) {
      float sum = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) sum += acc[sl * 4 + q] * (Uv[sl] + Vv[q]);
      if (NTP == 8) sum += __shfl_xor(sum, 32, 64);  // the other four species of the atom sit in the partner lane (l ^ 32)
      Sl[wave][g][sl][lane] = sum;
    }
    cur = nxt;
  }
#pragma unroll
  for (int sl = 0; sl < NS; ++sl) {
    const int atom = a0 + (NTP == 4 ? 2 * sl + half : sl);
    if (NTP == 8 && (sl & 1) != half) continue;  // both halves hold the sums: each writes half of the atoms
    if (atom >= N) continue;
    float S[10];
#pragma unroll
    for (int g = 0; g < 10; ++g) S[g] = Sl[wave][g][sl][lane];
    const float tr3 = (S[4] + S[7] + S[9]) * (1.0f / 3.0f);
    float u[9] = {S[0], S[1], S[2], S[3], S[4] - tr3, S[5], S[6], S[7] - tr3, S[8]};
    float* o = u0 + (int64_t)atom * 9 * F + f;
#pragma unroll
    for (int cc = 0; cc < 9; ++cc) o[cc * F] = u[cc];
    s0n[(int64_t)atom * F + f] = quad(u);
  }
}

// ------------------------------------------------------------------------------------------------ reverse contraction
// g_m[i, t, g, k] = sum_f gA[i, g, f] Z[i, t][f] P_c(g)[f, k]  (k < K),   g_m[.., K] = sum_f gA Z b_c[f].
// Wave = 32 rows (row = atom_local * NTP + t), K output columns (K / 32 MFMA column tiles), contraction over F; block = 4 waves.
// LDS: the weight fragments of the current component type (image [c][column tile][k-step over f][plane][lane] of P_c^T,
// reloaded when c changes: twice), the U rows of the block's atoms, the V rows of the species and b_c - so a wave keeps only
// its gA row (all of F, requested at once) and the accumulators in registers.  The bias column is a lane-local dot product.
template <int NTP, int KS, int FS>  // KS = K / 32, FS = F / 16
__global__ __launch_bounds__(256) void k_embed_gm(int N, const int64_t* __restrict__ z, const int* __restrict__ tz,
                                                  const float* __restrict__ Utab, const float* __restrict__ Vtab,
                                                  const uint4* __restrict__ B2img, const float* __restrict__ bdp,
                                                  const float* __restrict__ gA, float* __restrict__ gm) {
  constexpr int WA = 32 / NTP;  // atoms per wave
  constexpr int F = FS * 16, K = KS * 32, KP = K + 4;
  constexpr int NW = KS * FS * 192;  // uint4 per component type
  __shared__ uint4 wl[NW];
  __shared__ __attribute__((aligned(16))) float ul[4 * WA][F];
  __shared__ __attribute__((aligned(16))) float vl[NTP][F];
  __shared__ __attribute__((aligned(16))) float bl[F];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
  const int b0 = blockIdx.x * 4 * WA;  // first atom of the block
  const int a0 = b0 + wave * WA;
  const int r = lane & 31;
  const int al = wave * WA + r / NTP, t = r % NTP;  // atom (block-local) and species of this lane's A row
  const int atom = b0 + al;
  const bool val = atom < N;
  for (int idx = tid; idx < 4 * WA * (F / 4); idx += 256) {
    const int a = idx / (F / 4), f4 = idx - a * (F / 4);
    const int at = b0 + a < N ? b0 + a : N - 1;
    *reinterpret_cast<float4*>(&ul[a][4 * f4]) = *reinterpret_cast<const float4*>(Utab + (int64_t)z[at] * F + 4 * f4);
  }
  for (int idx = tid; idx < NTP * (F / 4); idx += 256) {
    const int a = idx / (F / 4), f4 = idx - a * (F / 4);
    *reinterpret_cast<float4*>(&vl[a][4 * f4]) = *reinterpret_cast<const float4*>(Vtab + (int64_t)tz[a] * F + 4 * f4);
  }
  int c_cur = -1;
#pragma unroll 1
  for (int g = 0; g < 10; ++g) {
    const int c = rb_type_of(g);
    // this lane's gA row: every k-step requested before the first use
    float4 x[FS][2];
    {
      const float* ga = gA + ((int64_t)(val ? atom : 0) * 10 + g) * F + 8 * half;
#pragma unroll
      for (int s = 0; s < FS; ++s) {
        x[s][0] = *reinterpret_cast<const float4*>(ga + 16 * s);
        x[s][1] = *reinterpret_cast<const float4*>(ga + 16 * s + 4);
      }
    }
    if (c != c_cur) {  // block-uniform
      c_cur = c;
      __syncthreads();  // every wave is done with the previous type's fragments
      for (int idx = tid; idx < NW; idx += 256) wl[idx] = B2img[(int64_t)c * NW + idx];
      for (int idx = tid; idx < F; idx += 256) bl[idx] = bdp[c * F + idx];
      __syncthreads();
    }
    floatx16 acc[KS];
#pragma unroll
    for (int ct = 0; ct < KS; ++ct)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[ct][e] = 0.f;
    float gb = 0.f;
#pragma unroll
    for (int s = 0; s < FS; ++s) {
      const int fo = 16 * s + 8 * half;
      const float4 u0 = *reinterpret_cast<const float4*>(&ul[al][fo]), u1 = *reinterpret_cast<const float4*>(&ul[al][fo + 4]);
      const float4 v0 = *reinterpret_cast<const float4*>(&vl[t][fo]), v1 = *reinterpret_cast<const float4*>(&vl[t][fo + 4]);
      const float4 b0_ = *reinterpret_cast<const float4*>(&bl[fo]), b1_ = *reinterpret_cast<const float4*>(&bl[fo + 4]);
      float4 x0 = x[s][0], x1 = x[s][1];
      x0.x *= u0.x + v0.x; x0.y *= u0.y + v0.y; x0.z *= u0.z + v0.z; x0.w *= u0.w + v0.w;
      x1.x *= u1.x + v1.x; x1.y *= u1.y + v1.y; x1.z *= u1.z + v1.z; x1.w *= u1.w + v1.w;
      gb += x0.x * b0_.x + x0.y * b0_.y + x0.z * b0_.z + x0.w * b0_.w + x1.x * b1_.x + x1.y * b1_.y + x1.z * b1_.z + x1.w * b1_.w;
      const Frag3 af = frag_split(x0, x1);
#pragma unroll
      for (int ct = 0; ct < KS; ++ct) {
        Frag3 bf;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          const uint4 v = wl[((ct * FS + s) * 3 + p) * 64 + lane];
          bf.p[p] = *reinterpret_cast<const bf16x8*>(&v);
        }
        mma6(af, bf, acc[ct]);
      }
    }
    gb += __shfl_xor(gb, 32, 64);  // the two k-halves of the row
    if (val && half == 0) gm[(((int64_t)atom * NTP + t) * 10 + g) * KP + K] = gb;
#pragma unroll
    for (int ct = 0; ct < KS; ++ct)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int rr = (e & 3) + 8 * (e >> 2) + 4 * half;
        const int oa = a0 + rr / NTP, ot = rr % NTP;
        if (oa < N) gm[(((int64_t)oa * NTP + ot) * 10 + g) * KP + ct * 32 + (lane & 31)] = acc[ct][e];
      }
  }
}

// ------------------------------------------------------------------------------------------------ weight images
// fragment images of the distance-projection weights Wdp [3F][K], bdp [3F], rebuilt by a kernel with every parameter upload AND
// every device-side update (one definition: a trained handle and a freshly loaded one hold the same bits)
//   forward  B[c][ct][s][plane][lane][8]: value(n = 32 ct + (l & 31), k = 16 s + 8 (l >> 5) + j) = Wdp[(c F + n) K + k];
//            last k-step (s = K / 16): k-half 0, j = 0 -> bdp[c F + n], zeros elsewhere
//   reverse  B2[c][ct][s][plane][lane][8]: value(n = 32 ct + (l & 31) (an index k' of the radial basis), f = 16 s + 8 (l >> 5) + j)
//            = Wdp[(c F + f) K + n]
__device__ __forceinline__ uint16_t rb_bf16(float x) {  // round to nearest even; infinities / NaNs truncated
  uint32_t u = __float_as_uint(x);
  if ((u & 0x7f800000u) == 0x7f800000u) return (uint16_t)(u >> 16);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float rb_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
size_t embed_rb_image_elems(int F, int K, bool reverse) {
  return reverse ? (size_t)3 * (K / 32) * (F / 16) * 1536 : (size_t)3 * (F / 32) * (K / 16 + 1) * 1536;
}
// one thread per (block of 1536, lane, j): the exact 3-way split of one weight into the three planes
__global__ void k_embed_rb_images(const float* __restrict__ Wdp, const float* __restrict__ bdp, int F, int K, uint16_t* __restrict__ fwd,
                                  uint16_t* __restrict__ rev) {
  const int CT = F / 32, NKS = K / 16, KS = K / 32, FS = F / 16;
  const int nf = 3 * CT * (NKS + 1) * 512, nr = 3 * KS * FS * 512;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nf + nr) return;
  float x = 0.f;
  uint16_t* blk;
  int l, j;
  if (idx < nf) {
    const int b = idx >> 9, e = idx & 511;
    l = e >> 3;
    j = e & 7;
    const int s = b % (NKS + 1), ct = (b / (NKS + 1)) % CT, c = b / ((NKS + 1) * CT);
    const int n = 32 * ct + (l & 31), k = 16 * s + 8 * (l >> 5) + j;
    if (s < NKS) x = Wdp[((size_t)c * F + n) * K + k];
    else if ((l >> 5) == 0 && j == 0) x = bdp[c * F + n];
    blk = fwd + (size_t)b * 1536;
  } else {
    const int r = idx - nf, b = r >> 9, e = r & 511;
    l = e >> 3;
    j = e & 7;
    const int s = b % FS, ct = (b / FS) % KS, c = b / (FS * KS);
    const int n = 32 * ct + (l & 31), f = 16 * s + 8 * (l >> 5) + j;
    x = Wdp[((size_t)c * F + f) * K + n];
    blk = rev + (size_t)b * 1536;
  }
  const uint16_t h = rb_bf16(x);
  const float r1 = x - rb_f32(h);
  const uint16_t m = rb_bf16(r1);
  const float r2 = r1 - rb_f32(m);
  blk[(0 * 64 + l) * 8 + j] = h;
  blk[(1 * 64 + l) * 8 + j] = m;
  blk[(2 * 64 + l) * 8 + j] = rb_bf16(r2);
}
void launch_embed_rb_images(const float* Wdp_dev, const float* bdp_dev, int F, int K, uint16_t* fwd_dev, uint16_t* rev_dev, hipStream_t s) {
  const int total = 3 * (F / 32) * (K / 16 + 1) * 512 + 3 * (K / 32) * (F / 16) * 512;
  hipLaunchKernelGGL(k_embed_rb_images, dim3((total + 255) / 256), dim3(256), 0, s, Wdp_dev, bdp_dev, F, K, fwd_dev, rev_dev);
}

bool embed_rb_shape_ok(int F, int K) { return (F == 64 || F == 128) && (K == 32 || K == 64); }
int embed_rb_ntp(int nt) { return nt <= 0 ? 0 : (nt <= 4 ? 4 : (nt <= 8 ? 8 : 0)); }
// forward moments: fragment-major tiles of 32 rows; their gradient (read per pair, lanes over k): plain rows [R][10][K + 4]
int64_t embed_rb_moment_elems(int64_t N, int ntp, int K) { return ((N * ntp + 31) / 32) * 10 * (K / 16 + 1) * 512; }
int64_t embed_rb_gmoment_elems(int64_t N, int ntp, int K) { return N * ntp * 10 * (K + 4); }

void launch_embed_moments(const Graph& g, int N, RadialParams rp, int ntp, const float* ps, float* m, hipStream_t s) {
  if (N <= 0) return;
  const dim3 grid(cdivr(N, 4)), block(256);
  if (ntp == 4 && rp.K == 32) hipLaunchKernelGGL((k_embed_moments<4, 1>), grid, block, 0, s, g, N, rp, ps, m);
  else if (ntp == 4) hipLaunchKernelGGL((k_embed_moments<4, 2>), grid, block, 0, s, g, N, rp, ps, m);
  else if (rp.K == 32) hipLaunchKernelGGL((k_embed_moments<8, 1>), grid, block, 0, s, g, N, rp, ps, m);
  else hipLaunchKernelGGL((k_embed_moments<8, 2>), grid, block, 0, s, g, N, rp, ps, m);
}

void launch_embed_combine(const Graph& g, int N, int F, int K, int ntp, const int64_t* z, const float* Utab, const float* Vtab,
                          const uint16_t* Bimg, const float* m, float* u0, float* s0n, hipStream_t s) {
  if (N <= 0) return;
  const dim3 grid(cdivr(N, 32 / ntp), cdivr(F / 32, 4)), block(256);
  const uint4* B = reinterpret_cast<const uint4*>(Bimg);
#define RB_COMBINE(NTP_, NKS_) hipLaunchKernelGGL((k_embed_combine<NTP_, NKS_>), grid, block, 0, s, N, F, z, g.tz, Utab, Vtab, B, m, u0, s0n)
  if (ntp == 4 && K == 32) RB_COMBINE(4, 2);
  else if (ntp == 4) RB_COMBINE(4, 4);
  else if (K == 32) RB_COMBINE(8, 2);
  else RB_COMBINE(8, 4);
#undef RB_COMBINE
}

void launch_embed_gm(const Graph& g, int N, int F, int K, int ntp, const int64_t* z, const float* Utab, const float* Vtab,
                     const uint16_t* B2img, const float* bdp, const float* gA, float* gm, hipStream_t s) {
  if (N <= 0) return;
  const dim3 grid(cdivr(N, 4 * (32 / ntp))), block(256);
  const uint4* B = reinterpret_cast<const uint4*>(B2img);
#define RB_GM(NTP_, KS_, FS_) hipLaunchKernelGGL((k_embed_gm<NTP_, KS_, FS_>), grid, block, 0, s, N, z, g.tz, Utab, Vtab, B, bdp, gA, gm)
  const int key = (ntp == 8 ? 4 : 0) | (K == 64 ? 2 : 0) | (F == 128 ? 1 : 0);
  switch (key) {
    case 0: RB_GM(4, 1, 4); break;
    case 1: RB_GM(4, 1, 8); break;
    case 2: RB_GM(4, 2, 4); break;
    case 3: RB_GM(4, 2, 8); break;
    case 4: RB_GM(8, 1, 4); break;
    case 5: RB_GM(8, 1, 8); break;
    case 6: RB_GM(8, 2, 4); break;
    default: RB_GM(8, 2, 8); break;
  }
#undef RB_GM
}

}  // namespace tn
