// Per-pair reverse kernels for the forward-tangent formulation of the edge MLP (gfx950).
//
// With dw/dd (and dQ/dd) propagated forward by the dual GEMM (tn_gemm_dual.hip), the distance gradient of a pair is
//     g_d[p] = sum_layers sum_{k,f} g_w[p,k,f] * dw[p,k,f]/dd  +  embedding terms ,
//     g_w[p,k,f] = sum_{c in k} gM[i,c,f] P[j,c,f] + gM[j,c,f] P[i,c,f]                   
// so the reverse pass of the edge side is pure gather + reduce: no edge GEMM, no g_w array in HBM.  Channel sums
// are wave-level xor-shuffle reductions inside the lanes that own a pair (16 bytes per lane when F allows it).
#include <cstdlib>

#include "tn_common.h"
#include "tn_kernels.h"

namespace tn {

static inline int cdivp(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// 16-byte-per-lane form of the per-pair kernels: F % 4 == 0, F / 4 a power of two <= 64 (else one channel per lane)
bool gather_v4_ok(int F) {
  static const bool off = getenv("TMDNET_NO_V4") != nullptr;  // developer switch: force the scalar kernels
  if (off || (F & 3)) return false;
  const int f4 = F >> 2;
  return f4 >= 1 && f4 <= 64 && (f4 & (f4 - 1)) == 0;
}
static inline int fthreads_p(int F) {
  int t = ((F + 63) / 64) * 64;
  return t > 256 ? 256 : t;
}

__device__ __forceinline__ float4 pld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float dot4(const float4& a, const float4& b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ __forceinline__ float pcomp(const float4& v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : (k == 2 ? v.z : v.w)); }
__device__ __forceinline__ float pgroup_sum(float v, int tpa) {
  for (int off = tpa >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
// sum over all threads of a block (blockDim multiple of 64, <= 256); result valid in thread 0
__device__ __forceinline__ float pblock_sum(float v, float* red) {
  v = wave_sum(v);
  const int nw = blockDim.x >> 6;
  if (nw == 1) return v;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = 0.f;
  for (int w = 0; w < nw; ++w) s += red[w];
  return s;
}

// ---- interaction layer: gd[p] += sum_{k,f} g_w * dw
__global__ __launch_bounds__(256) void k_pair_gd_v4(Graph g, int F, const float* __restrict__ gMi, const float* __restrict__ Pn,
                                                    const float* __restrict__ dw, float* __restrict__ gd) {
  const int tpa = F >> 2, ppb = 256 / tpa;
  if (g.counts[2]) return;
  const int item = xcd_chunk_act(blockIdx.x, (g.counts[0] + ppb - 1) / ppb);  // grid: pair capacity; items: existing pairs
  if (item < 0) return;
  const int p = item * ppb + threadIdx.x / tpa;
  if (p >= g.counts[0]) return;
  const int f = (threadIdx.x % tpa) << 2;
  const int i = g.pair_i[p], j = g.pair_j[p];
  const int F9 = 9 * F, F3 = 3 * F;
  const float* gi = gMi + (int64_t)i * F9 + f;
  const float* gj = gMi + (int64_t)j * F9 + f;
  const float* pi = Pn + (int64_t)i * F9 + f;
  const float* pj = Pn + (int64_t)j * F9 + f;
  float4 gw[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) gw[k] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int c = 0; c < 9; ++c) {
    const float4 a = pld4(gi + c * F), b = pld4(pj + c * F), cc = pld4(gj + c * F), d = pld4(pi + c * F);
    float4& o = gw[c == 0 ? 0 : (c < 4 ? 1 : 2)];
    o.x += a.x * b.x + cc.x * d.x;
    o.y += a.y * b.y + cc.y * d.y;
    o.z += a.z * b.z + cc.z * d.z;
    o.w += a.w * b.w + cc.w * d.w;
  }
  float part = 0.f;
#pragma unroll
  for (int k = 0; k < 3; ++k) part += dot4(gw[k], pld4(dw + (int64_t)p * F3 + k * F + f));
  part = pgroup_sum(part, tpa);
  if ((threadIdx.x % tpa) == 0) gd[p] += part;
}
__global__ void k_pair_gd(Graph g, int F, const float* __restrict__ gMi, const float* __restrict__ Pn, const float* __restrict__ dw,
                          float* __restrict__ gd) {
  __shared__ float red[4];
  if (g.counts[2]) return;
  const int p = xcd_chunk_act(blockIdx.x, g.counts[0]);
  if (p < 0) return;
  const int i = g.pair_i[p], j = g.pair_j[p];
  const int F9 = 9 * F, F3 = 3 * F;
  float part = 0.f;
  for (int f = threadIdx.x; f < F; f += blockDim.x) {
    float gw[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 9; ++c)
      gw[type_of(c)] += gMi[(int64_t)i * F9 + c * F + f] * Pn[(int64_t)j * F9 + c * F + f] +
                        gMi[(int64_t)j * F9 + c * F + f] * Pn[(int64_t)i * F9 + c * F + f];
#pragma unroll
    for (int k = 0; k < 3; ++k) part += gw[k] * dw[(int64_t)p * F3 + k * F + f];
  }
  part = pblock_sum(part, red);
  if (threadIdx.x == 0) gd[p] += part;
}
void launch_pair_gd(const Graph& g, int Pcap, int F, const float* gMi, const float* Pn, const float* dw, float* gd, hipStream_t s) {
  if (Pcap <= 0) return;
  if (gather_v4_ok(F)) {
    const int ppb = 256 / (F >> 2);
    hipLaunchKernelGGL(k_pair_gd_v4, dim3(cdivp(Pcap, ppb)), dim3(256), 0, s, g, F, gMi, Pn, dw, gd);
  } else {
    hipLaunchKernelGGL(k_pair_gd, dim3(Pcap), dim3(fthreads_p(F)), 0, s, g, F, gMi, Pn, dw, gd);
  }
}

// ---- embedding: W_c = C(d) Z Q_c(d):  gd[p] += sum_c s_c (C' Q_c + C dQ_c) ;  g_rhat[p] as before
// s_c = Z1 gW_c(i<-j) + Z2 gW_c(j<-i)  (both directions of the pair), gA = gradients wrt (I0, v[3], T[6]) per atom
template <bool V4>
__device__ __forceinline__ void embed_pair_channel(float Z1, float Z2, float q0, float q1, float q2, float dq0, float dq1, float dq2,
                                                   const float ai[10], const float aj[10], float r0, float r1, float r2, float cp,
                                                   float dcp, float acc[4]) {
  const float gW1_1 = ai[1] * r0 + ai[2] * r1 + ai[3] * r2;
  const float gW1_2 = -(aj[1] * r0 + aj[2] * r1 + aj[3] * r2);
  const float gW2_1 = ai[4] * r0 * r0 + ai[5] * r0 * r1 + ai[6] * r0 * r2 + ai[7] * r1 * r1 + ai[8] * r1 * r2 + ai[9] * r2 * r2;
  const float gW2_2 = aj[4] * r0 * r0 + aj[5] * r0 * r1 + aj[6] * r0 * r2 + aj[7] * r1 * r1 + aj[8] * r1 * r2 + aj[9] * r2 * r2;
  const float s0 = Z1 * ai[0] + Z2 * aj[0], s1 = Z1 * gW1_1 + Z2 * gW1_2, s2 = Z1 * gW2_1 + Z2 * gW2_2;
  acc[0] += dcp * (q0 * s0 + q1 * s1 + q2 * s2) + cp * (dq0 * s0 + dq1 * s1 + dq2 * s2);
  const float W1_1 = cp * Z1 * q1, W2_1 = cp * Z1 * q2, W1_2 = cp * Z2 * q1, W2_2 = cp * Z2 * q2;
  const float di0 = 2.f * ai[4] * r0 + ai[5] * r1 + ai[6] * r2, dj0 = 2.f * aj[4] * r0 + aj[5] * r1 + aj[6] * r2;
  const float di1 = ai[5] * r0 + 2.f * ai[7] * r1 + ai[8] * r2, dj1 = aj[5] * r0 + 2.f * aj[7] * r1 + aj[8] * r2;
  const float di2 = ai[6] * r0 + ai[8] * r1 + 2.f * ai[9] * r2, dj2 = aj[6] * r0 + aj[8] * r1 + 2.f * aj[9] * r2;
  acc[1] += ai[1] * W1_1 - aj[1] * W1_2 + di0 * W2_1 + dj0 * W2_2;
  acc[2] += ai[2] * W1_1 - aj[2] * W1_2 + di1 * W2_1 + dj1 * W2_2;
  acc[3] += ai[3] * W1_1 - aj[3] * W1_2 + di2 * W2_1 + dj2 * W2_2;
}

// the pair's last step, g_delta = (g_r - (g_r . r) r) / d + g_d r with g_d = this kernel's part + the sweeps' slots
// (k_geom_gd's arithmetic and summation order)
__device__ __forceinline__ void pair_finish(const Graph& g, int p, float gd_embed, float a0, float a1, float a2, float r0, float r1,
                                            float r2, const float* __restrict__ slots, int n_slots, int64_t slot_stride,
                                            float* __restrict__ g_delta) {
  const float d = g.pd[p];
  const float inv = d > 0.f ? 1.0f / d : 0.f;
  const float dot = a0 * r0 + a1 * r1 + a2 * r2;
  float s = gd_embed;
  for (int k = 0; k < n_slots; ++k) s += slots[k * slot_stride + 2 * (int64_t)p] + slots[k * slot_stride + 2 * (int64_t)p + 1];
  g_delta[p * 3] = (a0 - dot * r0) * inv + s * r0;
  g_delta[p * 3 + 1] = (a1 - dot * r1) * inv + s * r1;
  g_delta[p * 3 + 2] = (a2 - dot * r2) * inv + s * r2;
}

__global__ __launch_bounds__(256) void k_embed_pair_gd_v4(Graph g, int F, const int64_t* __restrict__ z, const float* __restrict__ Utab,
                                                          const float* __restrict__ Vtab, const float* __restrict__ Q,
                                                          const float* __restrict__ dQ, const float* __restrict__ C,
                                                          const float* __restrict__ dC, const float* __restrict__ gA,
                                                          float* __restrict__ gd, float* __restrict__ g_rhat,
                                                          float* __restrict__ g_delta, const float* __restrict__ slots, int n_slots,
                                                          int64_t slot_stride) {
  const int tpa = F >> 2, ppb = 256 / tpa;
  if (g.counts[2]) return;
  const int item = xcd_chunk_act(blockIdx.x, (g.counts[0] + ppb - 1) / ppb);  // grid: pair capacity; items: existing pairs
  if (item < 0) return;
  const int p = item * ppb + threadIdx.x / tpa;
  if (p >= g.counts[0]) return;
  const int f = (threadIdx.x % tpa) << 2;
  const int i = g.pair_i[p], j = g.pair_j[p];
  const int64_t zi = z[i], zj = z[j];
  const float r0 = g.prhat[p * 3], r1 = g.prhat[p * 3 + 1], r2 = g.prhat[p * 3 + 2];
  const float cp = C[p], dcp = dC[p];
  const int F3 = 3 * F, F10 = 10 * F;
  const float4 Ui = pld4(Utab + zi * F + f), Vi = pld4(Vtab + zi * F + f), Uj = pld4(Utab + zj * F + f), Vj = pld4(Vtab + zj * F + f);
  const float* qp = Q + (int64_t)p * F3 + f;
  const float* dqp = dQ + (int64_t)p * F3 + f;
  const float4 q0v = pld4(qp), q1v = pld4(qp + F), q2v = pld4(qp + 2 * F);
  const float4 d0v = pld4(dqp), d1v = pld4(dqp + F), d2v = pld4(dqp + 2 * F);
  float4 aiv[10], ajv[10];
#pragma unroll
  for (int c = 0; c < 10; ++c) {
    aiv[c] = pld4(gA + (int64_t)i * F10 + c * F + f);
    ajv[c] = pld4(gA + (int64_t)j * F10 + c * F + f);
  }
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float ai[10], aj[10];
#pragma unroll
    for (int c = 0; c < 10; ++c) {
      ai[c] = pcomp(aiv[c], k);
      aj[c] = pcomp(ajv[c], k);
    }
    embed_pair_channel<true>(pcomp(Ui, k) + pcomp(Vj, k), pcomp(Uj, k) + pcomp(Vi, k), pcomp(q0v, k), pcomp(q1v, k), pcomp(q2v, k),
                             pcomp(d0v, k), pcomp(d1v, k), pcomp(d2v, k), ai, aj, r0, r1, r2, cp, dcp, acc);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) acc[k] = pgroup_sum(acc[k], tpa);
  if ((threadIdx.x % tpa) == 0) {
    if (g_delta) {
      pair_finish(g, p, acc[0], acc[1], acc[2], acc[3], r0, r1, r2, slots, n_slots, slot_stride, g_delta);
    } else {
      gd[p] += acc[0];
      g_rhat[p * 3] = acc[1];
      g_rhat[p * 3 + 1] = acc[2];
      g_rhat[p * 3 + 2] = acc[3];
    }
  }
}
__global__ void k_embed_pair_gd(Graph g, int F, const int64_t* __restrict__ z, const float* __restrict__ Utab,
                                const float* __restrict__ Vtab, const float* __restrict__ Q, const float* __restrict__ dQ,
                                const float* __restrict__ C, const float* __restrict__ dC, const float* __restrict__ gA,
                                float* __restrict__ gd, float* __restrict__ g_rhat, float* __restrict__ g_delta,
                                const float* __restrict__ slots, int n_slots, int64_t slot_stride) {
  __shared__ float red[4];
  if (g.counts[2]) return;
  const int p = xcd_chunk_act(blockIdx.x, g.counts[0]);
  if (p < 0) return;
  const int i = g.pair_i[p], j = g.pair_j[p];
  const int64_t zi = z[i], zj = z[j];
  const float r0 = g.prhat[p * 3], r1 = g.prhat[p * 3 + 1], r2 = g.prhat[p * 3 + 2];
  const float cp = C[p], dcp = dC[p];
  const int F3 = 3 * F, F10 = 10 * F;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int f = threadIdx.x; f < F; f += blockDim.x) {
    float ai[10], aj[10];
#pragma unroll
    for (int c = 0; c < 10; ++c) {
      ai[c] = gA[(int64_t)i * F10 + c * F + f];
      aj[c] = gA[(int64_t)j * F10 + c * F + f];
    }
    const float* qp = Q + (int64_t)p * F3 + f;
    const float* dqp = dQ + (int64_t)p * F3 + f;
    embed_pair_channel<false>(Utab[zi * F + f] + Vtab[zj * F + f], Utab[zj * F + f] + Vtab[zi * F + f], qp[0], qp[F], qp[2 * F], dqp[0],
                              dqp[F], dqp[2 * F], ai, aj, r0, r1, r2, cp, dcp, acc);
  }
  float tot[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) tot[k] = pblock_sum(acc[k], red);
  if (threadIdx.x == 0) {
    if (g_delta) {
      pair_finish(g, p, tot[0], tot[1], tot[2], tot[3], r0, r1, r2, slots, n_slots, slot_stride, g_delta);
    } else {
      gd[p] += tot[0];
      g_rhat[p * 3] = tot[1];
      g_rhat[p * 3 + 1] = tot[2];
      g_rhat[p * 3 + 2] = tot[3];
    }
  }
}
void launch_embed_pair_gd(const Graph& g, int Pcap, int F, const int64_t* z, const float* Utab, const float* Vtab, const float* Q,
                          const float* dQ, const float* C, const float* dC, const float* gA, float* gd, float* g_rhat, hipStream_t s,
                          float* g_delta, const float* slots, int n_slots, int64_t slot_stride) {
  if (Pcap <= 0) return;
  if (gather_v4_ok(F)) {
    const int ppb = 256 / (F >> 2);
    hipLaunchKernelGGL(k_embed_pair_gd_v4, dim3(cdivp(Pcap, ppb)), dim3(256), 0, s, g, F, z, Utab, Vtab, Q, dQ, C, dC, gA, gd, g_rhat, g_delta, slots,
                       n_slots, slot_stride);
  } else {
    hipLaunchKernelGGL(k_embed_pair_gd, dim3(Pcap), dim3(fthreads_p(F)), 0, s, g, F, z, Utab, Vtab, Q, dQ, C, dC, gA, gd, g_rhat, g_delta, slots,
                       n_slots, slot_stride);
  }
}

// ---- embedding in the radial basis (tn_embed_rb.hip): the per-pair part of its reverse pass: gather g_m[i, species(j), :, k] and
// g_m[j, species(i), :, k] (2 x 10 numbers per radial function k), contract with psi_k(d) = C(d) phi_k(d), psi_k'(d) and the pair
// geometry, reduce over k.  It is embed_pair_channel with Z1 = Z2 = 1, q0 = q1 = q2 = psi and the cutoff folded into psi:
// with b[g] = ai[g] + s_g aj[g] (s_g = -1 for the three components odd in rhat)
//   acc0 += psi' * sum_g b[g] geom_g(r) ,   acc_a += psi * sum_g b[g] d geom_g / d r_a          (33 operations instead of ~110).
// Two earlier forms - 32 lanes per pair (one radial function per lane), and a wave per atom walking its pairs two at a time -
// kept 2 pairs per wave in flight and were bound by the latency of their dependent load phases (115 - 160 us at 256 x 64 atoms
// for 0.5 GB of gathers; profiles/r03_notes.md).
// K / 8 lanes per pair, EIGHT radial functions per lane (two 16-byte loads per moment row), 16 pairs per
// wave (K = 32).  Every lane fetches its own pair's record, so there is no shuffle, and the 40 row loads of all 16 pairs are in
// flight together: the kernel is a gather (0.5 GB at 256 x 64 atoms) and needs that many requests outstanding; that is what makes it fast.
template <int NTP, int LPP>  // LPP = K / 8 lanes per pair
__global__ __launch_bounds__(256) void k_embed_pair_rb8(Graph g, RadialParams rp, const float* __restrict__ ps,
                                                        const float* __restrict__ gm, float* __restrict__ gd,
                                                        float* __restrict__ g_rhat, float* __restrict__ g_delta,
                                                        const float* __restrict__ slots, int n_slots, int64_t slot_stride) {
  constexpr int PPB = 256 / LPP;  // pairs per block
  constexpr int K = LPP * 8, KP = K + 4;
  if (g.counts[2]) return;
  const int P = g.counts[0];
  const int item = xcd_chunk_act(blockIdx.x, (P + PPB - 1) / PPB);  // grid: pair capacity; items: existing pairs
  if (item < 0) return;
  const int p = item * PPB + threadIdx.x / LPP;
  if (p >= P) return;
  const int c8 = threadIdx.x % LPP;  // this lane's chunk of eight radial functions
  const int i = g.pair_i[p], j = g.pair_j[p];
  const int ti = g.tix[i], tj = g.tix[j];
  const float r0 = g.prhat[p * 3], r1 = g.prhat[p * 3 + 1], r2 = g.prhat[p * 3 + 2];
  const float4 sc = *reinterpret_cast<const float4*>(ps + (int64_t)p * 8);  // C, C', C0, C0' (k_pair_scalars)
  const float c = sc.x, dc = sc.y, c0 = sc.z, dc0 = sc.w;
  const float u = ps[(int64_t)p * 8 + 4];
  const float alpha = 5.0f / (rp.up - rp.lo);
  const float* gi = gm + ((int64_t)i * NTP + tj) * 10 * KP + 8 * c8;
  const float* gj = gm + ((int64_t)j * NTP + ti) * 10 * KP + 8 * c8;
  // b[g][k] = ai[g][k] + s_g aj[g][k] for this lane's eight k (rb_pair_channel); lane c8 == 0 also takes the bias moment
  float b[10][8], bb[10];
#pragma unroll
  for (int w = 0; w < 10; ++w) {
    const float4 a0 = pld4(gi + w * KP), a1 = pld4(gi + w * KP + 4), e0 = pld4(gj + w * KP), e1 = pld4(gj + w * KP + 4);
    const float sg = (w >= 1 && w <= 3) ? -1.f : 1.f;
    b[w][0] = a0.x + sg * e0.x; b[w][1] = a0.y + sg * e0.y; b[w][2] = a0.z + sg * e0.z; b[w][3] = a0.w + sg * e0.w;
    b[w][4] = a1.x + sg * e1.x; b[w][5] = a1.y + sg * e1.y; b[w][6] = a1.z + sg * e1.z; b[w][7] = a1.w + sg * e1.w;
    bb[w] = c8 == 0 ? gi[w * KP + K] + sg * gj[w * KP + K] : 0.f;  // gi / gj carry the lane offset 8 c8 = 0 here
  }
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  auto add = [&](float psi, float dpsi, float b0, float b1, float b2, float b3, float b4, float b5, float b6, float b7, float b8_,
                 float b9) {
    const float h0 = b1 + (2.f * b4 * r0 + b5 * r1 + b6 * r2);
    const float h1 = b2 + (b5 * r0 + 2.f * b7 * r1 + b8_ * r2);
    const float h2 = b3 + (b6 * r0 + b8_ * r1 + 2.f * b9 * r2);
    const float G = b0 + r0 * (b1 + 0.5f * (h0 - b1)) + r1 * (b2 + 0.5f * (h1 - b2)) + r2 * (b3 + 0.5f * (h2 - b3));
    acc[0] += dpsi * G;
    acc[1] += psi * h0;
    acc[2] += psi * h1;
    acc[3] += psi * h2;
  };
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int k = 8 * c8 + q;
    const float mu = rp.means[k], beta = rp.betas[k];
    const float gk = __expf(-beta * (u - mu) * (u - mu));
    const float phi = c0 * gk, dphi = dc0 * gk + c0 * gk * (-2.0f * beta * (u - mu)) * (-alpha * u);
    add(c * phi, dc * phi + c * dphi, b[0][q], b[1][q], b[2][q], b[3][q], b[4][q], b[5][q], b[6][q], b[7][q], b[8][q], b[9][q]);
  }
  add(c, dc, bb[0], bb[1], bb[2], bb[3], bb[4], bb[5], bb[6], bb[7], bb[8], bb[9]);  // zero except in lane c8 == 0
#pragma unroll
  for (int q = 0; q < 4; ++q) acc[q] = row_sum(acc[q], LPP);
  if (c8 == 0) {
    if (g_delta) {
      pair_finish(g, p, acc[0], acc[1], acc[2], acc[3], r0, r1, r2, slots, n_slots, slot_stride, g_delta);
    } else {
      gd[p] += acc[0];
      g_rhat[p * 3] = acc[1];
      g_rhat[p * 3 + 1] = acc[2];
      g_rhat[p * 3 + 2] = acc[3];
    }
  }
}
void launch_embed_pair_rb(const Graph& g, int Pcap, int N, RadialParams rp, int ntp, const float* ps, const float* gm, float* gd,
                          float* g_rhat, hipStream_t s, float* g_delta, const float* slots, int n_slots, int64_t slot_stride) {
  if (Pcap <= 0 || N <= 0) return;
  const int lpp = rp.K / 8;
  const dim3 grid(cdivp(Pcap, 256 / lpp)), block(256);
#define RB_P8(NTP_, LPP_) \
  hipLaunchKernelGGL((k_embed_pair_rb8<NTP_, LPP_>), grid, block, 0, s, g, rp, ps, gm, gd, g_rhat, g_delta, slots, n_slots, slot_stride)
  if (ntp == 4 && lpp == 4) RB_P8(4, 4);
  else if (ntp == 4) RB_P8(4, 8);
  else if (lpp == 4) RB_P8(8, 4);
  else RB_P8(8, 8);
#undef RB_P8
}

// g_delta = (g_r - (g_r . r) r) / d + g_d r     (reference neighbor_utils.py:11-46; zero for d = 0)
__global__ void k_geom_gd(Graph g, const float* __restrict__ gd, const float* __restrict__ g_rhat, float* __restrict__ g_delta,
                          const float* __restrict__ slots, int n_slots, int64_t slot_stride) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= g.counts[0] || g.counts[2]) return;
  const float r0 = g.prhat[p * 3], r1 = g.prhat[p * 3 + 1], r2 = g.prhat[p * 3 + 2];
  const float d = g.pd[p];
  const float inv = d > 0.f ? 1.0f / d : 0.f;
  const float a0 = g_rhat[p * 3], a1 = g_rhat[p * 3 + 1], a2 = g_rhat[p * 3 + 2];
  const float dot = a0 * r0 + a1 * r1 + a2 * r2;
  float s = gd[p];
  for (int k = 0; k < n_slots; ++k)  // per-(layer, wave) partial sums of the merged adjoint sweeps, both directions
    s += slots[k * slot_stride + 2 * (int64_t)p] + slots[k * slot_stride + 2 * (int64_t)p + 1];
  g_delta[p * 3] = (a0 - dot * r0) * inv + s * r0;
  g_delta[p * 3 + 1] = (a1 - dot * r1) * inv + s * r1;
  g_delta[p * 3 + 2] = (a2 - dot * r2) * inv + s * r2;
}
void launch_geom_gd(const Graph& g, int Pcap, const float* gd, const float* g_rhat, float* g_delta, hipStream_t s, const float* slots,
                    int n_slots, int64_t slot_stride) {
  if (Pcap <= 0) return;
  hipLaunchKernelGGL(k_geom_gd, dim3(cdivp(Pcap, 256)), dim3(256), 0, s, g, gd, g_rhat, g_delta, slots, n_slots, slot_stride);
}

}  // namespace tn
