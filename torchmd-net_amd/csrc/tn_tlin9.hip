// Nine-component tensor linear with the per-atom 3x3 algebra fused in (gfx950):
//
//     C[atom, c, n] = epilogue( sum_k prologue(A)[atom, c, k] * W_type(c)[n, k] ),   c = 0 (I) | 1..3 (A) | 4..8 (S)
//
// reference tensornet.py:595-617, 745-754, 808-812 and the hand-written adjoints of SURVEY Appendix C.  The grouped GEMM of
// tn_gemm_sb1.hip treats the nine components as nine independent row panels, so everything that couples the components of one
// atom - X / (||X||^2 + 1), X_hat + dX + dX.dX, the adjoint of the group product, the normalisation adjoint - ran as separate
// elementwise launches around it, each a round trip of [N, 9, F] tensors through HBM (0.58 ms of a 2.86 ms step in round 3).
// Here a block (8 waves, one per CU, persistent over its tiles) owns 32 ATOMS x all 9 components x 128 output channels:
//   * staging (prologue): a thread holds the nine components of (atom, 2 input channels) of a DOUBLE chunk (32 channels = one
//     128-byte line per row), so the per-(atom, channel) algebra runs in registers while the pair is split into its three bf16
//     planes on the way to LDS; two register slots keep the next two double chunks in flight;
//   * product: exact 3-way bf16 split, six MFMA products per fp32 product (tn_gemm_sb.hip); waves 0-3 own components 0-3,
//     waves 4-7 components 4-8, each for one 32-column block: an accumulator row block = the 32 atoms of one component, so the
//     nine components of (atom, column) sit at the same lane and element of nine accumulators.  Weights never touch LDS:
//     their fragment-major image (split_weight_fm) is read straight into MFMA operand registers, one chunk ahead, every
//     fragment by exactly one wave;
//   * the two wave classes run the steps of a double chunk in rotated order (stage - multiply / multiply - stage), as two
//     separate loops: the staging of one wave of a SIMD runs beside the products of the other;
//   * epilogue: 8 atoms at a time go through LDS (aliasing two free chunk buffers) into (atom, 2 columns)-per-lane order - nine
//     components in one lane, a wave = one atom = 512 contiguous bytes per component - the extra operands of the fused neighbour
//     were requested before the dump.
// Measured (C2, 9 x 16384 x 128 x 128, profiles/r04_notes.md): 50 us for the plain product (the grouped GEMM: 47), 70 us with the
// update in the epilogue (was 47 + 43), 73 us with the normalisation adjoint (was 47 + 74).  What the kernel is NOT: a
// streaming kernel - memory-only it runs at 5.1 TB/s, products-only in 23 us, together in 50: see the notes for the timeline.
#include <cstdlib>

#include "tn_common.h"
#include "tn_gemm_sb.h"
#include "tn_tlin9.h"

namespace tn {

constexpr int T9_RA = 32;                  // atoms per tile
constexpr int T9_PLANE = 9 * T9_RA * 32;   // one bf16 plane of a [288 rows][16 k] chunk
constexpr int T9_STAGE = 3 * T9_PLANE;     // 27 648 B
constexpr int T9_EG = 8;                   // atoms per epilogue group
constexpr int T9_NT = 128;                 // output channels per block

typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2v ld2(const float* p) { return *reinterpret_cast<const f2v*>(p); }
__device__ __forceinline__ void st2(float* p, f2v v) { *reinterpret_cast<f2v*>(p) = v; }
// wave-uniform base + 32-bit per-lane element offset: the form the compiler turns into `global_load ... v_off, s[base:base+1]`
// (one offset register for all nine components instead of a 64-bit address pair each)
__device__ __forceinline__ f2v ldu2(const float* ubase, unsigned off) {
  return *reinterpret_cast<const f2v*>(reinterpret_cast<const char*>(ubase) + (size_t)(off * 4u));
}
__device__ __forceinline__ float ldu1(const float* ubase, unsigned off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(ubase) + (size_t)(off * 4u));
}
__device__ __forceinline__ void stu2(float* ubase, unsigned off, f2v v) {
  *reinterpret_cast<f2v*>(reinterpret_cast<char*>(ubase) + (size_t)(off * 4u)) = v;
}

// ---- per-(atom, channel) algebra of the fused neighbours; same arithmetic as the stand-alone kernels of tn_kernels.hip
__device__ __forceinline__ void t9_norm(float u[9]) {
  const float inv = 1.0f / (quad(u) + 1.0f);
#pragma unroll
  for (int c = 0; c < 9; ++c) u[c] *= inv;
}
__device__ __forceinline__ void t9_update_bwd(const float gg[9], const float d[9], float kap, float o[9]) {
  const M3 Gf = decompose_T(gg);
  const M3 dXt = transpose(compose(d));
  const M3 gdx = add(Gf, scale(add(matmul(Gf, dXt), matmul(dXt, Gf)), kap));
  compose_T(gdx, o);
}
__device__ __forceinline__ void t9_update(const float x[9], const float d[9], float kap, float o[9]) {
  float xh[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) xh[c] = x[c];
  t9_norm(xh);
  const M3 dX = compose(d);
  const M3 Xf = add(add(compose(xh), dX), scale(matmul(dX, dX), kap));
  decompose(Xf, o);
}
__device__ __forceinline__ void t9_norm_bwd(const float u[9], float gx[9] /* in: G + g_lin, out: new G */) {
  float dq[9];
  dquad(u, dq);
  const float inv = 1.0f / (quad(u) + 1.0f);
  float dot = 0.f;
#pragma unroll
  for (int c = 0; c < 9; ++c) dot += gx[c] * u[c];
  const float g_s = -dot * inv * inv;
#pragma unroll
  for (int c = 0; c < 9; ++c) gx[c] = gx[c] * inv + dq[c] * g_s;
}

template <int N_> struct IC { static constexpr int value = N_; };

// Why the loop looks the way it does (measured, profiles/r04_notes.md):
//  * one instruction stream for all waves and NO branch around a global load in the steady state: at the join of a wave-uniform
//    branch that issued loads hipcc merges the outstanding-load counters and waits vmcnt(0), which drained the prefetch every
//    chunk (first version: wave pairs took turns staging, 2.6 TB/s);
//  * a thread requests FULL 128-byte lines: two consecutive 16-channel chunks of a row are one line, and requesting the halves
//    one iteration apart parks the second request behind the first one's miss in the vector L1 (TCP_PENDING_STALL 55 % of the
//    kernel, second version: 1.3 TB/s).  So the unit of the pipeline is a DOUBLE chunk (32 input channels): 16 lanes x 8 bytes
//    per (atom, component), a lane's two channels are one bf16 pair of the LDS image (no lane exchange);
//  * vmcnt counts in order: the wait for the next chunk's weights (L2 hits, needed at once) also waits for every older request,
//    so the input prefetch is at most one double chunk deep whatever the ring size - two ring slots.
template <int PRO, int EPI>
__global__ __launch_bounds__(512, 2) void k_tlin9(Tl9Args a, int tiles_m, int tiles_n) {
  // four chunk buffers: double chunk d lives in buffers 2 (d & 1), 2 (d & 1) + 1; the epilogue's transposition buffer aliases
  // the two buffers of a tile's last double chunk (free by then; the other two already hold the next tile's first double chunk)
  __shared__ __attribute__((aligned(16))) unsigned char sA[4 * T9_STAGE];
  float* const sE = reinterpret_cast<float*>(sA + 2 * T9_STAGE);  // [8 atoms][9][128 columns] = 36 864 B <= 2 stages
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);  // scalar: the class branches below become s_cbranch
  const int wn = wave & 3, wc = wave >> 2;  // NOT the scalar copy: with a scalar class branch hipcc merges the two product blocks and indexes the accumulators through scratch
  const int sa = tid >> 4, skp = tid & 15;  // staging role: atom of the tile, pair of the double chunk's 32 input channels
  const int N = a.N, F = a.F, nd = F >> 5, NB = F >> 5;  // nd: double chunks per tile (even: F % 128 == 0 -> multiple of 4)
  const int F9 = 9 * F;
  const int total = tiles_m * tiles_n;
  if ((int)blockIdx.x >= total) return;
  const int my_tiles = (total - 1 - (int)blockIdx.x) / (int)gridDim.x + 1;
  const int ndt = my_tiles * nd;  // double chunks of this block, numbered across its tiles

  auto tile_of = [&](int ts, int& m0, int& n0) __attribute__((always_inline)) {  // column tiles of one atom tile are consecutive
    const int t = (int)blockIdx.x + ts * (int)gridDim.x;
    const int tn_ = t % tiles_n;
    m0 = (t / tiles_n) * T9_RA;
    n0 = tn_ * T9_NT;
  };

  // ---------------------------------------------------------------- staging: global -> register ring -> (algebra, split) -> LDS
  f2v sr[2][9];
  f2v sr2[2][9];  // second operand (TL9_PRO_UPDBWD only; dead otherwise)
  float s_kap[2];
  auto load_dbl = [&](auto S_, int dd) __attribute__((always_inline)) {
    constexpr int S = decltype(S_)::value;
    dd = dd < ndt ? dd : ndt - 1;  // past the end: re-request the last one (no branch around loads)
    const int ts = dd / nd, kd = dd - ts * nd;
    int m0, n0;
    tile_of(ts, m0, n0);
    int n = m0 + sa;
    if (n >= N) n = N - 1;  // rows past the end: any valid row (their outputs are not stored)
    if (PRO == TL9_PRO_UPDBWD) s_kap[S] = a.kap ? a.kap[n] : 1.0f;
    const unsigned voff = (unsigned)((n - m0) * F9 + skp * 2);
    const float* p = a.A + (int64_t)m0 * F9 + kd * 32;  // wave-uniform
#pragma unroll
    for (int c = 0; c < 9; ++c) sr[S][c] = ldu2(p + c * F, voff);
    if (PRO == TL9_PRO_UPDBWD) {
      const float* p2 = a.A2 + (int64_t)m0 * F9 + kd * 32;
#pragma unroll
      for (int c = 0; c < 9; ++c) sr2[S][c] = ldu2(p2 + c * F, voff);
    }
  };
  // lanes with skp < 8 hold channels of the double chunk's first chunk, the others of its second: two of the four buffers
  const int st_off = (skp >> 3) * T9_STAGE + (skp & 3) * 4;
  const int st_h = (skp >> 2) & 1;
  auto stage_dbl = [&](auto S_) __attribute__((always_inline)) {  // ring slot S = double chunk parity = buffer pair
    constexpr int S = decltype(S_)::value;
    unsigned char* buf = sA + 2 * S * T9_STAGE + st_off;
    if (PRO == TL9_PRO_NORM) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        __builtin_amdgcn_sched_barrier(0);  // one channel at a time: interleaving the two doubles the live temporaries
        float u[9];
#pragma unroll
        for (int c = 0; c < 9; ++c) u[c] = sr[S][c][j];
        t9_norm(u);
#pragma unroll
        for (int c = 0; c < 9; ++c) sr[S][c][j] = u[c];
      }
    } else if (PRO == TL9_PRO_UPDBWD) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        __builtin_amdgcn_sched_barrier(0);
        float gg[9], d[9], o[9];
#pragma unroll
        for (int c = 0; c < 9; ++c) {
          gg[c] = sr[S][c][j];
          d[c] = sr2[S][c][j];
        }
        t9_update_bwd(gg, d, s_kap[S], o);
#pragma unroll
        for (int c = 0; c < 9; ++c) sr[S][c][j] = o[c];
      }
    }
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      uint32_t h, m, l;
      split2(sr[S][c][0], sr[S][c][1], h, m, l);
      const int off = sb_piece(c * T9_RA + sa, st_h);
      *reinterpret_cast<uint32_t*>(buf + 0 * T9_PLANE + off) = h;
      *reinterpret_cast<uint32_t*>(buf + 1 * T9_PLANE + off) = m;
      *reinterpret_cast<uint32_t*>(buf + 2 * T9_PLANE + off) = l;
    }
  };

  // ---------------------------------------------------------------- weights: fragment-major image -> operand registers
  // wave (wn, wc = 0): components 0..3 (types 0, 1) ; (wn, wc = 1): components 4..8 (type 2 only).  The two classes run separate
  // loops (below), so their request counts may differ: every weight fragment is read by exactly one wave of the block
  bf16x8 bfr[2][2][3];  // [chunk parity][slot][plane]; slot 0: type 0 (wc 0) / type 2 (wc 1), slot 1: type 1 (wc 0 only)
  const uint16_t* const wsl0 = a.Wfm[wc == 0 ? 0 : 2] + lane * 8;
  const uint16_t* const wsl1 = a.Wfm[1] + lane * 8;
  auto load_b = [&](auto PB_, auto WC_, int kt, int n0) __attribute__((always_inline)) {  // weights of chunk kt (16 channels), column tile n0
    constexpr int PB = decltype(PB_)::value, WC = decltype(WC_)::value;
    const int64_t o = ((int64_t)kt * NB + (n0 >> 5) + wn) * 3 * 512;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      bfr[PB][0][p] = *reinterpret_cast<const bf16x8*>(wsl0 + o + p * 512);
      if (WC == 0) bfr[PB][1][p] = *reinterpret_cast<const bf16x8*>(wsl1 + o + p * 512);
    }
  };

  floatx16 acc[5];
  auto mma_chunk = [&](auto PB_, auto WC_, const unsigned char* cur) __attribute__((always_inline)) {
    constexpr int PB = decltype(PB_)::value, WC = decltype(WC_)::value;
#define T9_MMA(ci, slot, pa_, pb_) \
  acc[ci] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ci][pa_], bfr[PB][slot][pb_], acc[ci], 0, 0, 0);
#define T9_AF(ci, comp) \
  _Pragma("unroll") for (int p = 0; p < 3; ++p) af[ci][p] = *reinterpret_cast<const bf16x8*>(cur + p * T9_PLANE + (comp) * (T9_RA * 32));
    if (WC == 0) {
      bf16x8 af[4][3];
      T9_AF(0, 0) T9_AF(1, 1) T9_AF(2, 2) T9_AF(3, 3)
      T9_MMA(0, 0, 0, 2) T9_MMA(1, 1, 0, 2) T9_MMA(2, 1, 0, 2) T9_MMA(3, 1, 0, 2)
      T9_MMA(0, 0, 2, 0) T9_MMA(1, 1, 2, 0) T9_MMA(2, 1, 2, 0) T9_MMA(3, 1, 2, 0)
      T9_MMA(0, 0, 1, 1) T9_MMA(1, 1, 1, 1) T9_MMA(2, 1, 1, 1) T9_MMA(3, 1, 1, 1)
      T9_MMA(0, 0, 0, 1) T9_MMA(1, 1, 0, 1) T9_MMA(2, 1, 0, 1) T9_MMA(3, 1, 0, 1)
      T9_MMA(0, 0, 1, 0) T9_MMA(1, 1, 1, 0) T9_MMA(2, 1, 1, 0) T9_MMA(3, 1, 1, 0)
      T9_MMA(0, 0, 0, 0) T9_MMA(1, 1, 0, 0) T9_MMA(2, 1, 0, 0) T9_MMA(3, 1, 0, 0)
    } else {
      bf16x8 af[5][3];
      T9_AF(0, 4) T9_AF(1, 5) T9_AF(2, 6) T9_AF(3, 7) T9_AF(4, 8)
      T9_MMA(0, 0, 0, 2) T9_MMA(1, 0, 0, 2) T9_MMA(2, 0, 0, 2) T9_MMA(3, 0, 0, 2) T9_MMA(4, 0, 0, 2)
      T9_MMA(0, 0, 2, 0) T9_MMA(1, 0, 2, 0) T9_MMA(2, 0, 2, 0) T9_MMA(3, 0, 2, 0) T9_MMA(4, 0, 2, 0)
      T9_MMA(0, 0, 1, 1) T9_MMA(1, 0, 1, 1) T9_MMA(2, 0, 1, 1) T9_MMA(3, 0, 1, 1) T9_MMA(4, 0, 1, 1)
      T9_MMA(0, 0, 0, 1) T9_MMA(1, 0, 0, 1) T9_MMA(2, 0, 0, 1) T9_MMA(3, 0, 0, 1) T9_MMA(4, 0, 0, 1)
      T9_MMA(0, 0, 1, 0) T9_MMA(1, 0, 1, 0) T9_MMA(2, 0, 1, 0) T9_MMA(3, 0, 1, 0) T9_MMA(4, 0, 1, 0)
      T9_MMA(0, 0, 0, 0) T9_MMA(1, 0, 0, 0) T9_MMA(2, 0, 0, 0) T9_MMA(3, 0, 0, 0) T9_MMA(4, 0, 0, 0)
    }
#undef T9_MMA
#undef T9_AF
  };

  // ---------------------------------------------------------------- epilogue: 8 atoms per group (a wave = one atom)
  const unsigned cp2 = 2 * (tid & 63);
  const int F3 = 3 * F;
  f2v x0[9], x1[(EPI == TL9_EPI_NORMBWD || EPI == TL9_EPI_NORMBWD_GATE) ? 9 : 1], gt[3];
  auto epi_loads = [&](int n, int n0) __attribute__((always_inline)) {  // operands of the fused neighbour for (atom n, 2 columns)
    const bool ok = n < N;
    const int64_t rowu = (int64_t)(ok ? n : 0) * F9 + n0;  // wave-uniform (a wave = one atom); lanes add cp2
    if (EPI == TL9_EPI_UPDATE || EPI == TL9_EPI_NORMBWD || EPI == TL9_EPI_NORMBWD_GATE ||
        EPI == TL9_EPI_EMBBWD) {
#pragma unroll
      for (int c = 0; c < 9; ++c) x0[c] = ldu2(a.e0 + rowu + c * F, cp2);
    }
    if (EPI == TL9_EPI_NORMBWD || EPI == TL9_EPI_NORMBWD_GATE) {
#pragma unroll
      for (int c = 0; c < 9; ++c) x1[c] = ldu2(a.e1 + rowu + c * F, cp2);
    }
    if (EPI == TL9_EPI_MULGATE) {
#pragma unroll
      for (int k = 0; k < 3; ++k) gt[k] = ldu2(a.e3 + (int64_t)(ok ? n : 0) * F3 + n0 + k * F, cp2);
    }
    if (EPI == TL9_EPI_EMBBWD) gt[0] = ldu2(a.e1 + (int64_t)(ok ? n : 0) * F + n0, cp2);
  };
  auto epi_pass = [&](int n, int n0, int ea) __attribute__((always_inline)) {  // the nine components of (atom n, 2 columns): LDS -> algebra -> stores
    const bool ok = n < N;
    const int64_t rowu = (int64_t)(ok ? n : 0) * F9 + n0;  // wave-uniform (a wave = one atom); lanes add cp2
    const float kap = (a.kap && ok) ? a.kap[n] : 1.0f;
    f2v v[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) v[c] = *reinterpret_cast<const f2v*>(sE + (ea * 9 + c) * T9_NT + cp2);
    if (EPI == TL9_EPI_PLAIN) {
      if (ok) {
#pragma unroll
        for (int c = 0; c < 9; ++c) stu2(a.C + rowu + c * F, cp2, v[c]);
      }
    } else if (EPI == TL9_EPI_MULGATE) {
      if (ok) {
#pragma unroll
        for (int c = 0; c < 9; ++c) {
          stu2(a.o1 + rowu + c * F, cp2, v[c]);
          stu2(a.C + rowu + c * F, cp2, v[c] * gt[type_of(c)]);
        }
      }
    } else if (EPI == TL9_EPI_UPDATE) {
      f2v ft[3];
      if (ok) {
#pragma unroll
        for (int c = 0; c < 9; ++c) stu2(a.C + rowu + c * F, cp2, v[c]);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        __builtin_amdgcn_sched_barrier(0);  // one column at a time: interleaving the two doubles the live 3x3 temporaries
        float d[9], x[9], o[9];
#pragma unroll
        for (int c = 0; c < 9; ++c) {
          d[c] = v[c][j];
          x[c] = x0[c][j];
        }
        t9_update(x, d, kap, o);
#pragma unroll
        for (int c = 0; c < 9; ++c) v[c][j] = o[c];
        const float t = o[4] + o[7];
        ft[0][j] = 3.0f * o[0] * o[0];
        ft[1][j] = 2.0f * (o[1] * o[1] + o[2] * o[2] + o[3] * o[3]);
        ft[2][j] = o[4] * o[4] + o[7] * o[7] + t * t + 2.0f * (o[5] * o[5] + o[6] * o[6] + o[8] * o[8]);
      }
      if (ok) {
#pragma unroll
        for (int c = 0; c < 9; ++c) stu2(a.o1 + rowu + c * F, cp2, v[c]);
        if (a.want_feat) {
#pragma unroll
          for (int k = 0; k < 3; ++k) stu2(a.o2 + (int64_t)n * F3 + n0 + k * F, cp2, ft[k]);
        }
      }
    } else if (EPI == TL9_EPI_NORMBWD || EPI == TL9_EPI_NORMBWD_GATE) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        __builtin_amdgcn_sched_barrier(0);  // one column at a time: interleaving the two doubles the live 3x3 temporaries
        float u[9], gx[9];
#pragma unroll
        for (int c = 0; c < 9; ++c) {
          u[c] = x0[c][j];
          gx[c] = x1[c][j] + v[c][j];
        }
        t9_norm_bwd(u, gx);
#pragma unroll
        for (int c = 0; c < 9; ++c) v[c][j] = gx[c];
      }
      if (EPI == TL9_EPI_NORMBWD) {
        if (ok) {
#pragma unroll
          for (int c = 0; c < 9; ++c) stu2(a.C + rowu + c * F, cp2, v[c]);
        }
      } else {
        // embedding gate adjoint: X1[c] = UX[c] * gate[type(c)]  (k_embed_gate_bwd); the new G itself is not stored
        const int64_t r3 = (int64_t)(ok ? n : 0) * F3 + n0;
        f2v ga[3] = {(f2v)(0.f), (f2v)(0.f), (f2v)(0.f)};
#pragma unroll
        for (int k = 0; k < 3; ++k) gt[k] = ldu2(a.e3 + r3 + k * F, cp2);
#pragma unroll
        for (int c = 0; c < 9; ++c) x0[c] = ldu2(a.e2 + rowu + c * F, cp2);  // UX
#pragma unroll
        for (int c = 0; c < 9; ++c) {
          ga[type_of(c)] += v[c] * x0[c];
          if (ok) stu2(a.C + rowu + c * F, cp2, v[c] * gt[type_of(c)]);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const f2v a2 = ldu2(a.e4 + r3 + k * F, cp2);
          f2v o;
#pragma unroll
          for (int j = 0; j < 2; ++j) o[j] = ga[k][j] * silu_grad(a2[j]);
          if (ok) stu2(a.o1 + r3 + k * F, cp2, o);
        }
      }
    } else if (EPI == TL9_EPI_EMBBWD) {
      f2v o9;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        __builtin_amdgcn_sched_barrier(0);  // one column at a time: interleaving the two doubles the live 3x3 temporaries
        float gl[9], u[9], dq[9];
#pragma unroll
        for (int c = 0; c < 9; ++c) {
          gl[c] = v[c][j];
          u[c] = x0[c][j];
        }
        dquad(u, dq);
        const float gs = gt[0][j];
#pragma unroll
        for (int c = 0; c < 9; ++c) gl[c] += dq[c] * gs;
        const float third = (gl[4] + gl[7]) * (1.0f / 3.0f);
#pragma unroll
        for (int c = 0; c < 9; ++c) v[c][j] = gl[c];
        v[4][j] = gl[4] - third;
        v[7][j] = gl[7] - third;
        o9[j] = -third;
      }
      if (ok) {
        float* op = a.o1 + (int64_t)n * 10 * F + n0;
#pragma unroll
        for (int c = 0; c < 9; ++c) stu2(op + c * F, cp2, v[c]);
        stu2(op + 9 * F, cp2, o9);
      }
    }
  };
  auto epi_group = [&](auto G_, int m0, int n0) __attribute__((always_inline)) {
    constexpr int G = decltype(G_)::value;
    const int n = m0 + T9_EG * G + wave_u;
    epi_loads(n, n0);  // requested before the dump so that their latency is covered by it
    {  // accumulators of this group's 8 atoms -> LDS [atom][component][128 columns]
      float* dst = sE + 32 * wn + (lane & 31) + (4 * (lane >> 5) * 9) * T9_NT;
      // one code path for both wave classes (the fifth accumulator of waves 0-3 is idle): a branch per class made the compiler
      // index the accumulators through scratch memory
#pragma unroll
      for (int ci = 0; ci < 5; ++ci) {
        if (ci < 4 || wc) {
          const int comp = (wc ? 4 : 0) + ci;
#pragma unroll
          for (int e = 0; e < 4; ++e) dst[(e * 9 + comp) * T9_NT] = acc[ci][4 * G + e];
        }
      }
    }
    __syncthreads();
    epi_pass(n, n0, wave_u);
    __syncthreads();  // the next group's dump (or the next tile's staging) overwrites the buffer
  };

  // ---------------------------------------------------------------- pipeline start-up
  load_dbl(IC<0>{}, 0);
  load_dbl(IC<1>{}, 1);
  stage_dbl(IC<0>{});
  load_dbl(IC<0>{}, 2);
  int m0c, n0c;
  tile_of(0, m0c, n0c);
  if (wave < 4) {
    load_b(IC<0>{}, IC<0>{}, 0, n0c);
    load_b(IC<1>{}, IC<0>{}, 1, n0c);
  } else {
    load_b(IC<0>{}, IC<1>{}, 0, n0c);
    load_b(IC<1>{}, IC<1>{}, 1, n0c);
  }
  __syncthreads();

  int d = 0;  // double chunk being multiplied

  // One double chunk = {S: the next one staged from its ring slot, L: the one after requested into that slot, M0 / M1: the two
  // chunks' products, each followed by the request for the weights two chunks on}, then a barrier.  The waves of the two halves
  // of the block (a SIMD hosts one wave of each) run the SAME steps in ROTATED order - S L M0 M1 and M0 M1 S L - as two separate
  // loops: while one wave of a SIMD stages (VALU, LDS writes, waits for its requests) the other one keeps the matrix pipe busy;
  // with one order for all waves every wave was in the same phase between two barriers and the phases added up (ablation: 23 us
  // of compute + 27 us of memory = 50 us).  Separate loops, not a branch inside one loop: no join of paths with different
  // outstanding requests per iteration (see the note at the top of the kernel).  No branch around a global load in either.
  auto dbl = [&](auto S_, auto ROT_, int kd, int n0_next) __attribute__((always_inline)) {
    constexpr int S = decltype(S_)::value, ROT = decltype(ROT_)::value;
    if (!ROT) {
      stage_dbl(IC<S ^ 1>{});
      load_dbl(IC<S ^ 1>{}, d + 3);
    }
    const unsigned char* cur = sA + 2 * S * T9_STAGE + sb_piece(lane & 31, lane >> 5);
    const bool wrap = kd + 1 == nd;  // the weights two chunks on belong to the next tile's first double chunk
    const int k2 = wrap ? 0 : 2 * kd + 2, n2 = wrap ? n0_next : n0c;
    mma_chunk(IC<0>{}, ROT_, cur);
    load_b(IC<0>{}, ROT_, k2, n2);
    mma_chunk(IC<1>{}, ROT_, cur + T9_STAGE);
    load_b(IC<1>{}, ROT_, k2 + 1, n2);
    if (ROT) {
      stage_dbl(IC<S ^ 1>{});
      load_dbl(IC<S ^ 1>{}, d + 3);
    }
    __syncthreads();
    ++d;
  };
  auto run = [&](auto ROT_) __attribute__((always_inline)) {
    for (int ts = 0; ts < my_tiles; ++ts) {
      int m0n = m0c, n0n = n0c;
      if (ts + 1 < my_tiles) tile_of(ts + 1, m0n, n0n);
#pragma unroll
      for (int ci = 0; ci < 5; ++ci)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[ci][e] = 0.f;
      for (int kd = 0; kd < nd; kd += 2) {
        dbl(IC<0>{}, ROT_, kd, n0n);
        dbl(IC<1>{}, ROT_, kd + 1, n0n);
      }
      epi_group(IC<0>{}, m0c, n0c);
      epi_group(IC<1>{}, m0c, n0c);
      epi_group(IC<2>{}, m0c, n0c);
      epi_group(IC<3>{}, m0c, n0c);
      m0c = m0n;
      n0c = n0n;
    }
  };
  if (wave < 4) run(IC<0>{});
  else run(IC<1>{});
}

// ------------------------------------------------------------------------------------------------ weight image
// [k chunk of 16][column block of 32][plane][lane][8 bf16]: lane l of a wave holds W[32 nb + (l & 31)][16 kc + 8 (l >> 5) + 0..7],
// i.e. the B operand of v_mfma_f32_32x32x16_bf16 as one contiguous 1 KB read per (chunk, column block, plane)
__global__ void k_split_weight_fm(const float* __restrict__ W, int n, int k, uint16_t* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int NB = n >> 5, nkc = k >> 4;
  if (idx >= nkc * NB * 64) return;
  const int lane = idx & 63, nb = (idx >> 6) % NB, kc = (idx >> 6) / NB;
  const float* src = W + (int64_t)(nb * 32 + (lane & 31)) * k + kc * 16 + 8 * (lane >> 5);
  const float4 u = *reinterpret_cast<const float4*>(src), v = *reinterpret_cast<const float4*>(src + 4);
  uint4 h, m, l;
  split8(u, v, h, m, l);
  uint16_t* o = out + (((int64_t)kc * NB + nb) * 3) * 512 + lane * 8;
  *reinterpret_cast<uint4*>(o) = h;
  *reinterpret_cast<uint4*>(o + 512) = m;
  *reinterpret_cast<uint4*>(o + 1024) = l;
}
size_t split_weight_fm_elems(int64_t n, int64_t k) { return (size_t)(n * k * 3); }
void launch_split_weight_fm(const float* W_dev, int64_t n, int64_t k, uint16_t* out_dev, hipStream_t s) {
  const int total = (int)((k >> 4) * (n >> 5) * 64);
  hipLaunchKernelGGL(k_split_weight_fm, dim3((total + 255) / 256), dim3(256), 0, s, W_dev, (int)n, (int)k, out_dev);
}

static int t9_num_cu() {
  static const int n_cu = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n;
  }();
  return n_cu;
}

bool tlin9_ok(int N, int F, bool any_rounds) {
  static const bool off = getenv("TMDNET_NO_TLIN9") != nullptr || getenv("TMDNET_NO_SPLIT_BF16") != nullptr;  // developer switches
  if (off || F < T9_NT || F % T9_NT) return false;  // (F % 128 == 0: the double-chunk count per tile is even)
  // batch scale only: below ~128 tiles the launch does not fill the chip and the split-K kernels of the small-system path win
  // (measured with the threshold at 1: 64 atoms 0.233 -> 0.391 ms per replayed step, 2048 atoms 0.65 -> 0.74, 4096 atoms equal)
  const int64_t tiles = (int64_t)((N + T9_RA - 1) / T9_RA) * (F / T9_NT);
  if (tiles < 128) return false;
  if (any_rounds) return true;  // the caller has no other schedule for these rows (the owned range of a halo exchange)
  // one persistent block per CU: the launch takes ceil(tiles / CUs) tile times, so a mostly empty last round is paid in full
  // (10 125 atoms = 317 tiles on 256 CUs: two rounds for 1.24 rounds of work, the water box stepped 3.02 -> 3.16 ms)
  const int64_t n_cu = t9_num_cu(), rounds = (tiles + n_cu - 1) / n_cu;
  return tiles <= n_cu || 5 * tiles >= 4 * rounds * n_cu;
}

int launch_tlin9(const Tl9Args& a, int pro, int epi, hipStream_t s) {
  const int tiles_m = (a.N + T9_RA - 1) / T9_RA, tiles_n = a.F / T9_NT;
  const int n_cu = t9_num_cu();
  const int total = tiles_m * tiles_n;
  const dim3 grid(total < n_cu ? total : n_cu), block(512);  // persistent: one block per CU
#define T9_LAUNCH(P, E) hipLaunchKernelGGL((k_tlin9<P, E>), grid, block, 0, s, a, tiles_m, tiles_n)
  if (pro == TL9_PRO_NORM && epi == TL9_EPI_PLAIN) T9_LAUNCH(TL9_PRO_NORM, TL9_EPI_PLAIN);
  else if (pro == TL9_PRO_PLAIN && epi == TL9_EPI_PLAIN) T9_LAUNCH(TL9_PRO_PLAIN, TL9_EPI_PLAIN);
  else if (pro == TL9_PRO_PLAIN && epi == TL9_EPI_MULGATE) T9_LAUNCH(TL9_PRO_PLAIN, TL9_EPI_MULGATE);
  else if (pro == TL9_PRO_PLAIN && epi == TL9_EPI_UPDATE) T9_LAUNCH(TL9_PRO_PLAIN, TL9_EPI_UPDATE);
  else if (pro == TL9_PRO_UPDBWD && epi == TL9_EPI_PLAIN) T9_LAUNCH(TL9_PRO_UPDBWD, TL9_EPI_PLAIN);
  else if (pro == TL9_PRO_PLAIN && epi == TL9_EPI_NORMBWD) T9_LAUNCH(TL9_PRO_PLAIN, TL9_EPI_NORMBWD);
  else if (pro == TL9_PRO_PLAIN && epi == TL9_EPI_NORMBWD_GATE) T9_LAUNCH(TL9_PRO_PLAIN, TL9_EPI_NORMBWD_GATE);
  else if (pro == TL9_PRO_PLAIN && epi == TL9_EPI_EMBBWD) T9_LAUNCH(TL9_PRO_PLAIN, TL9_EPI_EMBBWD);
  else return (int)hipErrorInvalidValue;
#undef T9_LAUNCH
  return (int)hipGetLastError();
}

}  // namespace tn
