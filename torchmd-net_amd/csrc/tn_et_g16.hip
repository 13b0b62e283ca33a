// Attention sweeps of the Equivariant Transformer, third generation (round 6): TILE sweeps.  A workgroup = one tile (a run of whole
// molecules of at most 64 rows, packed on the device every step: k_et_tile_pack) x one 32-channel slice; a 16-lane group per
// (row atom, slice), two channels per lane (reference torchmd_et.py:376-426 and its adjoint; the arithmetic per
// edge is that of the row sweeps k_et_attn_fwd_p / k_et_attn_bwd_p of tn_et.hip, which stay for systems whose tiles are not closed).
//
// Why (profiles/r06_notes.md, counters of the row sweeps on ET-SPICE 256 x 64 atoms):
//   * the reverse row sweep was VALU-bound, not byte-bound: 203 vector instructions per edge and wave, the vector pipe busy 94 % of
//     the kernel, which is why storing the pair rows as bf16 had bought 8 % only.  A third of the stream was not arithmetic: 64-bit
//     address pairs per load (34), register copies of the prefetched edge (26), IEEE divides in the two sigmoids (20), one head sum
//     per CHANNEL lane (4 x 4 DPP steps).  Here the products are v_pk_mul / v_pk_fma (two channels per instruction), a head of 16
//     channels is 8 lanes, the sigmoids are once per lane, loads come from a wave-uniform base + one 32-bit offset register;
//   * two dependent request latencies per step (index -> row) at four waves per SIMD: the tile's node rows (12 values x 32 channels
//     x 64 atoms = 98 KB; forward: 7 values) are staged in LDS once, and everything else an edge needs is one 32-byte record;
//   * both directed edges of a pair fetched the pair's filter rows (1.87 x the distinct bytes): in tiles that are at least 3/4 full
//     the rows walk in SLOT order - step `it` of row r is tile column (it - r) mod 64 - so the two rows of a pair are in the same
//     step; with fp32 rows each requests half of the pair's rows and they swap through an LDS mailbox (1.10 x the distinct bytes).
// A lane group owns its row's accumulators in registers (fixed order of edges: deterministic).  The distance / direction slots of
// the reverse sweep are per 32-channel slice (et_sweep_waves = F / 32 arrays per pair direction).  Which generation runs is a flag
// on the device (k_et_tile_open: closed tiles; k_et_tile_pack: full enough to pay), both are enqueued, one returns at once.
#include <cstdlib>
#include <type_traits>

#include "tn_common.h"
#include "tn_et.h"

namespace tn {

namespace g16 {  // (a named namespace: rocprofv3 tools of this repo cut kernel names at the first parenthesis)

typedef float f2v __attribute__((ext_vector_type(2)));

template <int CTRL>
__device__ __forceinline__ float dppf(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// sum over aligned groups of HL lanes (HL in {1, 2, 4, 8, 16}), every lane of the group gets it
template <int HL>
__device__ __forceinline__ float grp_sum(float v) {
  if (HL >= 2) v += dppf<0xB1>(v);    // quad_perm [1,0,3,2]
  if (HL >= 4) v += dppf<0x4E>(v);    // quad_perm [2,3,0,1]
  if (HL >= 8) v += dppf<0x141>(v);   // row_half_mirror
  if (HL >= 16) v += dppf<0x140>(v);  // row_mirror
  return v;
}
// four sums over the 16 lanes of a group for the price of five exchanges: totals in lanes 0-3 (a), 4-7 (b), 8-11 (c), 12-15 (d)
__device__ __forceinline__ float grp_sum4(float a, float b, float c, float d, int gl) {
  const bool hi = gl & 8;
  float k0 = hi ? c : a, k1 = hi ? d : b;
  const float s0 = hi ? a : c, s1 = hi ? b : d;
  k0 += dppf<0x140>(s0);
  k1 += dppf<0x140>(s1);
  const bool q = gl & 4;
  float v = q ? k1 : k0;
  v += dppf<0x141>(q ? k0 : k1);
  v += dppf<0xB1>(v);
  v += dppf<0x4E>(v);
  return v;
}
__device__ __forceinline__ float hadd(f2v v) { return v.x + v.y; }
__device__ __forceinline__ f2v splat(float s) { return f2v{s, s}; }
// sigmoid with the hardware reciprocal (1 ulp) instead of an IEEE divide (10 instructions)
__device__ __forceinline__ float sigm(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// wave-uniform base + 32-bit per-lane byte offset: `global_load ... v_off, s[base:base+1]`, one offset register for all the loads
// of a row instead of a 64-bit address pair each (the launcher takes this path only while every array stays below 2 GiB)
typedef uint32_t OffT;
__device__ __forceinline__ f2v ld2(const float* ubase, OffT boff) {
  return *reinterpret_cast<const f2v*>(reinterpret_cast<const char*>(ubase) + (size_t)boff);
}
template <typename T>
__device__ __forceinline__ T ld1(const T* ubase, int idx) {  // element idx (4-byte type) of a wave-uniform array
  return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(ubase) + (size_t)((uint32_t)idx * 4u));
}
// two consecutive channels of a pair row: fp32 (8 bytes) or bf16 (one dword: low half = even channel)
template <bool BF>
__device__ __forceinline__ f2v ldp2(const float* ubase, OffT boff) {
  if (BF) {
    const uint32_t w = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(ubase) + (size_t)boff);
    return f2v{__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
  }
  return *reinterpret_cast<const f2v*>(reinterpret_cast<const char*>(ubase) + (size_t)boff);
}

// first edge of row i's list in the order (row + column) mod 64 (et_rot_start of tn_et.hip, per lane group)
__device__ __forceinline__ int rot_start(const int* __restrict__ col, int e0, int e1, int i) {
  if (e1 <= e0) return e0;
  const int j0 = col[e0];
  const int target = j0 + ((64 - ((i + j0) & 63)) & 63);
  int lo = e0, hi = e1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (col[mid] < target) lo = mid + 1; else hi = mid;
  }
  return lo < e1 ? lo : e0;
}

// longest row of the workgroup (every wave runs that many steps so that the barriers match)
__device__ __forceinline__ int wg_max(int v, int* s_max) {
  if (threadIdx.x == 0) *s_max = 0;
  __syncthreads();
  if ((threadIdx.x & 15) == 0 && v > 0) atomicMax(s_max, v);
  __syncthreads();
  return *s_max;
}
// head sum: HL = 8 (16 channels per head) at compile time, any other width from the argument (wave-uniform branches)
template <int HL>
__device__ __forceinline__ float head_sum2(float v, int hl) {
  if (HL > 0) return grp_sum<HL>(v);
  if (hl >= 2) v += dppf<0xB1>(v);
  if (hl >= 4) v += dppf<0x4E>(v);
  if (hl >= 8) v += dppf<0x141>(v);
  if (hl >= 16) v += dppf<0x140>(v);
  return v;
}

// ---- the tile's node rows in LDS: value v of local atom a, channel pair gl at sN[(v * 64 + a) * 16 + gl].  A wave's four lane groups
// read four atoms (consecutive in the symmetric order): two even, two odd -> the 512 bytes of a ds_read_b64 in two passes (optimal)
constexpr int G16_ROWS = 64;
constexpr int G16_V = G16_ROWS * 16;  // f2v elements per value plane
constexpr int G16_RING = 8;           // steps whose (distance, direction) sums are kept in LDS before they are stored

// the row's edge of step it, clamped to a valid edge when the row has none in that step (nothing is accumulated for it: act = 0).
//   list order (sparse tiles): the row's list rotated to start at (row + column) mod 64 = 0, step it = its it-th edge;
//   slot order (tiles at least 3/4 full, WG-uniform choice): step it = the column (it - row) mod 64 of the tile, whether the row has
//   it or not - the two rows of a pair are then in the SAME step whatever the gaps in their lists (with list order a missing
//   column shifts the rest of the row by a step, and the L2 holds about two steps of the XCD's traffic: the partner's request for
//   the pair's filter rows missed, 1.8 x the distinct bytes).  The slot table lives in LDS: pair, edge number, sign of every tile
//   column of every row, so nothing of a step's requests depends on another request.
struct G16Idx {  // the edge's record (k_et_edge_records): one 32-byte request instead of eight 4-byte ones
  int j, p;
  float sg, C, dC, h0, h1, h2;
  int act;
};
constexpr int G16_NONE = -1;  // slot table: no such neighbour; else pair (24 bits) | edge number in the row << 24 | (sign < 0) << 30
struct G16Walk {
  const float* erec;  // [E][8] col | pair | sign | C | dC | prhat
  int e0, e1, len, rot, rl;
  const int* tbl;  // LDS slot table (slot order), null: list order
};
typedef float f4v_ __attribute__((ext_vector_type(4)));
__device__ __forceinline__ G16Idx g16_rec(const G16Walk& w, int e, int act) {
  const f4v_* rp = reinterpret_cast<const f4v_*>(reinterpret_cast<const char*>(w.erec) + (size_t)((uint32_t)e * 32u));
  const f4v_ r0 = rp[0], r1 = rp[1];
  return G16Idx{__float_as_int(r0.x), __float_as_int(r0.y), r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, act};
}
__device__ __forceinline__ int g16_slot(const G16Walk& w, int it) {  // slot-table entry of step it
  return it < G16_ROWS ? w.tbl[w.rl * G16_ROWS + ((it - w.rl) & (G16_ROWS - 1))] : G16_NONE;
}
__device__ __forceinline__ G16Idx g16_idx(const Graph& g, int it, const G16Walk& w) {
  int e, act;
  if (w.tbl) {
    const int t = g16_slot(w, it);
    act = t != G16_NONE;
    e = act ? w.e0 + ((t >> 24) & 63) : (w.len ? w.e0 : 0);
  } else {
    e = it + w.rot;
    if (e >= w.e1) e -= w.len;
    act = it < w.len;
    if (!act) e = w.len ? w.e0 : 0;
  }
  return g16_rec(w, e, act);
}
// the walk of a lane group's row; fills the slot table when the tile is dense.  Every thread of the block calls (barriers inside).
__device__ __forceinline__ G16Walk g16_walk(const Graph& g, const float* erec, int e0, int len, int rr, int rl, int gl, int tile0,
                                            int slot_min, int* s_max, int* s_tbl, int& steps) {
  G16Walk w;
  w.erec = erec;
  w.e0 = e0;
  w.len = len;
  w.e1 = e0 + len;
  w.rl = rl;
  const int longest = wg_max(len, s_max);
  const bool slot = slot_min > 0 && longest >= slot_min;  // WG-uniform
  if (slot) {
    for (int k = threadIdx.x; k < G16_ROWS * G16_ROWS; k += blockDim.x) s_tbl[k] = G16_NONE;
    __syncthreads();
    for (int k = gl; k < len; k += 16)
      s_tbl[rl * G16_ROWS + (g.col[e0 + k] - tile0)] = g.epair[e0 + k] | (k << 24) | (g.esign[e0 + k] < 0.f ? 1 << 30 : 0);
    __syncthreads();
  }
  w.tbl = slot ? s_tbl : nullptr;
  w.rot = slot ? e0 : rot_start(g.col, e0, e0 + len, rr);
  steps = slot ? G16_ROWS : longest;
  return w;
}
// ================================================================================================= forward
// xagg[t,c] = sum_e sx * A_h ; vagg[t,a,c] = sum_e vec[s,a,c] * s1 + s2 * rhat_a
struct FwdPair {  // what arrives from memory for one edge: the pair's filter rows and scalars (node rows come from LDS)
  f2v dk, dvx, dv1, dv2;
  float C, h0, h1, h2, sg;
  int j, act;
};

template <bool HAS_DK, bool HAS_DV, bool VCUT, int HL, bool BF>
__global__ __launch_bounds__(1024, BF ? 8 : 4) void k_et_attn_fwd_g16(Graph g, EtAttnArgs a, int N, int sync_every, int slot_min, float* __restrict__ xagg,
                                                          float* __restrict__ vagg) {
  __shared__ __attribute__((aligned(16))) f2v sN[7 * G16_V];  // k | vx | v1 | v2 | vec0 | vec1 | vec2 : 57 344 B
  __shared__ int s_max;
  __shared__ __attribute__((aligned(16))) int s_tbl[G16_ROWS * G16_ROWS];  // slot table, 16 KB
  // mailbox of the slot-order walk (see k_et_attn_bwd_g16): the pair's i requests dk | dvx, its j dv1 | dv2; two buffers taking turns.
  // fp32 rows only: with bf16 rows the exchange measured slower than fetching twice (7.95 vs 8.19 ms per step), and without the
  // mailbox two workgroups fit a CU (73 KB of LDS, 64 registers)
  typedef typename std::conditional<BF, uint32_t, f2v>::type HalfT;
  constexpr bool MB = !BF;
  __shared__ __attribute__((aligned(16))) HalfT s_mb[MB ? 2 * 4 * G16_V : 4];
  if (g.counts[2] || a.tile_open[0]) return;
  const int F = a.F, nsl = F >> 5, hl = a.hd >> 1;
  const int tile = (int)blockIdx.x / nsl, sl = (int)blockIdx.x - tile * nsl;
  if (tile >= a.tile_open[1]) return;  // (the grid covers the largest possible number of tiles)
  const int tile0 = a.tile_start[tile], nrows = a.tile_start[tile + 1] - tile0;  // whole molecules, at most 64 rows
  const int gl = threadIdx.x & 15, rl = (int)threadIdx.x >> 4;
  const int r = tile0 + rl;
  const bool rowok = rl < nrows && r < N;
  const int rr = rowok ? r : N - 1;
  const int c = sl * 32 + gl * 2;
  const int e0 = g.rowptr[rr], len = rowok ? g.rowptr[rr + 1] - e0 : 0, e1 = e0 + len;
  const f2v qt = *reinterpret_cast<const f2v*>(a.qkv + (int64_t)rr * 5 * F + c);
  {  // own row -> LDS (rows past the end: zeros)
    const float* rq = a.qkv + (int64_t)rr * 5 * F + c;
    const float* rv = a.vec + (int64_t)rr * 3 * F + c;
    f2v* mine = sN + rl * 16 + gl;
    const f2v z = splat(0.f);
    mine[0 * G16_V] = rowok ? *reinterpret_cast<const f2v*>(rq + F) : z;
    mine[1 * G16_V] = rowok ? *reinterpret_cast<const f2v*>(rq + 2 * F) : z;
    mine[2 * G16_V] = rowok ? *reinterpret_cast<const f2v*>(rq + 3 * F) : z;
    mine[3 * G16_V] = rowok ? *reinterpret_cast<const f2v*>(rq + 4 * F) : z;
    mine[4 * G16_V] = rowok ? *reinterpret_cast<const f2v*>(rv) : z;
    mine[5 * G16_V] = rowok ? *reinterpret_cast<const f2v*>(rv + F) : z;
    mine[6 * G16_V] = rowok ? *reinterpret_cast<const f2v*>(rv + 2 * F) : z;
  }
  int steps;
  const G16Walk w = g16_walk(g, a.erec, e0, len, rr, rl, gl, tile0, slot_min, &s_max, s_tbl, steps);  // (its barriers publish the tile)
  const int esz = BF ? 2 : 4;
  const OffT rowb = (OffT)a.Wd * esz;
  const int dko = a.dk_off < 0 ? 0 : a.dk_off, dvo = a.dv_off < 0 ? 0 : a.dv_off;
  const char* const dkvb = reinterpret_cast<const char*>(a.dkv);
  const float* const pdk = reinterpret_cast<const float*>(dkvb + (int64_t)dko * esz);
  const float* const pdx = reinterpret_cast<const float*>(dkvb + (int64_t)dvo * esz);
  const float* const pd1 = reinterpret_cast<const float*>(dkvb + (int64_t)(dvo + F) * esz);
  const float* const pd2 = reinterpret_cast<const float*>(dkvb + (int64_t)(dvo + 2 * F) * esz);

  auto rows = [&](const G16Idx& ix, FwdPair& u) __attribute__((always_inline)) {
    const OffT pr = (OffT)ix.p * rowb + (OffT)(c * esz);
    u.dk = HAS_DK ? ldp2<BF>(pdk, pr) : splat(1.f);
    u.dvx = HAS_DV ? ldp2<BF>(pdx, pr) : splat(1.f);
    u.dv1 = HAS_DV ? ldp2<BF>(pd1, pr) : splat(1.f);
    u.dv2 = HAS_DV ? ldp2<BF>(pd2, pr) : splat(1.f);
    u.C = ix.C;
    u.h0 = ix.h0;
    u.h1 = ix.h1;
    u.h2 = ix.h2;
    u.sg = ix.sg;
    u.j = ix.j;
    u.act = ix.act;
  };
  f2v xa = splat(0.f), va0 = splat(0.f), va1 = splat(0.f), va2 = splat(0.f);
  auto add = [&](const FwdPair& u) __attribute__((always_inline)) {
    const f2v* nb = sN + (u.j - tile0) * 16 + gl;
    const float cv = VCUT ? u.C : 1.0f, ca = VCUT ? 1.0f : u.C;
    const float m = u.sg != 0.f ? -u.sg : 0.f;  // rhat(t <- s) = -esign * prhat; self edge: 0
    const float ak = head_sum2<HL>(hadd(qt * nb[0] * u.dk), hl);
    const float A = ak * sigm(ak) * ca;
    const f2v sx = nb[1 * G16_V] * cv * u.dvx, s1 = nb[2 * G16_V] * cv * u.dv1, s2 = nb[3 * G16_V] * cv * u.dv2;
    xa += sx * A;
    va0 += nb[4 * G16_V] * s1 + s2 * (m * u.h0);
    va1 += nb[5 * G16_V] * s1 + s2 * (m * u.h1);
    va2 += nb[6 * G16_V] * s1 + s2 * (m * u.h2);
  };

  if (MB && w.tbl && a.mailbox) {
    // ---- slot order with the exchange: a step's requests depend on the LDS slot table only, four steps ahead of their use (the
    // sweep has little arithmetic per step: what hides the latency is the distance, not the other waves)
    struct Step { HalfT h[2]; float C; f4v_ r1; int t; };
    auto ldh = [&](const float* ubase, OffT off) __attribute__((always_inline)) -> HalfT {
      if constexpr (BF) return *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(ubase) + (size_t)off);
      else return *reinterpret_cast<const f2v*>(reinterpret_cast<const char*>(ubase) + (size_t)off);
    };
    auto widen = [&](HalfT x) __attribute__((always_inline)) -> f2v {
      if constexpr (BF) return f2v{__uint_as_float(x << 16), __uint_as_float(x & 0xffff0000u)};
      else return x;
    };
    {  // the self edge first, on its own (it needs all four rows: no branch around a request inside the loop), then out of the table
      const int t = s_tbl[rl * G16_ROWS + rl];
      if (t != G16_NONE) {
        const G16Idx ix = g16_rec(w, e0 + ((t >> 24) & 63), 1);
        FwdPair u;
        rows(ix, u);
        add(u);
        s_tbl[rl * G16_ROWS + rl] = G16_NONE;  // (read by this lane group only)
      }
    }
    // the pair's i requests dk | dvx, its j dv1 | dv2 = the same two requests 2 F elements further (the launcher takes this path
    // only when both projections exist: dk at 0, dvx | dv1 | dv2 at F, 2 F, 3 F of the row)
    const OffT j_delta = (OffT)(2 * F * esz);
    auto request = [&](int it, Step& q) __attribute__((always_inline)) {
      const int t = g16_slot(w, it);
      q.t = t;
      const bool act = t != G16_NONE;
      const int e = act ? e0 + ((t >> 24) & 63) : (len ? e0 : 0);
      const int p = act ? (t & 0xFFFFFF) : 0;
      const char* rp = reinterpret_cast<const char*>(a.erec) + (size_t)((uint32_t)e * 32u);
      q.C = *reinterpret_cast<const float*>(rp + 12);
      q.r1 = *reinterpret_cast<const f4v_*>(rp + 16);
      const OffT pr = (OffT)p * rowb + (OffT)(c * esz) + ((t >> 30) & 1 ? j_delta : (OffT)0);
      q.h[0] = ldh(pdk, pr);
      q.h[1] = ldh(pdx, pr);
    };
    auto finish = [&](int it, const Step& q) __attribute__((always_inline)) {
      const bool act = q.t != G16_NONE;
      const int cl = (it - rl) & (G16_ROWS - 1);
      const bool neg = (q.t >> 30) & 1;
      HalfT* mb = s_mb + (it & 1) * 4 * G16_V;
      if (act) {
        const int o = (neg ? 2 * G16_V : 0) + rl * 16 + gl;
        mb[o] = q.h[0];
        mb[G16_V + o] = q.h[1];
      }
      __syncthreads();
      FwdPair u;
      const int oi = (neg ? cl : rl) * 16 + gl, oj = (neg ? rl : cl) * 16 + gl;
      u.dk = widen(mb[oi]);
      u.dvx = widen(mb[1 * G16_V + oi]);
      u.dv1 = widen(mb[2 * G16_V + oj]);
      u.dv2 = widen(mb[3 * G16_V + oj]);
      if (act) {
        u.C = q.C; u.h0 = q.r1.y; u.h1 = q.r1.z; u.h2 = q.r1.w;
        u.sg = neg ? -1.f : 1.f;
        u.j = tile0 + cl; u.act = 1;
        add(u);
      }
    };
    Step qA, qB, qC, qD;
    request(0, qA);
    request(1, qB);
    request(2, qC);
    request(3, qD);
    for (int it = 0; it < steps; it += 4) {  // (steps = 64 in slot order)
      finish(it, qA);
      request(it + 4, qA);
      finish(it + 1, qB);
      request(it + 5, qB);
      finish(it + 2, qC);
      request(it + 6, qC);
      finish(it + 3, qD);
      request(it + 7, qD);
    }
  } else {
  // list order: two steps per trip, the edges' rows in two register sets that take turns (no copies): rows one step ahead, indices two
  FwdPair uA, uB;
  G16Idx i1 = {0, 0, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0};
  if (steps > 0) {
    rows(g16_idx(g, 0, w), uA);
    i1 = g16_idx(g, 1, w);
  }
  int until_sync = sync_every;
  for (int it = 0; it < steps; it += 2) {
    const G16Idx i2 = g16_idx(g, it + 2, w);
    rows(i1, uB);
    if (uA.act) add(uA);
    const G16Idx i3 = g16_idx(g, it + 3, w);
    rows(i2, uA);
    if (uB.act) add(uB);
    i1 = i3;
    if (sync_every > 0 && (until_sync -= 2) <= 0) {
      until_sync = sync_every;
      __syncthreads();
    }
  }
  }
  if (rowok) {
    *reinterpret_cast<f2v*>(xagg + (int64_t)r * F + c) = xa;
    float* o = vagg + (int64_t)r * 3 * F + c;
    *reinterpret_cast<f2v*>(o) = va0;
    *reinterpret_cast<f2v*>(o + F) = va1;
    *reinterpret_cast<f2v*>(o + 2 * F) = va2;
  }
}

// ================================================================================================= reverse
struct BwdPair {
  f2v dk, tk, dvx, dv1, dv2, tvx, tv1, tv2;
  float C, dC, h0, h1, h2, sg;
  int j, p, act;
};

template <bool HAS_DK, bool HAS_DV, bool VCUT, int HL, bool BF>
__global__ __launch_bounds__(1024) void k_et_attn_bwd_g16(Graph g, EtAttnArgs a, int N, int sync_every, int slot_min, const float* __restrict__ g_xagg,
                                                          const float* __restrict__ g_vagg, float* __restrict__ g_qkv,
                                                          float* __restrict__ g_vec, float* __restrict__ gd2, float* __restrict__ gr2) {
  // q | k | vx | v1 | v2 | vec0..2 | g_xagg | g_vagg0..2 of the tile's 64 atoms: 98 304 B
  __shared__ __attribute__((aligned(16))) f2v sN[12 * G16_V];
  __shared__ int s_max;
  __shared__ __attribute__((aligned(16))) int s_tbl[G16_ROWS * G16_ROWS];  // slot table, 16 KB
  // the mailbox of the slot-order walk: the two rows of a pair are in the same step, each requests HALF of the pair's filter rows
  // (the pair's i: dkv, its j: the tangents tkv) and leaves it here for the other: every row is fetched once per tile.  
  // one buffer (32 KB: what the 160 KB leave) and two barriers per step.
  typedef typename std::conditional<BF, uint32_t, f2v>::type HalfT;
  constexpr int NBUF = 1;
  constexpr bool MB = !BF;  // (bf16 rows: the exchange measured slower than fetching twice)
  __shared__ __attribute__((aligned(16))) HalfT s_mb[MB ? NBUF * 4 * G16_V : 4];
  // the (distance, direction) sums of the last G16_RING steps wait here and leave in one burst: a store between two requests makes
  // the wait for the older request a wait for everything (loads and stores share one counter and retire out of order)
  __shared__ float s_ring[MB ? G16_RING * G16_ROWS * 4 : 4];
  if (g.counts[2] || a.tile_open[0]) return;
  const int F = a.F, nsl = F >> 5, hl = a.hd >> 1;
  const int tile = (int)blockIdx.x / nsl, sl = (int)blockIdx.x - tile * nsl;
  if (tile >= a.tile_open[1]) return;  // (the grid covers the largest possible number of tiles)
  const int tile0 = a.tile_start[tile], nrows = a.tile_start[tile + 1] - tile0;  // whole molecules, at most 64 rows
  const int gl = threadIdx.x & 15, rl = (int)threadIdx.x >> 4;
  const int r = tile0 + rl;
  const bool rowok = rl < nrows && r < N;
  const int rr = rowok ? r : N - 1;
  const int c = sl * 32 + gl * 2;
  const int e0 = g.rowptr[rr], len = rowok ? g.rowptr[rr + 1] - e0 : 0, e1 = e0 + len;
  {  // own row -> LDS (rows past the end: zeros)
    const float* rq = a.qkv + (int64_t)rr * 5 * F + c;
    const float* rv = a.vec + (int64_t)rr * 3 * F + c;
    const float* gv = g_vagg + (int64_t)rr * 3 * F + c;
    f2v* mine = sN + rl * 16 + gl;
    const f2v z = splat(0.f);
#pragma unroll
    for (int k = 0; k < 5; ++k) mine[k * G16_V] = rowok ? *reinterpret_cast<const f2v*>(rq + k * F) : z;
#pragma unroll
    for (int k = 0; k < 3; ++k) mine[(5 + k) * G16_V] = rowok ? *reinterpret_cast<const f2v*>(rv + k * F) : z;
    mine[8 * G16_V] = rowok ? *reinterpret_cast<const f2v*>(g_xagg + (int64_t)rr * F + c) : z;
#pragma unroll
    for (int k = 0; k < 3; ++k) mine[(9 + k) * G16_V] = rowok ? *reinterpret_cast<const f2v*>(gv + k * F) : z;
  }
  int steps;
  const G16Walk w = g16_walk(g, a.erec, e0, len, rr, rl, gl, tile0, slot_min, &s_max, s_tbl, steps);  // (its barriers publish the tile)
  const bool head0 = (gl & (hl - 1)) == 0;  // one lane per head (hl is a power of two)
  const int esz = BF ? 2 : 4;
  const OffT rowb = (OffT)a.Wd * esz;
  const int dko = a.dk_off < 0 ? 0 : a.dk_off, dvo = a.dv_off < 0 ? 0 : a.dv_off;
  const char* const dkvb = reinterpret_cast<const char*>(a.dkv);
  const char* const tkvb = reinterpret_cast<const char*>(a.tkv);
  const float* const pdk = reinterpret_cast<const float*>(dkvb + (int64_t)dko * esz);
  const float* const pdx = reinterpret_cast<const float*>(dkvb + (int64_t)dvo * esz);
  const float* const pd1 = reinterpret_cast<const float*>(dkvb + (int64_t)(dvo + F) * esz);
  const float* const pd2 = reinterpret_cast<const float*>(dkvb + (int64_t)(dvo + 2 * F) * esz);
  const float* const ptk = reinterpret_cast<const float*>(tkvb + (int64_t)dko * esz);
  const float* const ptx = reinterpret_cast<const float*>(tkvb + (int64_t)dvo * esz);
  const float* const pt1 = reinterpret_cast<const float*>(tkvb + (int64_t)(dvo + F) * esz);
  const float* const pt2 = reinterpret_cast<const float*>(tkvb + (int64_t)(dvo + 2 * F) * esz);

  auto rows = [&](const G16Idx& ix, BwdPair& u) __attribute__((always_inline)) {
    const OffT pr = (OffT)ix.p * rowb + (OffT)(c * esz);
    u.dk = HAS_DK ? ldp2<BF>(pdk, pr) : splat(1.f);
    u.tk = HAS_DK ? ldp2<BF>(ptk, pr) : splat(0.f);
    u.dvx = HAS_DV ? ldp2<BF>(pdx, pr) : splat(1.f);
    u.dv1 = HAS_DV ? ldp2<BF>(pd1, pr) : splat(1.f);
    u.dv2 = HAS_DV ? ldp2<BF>(pd2, pr) : splat(1.f);
    u.tvx = HAS_DV ? ldp2<BF>(ptx, pr) : splat(0.f);
    u.tv1 = HAS_DV ? ldp2<BF>(pt1, pr) : splat(0.f);
    u.tv2 = HAS_DV ? ldp2<BF>(pt2, pr) : splat(0.f);
    u.C = ix.C;
    u.dC = ix.dC;
    u.h0 = ix.h0;
    u.h1 = ix.h1;
    u.h2 = ix.h2;
    u.sg = ix.sg;
    u.j = ix.j;
    u.p = ix.p;
    u.act = ix.act;
  };

  f2v gq = splat(0.f), gk = splat(0.f), gvx = splat(0.f), gv1 = splat(0.f), gv2 = splat(0.f);
  f2v gvec0 = splat(0.f), gvec1 = splat(0.f), gvec2 = splat(0.f);
  int own_off = rl * 16 + gl;
  int ring_step = -1;  // >= 0: the step whose sums go to the ring (slot order), -1: stored at once
  auto add = [&](const BwdPair& u) __attribute__((always_inline)) {
    asm volatile("" : "+v"(own_off));  // the row's own values are re-read from LDS every step (22 registers otherwise)
    const f2v* own = sN + own_off;
    const f2v* nb = sN + (u.j - tile0) * 16 + gl;
    const float cv = VCUT ? u.C : 1.0f, ca = VCUT ? 1.0f : u.C;
    // prhat with the edge's sign: rhat(j <- r); rhat(r <- j) is its negative
    const float m = u.sg != 0.f ? u.sg : 0.f;
    const float p0 = m * u.h0, p1 = m * u.h1, p2 = m * u.h2;
    // ---- role TARGET: message j -> r
    {
      const f2v qr = own[0], gxr = own[8 * G16_V], gr0 = own[9 * G16_V], gr1 = own[10 * G16_V], gr2_ = own[11 * G16_V];
      const f2v kj = nb[1 * G16_V], vxj = nb[2 * G16_V], v1j = nb[3 * G16_V], v2j = nb[4 * G16_V];
      const float at = head_sum2<HL>(hadd(qr * kj * u.dk), hl);
      const float sg_ = sigm(at), sil = at * sg_, sgr = sg_ * (1.0f + at * (1.0f - sg_));
      const float A = sil * ca;
      const f2v sx = vxj * cv * u.dvx, s2 = v2j * cv * u.dv2;
      const f2v g_sx = gxr * A;
      const float g_A = head_sum2<HL>(hadd(gxr * sx), hl);
      const f2v g_s1 = gr0 * nb[5 * G16_V] + gr1 * nb[6 * G16_V] + gr2_ * nb[7 * G16_V];
      const f2v g_s2 = -(gr0 * p0 + gr1 * p1 + gr2_ * p2);
      const float g_a = g_A * sgr * ca;
      gq += g_a * kj * u.dk;
      const f2v gd_v = cv * (g_sx * vxj * u.tvx + g_s1 * v1j * u.tv1 + g_s2 * v2j * u.tv2) + g_a * qr * kj * u.tk;
      float gd = hadd(gd_v);
      if (VCUT) gd += hadd(g_sx * vxj * u.dvx + g_s1 * v1j * u.dv1 + g_s2 * v2j * u.dv2) * u.dC;
      else gd += (head0 ? g_A * sil : 0.f) * u.dC;
      const float tot = grp_sum4(gd, hadd(gr0 * s2), hadd(gr1 * s2), hadd(gr2_ * s2), gl);
      if (ring_step >= 0) {
        if ((gl & 3) == 0) s_ring[((ring_step & (G16_RING - 1)) * G16_ROWS + rl) * 4 + (gl >> 2)] = tot;
      } else if (u.sg != 0.f && (gl & 3) == 0) {
        const int64_t slot = (int64_t)sl * a.slot_stride + 2 * (int64_t)u.p + (u.sg > 0.f ? 0 : 1);
        const int comp = gl >> 2;
        if (comp == 0) gd2[slot] = tot;
        else gr2[slot * 3 + comp - 1] = tot;
      }
    }
    // ---- role SOURCE: message r -> j
    {
      const f2v kr = own[1 * G16_V], vxr = own[2 * G16_V], v1r = own[3 * G16_V];
      const f2v qj = nb[0], gxj = nb[8 * G16_V], gj0 = nb[9 * G16_V], gj1 = nb[10 * G16_V], gj2 = nb[11 * G16_V];
      const float as = head_sum2<HL>(hadd(qj * kr * u.dk), hl);
      const float sg_ = sigm(as), sil = as * sg_, sgr = sg_ * (1.0f + as * (1.0f - sg_));
      const float A = sil * ca;
      const f2v sx = vxr * cv * u.dvx, s1 = v1r * cv * u.dv1;
      const float g_A = head_sum2<HL>(hadd(gxj * sx), hl);
      const float g_a = g_A * sgr * ca;
      gk += g_a * qj * u.dk;
      gvx += gxj * A * cv * u.dvx;
      gv1 += (gj0 * own[5 * G16_V] + gj1 * own[6 * G16_V] + gj2 * own[7 * G16_V]) * cv * u.dv1;
      gv2 += (gj0 * p0 + gj1 * p1 + gj2 * p2) * cv * u.dv2;
      gvec0 += gj0 * s1;
      gvec1 += gj1 * s1;
      gvec2 += gj2 * s1;
    }
  };
  if (MB && w.tbl && a.mailbox) {
    // ---- slot order with the exchange.  The half a lane group requests: the rows of dkv (sign >= 0, the self edge too) or of tkv
    // (sign < 0), one offset register for both (tkv = dkv + a constant number of bytes, checked by the launcher); it lands in
    // the group's mailbox row, and both halves are read back from the mailbox rows of the pair's i (dkv) and j (tkv) - no
    // per-lane selects.  A self edge reads its own dkv rows in place of the tangents: they only feed slots that are not written.
    const OffT tk_delta = (OffT)(tkvb - dkvb);
    struct Half { HalfT v[4]; };
    struct Step { f4v_ r0, r1; Half h; int t; };  // one step's requests: its record, its half rows, its slot-table entry
    auto ldh = [&](const float* ubase, OffT off) __attribute__((always_inline)) -> HalfT {
      if constexpr (BF) return *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(ubase) + (size_t)off);
      else return *reinterpret_cast<const f2v*>(reinterpret_cast<const char*>(ubase) + (size_t)off);
    };
    auto widen = [&](HalfT x) __attribute__((always_inline)) -> f2v {
      if constexpr (BF) return f2v{__uint_as_float(x << 16), __uint_as_float(x & 0xffff0000u)};
      else return x;
    };
    auto request = [&](int it, Step& q) __attribute__((always_inline)) {
      const int t = g16_slot(w, it);
      q.t = t;
      const bool act = t != G16_NONE;
      const int e = act ? e0 + ((t >> 24) & 63) : (len ? e0 : 0);
      const int p = act ? (t & 0xFFFFFF) : 0;
      const f4v_* rp = reinterpret_cast<const f4v_*>(reinterpret_cast<const char*>(a.erec) + (size_t)((uint32_t)e * 32u));
      q.r0 = rp[0];
      q.r1 = rp[1];
      const OffT pr = (OffT)p * rowb + (OffT)(c * esz) + ((t >> 30) & 1 ? tk_delta : (OffT)0);
      if (HAS_DK) q.h.v[0] = ldh(pdk, pr);
      if (HAS_DV) {
        q.h.v[1] = ldh(pdx, pr);
        q.h.v[2] = ldh(pd1, pr);
        q.h.v[3] = ldh(pd2, pr);
      }
    };
    int buf = 0;
    auto finish = [&](int it, const Step& q) __attribute__((always_inline)) {
      const bool act = q.t != G16_NONE;
      const int cl = (it - rl) & (G16_ROWS - 1);  // the partner's row of the tile
      const bool neg = (q.t >> 30) & 1;
      HalfT* mb = s_mb + buf * 4 * G16_V;
      if (act) {  // (a self edge too: its own row is the only one that reads it back)
#pragma unroll
        for (int k = HAS_DK ? 0 : 1; k < (HAS_DV ? 4 : 1); ++k) mb[k * G16_V + rl * 16 + gl] = q.h.v[k];
      }
      __syncthreads();
      BwdPair u;
      {
        const int od = ((neg ? cl : rl) * 16 + gl), ot = ((neg ? rl : cl) * 16 + gl);
        const f2v one = splat(1.f), zero = splat(0.f);
        u.dk = HAS_DK ? widen(mb[od]) : one;
        u.tk = HAS_DK ? widen(mb[ot]) : zero;
        u.dvx = HAS_DV ? widen(mb[1 * G16_V + od]) : one;
        u.dv1 = HAS_DV ? widen(mb[2 * G16_V + od]) : one;
        u.dv2 = HAS_DV ? widen(mb[3 * G16_V + od]) : one;
        u.tvx = HAS_DV ? widen(mb[1 * G16_V + ot]) : zero;
        u.tv1 = HAS_DV ? widen(mb[2 * G16_V + ot]) : zero;
        u.tv2 = HAS_DV ? widen(mb[3 * G16_V + ot]) : zero;
      }
      if (NBUF == 1) __syncthreads();
      else buf ^= 1;
      if (act) {
        u.C = q.r0.w; u.dC = q.r1.x; u.h0 = q.r1.y; u.h1 = q.r1.z; u.h2 = q.r1.w;
        u.sg = cl == rl ? 0.f : (neg ? -1.f : 1.f);
        u.j = tile0 + cl; u.p = q.t & 0xFFFFFF; u.act = 1;
        ring_step = it;
        add(u);
      }
    };
    auto flush = [&](int last) __attribute__((always_inline)) {  // the ring's steps last - 7 .. last (each lane reads back what it wrote)
      if ((gl & 3) == 0) {
        const int comp = gl >> 2;
#pragma unroll
        for (int k = 0; k < G16_RING; ++k) {
          const int st = last - (G16_RING - 1) + k;
          const int t = g16_slot(w, st);
          if (t != G16_NONE && ((st - rl) & (G16_ROWS - 1)) != rl) {
            const float v = s_ring[((st & (G16_RING - 1)) * G16_ROWS + rl) * 4 + comp];
            const int64_t slot = (int64_t)sl * a.slot_stride + 2 * (int64_t)(t & 0xFFFFFF) + ((t >> 30) & 1);
            if (comp == 0) gd2[slot] = v;
            else gr2[slot * 3 + comp - 1] = v;
          }
        }
      }
    };
    // two steps per trip, two request sets taking turns (no copies): everything of step it + 1 is requested before step it's
    // exchange and arithmetic
    Step qA, qB;
    request(0, qA);
    for (int it = 0; it < steps; it += 2) {
      request(it + 1, qB);
      finish(it, qA);
      request(it + 2, qA);
      finish(it + 1, qB);
      if (((it + 2) & (G16_RING - 1)) == 0) flush(it + 1);
    }
  } else {
  BwdPair uA, uB;
  G16Idx i1 = {0, 0, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0};
  if (steps > 0) {
    rows(g16_idx(g, 0, w), uA);
    i1 = g16_idx(g, 1, w);
  }
  int until_sync = sync_every;
  for (int it = 0; it < steps; it += 2) {
    const G16Idx i2 = g16_idx(g, it + 2, w);
    rows(i1, uB);
    if (uA.act) add(uA);
    const G16Idx i3 = g16_idx(g, it + 3, w);
    rows(i2, uA);
    if (uB.act) add(uB);
    i1 = i3;
    if (sync_every > 0 && (until_sync -= 2) <= 0) {
      until_sync = sync_every;
      __syncthreads();
    }
  }
  }
  if (rowok) {
    float* o = g_qkv + (int64_t)r * 5 * F + c;
    *reinterpret_cast<f2v*>(o) = gq;
    *reinterpret_cast<f2v*>(o + F) = gk;
    *reinterpret_cast<f2v*>(o + 2 * F) = gvx;
    *reinterpret_cast<f2v*>(o + 3 * F) = gv1;
    *reinterpret_cast<f2v*>(o + 4 * F) = gv2;
    float* gv = g_vec + (int64_t)r * 3 * F + c;
    *reinterpret_cast<f2v*>(gv) += gvec0;
    *reinterpret_cast<f2v*>(gv + F) += gvec1;
    *reinterpret_cast<f2v*>(gv + 2 * F) += gvec2;
  }
}

// ---- tiles = runs of WHOLE molecules of at most 64 rows (round 6, second step: a batch of molecules of any sizes <= 64 has closed
// tiles, not only 64-atom molecules aligned to 64 rows).  One block: the rows at which a tile may start (first atom of a
// molecule; row N) as a bitmap in LDS, then one wave packs greedily - tile after tile, the last allowed start within 64 rows.
// meta[0] = 1 (open: the row sweeps run) when a molecule has more than 64 atoms, meta[1] = number of tiles, tile_start[0 .. n].
constexpr int G16_BITMAP_WORDS = 8192;  // rows up to 64 * 8192 - 64
__global__ __launch_bounds__(1024) void k_et_tile_pack(Graph g, const int64_t* __restrict__ batch, int N, int max_tiles, float min_fill,
                                                      int* __restrict__ tile_start, int* __restrict__ meta) {
  __shared__ unsigned long long s_b[G16_BITMAP_WORDS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (g.counts[2] || g.counts[3] || N > 64 * G16_BITMAP_WORDS - 128) {  // (overflow / unsorted batch: the step fails anyway)
    if (tid == 0) {
      meta[0] = 1;
      meta[1] = 0;
    }
    return;
  }
  const int nw = (N >> 6) + 2;
  for (int wd = wave; wd < nw; wd += 16) {
    const int r = wd * 64 + lane;
    const bool ok = r == N || r == 0 || (r < N && batch && batch[r] != batch[r - 1]);  // (no batch vector: one molecule)
    const unsigned long long bits = __ballot(ok);
    if (lane == 0) s_b[wd] = bits;
  }
  __syncthreads();
  if (wave != 0) return;
  int s0 = 0, n = 0;
  bool open = false;
  while (s0 < N) {
    const int lo = s0 + 1, wd = lo >> 6, sh = lo & 63;  // candidates lo .. lo + 63: bit k of the window = row lo + k
    const unsigned long long win = (s_b[wd] >> sh) | (sh ? (s_b[wd + 1] << (64 - sh)) : 0ull);
    int nxt;
    if (win) {
      nxt = lo + 63 - __clzll(win);
    } else {  // a molecule of more than 64 atoms: not closed; keep a valid partition
      nxt = s0 + G16_ROWS < N ? s0 + G16_ROWS : N;
      open = true;
    }
    if (lane == 0) tile_start[n] = s0;
    ++n;
    s0 = nxt;
  }
  if (lane == 0) {
    tile_start[n] = N;
    meta[1] = n;
    if (open || n > max_tiles) meta[0] = 1;  // (more tiles than the sweeps' grids cover cannot happen with closed tiles: n <= molecules)
    // sparse tiles (small molecules, short cutoffs): the row sweeps are the faster generation - a ragged batch of 10 .. 64-atom
    // molecules at half-full tiles measured 8.87 (rows) against 9.44 ms (tiles) with fp32 pair rows, 8.18 against 7.36 with bf16
    if ((float)g.counts[1] < min_fill * (float)n * (float)(G16_ROWS * G16_ROWS)) meta[0] = 1;
  }
}
// meta[0] |= 1 when a row has a neighbour outside its tile (one wave per tile; after k_et_tile_pack)
__global__ __launch_bounds__(256) void k_et_tile_open(Graph g, int N, const int* __restrict__ tile_start, int* __restrict__ meta) {
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (g.counts[2] || meta[0] || t >= meta[1]) return;
  const int t0 = tile_start[t], t1 = tile_start[t + 1];
  const int r = t0 + lane;
  bool bad = t1 - t0 > G16_ROWS;
  if (!bad && r < t1) {
    const int e0 = g.rowptr[r], len = g.rowptr[r + 1] - e0;
    if (len > 0) bad = g.col[e0] < t0 || g.col[e0 + len - 1] >= t1;  // (columns ascend within a row)
  }
  if (bad) atomicOr(meta, 1);
}
// the per-edge records of the tile sweeps: everything an edge needs besides rows, in one 32-byte piece
__global__ void k_et_edge_records(Graph g, const float* __restrict__ C, const float* __restrict__ dC, int64_t cap,
                                  float* __restrict__ erec) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g.counts[2] || e >= g.counts[1] || e >= cap) return;
  const int p = g.epair[e];
  const float sg = g.esign[e];
  typedef float f4v __attribute__((ext_vector_type(4)));
  f4v* o = reinterpret_cast<f4v*>(erec + e * 8);
  // self edge (sign 0): its geometry slot is not written by the graph phase - the sweeps multiply it by the sign, keep it finite
  o[0] = f4v{__int_as_float(g.col[e]), __int_as_float(p), sg, C[p]};
  o[1] = f4v{dC[p], sg != 0.f ? g.prhat[(int64_t)p * 3] : 0.f, sg != 0.f ? g.prhat[(int64_t)p * 3 + 1] : 0.f,
             sg != 0.f ? g.prhat[(int64_t)p * 3 + 2] : 0.f};
}

int g16_sync_every() {
  static const int v = getenv("TMDNET_ET_G16_SYNC") ? atoi(getenv("TMDNET_ET_G16_SYNC")) : 4;
  return v;
}
bool g16_mailbox() {  // developer switch: every lane group requests all rows of its pair (no exchange)
  static const bool off = getenv("TMDNET_ET_G16_NO_MAILBOX") != nullptr;
  return !off;
}
int g16_slot_min() {  // slot order from this row length on (0: list order always)
  static const int v = getenv("TMDNET_ET_G16_SLOT_MIN") ? atoi(getenv("TMDNET_ET_G16_SLOT_MIN")) : 48;
  return v;
}

}  // namespace g16
using namespace g16;

// the 16-lane-group sweeps apply: channel slices of 32, a head inside a lane group, and every array the kernels index with a
// 32-bit byte offset below 2 GiB (P1 = rows of the per-pair arrays)
bool et_g16_ok(int N, int64_t P1, const EtAttnArgs& a) {
  static const bool off = getenv("TMDNET_ET_NO_G16") != nullptr;  // developer switch: the one-channel-per-lane kernels
  const int64_t lim = (int64_t)1 << 31;
  const int hd = a.hd;
  return !off && a.tile_open && a.tile_start && a.max_tiles > 0 && a.erec && a.F % 32 == 0 && a.F <= 1024 && (hd == 2 || hd == 4 || hd == 8 || hd == 16 || hd == 32) &&
         (int64_t)N * 5 * a.F * 4 < lim && P1 * a.Wd * 4 < lim && (2 * P1 + N) * 32 < lim;
}

#define G16_LAUNCH(KERNEL, DK, DV, VC, HL, ...)                                                                    \
  if (a.pair_bf16) hipLaunchKernelGGL((KERNEL<DK, DV, VC, HL, true>), grid, block, 0, s, g, a, N, se, sm, __VA_ARGS__); \
  else hipLaunchKernelGGL((KERNEL<DK, DV, VC, HL, false>), grid, block, 0, s, g, a, N, se, sm, __VA_ARGS__);
#define G16_KEY(KERNEL, HL, ...)                                                  \
  switch (key) {                                                                  \
    case 0: G16_LAUNCH(KERNEL, false, false, false, HL, __VA_ARGS__) break;       \
    case 1: G16_LAUNCH(KERNEL, false, false, true, HL, __VA_ARGS__) break;        \
    case 2: G16_LAUNCH(KERNEL, false, true, false, HL, __VA_ARGS__) break;        \
    case 3: G16_LAUNCH(KERNEL, false, true, true, HL, __VA_ARGS__) break;         \
    case 4: G16_LAUNCH(KERNEL, true, false, false, HL, __VA_ARGS__) break;        \
    case 5: G16_LAUNCH(KERNEL, true, false, true, HL, __VA_ARGS__) break;         \
    case 6: G16_LAUNCH(KERNEL, true, true, false, HL, __VA_ARGS__) break;         \
    default: G16_LAUNCH(KERNEL, true, true, true, HL, __VA_ARGS__) break;         \
  }
#define G16_DISPATCH(KERNEL, ...)                                                                  \
  {                                                                                                \
    const dim3 grid(a.max_tiles * (a.F / 32)), block(1024);                                    \
    const int key = (a.dk_off >= 0 ? 4 : 0) | (a.dv_off >= 0 ? 2 : 0) | (a.vector_cutoff ? 1 : 0); \
    const int se = g16_sync_every(), sm = g16_slot_min();                                          \
    if (a.hd == 16) {                                                                              \
      G16_KEY(KERNEL, 8, __VA_ARGS__)                                                              \
    } else {                                                                                       \
      G16_KEY(KERNEL, 0, __VA_ARGS__)                                                              \
    }                                                                                              \
  }

int et_g16_max_tiles(int N, int B) {  // closed tiles hold whole molecules, and two consecutive tiles more than 64 rows
  const int by_rows = 2 * ((N + G16_ROWS - 1) / G16_ROWS) + 1;
  const int t = B < by_rows ? B : by_rows;
  return t < 1 ? 1 : t;
}
void launch_et_tile_prep(const Graph& g, int N, int B, const int64_t* batch, const float* C, const float* dC, int64_t ecap, int pair_bf16,
                         int* meta, int* tile_start, float* erec, hipStream_t s) {
  if (N <= 0) return;
  const int max_tiles = et_g16_max_tiles(N, B);
  // directed edges per tile slot (64 x 64) below which the step takes the row sweeps (developer switch, in percent)
  static const int fill_env = getenv("TMDNET_ET_G16_MIN_FILL") ? atoi(getenv("TMDNET_ET_G16_MIN_FILL")) : -1;
  const float min_fill = fill_env >= 0 ? 0.01f * (float)fill_env : (pair_bf16 ? 0.25f : 0.70f);
  launch_fill(reinterpret_cast<float*>(meta), 0.f, 2, s);
  hipLaunchKernelGGL(k_et_tile_pack, dim3(1), dim3(1024), 0, s, g, batch, N, max_tiles, min_fill, tile_start, meta);
  hipLaunchKernelGGL(k_et_tile_open, dim3((max_tiles + 3) / 4), dim3(256), 0, s, g, N, tile_start, meta);
  hipLaunchKernelGGL(k_et_edge_records, dim3((unsigned)((ecap + 255) / 256)), dim3(256), 0, s, g, C, dC, ecap, erec);
}
void launch_et_attn_fwd_g16(const Graph& g, int N, const EtAttnArgs& a_in, float* xagg, float* vagg, hipStream_t s) {
  if (N <= 0) return;
  EtAttnArgs a = a_in;
  a.mailbox = (g16_mailbox() && !a.pair_bf16 && a.dk_off == 0 && a.dv_off == a.F) ? 1 : 0;
  G16_DISPATCH(k_et_attn_fwd_g16, xagg, vagg)
}
void launch_et_attn_bwd_g16(const Graph& g, int N, const EtAttnArgs& a_in, const float* g_xagg, const float* g_vagg, float* g_qkv,
                            float* g_vec, float* gd2, float* gr2, hipStream_t s) {
  if (N <= 0) return;
  EtAttnArgs a = a_in;
  {  // the exchange addresses tkv as dkv + a 32-bit byte offset
    const int64_t delta = reinterpret_cast<const char*>(a.tkv) - reinterpret_cast<const char*>(a.dkv);
    const int64_t rows_b = (a.slot_stride / 2) * (int64_t)a.Wd * (a.pair_bf16 ? 2 : 4);
    a.mailbox = (g16_mailbox() && !a.pair_bf16 && delta >= 0 && delta + rows_b < ((int64_t)1 << 32)) ? 1 : 0;  // (bf16 rows: see the forward kernel)
  }
  G16_DISPATCH(k_et_attn_bwd_g16, g_xagg, g_vagg, g_qkv, g_vec, gd2, gr2)
}

}  // namespace tn
