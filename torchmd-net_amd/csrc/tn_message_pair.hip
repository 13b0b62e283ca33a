// LDS-staged message sweeps with 16 bytes per lane (reference tensornet.py:757-806 and its adjoint), gfx950.
//
//     acc[i, c, f] = sum_{e in row(i)} w[pair(e), type(c), f] * src[col(e), c, f]
//
// Same tiling as tn_message_tile.hip - a block owns 64 consecutive rows x 32 channels and stages the rows of its
// column window [<= 64][9][32] (72 KB) in LDS once - but a different thread layout.  There a half-wave owned a row
// with one channel per lane: every edge cost a wave 3 dword loads, 9 ds_read_b32 and 9 v_fma for 2 rows x 32 channels, and
// the sweep was bound by instruction issue (4 cycles per wave64 VALU / LDS instruction), not by HBM or LDS bandwidth
// (profiles/r02_notes.md: 113 us of the 200 us remained with the weight loads removed, against 29 us of LDS time).
// Here EIGHT lanes own a row and each lane carries FOUR channels: an edge costs a wave 3 global_load_dwordx4,
// 9 ds_read_b128 and 18 v_pk_fma_f32 for 8 rows x 32 channels - a quarter of the instructions per channel - and all 64
// rows of the tile advance together (8 waves), so the two reads of a pair's weights (by its row i and by its row j) fall
// within the few microseconds a tile takes instead of being spread over a long block lifetime.
//
// The tile's slice of the adjacency (col / epair of rows r0 .. r1, contiguous) sits in LDS next to the window: the (column,
// pair id) record of a trip is an LDS broadcast read, fetched one step ahead; weights are loaded two edges ahead.  The rows of
// a tile are balanced: the k-th longest row hands the tail of its list to the group of the k-th shortest (see "Balanced
// walk" in the kernel).  Deterministic, no atomics; the summation order of a long row differs from the other sweeps' by that
// one split.  A tile whose column window is wider than 64 rows (large systems in cell order, ragged molecules straddling
// the tile) gathers its sources from global memory instead of LDS, one whose slice exceeds 4096 entries walks it from global
// memory with the 8 lanes of a row fetching 8 records at a time; both choices are block-uniform.
//
// In-kernel timestamps (round 4, C2: 1024 blocks in 4 rounds): window + slice staging 5.7 us, edge loop 20.9 us (27.9 before
// the slice was staged and the rows balanced), epilogue 2.1 us per block; with the weight loads removed the loop still took
// 13 us, with the window reads removed as well the same - the loop moves ~590 KB of weight pieces per block, 7.2 TB/s over all
// CUs while it runs, the rate tools/microbench/gather_bw.hip measures for random 128-byte pieces.
//
// Forward message + group product + normalisation -> Mi, Ch.  (A reverse mode of this layout existed in round 2 and again, with the symmetric
// walk order, in round 4: 216 VGPRs, one block per CU, 325 us against 283 for the row kernel k_message_adjoint_gd; removed,
// profiles/r04_notes.md, profiles/r04_experiments/.)
#include <cstdlib>

#include "tn_common.h"
#include "tn_kernels.h"

namespace tn {

constexpr int MP_TA = 64;        // rows per tile
constexpr int MP_FC = 32;  // channels per block
constexpr int MP_W = 64;   // source-window capacity (rows staged in LDS)
constexpr int MP_E = 4096;  // adjacency entries of a tile kept in LDS (32 KB)
constexpr int MP_U = 2;    // edges whose weights are loaded ahead of their use

typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));
template <int VW> struct VecOf;
template <> struct VecOf<4> { typedef f4v T; };
template <> struct VecOf<2> { typedef f2v T; };

__device__ __forceinline__ f4v ldg4(const float* p) { return *reinterpret_cast<const f4v*>(p); }

// Order in which a row walks its (ascending) neighbour list.  Both rows of a pair read the pair's weight row; in list order
// the two reads are up to a tile lifetime apart and the second one misses the 4 MB L2 two times out of three (PMC: 1.30x the
// algorithmic bytes).  Walking in the order of the symmetric key (i + j) mod 64 - a ROTATION of the ascending list: start at the
// first neighbour j >= J, J = -i (mod 64), wrap at the end - puts the pair (i, j) at about the same trip of both rows, so the
// second read finds the line in L2.  Any rotation is a valid order; molecules wider than 64 atoms only lose the symmetry.
// The start is the position of the smallest key in the row; the LPR lanes of a row look at every LPR-th entry (independent
// loads, one round trip for rows of up to 8 LPR entries) and combine with a min over the group.  Group-uniform result.
template <int LPR>
__device__ __forceinline__ int row_rotation(const Graph& g, int i, int e0, int len, int ql) {
  unsigned best = 0xffffffffu;  // (key << 16) | position
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const int pos = ql + m * LPR;
    if (pos < len) best = min(best, ((unsigned)((i + g.col[e0 + pos]) & 63) << 16) | (unsigned)pos);
  }
  for (int pos = ql + 8 * LPR; pos < len; pos += LPR)
    best = min(best, ((unsigned)((i + g.col[e0 + pos]) & 63) << 16) | (unsigned)pos);
#pragma unroll
  for (int off = 1; off < LPR; off <<= 1) best = min(best, (unsigned)__shfl_xor((int)best, off, 64));
  return len > 0 ? (int)(best & 0xffffu) : 0;
}
__device__ __forceinline__ int walk_edge(int e0, int len, int rot, int t) {  // edge of trip t (clamped to the last trip)
  int pos = rot + min(t, len - 1);
  if (pos >= len) pos -= len;
  return e0 + max(pos, 0);
}

// LPR lanes own a row, VW channels per lane (LPR * VW = 32 channels per block).  8 x 4: 512 threads, 16-byte accesses, the
// fewest instructions per channel - but the reverse sweep's per-lane state (two 9-vectors, weights and their derivatives
// two edges ahead) is then 216 VGPRs = two waves per SIMD, too few to cover the loads.  16 x 2: 1024 threads, 8-byte
// accesses, half the state per lane (four waves per SIMD in one block per CU).
template <int LPR, int VW>
__global__ __launch_bounds__(64 * LPR) void k_message_rows8(Graph g, int N, int F, const float* __restrict__ w,
                                                                const float* __restrict__ src, const float* __restrict__ q,
                                                                const int64_t* __restrict__ batch, int o3,
                                                                float* __restrict__ Mi, float* __restrict__ out, int nchunks,
                                                                int balance) {
  constexpr int MP_FC = VW * LPR, MP_THREADS = 64 * LPR, PIECES = MP_FC / 4;
  constexpr int U = MP_U;
  static_assert(MP_TA == 64 && MP_E % MP_THREADS == 0, "one wave ranks the tile's rows");
  typedef typename VecOf<VW>::T vf;
  auto ldv = [](const float* p) { return *reinterpret_cast<const vf*>(p); };
  __shared__ __attribute__((aligned(16))) float win[MP_W * 9 * MP_FC];
  __shared__ __attribute__((aligned(16))) float s_help[(MP_TA / 2) * 9 * MP_FC];  // helper partial sums of the long rows
  __shared__ int s_col[MP_E], s_pair[MP_E];                                          // the tile's slice of col / epair
  __shared__ int s_lo[MP_THREADS / 64], s_hi[MP_THREADS / 64], s_rp[MP_TA], s_len[MP_TA], s_rank[MP_TA], s_byrank[MP_TA];
  if (g.counts[2]) return;  // pair overflow: the adjacency was not filled (the host reports the error)
  const int b = xcd_chunk(blockIdx.x, gridDim.x);
  const int tile = b / nchunks, chunk = b - tile * nchunks;
  const int r0 = tile * MP_TA, r1 = min(N, r0 + MP_TA);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int F9 = 9 * F, F3 = 3 * F, c0 = chunk * MP_FC;

  // ---- the tile's slice of the adjacency -> LDS.  Timestamps inside the kernel (round 4) showed the edge loop at 0.35 us per
  // trip with the weight loads AND the window reads removed: it waited for the (column, pair id) records, a dependent global
  // load in front of every eight trips' weight loads.  Rows r0 .. r1 are one contiguous range of col / epair (~1500 entries
  // for 64-atom molecules); it is requested together with the window bounds and read from LDS afterwards.
  const int eT0 = g.rowptr[r0], nE = g.rowptr[r1] - eT0;
  const bool csr_lds = nE <= MP_E;  // block-uniform
  int lo = 0x7fffffff, hi = -1;
  {
    int cbuf[MP_E / MP_THREADS], pbuf[MP_E / MP_THREADS];
    if (csr_lds) {
#pragma unroll
      for (int k = 0; k < MP_E / MP_THREADS; ++k) {
        const int idx = tid + k * MP_THREADS;
        if (idx < nE) {
          cbuf[k] = g.col[eT0 + idx];
          pbuf[k] = g.epair[eT0 + idx];
        }
      }
    }
    // column window of the tile (rows are sorted ascending: first / last entry of each row)
    if (tid < MP_TA) {
      const int rr = min(r0 + tid, r1);
      const int a0 = g.rowptr[rr], a1 = g.rowptr[min(rr + 1, r1)];
      s_rp[tid] = a0 - eT0;
      s_len[tid] = a1 - a0;
      if (a1 > a0) {
        lo = g.col[a0];
        hi = g.col[a1 - 1];
      }
    }
    if (csr_lds) {
#pragma unroll
      for (int k = 0; k < MP_E / MP_THREADS; ++k) {
        const int idx = tid + k * MP_THREADS;
        if (idx < nE) {
          s_col[idx] = cbuf[k];
          s_pair[idx] = pbuf[k];
        }
      }
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    lo = min(lo, __shfl_xor(lo, off, 64));
    hi = max(hi, __shfl_xor(hi, off, 64));
  }
  if (lane == 0) {
    s_lo[wave] = lo;
    s_hi[wave] = hi;
  }
  __syncthreads();
  lo = s_lo[0];
  hi = s_hi[0];
#pragma unroll
  for (int k = 1; k < MP_THREADS / 64; ++k) {
    lo = min(lo, s_lo[k]);
    hi = max(hi, s_hi[k]);
  }
  const int wn = hi - lo + 1;
  const bool staged = hi >= lo && wn <= MP_W;  // block-uniform

  {
    // window rows -> LDS as [row][9][32]: 8 lanes x 16 B cover the 32 channels of one (row, component)
    // all of a thread's pieces are requested before the first one is stored: as a plain loop (trip count unknown to the
    // compiler) this was nine dependent load -> store round trips, 9 of the 44 us a block lives
    const int pieces = staged ? wn * 9 * PIECES : 0;
    constexpr int NIT = (MP_W * 9 * PIECES + MP_THREADS - 1) / MP_THREADS;
    f4v tmp[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int idx = tid + k * MP_THREADS;
      if (idx < pieces) {
        const int rc = idx / PIECES, f4 = (idx % PIECES) << 2;
        const int row = rc / 9, c = rc - row * 9;
        tmp[k] = ldg4(src + (int64_t)(lo + row) * F9 + c * F + c0 + f4);
      }
    }
    // while the window is on its way: rank of every row by length (0 = longest; ties by row), one wave
    if (tid < MP_TA) {
      const int mylen = s_len[tid];
      int rank = 0;
      for (int r = 0; r < MP_TA; ++r) {
        const int l = s_len[r];
        rank += (l > mylen || (l == mylen && r < tid)) ? 1 : 0;
      }
      s_rank[tid] = rank;
      s_byrank[rank] = tid;
    }
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int idx = tid + k * MP_THREADS;
      if (idx < pieces) {
        const int rc = idx / PIECES, f4 = (idx % PIECES) << 2;
        *reinterpret_cast<f4v*>(&win[rc * MP_FC + f4]) = tmp[k];
      }
    }
    __syncthreads();
  }

  const int rl = tid / LPR, i = r0 + rl, ql = tid & (LPR - 1), f = c0 + VW * ql;  // row, lane within the row's group, first channel
  const bool live = i < r1;
  vf acc[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) acc[c] = (vf)(0.f);
  int help_slot = -1;  // >= 0: this row is a long row, the partial sum of its last entries is in s_help[help_slot]

  if (csr_lds) {
    // ---- Balanced walk.  The 8 rows of a wave advance together and the block lives as long as its longest row: 45 trips at
    // C2 against a mean row length of 24.  The k-th longest row hands the last h = (len_long - len_short) / 2 entries of its
    // list to the group of the k-th shortest row, which sums them FIRST (segment A), parks that partial sum in LDS and then walks
    // its own row (segment B); the long row's group adds the parked sum to its own in the epilogue.  Fixed order, one
    // writer per slot: deterministic (the order differs from the one-lane-per-channel sweeps' by that one split).
    const int rank = s_rank[rl], partner = s_byrank[MP_TA - 1 - rank];
    const int mylen = s_len[rl], plen = s_len[partner];
    int baseA = 0, lenA = 0, rowA = 0, baseB = s_rp[rl], lenB = mylen, slotA = 0;
    if (balance) {
      if (rank < MP_TA / 2) {
        const int h = ((mylen - plen) / 2) & ~(U - 1);
        lenB = mylen - h;
        if (h > 0) help_slot = rank;
      } else {
        const int h = ((plen - mylen) / 2) & ~(U - 1);
        lenA = h;
        baseA = s_rp[partner] + plen - h;
        rowA = r0 + partner;
        slotA = MP_TA - 1 - rank;
      }
    }
    // rotation of a segment (see row_rotation): position of its smallest key (row + column) mod 64
    auto seg_rot = [&](int base, int len, int row) __attribute__((always_inline)) {
      unsigned best = 0xffffffffu;
      for (int pos = ql; pos < len; pos += LPR) best = min(best, ((unsigned)((row + s_col[base + pos]) & 63) << 16) | (unsigned)pos);
#pragma unroll
      for (int off = 1; off < LPR; off <<= 1) best = min(best, (unsigned)__shfl_xor((int)best, off, 64));
      return len > 0 ? (int)(best & 0xffffu) : 0;
    };
    const int rotA = seg_rot(baseA, lenA, rowA), rotB = seg_rot(baseB, lenB, i);
    const int ntrip = lenA + lenB;
    int nmax = ntrip;  // the 8 rows of a wave advance together: trip count = the longest of them (past its end a row adds zeros)
#pragma unroll
    for (int off = LPR; off <= 32; off <<= 1) nmax = max(nmax, __shfl_xor(nmax, off, 64));

    // every lane of a row reads the (column, pair id) of a trip from LDS (one address per row: a broadcast)
    auto idx_of = [&](int t, int& j, int& p) __attribute__((always_inline)) {
      const bool inA = t < lenA;
      const int tt = inA ? t : t - lenA, L = inA ? lenA : lenB;
      int pos = (inA ? rotA : rotB) + min(tt, L - 1);
      if (pos >= L) pos -= L;
      const int entry = L > 0 ? (inA ? baseA : baseB) + pos : 0;  // a row without entries: entry 0 of the slice, masked
      j = s_col[entry];
      p = s_pair[entry];
    };
    auto park = [&]() __attribute__((always_inline)) {
      float* hp = s_help + slotA * (9 * MP_FC) + VW * ql;
#pragma unroll
      for (int c = 0; c < 9; ++c) {
        *reinterpret_cast<vf*>(hp + c * MP_FC) = acc[c];
        acc[c] = (vf)(0.f);
      }
    };
    bool parked = lenA == 0;
    int jn[U], pn[U];
#pragma unroll
    for (int u = 0; u < U; ++u) idx_of(u, jn[u], pn[u]);
    for (int t = 0; t < nmax; t += U) {
      if (!parked && t == lenA) {  // lenA is a multiple of U
        park();
        parked = true;
      }
      int jj[U];
      vf wv[U][3];
      float msk[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        jj[u] = jn[u];
        msk[u] = (t + u < ntrip) ? 1.0f : 0.0f;
        const float* wp = w + (int64_t)pn[u] * F3 + f;
        wv[u][0] = ldv(wp);
        wv[u][1] = ldv(wp + F);
        wv[u][2] = ldv(wp + 2 * F);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) idx_of(t + U + u, jn[u], pn[u]);  // the next step's records (clamped past the end)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        vf s9[9];
        if (staged) {
          const float* sp = win + (jj[u] - lo) * (9 * MP_FC) + VW * ql;
#pragma unroll
          for (int c = 0; c < 9; ++c) s9[c] = ldv(sp + c * MP_FC);
        } else {
          const float* sp = src + (int64_t)jj[u] * F9 + f;
#pragma unroll
          for (int c = 0; c < 9; ++c) s9[c] = ldv(sp + c * F);
        }
        const vf w0 = wv[u][0] * msk[u], w1 = wv[u][1] * msk[u], w2 = wv[u][2] * msk[u];
        acc[0] += w0 * s9[0];
        acc[1] += w1 * s9[1];
        acc[2] += w1 * s9[2];
        acc[3] += w1 * s9[3];
        acc[4] += w2 * s9[4];
        acc[5] += w2 * s9[5];
        acc[6] += w2 * s9[6];
        acc[7] += w2 * s9[7];
        acc[8] += w2 * s9[8];
      }
    }
    if (!parked) park();  // a helper whose own row is empty
  } else {
    const int e0 = live ? g.rowptr[i] : 0, e1 = live ? g.rowptr[i + 1] : 0;
    const int grp = lane & ~(LPR - 1);  // first lane of this row's group within the wave
    const int len = e1 - e0, rot = row_rotation<LPR>(g, i, e0, len, ql);
    int nmax = e1 - e0;
#pragma unroll
    for (int off = LPR; off <= 32; off <<= 1) nmax = max(nmax, __shfl_xor(nmax, off, 64));
    for (int eb = 0; eb < nmax; eb += LPR) {
      const int me = walk_edge(e0, len, rot, eb + ql);  // clamped: lanes past the row's end repeat its last edge with zero weights
      const int myc = (e1 > e0) ? g.col[me] : 0, myp = (e1 > e0) ? g.epair[me] : 0;
      const int n = min(LPR, nmax - eb);
      for (int k = 0; k < n; k += MP_U) {
        int jj[MP_U], pp[MP_U];
        vf wv[MP_U][3];
        float msk[MP_U];
#pragma unroll
        for (int u = 0; u < MP_U; ++u) {
          const bool valid = e0 + eb + k + u < e1;
          const int srcl = grp + min(k + u, LPR - 1);
          jj[u] = __shfl(myc, srcl, 64);
          pp[u] = __shfl(myp, srcl, 64);
          msk[u] = valid ? 1.0f : 0.0f;
          const float* wp = w + (int64_t)pp[u] * F3 + f;
          wv[u][0] = ldv(wp);
          wv[u][1] = ldv(wp + F);
          wv[u][2] = ldv(wp + 2 * F);
        }
#pragma unroll
        for (int u = 0; u < MP_U; ++u) {
          vf s9[9];
          if (staged) {
            const float* sp = win + (jj[u] - lo) * (9 * MP_FC) + VW * ql;
#pragma unroll
            for (int c = 0; c < 9; ++c) s9[c] = ldv(sp + c * MP_FC);
          } else {
            const float* sp = src + (int64_t)jj[u] * F9 + f;
#pragma unroll
            for (int c = 0; c < 9; ++c) s9[c] = ldv(sp + c * F);
          }
          const vf w0 = wv[u][0] * msk[u], w1 = wv[u][1] * msk[u], w2 = wv[u][2] * msk[u];
          acc[0] += w0 * s9[0];
          acc[1] += w1 * s9[1];
          acc[2] += w1 * s9[2];
          acc[3] += w1 * s9[3];
          acc[4] += w2 * s9[4];
          acc[5] += w2 * s9[5];
          acc[6] += w2 * s9[6];
          acc[7] += w2 * s9[7];
          acc[8] += w2 * s9[8];
        }
      }
    }
  }
  __syncthreads();  // the parked sums are complete
  if (!live) return;
  if (help_slot >= 0) {
    const float* hp = s_help + help_slot * (9 * MP_FC) + VW * ql;
#pragma unroll
    for (int c = 0; c < 9; ++c) acc[c] += ldv(hp + c * MP_FC);
  }

  float* o = out + (int64_t)i * F9 + f;
  vf yy[9];  // the row's own source row; overwritten component by component with the result
  if (staged && i >= lo && i <= hi) {  // the row's own source row is in its window whenever it has a self edge
    const float* yp = win + (i - lo) * (9 * MP_FC) + VW * ql;
#pragma unroll
    for (int c = 0; c < 9; ++c) yy[c] = ldv(yp + c * MP_FC);
  } else {
    const float* yp = src + (int64_t)i * F9 + f;
#pragma unroll
    for (int c = 0; c < 9; ++c) yy[c] = ldv(yp + c * F);
  }
  float* mo = Mi + (int64_t)i * F9 + f;
#pragma unroll
  for (int c = 0; c < 9; ++c) *reinterpret_cast<vf*>(mo + c * F) = acc[c];
  const float kap = q ? (batch ? 1.0f + 0.1f * q[batch[i]] : q[i]) : 1.0f;
#pragma unroll
  for (int t = 0; t < VW; ++t) {
    float m9[9], y9[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      m9[c] = acc[c][t];
      y9[c] = yy[c][t];
    }
    const M3 Y = compose(y9), M = compose(m9);
    M3 Cm = o3 ? scale(add(matmul(Y, M), matmul(M, Y)), kap) : scale(matmul(Y, M), 2.0f);
    float uc[9];
    decompose(Cm, uc);
    const float inv = 1.0f / (frob2(Cm) + 1.0f);
#pragma unroll
    for (int c = 0; c < 9; ++c) yy[c][t] = uc[c] * inv;
  }
#pragma unroll
  for (int c = 0; c < 9; ++c) *reinterpret_cast<vf*>(o + c * F) = yy[c];
}

// Reverse sweep in the same tile layout with the same two devices (adjacency slice in LDS, balanced rows): the adjoint of the
// message sum,  gPn[i] += sum_e w[p] * gMi[col(e)],  and the layer's distance-gradient halves
// h(i <- j) = sum_{k,f} dw[p,k,f] * sum_{c in k} gMi[j,c,f] * Pn[i,c,f]  (k_message_adjoint_gd, tn_kernels.hip).  The gMi rows of
// the tile's column window sit in LDS, the Pn row of the row being walked in registers (a helper group holds the LONG row's Pn
// while it walks that row's tail, then its own); per edge a lane loads six 16-byte pieces (w, dw) and reads nine from LDS; the 8
// lanes of a row reduce their channel products and lane 0 writes the slot (channel chunk, pair, direction): one writer per slot,
// summed in fixed order by the embedding's pair kernel.  Tiles whose window or adjacency slice does not fit LDS are not taken by
// this kernel at all (message_adjoint_pair_ok looks at the batch; the row kernel serves them).
__global__ __launch_bounds__(512) void k_message_adjoint_rows8(Graph g, int N, int F, const float* __restrict__ w,
                                                               const float* __restrict__ dw, const float* __restrict__ gMi,
                                                               const float* __restrict__ Pn, float* __restrict__ gPn,
                                                               float* __restrict__ slots, int64_t slot_stride, int nchunks) {
  constexpr int LPR = 8, VW = 4, FC = 32, THREADS = 512, PIECES = FC / 4, U = 2;
  typedef f4v vf;
  auto ldv = [](const float* p) { return *reinterpret_cast<const vf*>(p); };
  __shared__ __attribute__((aligned(16))) float win[MP_W * 9 * FC];
  __shared__ __attribute__((aligned(16))) float s_help[(MP_TA / 2) * 9 * FC];
  __shared__ int s_col[MP_E], s_pair[MP_E];
  __shared__ int s_lo[THREADS / 64], s_hi[THREADS / 64], s_rp[MP_TA], s_len[MP_TA], s_rank[MP_TA], s_byrank[MP_TA];
  if (g.counts[2]) return;
  const int b = xcd_chunk(blockIdx.x, gridDim.x);
  const int tile = b / nchunks, chunk = b - tile * nchunks;
  const int r0 = tile * MP_TA, r1 = min(N, r0 + MP_TA);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int F9 = 9 * F, F3 = 3 * F, c0 = chunk * FC;

  const int eT0 = g.rowptr[r0], nE = g.rowptr[r1] - eT0;
  const bool csr_lds = nE <= MP_E;  // block-uniform
  int lo = 0x7fffffff, hi = -1;
  {
    int cbuf[MP_E / THREADS], pbuf[MP_E / THREADS];
    if (csr_lds) {
#pragma unroll
      for (int k = 0; k < MP_E / THREADS; ++k) {
        const int idx = tid + k * THREADS;
        if (idx < nE) {
          cbuf[k] = g.col[eT0 + idx];
          pbuf[k] = g.epair[eT0 + idx];
        }
      }
    }
    if (tid < MP_TA) {
      const int rr = min(r0 + tid, r1);
      const int a0 = g.rowptr[rr], a1 = g.rowptr[min(rr + 1, r1)];
      s_rp[tid] = a0 - eT0;
      s_len[tid] = a1 - a0;
      if (a1 > a0) {
        lo = g.col[a0];
        hi = g.col[a1 - 1];
      }
    }
    if (csr_lds) {
#pragma unroll
      for (int k = 0; k < MP_E / THREADS; ++k) {
        const int idx = tid + k * THREADS;
        if (idx < nE) {
          s_col[idx] = cbuf[k];
          s_pair[idx] = pbuf[k];
        }
      }
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    lo = min(lo, __shfl_xor(lo, off, 64));
    hi = max(hi, __shfl_xor(hi, off, 64));
  }
  if (lane == 0) {
    s_lo[wave] = lo;
    s_hi[wave] = hi;
  }
  __syncthreads();
  lo = s_lo[0];
  hi = s_hi[0];
#pragma unroll
  for (int k = 1; k < THREADS / 64; ++k) {
    lo = min(lo, s_lo[k]);
    hi = max(hi, s_hi[k]);
  }
  const int wn = hi - lo + 1;
  const bool staged = hi >= lo && wn <= MP_W;  // block-uniform
  {
    const int pieces = staged ? wn * 9 * PIECES : 0;
    constexpr int NIT = (MP_W * 9 * PIECES + THREADS - 1) / THREADS;
    f4v tmp[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int idx = tid + k * THREADS;
      if (idx < pieces) {
        const int rc = idx / PIECES, f4 = (idx % PIECES) << 2;
        const int row = rc / 9, c = rc - row * 9;
        tmp[k] = ldg4(gMi + (int64_t)(lo + row) * F9 + c * F + c0 + f4);
      }
    }
    if (tid < MP_TA) {
      const int mylen = s_len[tid];
      int rank = 0;
      for (int r = 0; r < MP_TA; ++r) {
        const int l = s_len[r];
        rank += (l > mylen || (l == mylen && r < tid)) ? 1 : 0;
      }
      s_rank[tid] = rank;
      s_byrank[rank] = tid;
    }
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int idx = tid + k * THREADS;
      if (idx < pieces) {
        const int rc = idx / PIECES, f4 = (idx % PIECES) << 2;
        *reinterpret_cast<f4v*>(&win[rc * FC + f4]) = tmp[k];
      }
    }
    __syncthreads();
  }

  const int rl = tid / LPR, i = r0 + rl, ql = tid & (LPR - 1), f = c0 + VW * ql;
  const bool live = i < r1;
  vf acc[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) acc[c] = (vf)(0.f);
  int help_slot = -1;
  float* const slot_base = slots + (int64_t)chunk * slot_stride;

  // segments: A = the tail of the partner's (long) row, walked first; B = the own row (or its head, when this row is a long one)
  const int rank = s_rank[rl], partner = s_byrank[MP_TA - 1 - rank];
  const int mylen = s_len[rl], plen = s_len[partner];
  int baseA = 0, lenA = 0, rowA = r0, baseB = s_rp[rl], lenB = mylen, slotA = 0;
  if (csr_lds) {
    if (rank < MP_TA / 2) {
      const int h = ((mylen - plen) / 2) & ~(U - 1);
      lenB = mylen - h;
      if (h > 0) help_slot = rank;
    } else {
      const int h = ((plen - mylen) / 2) & ~(U - 1);
      lenA = h;
      baseA = s_rp[partner] + plen - h;
      rowA = r0 + partner;
      slotA = MP_TA - 1 - rank;
    }
  }
  auto seg_rot = [&](int base, int len, int row) __attribute__((always_inline)) {
    unsigned best = 0xffffffffu;
    if (csr_lds)
      for (int pos = ql; pos < len; pos += LPR) best = min(best, ((unsigned)((row + s_col[base + pos]) & 63) << 16) | (unsigned)pos);
    else
      for (int pos = ql; pos < len; pos += LPR) best = min(best, ((unsigned)((row + g.col[eT0 + base + pos]) & 63) << 16) | (unsigned)pos);
#pragma unroll
    for (int off = 1; off < LPR; off <<= 1) best = min(best, (unsigned)__shfl_xor((int)best, off, 64));
    return len > 0 ? (int)(best & 0xffffu) : 0;
  };
  const int rotA = seg_rot(baseA, lenA, rowA), rotB = seg_rot(baseB, lenB, i);
  const int ntrip = lenA + lenB;
  int nmax = ntrip;
#pragma unroll
  for (int off = LPR; off <= 32; off <<= 1) nmax = max(nmax, __shfl_xor(nmax, off, 64));

  // (column, pair id) of trip t: LDS broadcast read (or, for a slice that did not fit, a global read)
  auto idx_of = [&](int t, int& j, int& p) __attribute__((always_inline)) {
    const bool inA = t < lenA;
    const int tt = inA ? t : t - lenA, L = inA ? lenA : lenB;
    int pos = (inA ? rotA : rotB) + min(tt, L - 1);
    if (pos >= L) pos -= L;
    const int entry = L > 0 ? (inA ? baseA : baseB) + pos : 0;
    if (csr_lds) {
      j = s_col[entry];
      p = s_pair[entry];
    } else {
      j = nE > 0 ? g.col[eT0 + entry] : 0;
      p = nE > 0 ? g.epair[eT0 + entry] : 0;
    }
  };
  vf y[9];  // Pn row of the row being walked
  auto load_y = [&](int row) __attribute__((always_inline)) {
    const float* yp = Pn + (int64_t)min(row, N - 1) * F9 + f;
#pragma unroll
    for (int c = 0; c < 9; ++c) y[c] = ldv(yp + c * F);
  };
  auto park = [&]() __attribute__((always_inline)) {
    float* hp = s_help + slotA * (9 * FC) + VW * ql;
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      *reinterpret_cast<vf*>(hp + c * FC) = acc[c];
      acc[c] = (vf)(0.f);
    }
    load_y(i);
  };
  bool parked = lenA == 0;
  load_y(parked ? i : rowA);
  int jn[U], pn[U];
#pragma unroll
  for (int u = 0; u < U; ++u) idx_of(u, jn[u], pn[u]);
  for (int t = 0; t < nmax; t += U) {
    if (!parked && t == lenA) {  // lenA is a multiple of U
      park();
      parked = true;
    }
    const int row_now = t < lenA ? rowA : i;
    int jj[U], pp[U];
    vf wv[U][3], dv[U][3];
    float msk[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      jj[u] = jn[u];
      pp[u] = pn[u];
      msk[u] = (t + u < ntrip) ? 1.0f : 0.0f;
      const float* wp = w + (int64_t)pn[u] * F3 + f;
      const float* dp = dw + (int64_t)pn[u] * F3 + f;
      wv[u][0] = ldv(wp);
      wv[u][1] = ldv(wp + F);
      wv[u][2] = ldv(wp + 2 * F);
      dv[u][0] = ldv(dp);
      dv[u][1] = ldv(dp + F);
      dv[u][2] = ldv(dp + 2 * F);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) idx_of(t + U + u, jn[u], pn[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      vf s9[9];
      if (staged) {
        const float* sp = win + (jj[u] - lo) * (9 * FC) + VW * ql;
#pragma unroll
        for (int c = 0; c < 9; ++c) s9[c] = ldv(sp + c * FC);
      } else {
        const float* sp = gMi + (int64_t)jj[u] * F9 + f;
#pragma unroll
        for (int c = 0; c < 9; ++c) s9[c] = ldv(sp + c * F);
      }
      const vf w0 = wv[u][0] * msk[u], w1 = wv[u][1] * msk[u], w2 = wv[u][2] * msk[u];
      acc[0] += w0 * s9[0];
      acc[1] += w1 * s9[1];
      acc[2] += w1 * s9[2];
      acc[3] += w1 * s9[3];
      acc[4] += w2 * s9[4];
      acc[5] += w2 * s9[5];
      acc[6] += w2 * s9[6];
      acc[7] += w2 * s9[7];
      acc[8] += w2 * s9[8];
      const vf hv = dv[u][0] * (s9[0] * y[0]) + dv[u][1] * (s9[1] * y[1] + s9[2] * y[2] + s9[3] * y[3]) +
                    dv[u][2] * (s9[4] * y[4] + s9[5] * y[5] + s9[6] * y[6] + s9[7] * y[7] + s9[8] * y[8]);
      const float h = row_sum((hv[0] + hv[1]) + (hv[2] + hv[3]), LPR);
      // slot (pair, direction): direction 0 when the walked row is the pair's i (its neighbour has the smaller index); self edge: none
      if (ql == 0 && t + u < ntrip && jj[u] != row_now) slot_base[2 * pp[u] + (jj[u] < row_now ? 0 : 1)] = h;
    }
  }
  if (!parked) park();  // a helper whose own row is empty
  __syncthreads();      // the parked sums are complete
  if (!live) return;
  if (help_slot >= 0) {
    const float* hp = s_help + help_slot * (9 * FC) + VW * ql;
#pragma unroll
    for (int c = 0; c < 9; ++c) acc[c] += ldv(hp + c * FC);
  }
  float* o = gPn + (int64_t)i * F9 + f;
#pragma unroll
  for (int c = 0; c < 9; ++c) {
    vf* op = reinterpret_cast<vf*>(o + c * F);
    *op = *op + acc[c];
  }
}

bool message_adjoint_pair_ok(const Graph& g, int N, int F) {
  static const bool off2 = getenv("TMDNET_NO_ADJ_ROWS8") != nullptr;  // developer switch: the row kernel k_message_adjoint_gd
  static const bool off = getenv("TMDNET_NO_MSG_ROWS8") != nullptr;   // developer switch: one-channel-per-lane sweeps
  if (off || off2 || !g.small_mols || F < MP_FC || F % MP_FC) return false;
  return (int64_t)((N + MP_TA - 1) / MP_TA) * (F / MP_FC) >= 512;
}
void launch_message_adjoint_pair(const Graph& g, int N, int F, const float* w, const float* dw, const float* gMi, const float* Pn,
                                 float* gPn, float* slots, int64_t slot_stride, hipStream_t s) {
  const int nchunks = F / MP_FC, tiles = (N + MP_TA - 1) / MP_TA;
  hipLaunchKernelGGL(k_message_adjoint_rows8, dim3(tiles * nchunks), dim3(512), 0, s, g, N, F, w, dw, gMi, Pn, gPn, slots, slot_stride,
                     nchunks);
}

bool message_pair_ok(int N, int F) {
  static const bool off = getenv("TMDNET_NO_MSG_ROWS8") != nullptr;  // developer switch: one-channel-per-lane sweeps
  if (off || F < MP_FC || F % MP_FC) return false;
  return (int64_t)((N + MP_TA - 1) / MP_TA) * (F / MP_FC) >= 512;
}
void launch_message_pair(const Graph& g, int N, int F, const float* w, const float* src, const float* q, const int64_t* batch,
                         int o3, float* Mi, float* Ch, hipStream_t s) {
  const int nchunks = F / MP_FC, tiles = (N + MP_TA - 1) / MP_TA;
  static const int balance = getenv("TMDNET_MSG_NOBALANCE") ? 0 : 1;  // developer switch: every row walked by its own group only
  hipLaunchKernelGGL((k_message_rows8<8, 4>), dim3(tiles * nchunks), dim3(512), 0, s, g, N, F, w, src, q, batch, o3, Mi, Ch, nchunks, balance);
}

}  // namespace tn
