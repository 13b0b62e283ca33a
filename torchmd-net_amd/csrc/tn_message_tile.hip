// LDS-staged message sweeps (reference tensornet.py:757-806 and its adjoint): the CSR segmented gather-sum
//     acc[i, c, f] = sum_{e in row(i)} w[pair(e), type(c), f] * src[col(e), c, f]
// for a TILE of 64 consecutive rows and 32 channels per block.  In a batch of molecules (sorted batch index) the
// neighbours of 64 consecutive atoms are a small contiguous window of atoms, so the window's source rows
// [W <= 64][9][32 channels] = 72 KB are staged in LDS once and every edge gathers its 9 components from LDS instead
// of L2 (the row-per-block kernel moved 2.5 GB through L2 per launch at C2, this one 0.7 GB).  A tile whose column
// window is wider (large single systems in cell order, ragged molecules straddling the tile) gathers from global
// memory exactly as k_message does; the choice is block-uniform.  Same edge order per (row, channel) as k_message:
// bit-identical sums, no atomics.
//
// Thread layout: MT_THREADS = row slots x 32 channels (one half-wave per row: its LDS reads are 32 consecutive dwords
// -> conflict free); a thread walks rows slot, slot + slots, ...  The 32 lanes of a row load 32 edge indices of
// the row with one coalesced access and hand them round by ds_bpermute, so no memory latency sits between two edges.
#include "tn_common.h"
#include "tn_kernels.h"

namespace tn {

constexpr int MT_TA = 64;   // rows per tile
constexpr int MT_FC = 32;   // channels per block
constexpr int MT_W = 64;    // window capacity (rows staged in LDS)
constexpr int MT_THREADS = 512;
constexpr int MT_SLOTS = MT_THREADS / 32;

template <int MODE>  // 0: forward message + group product + normalisation -> Mi, Ch ; 1: adjoint, out += gather
__global__ __launch_bounds__(MT_THREADS, 2) void k_message_tile(Graph g, int N, int F, const float* __restrict__ w,
                                                         const float* __restrict__ src, const float* __restrict__ q,
                                                         const int64_t* __restrict__ batch, int o3, float* __restrict__ Mi,
                                                         float* __restrict__ out, int nchunks) {
  __shared__ __attribute__((aligned(16))) float win[MT_W * 9 * MT_FC];
  __shared__ int s_lo[MT_THREADS / 64], s_hi[MT_THREADS / 64];
  if (g.counts[2]) return;  // pair overflow: the adjacency was not filled (the host reports the error)
  const int b = xcd_chunk(blockIdx.x, gridDim.x);
  const int tile = b / nchunks, chunk = b - tile * nchunks;
  const int r0 = tile * MT_TA, r1 = min(N, r0 + MT_TA);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int F9 = 9 * F, F3 = 3 * F, c0 = chunk * MT_FC;

  // column window of the tile: rows are sorted ascending, so first / last entry of each row
  int lo = 0x7fffffff, hi = -1;
  if (tid < r1 - r0) {  // MT_TA <= MT_THREADS
    const int e0 = g.rowptr[r0 + tid], e1 = g.rowptr[r0 + tid + 1];
    if (e1 > e0) {
      lo = g.col[e0];
      hi = g.col[e1 - 1];
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
  for (int k = 1; k < MT_THREADS / 64; ++k) {
    lo = min(lo, s_lo[k]);
    hi = max(hi, s_hi[k]);
  }
  const int wn = hi - lo + 1;
  const bool staged = hi >= lo && wn <= MT_W;  // block-uniform
  if (staged) {
    // window rows -> LDS as [row][9][32]: 8 lanes x 16 B cover the 32 channels of one (row, component)
    const int pieces = wn * 9 * (MT_FC / 4);
    for (int idx = tid; idx < pieces; idx += MT_THREADS) {
      const int rc = idx >> 3, f4 = (idx & 7) << 2;
      const int row = rc / 9, c = rc - row * 9;
      *reinterpret_cast<float4*>(&win[rc * MT_FC + f4]) =
          *reinterpret_cast<const float4*>(src + (int64_t)(lo + row) * F9 + c * F + c0 + f4);
    }
    __syncthreads();
  }

  const int fl = tid & 31, f = c0 + fl;
  for (int i = r0 + (tid >> 5); i < r1; i += MT_SLOTS) {
    const int e0 = g.rowptr[i], e1 = g.rowptr[i + 1];
    float acc[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) acc[c] = 0.f;
    for (int eb = e0; eb < e1; eb += 32) {
      const int me = eb + fl < e1 ? eb + fl : e1 - 1;
      const int myc = g.col[me], myp = g.epair[me];
      const int n = min(32, e1 - eb);
      // groups of 4 edges: all 12 weight loads of a group are issued before the first is used.  A slot past the
      // row's end repeats the last edge with zero weights (adds +0, sums unchanged).
      for (int k = 0; k < n; k += 4) {
        int jj[4];
        float wv[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const bool valid = k + u < n;
          const int kc = valid ? k + u : n - 1;
          jj[u] = __shfl(myc, kc, 32);
          const int p = __shfl(myp, kc, 32);
          const float* wp = w + (int64_t)p * F3 + f;
          const float msk = valid ? 1.0f : 0.0f;
          wv[u][0] = wp[0] * msk;
          wv[u][1] = wp[F] * msk;
          wv[u][2] = wp[2 * F] * msk;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (staged) {
            const float* sp = win + (jj[u] - lo) * (9 * MT_FC) + fl;
            acc[0] += wv[u][0] * sp[0];
            acc[1] += wv[u][1] * sp[1 * MT_FC];
            acc[2] += wv[u][1] * sp[2 * MT_FC];
            acc[3] += wv[u][1] * sp[3 * MT_FC];
            acc[4] += wv[u][2] * sp[4 * MT_FC];
            acc[5] += wv[u][2] * sp[5 * MT_FC];
            acc[6] += wv[u][2] * sp[6 * MT_FC];
            acc[7] += wv[u][2] * sp[7 * MT_FC];
            acc[8] += wv[u][2] * sp[8 * MT_FC];
          } else {
            const float* sp = src + (int64_t)jj[u] * F9 + f;
            acc[0] += wv[u][0] * sp[0];
            acc[1] += wv[u][1] * sp[F];
            acc[2] += wv[u][1] * sp[2 * F];
            acc[3] += wv[u][1] * sp[3 * F];
            acc[4] += wv[u][2] * sp[4 * F];
            acc[5] += wv[u][2] * sp[5 * F];
            acc[6] += wv[u][2] * sp[6 * F];
            acc[7] += wv[u][2] * sp[7 * F];
            acc[8] += wv[u][2] * sp[8 * F];
          }
        }
      }
    }
    float* o = out + (int64_t)i * F9 + f;
    if (MODE == 1) {
#pragma unroll
      for (int c = 0; c < 9; ++c) o[c * F] += acc[c];
    } else {
      float y[9];
      const float* yp = src + (int64_t)i * F9 + f;
#pragma unroll
      for (int c = 0; c < 9; ++c) y[c] = yp[c * F];
      float* mo = Mi + (int64_t)i * F9 + f;
#pragma unroll
      for (int c = 0; c < 9; ++c) mo[c * F] = acc[c];
      const float kap = q ? (batch ? 1.0f + 0.1f * q[batch[i]] : q[i]) : 1.0f;
      const M3 Y = compose(y), M = compose(acc);
      M3 Cm = o3 ? scale(add(matmul(Y, M), matmul(M, Y)), kap) : scale(matmul(Y, M), 2.0f);
      float uc[9];
      decompose(Cm, uc);
      const float inv = 1.0f / (frob2(Cm) + 1.0f);
#pragma unroll
      for (int c = 0; c < 9; ++c) o[c * F] = uc[c] * inv;
    }
  }
}

bool message_tile_ok(int N, int F) {
  static const bool off = getenv("TMDNET_NO_MSG_TILE") != nullptr;  // developer switch: row-per-block kernels
  if (off || F < MT_FC || F % MT_FC) return false;
  // a tile is one block of 16 waves walking 64 rows: it needs >= 2 blocks per CU to beat one block per row
  return (int64_t)((N + MT_TA - 1) / MT_TA) * (F / MT_FC) >= 512;
}

void launch_message_tile(const Graph& g, int N, int F, const float* w, const float* src, const float* q, const int64_t* batch,
                         int o3, float* Mi, float* Ch, hipStream_t s) {
  const int nchunks = F / MT_FC, tiles = (N + MT_TA - 1) / MT_TA;
  hipLaunchKernelGGL((k_message_tile<0>), dim3(tiles * nchunks), dim3(MT_THREADS), 0, s, g, N, F, w, src, q, batch, o3, Mi, Ch, nchunks);
}
void launch_message_adjoint_tile(const Graph& g, int N, int F, const float* w, const float* gMi, float* gPn, hipStream_t s) {
  const int nchunks = F / MT_FC, tiles = (N + MT_TA - 1) / MT_TA;
  hipLaunchKernelGGL((k_message_tile<1>), dim3(tiles * nchunks), dim3(MT_THREADS), 0, s, g, N, F, w, gMi, nullptr, nullptr, 0, nullptr,
                     gPn, nchunks);
}

}  // namespace tn
