// Launch wrappers of the Equivariant Transformer kernels (tn_et.hip).  Layouts: see tn_et.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tn_kernels.h"

namespace tn {

struct EtAttnArgs {   // per-layer operands of the attention sweeps
  const float* qkv;   // [N][5F]  q | k | vx | v1 | v2
  const float* vec;   // [N][3][F] layer input
  const float* dkv;   // [P+1][Wd] silu(dk_proj phi) | silu(dv_proj phi) (thirds)
  const float* tkv;   // [P+1][Wd] d/dd of dkv (reverse pass only)
  const float* C;     // [P+1] cosine cutoff, dC its derivative
  const float* dC;
  int F, hd, Wd, dk_off, dv_off;  // *_off = -1: the model has no such projection (factor 1)
  int vector_cutoff;
  int64_t slot_stride;  // reverse target sweep: distance between the per-wave slot arrays of gd2 / gr2 (= 2 (P + 1))
  int pair_bf16;        // 1: dkv / tkv rows are stored as bf16 (option "pair_rows_bf16"): the sweeps read half the bytes
  // written on the device by launch_et_tile_prep: tile_open[0] = 0: every row's neighbours lie inside its tile (the tile sweeps of
  // tn_et_g16.hip run and the row sweeps return at once), else the other way round; tile_open[1] = number of tiles; null: row
  // sweeps only.  Tiles = runs of whole molecules of at most 64 rows: rows tile_start[t] .. tile_start[t + 1]
  const int* tile_open;
  const int* tile_start;
  int max_tiles;      // the tile sweeps' grids cover this many tiles (et_g16_max_tiles)
  const float* erec;  // [E][8] per-edge records of the tile sweeps (launch_et_tile_prep)
  int mailbox;        // set by the launchers of tn_et_g16.hip: the rows of a pair are fetched once per tile and exchanged through LDS
};

// element `idx` of a per-pair row array kept in fp32 or, in the reduced-precision storage mode, bf16.  Branch-free on purpose:
// a (wave-uniform) branch around every load stops the compiler from issuing an edge's loads together, which doubled the
// sweeps' time (profiles/r03_notes.md); both modes load one aligned dword (two lanes share it in bf16 mode) and select.
__device__ __forceinline__ float ldpair(const float* base, int64_t idx, int bf16) {
  const uint32_t w = reinterpret_cast<const uint32_t*>(base)[bf16 ? (idx >> 1) : idx];
  const uint32_t hi = w & 0xffff0000u, lo = w << 16;
  return __uint_as_float(bf16 ? ((idx & 1) ? hi : lo) : w);
}

void launch_et_embed(const int64_t* z, const float* emb, int N, int F, float* x, hipStream_t s);
void launch_et_nbr_embed(const Graph& g, int N, int F, const int64_t* z, const float* emb, const float* embN, const float* Wn,
                         float* xcat, hipStream_t s);
void launch_et_nbr_embed_bwd(const Graph& g, int Pcap, int F, const int64_t* z, const float* embN, const float* g_xcat,
                             const float* dWn, float* gd2, hipStream_t s);
void launch_et_attn_fwd(const Graph& g, int N, const EtAttnArgs& a, float* xagg, float* vagg, hipStream_t s);
void launch_et_update(const float* x, const float* vec, const float* vp, const float* o, const float* vagg, int N, int F, float* xn,
                      float* vecn, float* vdot, hipStream_t s);
void launch_et_update_bwd(const float* g_x, const float* g_vec, const float* vp, const float* o, const float* vdot, int N, int F,
                          float* g_o, float* g_vp, hipStream_t s);
// both roles of the row atom in one sweep (replaces the two launches above: the per-pair rows are read once per directed edge)
void launch_et_attn_bwd(const Graph& g, int N, const EtAttnArgs& a, const float* g_xagg, const float* g_vagg, float* g_qkv,
                        float* g_vec, float* gd2, float* gr2, hipStream_t s);
void launch_et_pair_combine(const Graph& g, int Pcap, const float* gd2, const float* gr2, int nw, int64_t stride,
                            const float* gd_extra, float* gd, float* g_rhat, hipStream_t s);
// parameter gradients: adjoints of the per-pair filter rows and of the neighbour embedding
void launch_et_train_filter(const Graph& g, int N, const EtAttnArgs& a, const float* g_xagg, const float* g_vagg, float* slots,
                            int64_t dir_stride, float* self_rows, hipStream_t s);
void launch_et_train_gpre(const Graph& g, int P, int Wd, const float* slots, int64_t dir_stride, const float* self_sum, const float* pre,
                          float* g_pre, hipStream_t s);
void launch_et_train_nbr(const Graph& g, int N, int F, const int64_t* z, const float* embN, const float* Wn, const float* g_xcat,
                         float* slots, int64_t dir_stride, float* gEN, hipStream_t s);
// third generation of the two attention sweeps (tn_et_g16.hip); et_g16_ok: the layout and the 32-bit offsets apply (P1 = pair rows)
bool et_g16_ok(int N, int64_t P1, const EtAttnArgs& a);
// once per step before the first sweep: the closed-tiles flag and the per-edge records (ecap = rows of erec)
void launch_et_tile_prep(const Graph& g, int N, int B, const int64_t* batch, const float* C, const float* dC, int64_t ecap, int pair_bf16,
                         int* meta, int* tile_start, float* erec, hipStream_t s);
int et_g16_max_tiles(int N, int B);
void launch_et_attn_fwd_g16(const Graph& g, int N, const EtAttnArgs& a, float* xagg, float* vagg, hipStream_t s);
void launch_et_attn_bwd_g16(const Graph& g, int N, const EtAttnArgs& a, const float* g_xagg, const float* g_vagg, float* g_qkv,
                            float* g_vec, float* gd2, float* gr2, hipStream_t s);
int et_sweep_waves(int F);  // waves per block of the attention sweeps = partial-sum slots per pair direction
void launch_et_cat_norm(const float* xsrc, int Fx, const float* u, int ldu, int Fn, int N, float* hcat, hipStream_t s);
void launch_et_norm_bwd(const float* g_n, int ldg, const float* u, int ldu, int Fn, int N, float* g_u, int ldgu, hipStream_t s);
void launch_et_head_mid(const float* y, const float* u2, int ldu, int F2, int N, float* hcat2, float* vq, hipStream_t s);
void launch_et_head_mid_bwd(const float* y, const float* u2, int ldu, const float* g_h2, const float* g_vq, int F2, int N, float* g_y,
                            float* g_u2, int ldgu, hipStream_t s);
void launch_et_copy2d(const float* src, int lds_, float* dst, int ldd, int rows, int cols, hipStream_t s);
void launch_et_add(const float* in, float* out, int64_t n, hipStream_t s);

}  // namespace tn
