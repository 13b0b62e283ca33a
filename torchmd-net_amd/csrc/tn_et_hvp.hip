// Kernels of the analytic second-order pass of the Equivariant Transformer (gfx950): index arithmetic around the bodies of
// tn_et_hvp_math.h (one logical thread = one (atom, channel), (atom, head), pair-row element or pair), as tn_hvp.hip is around
// tn_hvp_math.h.  Specification oracle/et_second_order.py; the same bodies run on the host in tests/et_hvp_host_mirror.py.
// Training path, written to be exact and simple.  The attention sweeps: a block per atom, a thread per channel, the head sums by DPP /
// shuffle over the head's hd lanes (the bodies' lane-group form with nc = 1; the host harness runs the same statements with a whole
// head per thread): coalesced rows, accumulators in registers, every output written once.
#include "tn_hvp.h"

#include "tn_common.h"
#include "tn_et_hvp_math.h"

namespace tn {
namespace hvp {

namespace {
constexpr int TB = 256;
inline dim3 grid_for(int64_t n) { return dim3((unsigned)((n + TB - 1) / TB)); }
struct EtRedHead {  // sum over the hd = 2^k <= 64 lanes of one head (all of them active together: a block = one atom, F % hd == 0)
  int hd;
  __device__ float operator()(float v) const {
    v = row_sum(v, hd < 16 ? hd : 16);
    if (hd >= 32) v += __shfl_xor(v, 16, 64);
    if (hd >= 64) v += __shfl_xor(v, 32, 64);
    return v;
  }
};
inline int et_block(int F) { return ((F + 63) / 64) * 64; }
#define IDX2(ROWS, W)                                                   \
  const int64_t idx_ = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  \
  if (idx_ >= (int64_t)(ROWS) * (W)) return;                            \
  const int n = (int)(idx_ / (W)), f = (int)(idx_ - (int64_t)n * (W));

__global__ __launch_bounds__(TB) void k_rowscale_dual(int64_t total, int W, const float* __restrict__ e, const float* __restrict__ e_t,
                                                      const float* __restrict__ C, const float* __restrict__ C_t, float* __restrict__ o,
                                                      float* __restrict__ o_t) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) rowscale_dual(i, W, e, e_t, C, C_t, o, o_t);
}
__global__ __launch_bounds__(TB) void k_et_nbr_embed_dual(Graph g, int N, int F, int P, const int64_t* __restrict__ z,
                                                          const float* __restrict__ emb, const float* __restrict__ embN,
                                                          const float* __restrict__ Wn, const float* __restrict__ Wn_t,
                                                          float* __restrict__ xcat, float* __restrict__ xcat_t) {
  IDX2(N, F)
  et_nbr_embed_dual(n, f, F, P, g.rowptr, g.col, g.epair, g.esign, z, emb, embN, Wn, Wn_t, xcat, xcat_t);
}
__global__ __launch_bounds__(TB) void k_et_embed_dual(int N, int F, const int64_t* __restrict__ z, const float* __restrict__ emb,
                                                      float* __restrict__ x, float* __restrict__ x_t) {
  IDX2(N, F)
  et_embed_dual(n, f, F, z, emb, x, x_t);
}
__global__ __launch_bounds__(512) void k_et_attn_fwd_dual(Graph g, int N, int H, EtAttn A_, float* __restrict__ xagg,
                                                         float* __restrict__ xagg_t, float* __restrict__ vagg, float* __restrict__ vagg_t) {
  (void)H;
  const int t = blockIdx.x, c = threadIdx.x;  // a block per target atom, a thread per channel
  if (t >= N || c >= A_.F) return;
  et_attn_fwd_dual_g<1>(t, c, 1, EtRedHead{A_.hd}, A_, g.rowptr, g.col, g.epair, g.esign, xagg, xagg_t, vagg, vagg_t);
}
__global__ __launch_bounds__(TB) void k_et_update_dual(int N, int F, const float* __restrict__ x, const float* __restrict__ x_t,
                                                       const float* __restrict__ vec, const float* __restrict__ vec_t,
                                                       const float* __restrict__ vp, const float* __restrict__ vp_t,
                                                       const float* __restrict__ o, const float* __restrict__ o_t,
                                                       const float* __restrict__ vagg, const float* __restrict__ vagg_t,
                                                       float* __restrict__ xn, float* __restrict__ xn_t, float* __restrict__ vecn,
                                                       float* __restrict__ vecn_t, float* __restrict__ vdot, float* __restrict__ vdot_t) {
  IDX2(N, F)
  et_update_dual(n, f, F, x, x_t, vec, vec_t, vp, vp_t, o, o_t, vagg, vagg_t, xn, xn_t, vecn, vecn_t, vdot, vdot_t);
}
__global__ __launch_bounds__(TB) void k_et_cat_norm_dual(int N, int W, const float* __restrict__ x, const float* __restrict__ x_t, int Fx,
                                                         const float* __restrict__ u, const float* __restrict__ u_t, int ldu, int Fn, int ldo,
                                                         float* __restrict__ out, float* __restrict__ out_t) {
  IDX2(N, W)
  et_cat_norm_dual(n, f, x, x_t, Fx, u, u_t, ldu, Fn, ldo, out, out_t);
}
__global__ __launch_bounds__(TB) void k_et_head_mid_dual(int N, int F2, const float* __restrict__ y, const float* __restrict__ y_t,
                                                         const float* __restrict__ u2, const float* __restrict__ u2_t, int ldu,
                                                         float* __restrict__ hcat2, float* __restrict__ hcat2_t, float* __restrict__ vq,
                                                         float* __restrict__ vq_t) {
  IDX2(N, F2)
  et_head_mid_dual(n, f, F2, y, y_t, u2, u2_t, ldu, hcat2, hcat2_t, vq, vq_t);
}
__global__ __launch_bounds__(TB) void k_et_norm_bwd_dual(int N, int W, const float* __restrict__ g_n, const float* __restrict__ g_n_t,
                                                         int ldg, const float* __restrict__ u, const float* __restrict__ u_t, int ldu,
                                                         float* __restrict__ g_u, float* __restrict__ g_u_t, int ldgu) {
  IDX2(N, W)
  et_norm_bwd_dual(n, f, g_n, g_n_t, ldg, u, u_t, ldu, g_u, g_u_t, ldgu);
}
__global__ __launch_bounds__(TB) void k_et_head_mid_bwd_dual(int N, int F2, const float* __restrict__ y, const float* __restrict__ y_t,
                                                             const float* __restrict__ u2, const float* __restrict__ u2_t, int ldu,
                                                             const float* __restrict__ g_h2, const float* __restrict__ g_h2_t,
                                                             const float* __restrict__ g_vq, const float* __restrict__ g_vq_t,
                                                             float* __restrict__ g_y, float* __restrict__ g_y_t, float* __restrict__ g_u2,
                                                             float* __restrict__ g_u2_t, int ldgu) {
  IDX2(N, F2)
  et_head_mid_bwd_dual(n, f, F2, y, y_t, u2, u2_t, ldu, g_h2, g_h2_t, g_vq, g_vq_t, g_y, g_y_t, g_u2, g_u2_t, ldgu);
}
__global__ __launch_bounds__(TB) void k_et_update_bwd_dual(int N, int F, const float* __restrict__ g_x, const float* __restrict__ g_x_t,
                                                           const float* __restrict__ g_vec, const float* __restrict__ g_vec_t,
                                                           const float* __restrict__ vp, const float* __restrict__ vp_t,
                                                           const float* __restrict__ o, const float* __restrict__ o_t,
                                                           const float* __restrict__ vdot, const float* __restrict__ vdot_t,
                                                           float* __restrict__ g_o, float* __restrict__ g_o_t, float* __restrict__ g_vp,
                                                           float* __restrict__ g_vp_t) {
  IDX2(N, F)
  et_update_bwd_dual(n, f, F, g_x, g_x_t, g_vec, g_vec_t, vp, vp_t, o, o_t, vdot, vdot_t, g_o, g_o_t, g_vp, g_vp_t);
}
__global__ __launch_bounds__(512) void k_et_attn_bwd_tgt_dual(Graph g, int N, int H, EtAttn A_, const float* __restrict__ g_xagg,
                                                             const float* __restrict__ g_xagg_t, const float* __restrict__ g_vagg,
                                                             const float* __restrict__ g_vagg_t, float* __restrict__ g_qkv,
                                                             float* __restrict__ g_qkv_t, float* __restrict__ gq, float* __restrict__ gq_t,
                                                             int64_t dir_stride, float* __restrict__ selfq, float* __restrict__ selfq_t,
                                                             float* __restrict__ slots, float* __restrict__ slots_t, int64_t slot_dir_stride) {
  const int t = blockIdx.x, c = threadIdx.x;
  if (t >= N || c >= A_.F) return;
  et_attn_bwd_tgt_dual_g<1>(t, c / A_.hd, c, 1, (c % A_.hd) == 0, EtRedHead{A_.hd}, H, A_, g.rowptr, g.col, g.epair, g.esign, g_xagg, g_xagg_t,
                            g_vagg, g_vagg_t, g_qkv, g_qkv_t, gq, gq_t, dir_stride, selfq, selfq_t, slots, slots_t, slot_dir_stride);
}
__global__ __launch_bounds__(512) void k_et_attn_bwd_src_dual(Graph g, int N, int H, EtAttn A_, const float* __restrict__ g_xagg,
                                                             const float* __restrict__ g_xagg_t, const float* __restrict__ g_vagg,
                                                             const float* __restrict__ g_vagg_t, float* __restrict__ g_qkv,
                                                             float* __restrict__ g_qkv_t, float* __restrict__ g_vec_in,
                                                             float* __restrict__ g_vec_in_t) {
  (void)H;
  const int s_ = blockIdx.x, c = threadIdx.x;
  if (s_ >= N || c >= A_.F) return;
  et_attn_bwd_src_dual_g<1>(s_, c, 1, EtRedHead{A_.hd}, A_, g.rowptr, g.col, g.epair, g.esign, g_xagg, g_xagg_t, g_vagg, g_vagg_t, g_qkv, g_qkv_t,
                            g_vec_in, g_vec_in_t);
}
__global__ __launch_bounds__(TB) void k_et_filter_gpre_dual(int64_t total, int P, int Wd, const float* __restrict__ gq,
                                                            const float* __restrict__ gq_t, int64_t dir_stride,
                                                            const float* __restrict__ self_g, const float* __restrict__ self_g_t,
                                                            const float* __restrict__ ekv, const float* __restrict__ ekv_t,
                                                            float* __restrict__ g_e, float* __restrict__ g_e_t) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) et_filter_gpre_dual(i, P, Wd, gq, gq_t, dir_stride, self_g, self_g_t, ekv, ekv_t, g_e, g_e_t);
}
__global__ __launch_bounds__(TB) void k_et_nbr_bwd_dual(Graph g, int N, int F, int P, const int64_t* __restrict__ z,
                                                        const float* __restrict__ embN, const float* __restrict__ Wn,
                                                        const float* __restrict__ Wn_t, const float* __restrict__ g_xcat,
                                                        const float* __restrict__ g_xcat_t, float* __restrict__ gq, float* __restrict__ gq_t,
                                                        int64_t dir_stride, float* __restrict__ gZ_t) {
  IDX2(N, F)
  et_nbr_bwd_dual(n, f, F, P, g.rowptr, g.col, g.epair, g.esign, z, embN, Wn, Wn_t, g_xcat, g_xcat_t, gq, gq_t, dir_stride, gZ_t);
}
__global__ __launch_bounds__(TB) void k_et_nbr_pair_dual(int64_t total, int F, const float* __restrict__ gq, const float* __restrict__ gq_t,
                                                         int64_t dir_stride, const float* __restrict__ C, const float* __restrict__ C_t,
                                                         float* __restrict__ g_Wn, float* __restrict__ g_Wn_t, float* __restrict__ g_en,
                                                         float* __restrict__ g_en_t) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) et_nbr_pair_dual(i, F, gq, gq_t, dir_stride, C, C_t, g_Wn, g_Wn_t, g_en, g_en_t);
}
__global__ __launch_bounds__(TB) void k_pair_rowdot2(int rows, int W, const float* __restrict__ x, const float* __restrict__ x_t,
                                                     const float* __restrict__ y, const float* __restrict__ y_t, int accumulate,
                                                     float* __restrict__ out, float* __restrict__ out_t) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < rows) pair_rowdot2(p, W, x, x_t, y, y_t, accumulate, out, out_t);
}
__global__ __launch_bounds__(TB) void k_et_pair_slots_dual(int P, int H, int nsets, const float* __restrict__ slots,
                                                           const float* __restrict__ slots_t, int64_t set_stride, int64_t slot_dir_stride,
                                                           float* __restrict__ g_cut, float* __restrict__ g_cut_t, float* __restrict__ g_rh,
                                                           float* __restrict__ g_rh_t) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < P) et_pair_slots_dual(p, H, nsets, slots, slots_t, set_stride, slot_dir_stride, g_cut, g_cut_t, g_rh, g_rh_t);
}
__global__ __launch_bounds__(TB) void k_et_geom_dual(Graph g, int P, const float* __restrict__ d_t, const float* __restrict__ rhat_t,
                                                     const float* __restrict__ dC, const float* __restrict__ d2C,
                                                     const float* __restrict__ g_cut, const float* __restrict__ g_cut_t,
                                                     const float* __restrict__ g_dphi, const float* __restrict__ g_dphi_t,
                                                     const float* __restrict__ g_rh, const float* __restrict__ g_rh_t,
                                                     float* __restrict__ gdel, float* __restrict__ gdel_t) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < P) et_geom_dual(p, g.pd, g.prhat, d_t, rhat_t, dC, d2C, g_cut, g_cut_t, g_dphi, g_dphi_t, g_rh, g_rh_t, gdel, gdel_t);
}
__global__ __launch_bounds__(TB) void k_add2(int64_t n, const float* __restrict__ a, float* __restrict__ o) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) o[i] += a[i];
}
}  // namespace

#define LAUNCH(kernel, count, ...)                                                            \
  do {                                                                                        \
    if ((count) > 0) hipLaunchKernelGGL(kernel, grid_for(count), dim3(TB), 0, s, __VA_ARGS__); \
  } while (0)

void launch_rowscale_dual(int64_t rows, int W, const float* e, const float* e_t, const float* C, const float* C_t, float* o, float* o_t,
                          hipStream_t s) {
  LAUNCH(k_rowscale_dual, rows * W, rows * W, W, e, e_t, C, C_t, o, o_t);
}
void launch_et_nbr_embed_dual(const Graph& g, int N, int F, int P, const int64_t* z, const float* emb, const float* embN, const float* Wn,
                              const float* Wn_t, float* xcat, float* xcat_t, hipStream_t s) {
  LAUNCH(k_et_nbr_embed_dual, (int64_t)N * F, g, N, F, P, z, emb, embN, Wn, Wn_t, xcat, xcat_t);
}
void launch_et_embed_dual(int N, int F, const int64_t* z, const float* emb, float* x, float* x_t, hipStream_t s) {
  LAUNCH(k_et_embed_dual, (int64_t)N * F, N, F, z, emb, x, x_t);
}
void launch_et_attn_fwd_dual(const Graph& g, int N, int H, const EtAttn& A_, float* xagg, float* xagg_t, float* vagg, float* vagg_t,
                             hipStream_t s) {
  if (N > 0) hipLaunchKernelGGL(k_et_attn_fwd_dual, dim3(N), dim3(et_block(A_.F)), 0, s, g, N, H, A_, xagg, xagg_t, vagg, vagg_t);
}
void launch_et_update_dual(int N, int F, const float* x, const float* x_t, const float* vec, const float* vec_t, const float* vp,
                           const float* vp_t, const float* o, const float* o_t, const float* vagg, const float* vagg_t, float* xn,
                           float* xn_t, float* vecn, float* vecn_t, float* vdot, float* vdot_t, hipStream_t s) {
  LAUNCH(k_et_update_dual, (int64_t)N * F, N, F, x, x_t, vec, vec_t, vp, vp_t, o, o_t, vagg, vagg_t, xn, xn_t, vecn, vecn_t, vdot, vdot_t);
}
void launch_et_cat_norm_dual(int N, int W, const float* x, const float* x_t, int Fx, const float* u, const float* u_t, int ldu, int Fn,
                             int ldo, float* out, float* out_t, hipStream_t s) {
  LAUNCH(k_et_cat_norm_dual, (int64_t)N * W, N, W, x, x_t, Fx, u, u_t, ldu, Fn, ldo, out, out_t);
}
void launch_et_head_mid_dual(int N, int F2, const float* y, const float* y_t, const float* u2, const float* u2_t, int ldu, float* hcat2,
                             float* hcat2_t, float* vq, float* vq_t, hipStream_t s) {
  LAUNCH(k_et_head_mid_dual, (int64_t)N * F2, N, F2, y, y_t, u2, u2_t, ldu, hcat2, hcat2_t, vq, vq_t);
}
void launch_et_norm_bwd_dual(int N, int W, const float* g_n, const float* g_n_t, int ldg, const float* u, const float* u_t, int ldu,
                             float* g_u, float* g_u_t, int ldgu, hipStream_t s) {
  LAUNCH(k_et_norm_bwd_dual, (int64_t)N * W, N, W, g_n, g_n_t, ldg, u, u_t, ldu, g_u, g_u_t, ldgu);
}
void launch_et_head_mid_bwd_dual(int N, int F2, const float* y, const float* y_t, const float* u2, const float* u2_t, int ldu,
                                 const float* g_h2, const float* g_h2_t, const float* g_vq, const float* g_vq_t, float* g_y, float* g_y_t,
                                 float* g_u2, float* g_u2_t, int ldgu, hipStream_t s) {
  LAUNCH(k_et_head_mid_bwd_dual, (int64_t)N * F2, N, F2, y, y_t, u2, u2_t, ldu, g_h2, g_h2_t, g_vq, g_vq_t, g_y, g_y_t, g_u2, g_u2_t, ldgu);
}
void launch_et_update_bwd_dual(int N, int F, const float* g_x, const float* g_x_t, const float* g_vec, const float* g_vec_t, const float* vp,
                               const float* vp_t, const float* o, const float* o_t, const float* vdot, const float* vdot_t, float* g_o,
                               float* g_o_t, float* g_vp, float* g_vp_t, hipStream_t s) {
  LAUNCH(k_et_update_bwd_dual, (int64_t)N * F, N, F, g_x, g_x_t, g_vec, g_vec_t, vp, vp_t, o, o_t, vdot, vdot_t, g_o, g_o_t, g_vp, g_vp_t);
}
void launch_et_attn_bwd_dual(const Graph& g, int N, int H, const EtAttn& A_, const float* g_xagg, const float* g_xagg_t, const float* g_vagg,
                             const float* g_vagg_t, float* g_qkv, float* g_qkv_t, float* g_vec_in, float* g_vec_in_t, float* gq, float* gq_t,
                             int64_t dir_stride, float* selfq, float* selfq_t, float* slots, float* slots_t, int64_t slot_dir_stride,
                             hipStream_t s) {
  if (N <= 0) return;
  hipLaunchKernelGGL(k_et_attn_bwd_tgt_dual, dim3(N), dim3(et_block(A_.F)), 0, s, g, N, H, A_, g_xagg, g_xagg_t, g_vagg, g_vagg_t, g_qkv,
                     g_qkv_t, gq, gq_t, dir_stride, selfq, selfq_t, slots, slots_t, slot_dir_stride);
  hipLaunchKernelGGL(k_et_attn_bwd_src_dual, dim3(N), dim3(et_block(A_.F)), 0, s, g, N, H, A_, g_xagg, g_xagg_t, g_vagg, g_vagg_t, g_qkv,
                     g_qkv_t, g_vec_in, g_vec_in_t);
}
void launch_et_filter_gpre_dual(int P, int Wd, const float* gq, const float* gq_t, int64_t dir_stride, const float* self_g,
                                const float* self_g_t, const float* ekv, const float* ekv_t, float* g_e, float* g_e_t, hipStream_t s) {
  LAUNCH(k_et_filter_gpre_dual, (int64_t)(P + 1) * Wd, (int64_t)(P + 1) * Wd, P, Wd, gq, gq_t, dir_stride, self_g, self_g_t, ekv, ekv_t, g_e,
         g_e_t);
}
void launch_et_nbr_bwd_dual(const Graph& g, int N, int F, int P, const int64_t* z, const float* embN, const float* Wn, const float* Wn_t,
                            const float* g_xcat, const float* g_xcat_t, float* gq, float* gq_t, int64_t dir_stride, float* gZ_t,
                            hipStream_t s) {
  LAUNCH(k_et_nbr_bwd_dual, (int64_t)N * F, g, N, F, P, z, embN, Wn, Wn_t, g_xcat, g_xcat_t, gq, gq_t, dir_stride, gZ_t);
}
void launch_et_nbr_pair_dual(int P, int F, const float* gq, const float* gq_t, int64_t dir_stride, const float* C, const float* C_t,
                             float* g_Wn, float* g_Wn_t, float* g_en, float* g_en_t, hipStream_t s) {
  LAUNCH(k_et_nbr_pair_dual, (int64_t)P * F, (int64_t)P * F, F, gq, gq_t, dir_stride, C, C_t, g_Wn, g_Wn_t, g_en, g_en_t);
}
void launch_pair_rowdot2(int rows, int W, const float* x, const float* x_t, const float* y, const float* y_t, bool accumulate, float* out,
                         float* out_t, hipStream_t s) {
  LAUNCH(k_pair_rowdot2, (int64_t)rows, rows, W, x, x_t, y, y_t, accumulate ? 1 : 0, out, out_t);
}
void launch_et_pair_slots_dual(int P, int H, int nsets, const float* slots, const float* slots_t, int64_t set_stride,
                               int64_t slot_dir_stride, float* g_cut, float* g_cut_t, float* g_rh, float* g_rh_t, hipStream_t s) {
  LAUNCH(k_et_pair_slots_dual, (int64_t)P, P, H, nsets, slots, slots_t, set_stride, slot_dir_stride, g_cut, g_cut_t, g_rh, g_rh_t);
}
void launch_et_geom_dual(const Graph& g, int P, const float* d_t, const float* rhat_t, const float* dC, const float* d2C, const float* g_cut,
                         const float* g_cut_t, const float* g_dphi, const float* g_dphi_t, const float* g_rh, const float* g_rh_t, float* gdel,
                         float* gdel_t, hipStream_t s) {
  LAUNCH(k_et_geom_dual, (int64_t)P, g, P, d_t, rhat_t, dC, d2C, g_cut, g_cut_t, g_dphi, g_dphi_t, g_rh, g_rh_t, gdel, gdel_t);
}
void launch_add2(int64_t n, const float* a, float* o, hipStream_t s) { LAUNCH(k_add2, n, n, a, o); }

}  // namespace hvp
}  // namespace tn
