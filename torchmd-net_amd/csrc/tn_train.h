// Kernels of the parameter-gradient pass (tn_train.hip); called from tn_api.hip when a training context is attached.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tn_kernels.h"

namespace tn {

// row r of a logical [R, *] operand lives at  base + (r / reps) * ld + (r % reps) * sub   (reps = 1: plain leading dimension;
// reps = 3 / 5: the (atom, component) rows of one irreducible type inside a [N, 9, F] tensor: ld = 9 F, sub = F)
struct RowMap {
  int64_t ld;
  int reps;
  int64_t sub;
};
inline RowMap rows_plain(int64_t ld) { return RowMap{ld, 1, 0}; }
inline RowMap rows_comp(int F, int ncomp) { return RowMap{(int64_t)9 * F, ncomp, (int64_t)F}; }

size_t train_part_floats(int R, int64_t out_elems);  // scratch of one launch_tn_gemm / launch_colsum
// out[n][k] (+)= sum_r A[r][n] * rowscale[r] * B[r][k]     (r_dev: optional device-side row count, min'ed with R)
void launch_tn_gemm_pair(hipStream_t s, const float* A1, const float* B1, const float* A2, const float* B2, RowMap ma, RowMap mb,
                         const int* r_dev, int R, int Nout, int Kin, float* out, bool accumulate, float* part);  // out (+)= A1^T B1 + A2^T B2
void launch_tn_gemm(hipStream_t s, const float* A, RowMap ma, const float* B, RowMap mb, const float* rowscale, const int* r_dev, int R,
                    int Nout, int Kin, float* out, bool accumulate, float* part);
// out[c] (+)= sum_r A[r][c] * (B ? B[r][c] : 1) * rowscale[r]
void launch_colsum(hipStream_t s, const float* A, RowMap ma, const float* B, RowMap mb, const float* rowscale, const int* r_dev, int R,
                   int ncol, float* out, bool accumulate, float* part);
void launch_train_seed(const float* ao, const float* gE, const int64_t* batch, int N, int H, float std, float* g_ao, float* head,
                       hipStream_t s);
void launch_onehot(const int64_t* z, int N, int Zp, float* oh, hipStream_t s);
void launch_transpose(const float* in, int rows, int cols, float* out, hipStream_t s);
void launch_train_gw(const Graph& g, int P, int F, const float* gMi, const float* Pn, const float* pre3, const float* C,
                     const float* self_gw, float* g_pre3, hipStream_t s);
void launch_train_embed(const Graph& g, int N, int F, const int64_t* z, const float* Utab, const float* Vtab, const float* Q, const float* C,
                        const float* gA, float* gq, int64_t dir_stride, float* selfq, float* gZu, float* gZv, hipStream_t s);

}  // namespace tn
