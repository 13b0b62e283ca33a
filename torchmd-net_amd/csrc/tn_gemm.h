// fp32 MFMA GEMM with fused epilogues for gfx950 (v_mfma_f32_32x32x2_f32: exact fp32 fmaf chain).
//
//   C[g][m, n] = epilogue( sum_k A[g][m, k] * W[g][n, k] )          "NT": W is [N_out, K] row-major
//
// Used for every dense contraction on the path: edge MLPs (reference tensornet.py:738-743),
// distance projections (:558-560), scalar MLPs (:590-593), the 9-component tensor linears
// (:595-617, :752-754, :808-810; grouped launch, one group per irreducible component), readout
// linear (:398) and output MLP (models/utils.py:552-580), plus all of their input-gradient
// (transposed-weight) counterparts in the hand-written reverse pass.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tn {

enum GemmFlags : int {
  GEMM_ACT_SILU = 1,       // out = silu(pre)
  GEMM_MUL_AUX = 2,        // out *= aux[m, n]
  GEMM_MUL_DSILU_AUX = 4,  // out *= silu'(aux[m, n])
  GEMM_ACCUM = 8,          // out += C_old[m, n]
  GEMM_ROWSCALE = 16,      // out *= rowscale[m]
};

constexpr int GEMM_MAX_GROUPS = 9;

struct GemmArgs {
  const float* A;
  const float* W[GEMM_MAX_GROUPS];
  float* C;
  const float* bias[GEMM_MAX_GROUPS];  // [N] or null
  float* pre;                          // optional: pre-activation (after bias) saved here
  const float* aux;                    // optional operand of MUL_AUX / MUL_DSILU_AUX
  const float* rowscale;               // [M]
  int64_t lda, ldw, ldc, ldpre, ldaux;
  int a_off[GEMM_MAX_GROUPS], c_off[GEMM_MAX_GROUPS], pre_off[GEMM_MAX_GROUPS], aux_off[GEMM_MAX_GROUPS];
  int M, N, K, groups, flags;
  // optional device-side row count: rows = min(M, *m_dev + m_add).  The grid is sized by M (the capacity) and
  // surplus blocks exit at once, so a schedule with data-dependent pair counts needs no host synchronisation.
  const int* m_dev;
  int m_add;
  // dual (value + tangent) launches, tn_gemm_dual.hip: second A operand, second output, second row vector
  const float* A2;
  float* C2;
  const float* rowscale2;
  // split-bf16 tile image of W (tn_gemm_sb.hip) or null
  const uint16_t* Wsb;
  // per-group split-bf16 tile images for launch_gemm (tn_gemm_sb1.hip) or null
  const uint16_t* Wsbg[GEMM_MAX_GROUPS];
};

// launches on `stream`; returns hipError_t as int
int launch_gemm(const GemmArgs& args, hipStream_t stream);
// value + d/dd tangent through one weight tile; kind 0: plain, 1: silu, 2: silu * C(d) (rowscale/rowscale2 = C, C')
int launch_gemm_dual(const GemmArgs& args, int kind, hipStream_t stream);
// split-bf16 MFMA variant (tn_gemm_sb.hip): taken by launch_gemm_dual when gemm_dual_sb_ok(args)
bool gemm_dual_sb_ok(const GemmArgs& args);
int launch_gemm_dual_sb(const GemmArgs& args, int kind, hipStream_t stream);
// single-product split-bf16 variant with the node-side epilogues (tn_gemm_sb1.hip): taken by launch_gemm when ok
bool gemm_sb1_ok(const GemmArgs& args);
int launch_gemm_sb1(const GemmArgs& args, hipStream_t stream);
size_t split_weight_elems(int64_t N, int64_t K);                                   // uint16 elements of the tile image
void split_weight_tiles(const float* W_host, int64_t N, int64_t K, uint16_t* out_host);
void launch_split_weight_tiles(const float* W_dev, int64_t N, int64_t K, uint16_t* out_dev, hipStream_t s);  // same image, on the device

}  // namespace tn
