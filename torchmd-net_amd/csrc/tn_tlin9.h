// Nine-component tensor linear with the per-atom 3x3 algebra of its neighbours in the prologue and epilogue (tn_tlin9.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tn {

// what happens to the [atom, 9, k] input while it is staged (per atom and input channel, all nine components at hand)
enum Tl9Pro : int {
  TL9_PRO_PLAIN = 0,   // A
  TL9_PRO_NORM,        // A / (||A||^2 + 1)                               (k_norm_x; reference tensornet.py:745)
  TL9_PRO_UPDBWD,      // compose^T(Gf + kappa (Gf dX^T + dX^T Gf)), Gf = dec^T(A), dX = compose(A2)     (k_update_bwd)
};
// what happens to the [atom, 9, n] product before it is stored (per atom and output channel, all nine components at hand)
enum Tl9Epi : int {
  TL9_EPI_PLAIN = 0,      // C = acc
  TL9_EPI_MULGATE,        // o1 = acc ; C = acc * gates[atom, type(c), n]                       (embedding, tensornet.py:595-617)
  TL9_EPI_UPDATE,         // C = acc (= dX) ; o1 = X_hat + dX + kappa dX.dX with X_hat = e0 / (||e0||^2 + 1) ; o2 = invariants of o1
  TL9_EPI_NORMBWD,        // C = normalisation adjoint of X = e0 with incoming e1 + acc                            (k_norm_bwd<0>)
  TL9_EPI_NORMBWD_GATE,   // the same, then the embedding gate adjoint: C = g_UX, o1 = g_a2 (e2 = UX, e3 = gates, e4 = a2)
  TL9_EPI_EMBBWD,         // o1[atom, 10, n] = embedding atom adjoint of acc + dquad(e0) e1[atom, n]               (k_embed_bwd_atom)
};

struct Tl9Args {
  const float* A;           // [N, 9, F]
  const float* A2;          // [N, 9, F]   second prologue operand (TL9_PRO_UPDBWD: dX)
  const uint16_t* Wfm[3];   // fragment-major split-bf16 images of the I / A / S weights (split_weight_fm)
  float* C;                 // [N, 9, F]
  const float* e0;
  const float* e1;
  const float* e2;
  const float* e3;
  const float* e4;
  float* o1;
  float* o2;
  const float* kap;         // [N] per-atom charge factor or null (= 1)
  int N, F, o3, want_feat;
};

bool tlin9_ok(int N, int F, bool any_rounds = false);  // any_rounds: accept a mostly empty last round of tiles (see the definition)
size_t split_weight_fm_elems(int64_t n, int64_t k);  // uint16 elements of one image
void launch_split_weight_fm(const float* W_dev, int64_t n, int64_t k, uint16_t* out_dev, hipStream_t s);
int launch_tlin9(const Tl9Args& a, int pro, int epi, hipStream_t s);

}  // namespace tn
