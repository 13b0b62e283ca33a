// Small systems (one molecule in MD, a rank's shard under strong scaling): the node side of the TensorNet step as FOUR kinds of
// per-atom kernels instead of ~35 launches (tn_small.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tn_kernels.h"

namespace tn {

// weights of one interaction layer; the forward products read the transposed copies ([in][out]: coalesced over the output
// channel a thread owns), the reverse products the plain ones
struct SmallLayerW {
  const float* V[6];
  const float* VT[6];
};

struct SmallEmbedArgs {  // embedding scatter -> LayerNorm -> gate MLP -> tensor linear x gates -> X0 (-> normalise -> linear -> Pn[0])
  Graph g;
  int N, F, L;
  const int64_t* z;
  const float *Utab, *Vtab, *Q, *C;
  const float *ln0_w, *ln0_b, *L1T, *bL1, *L2T, *bL2;
  const float* UeT[3];
  const float* V0T[3];  // layer 0's first linear (L > 0)
  float *u0, *xh0, *rstd0, *a1, *a2, *gates, *UX, *X0, *Pn0;
};

struct SmallLayerArgs {  // message sweep -> group product -> linear -> update (-> next layer's normalise + linear | readout + reverse start)
  Graph g;
  int N, F, H, o3, want_forces;
  const int64_t* z;
  const float* kap;  // per-atom charge factor or null
  const float* atom_w;  // weights of the energy sum or null
  const int* perm;  // internal -> caller's atom index (cell order) or null
  const float *w, *Pn, *X;
  float *Mi, *D, *Xn;
  const float* VbT[3];  // this layer's second linear, transposed
  // not the last layer
  const float* VnT[3];  // next layer's first linear, transposed
  float* Pn_next;
  // last layer: readout, head, and the start of the reverse pass down to this layer's message adjoint inputs
  const float *lnr_w, *lnr_b, *LinT, *bLin, *O1T, *bO1, *O2, *bO2, *atomref, *Lin, *O1;
  float std_;
  const float* Vb[3];  // this layer's second linear, plain (reverse)
  float *xhr, *rstdr, *al, *x, *ea, *G, *gMi, *gPn;
};

struct SmallRevArgs {  // adjoint sweep (+ distance-gradient slots) -> linear^T -> normalisation adjoint -> (update adjoint + linear^T + group
                       // product adjoint of the layer below | gate adjoint + gate MLP adjoint + LayerNorm adjoint + embedding adjoint)
  Graph g;
  int N, F, B, o3, first;  // first: this is layer 0
  const float* kap;
  const float *w, *dw, *gMi_in, *Pn, *X;
  float *gPn, *G, *slots;
  int64_t slot_stride;
  const float* Va[3];  // this layer's first linear, plain
  // layer below (first == 0)
  const float* Vb_prev[3];
  const float *D_prev, *Pn_prev, *Mi_prev;
  float* gMi_out;
  // embedding (first == 1)
  const float *UX, *gates, *a2, *a1, *L2, *L1, *xh0, *rstd0, *ln0_w, *u0;
  const float* Ue[3];
  float* gA;
  // per-molecule energies (done by the first B blocks of the top layer's launch; null otherwise)
  const float* ea;
  const int64_t* batch;
  float mean;
  float* energy;
};

bool small_fused_ok(int N, int F, int H, int L);
// four atoms per block (tn_mid.hip): same argument structs, same cut of the step; systems of 513 .. 1 024 atoms
bool mid_fused_ok(int N, int F, int H, int L);
void launch_mid_embed(const SmallEmbedArgs& a, hipStream_t s);
void launch_mid_layer(const SmallLayerArgs& a, bool last, hipStream_t s);
void launch_mid_rev(const SmallRevArgs& a, hipStream_t s);
void launch_small_embed(const SmallEmbedArgs& a, hipStream_t s);
void launch_small_layer(const SmallLayerArgs& a, bool last, hipStream_t s);
void launch_small_rev(const SmallRevArgs& a, hipStream_t s);

}  // namespace tn
