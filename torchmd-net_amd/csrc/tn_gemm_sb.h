// Shared pieces of the split-bf16 MFMA GEMM kernels (tn_gemm_sb.hip: value + tangent on pair rows,
// tn_gemm_sb1.hip: single product with the node-side epilogues).  See tn_gemm_sb.hip for the arithmetic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tn {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// two fp32 -> three packed bf16 pairs (round to nearest even at every level; low half = first element)
__device__ __forceinline__ void split2(float x0, float x1, uint32_t& h, uint32_t& m, uint32_t& l) {
  union { bf16x2 v; uint32_t u; } c;
  c.v = __builtin_convertvector((floatx2){x0, x1}, bf16x2);
  h = c.u;
  const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
  c.v = __builtin_convertvector((floatx2){r0, r1}, bf16x2);
  m = c.u;
  const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
  c.v = __builtin_convertvector((floatx2){s0, s1}, bf16x2);
  l = c.u;
}
__device__ __forceinline__ void split8(const float4& u, const float4& v, uint4& h, uint4& m, uint4& l) {
  split2(u.x, u.y, h.x, m.x, l.x);
  split2(u.z, u.w, h.y, m.y, l.y);
  split2(v.x, v.y, h.z, m.z, l.z);
  split2(v.z, v.w, h.w, m.w, l.w);
}

constexpr int SB2_PLANE = 128 * 32;  // one bf16 plane of a [128 rows][16 k] chunk: 4096 bytes, no padding

// LDS byte offset of the 16-byte piece (row r, k-half h) inside a plane: 16 * (2 r + (h ^ f(r))), f(r) = ((r+4)>>3) & 1.
// ds_read_b128 serves the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32) in one LDS cycle each over 64
// banks: with this f every group touches 16 distinct 16-byte slots; ds_write_b128 (8 contiguous lanes) is
// conflict-free as well.  f(r) = f(r + 32).
__device__ __forceinline__ int sb_piece(int r, int h) { return (2 * r + (h ^ (((r + 4) >> 3) & 1))) * 16; }

}  // namespace tn
