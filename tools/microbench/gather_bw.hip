// Micro-benchmark (developer tool): achievable HBM throughput of RANDOM PIECES of S bytes on gfx950 - the access pattern of the
// message sweeps (per-pair rows gathered in data-dependent order).  Each wave reads `pieces_per_wave` pieces of S bytes at
// pseudo-random S-aligned offsets of a large buffer with D independent 16-byte-per-lane loads in flight; a piece of S bytes is
// read by S/16 consecutive lanes (so one wave-load covers 1024/S pieces).  Output: GB/s per (S, D, waves per CU).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench/gather_bw tools/microbench/gather_bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4v __attribute__((ext_vector_type(4)));

template <int D>
__global__ __launch_bounds__(256) void k_gather(const f4v* __restrict__ buf, uint64_t n_pieces_total, int S16 /* 16-byte units per piece */,
                                              int pieces_per_lane_group, float* __restrict__ out, uint32_t seed) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane_in_piece = threadIdx.x % S16;           // which 16 bytes of the piece
  const uint32_t group = tid / S16;                       // lane group = one piece per load
  f4v acc = (f4v)(0.f);
  uint32_t x = seed ^ (group * 2654435761u);
  for (int it = 0; it < pieces_per_lane_group; it += D) {
    f4v v[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
      x = x * 1664525u + 1013904223u;                     // LCG: the next piece of this lane group
      const uint64_t piece = ((uint64_t)x * n_pieces_total) >> 32;
      v[d] = buf[piece * S16 + lane_in_piece];
    }
#pragma unroll
    for (int d = 0; d < D; ++d) acc += v[d];
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[tid] = acc[0];
}

template <int D>
static double run(const f4v* buf, uint64_t bytes, int S, int blocks, int pieces, float* out) {
  const int S16 = S / 16;
  const uint64_t n_pieces = bytes / S;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k_gather<D>), dim3(blocks), dim3(256), 0, 0, buf, n_pieces, S16, pieces, out, 1u);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k_gather<D>), dim3(blocks), dim3(256), 0, 0, buf, n_pieces, S16, pieces, out, 7u + r);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double moved = 3.0 * (double)blocks * 256 * 16.0 * pieces;  // every lane loads 16 B per piece of its group
  return moved / (ms * 1e-3) / 1e9;
}

int main() {
  const uint64_t bytes = 2ull << 30;  // 2 GiB: far beyond L2 (32 MB) and the Infinity Cache (256 MB)
  f4v* buf; float* out;
  hipMalloc(&buf, bytes); hipMalloc(&out, 64 << 20);
  hipMemset(buf, 0, bytes);
  int cus = 256;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  printf("random pieces of S bytes, 2 GiB buffer, %d CUs; GB/s\n", cus);
  printf("%6s %4s | %8s %8s %8s %8s\n", "S", "D", "2 w/SIMD", "4 w/SIMD", "6 w/SIMD", "8 w/SIMD");
  for (int S : {64, 128, 256, 512, 1024, 4096}) {
    for (int D : {2, 4, 8, 16}) {
      printf("%6d %4d |", S, D);
      for (int wps : {2, 4, 6, 8}) {
        const int blocks = cus * wps;  // 256 threads = 4 waves = one wave per SIMD per block
        const int pieces = 4096 / (wps * 1);
        double g = 0;
        switch (D) {
          case 2: g = run<2>(buf, bytes, S, blocks, pieces * 2, out); break;
          case 4: g = run<4>(buf, bytes, S, blocks, pieces * 2, out); break;
          case 8: g = run<8>(buf, bytes, S, blocks, pieces * 2, out); break;
          default: g = run<16>(buf, bytes, S, blocks, pieces * 2, out); break;
        }
        printf(" %8.0f", g);
      }
      printf("\n");
    }
  }
  return 0;
}
