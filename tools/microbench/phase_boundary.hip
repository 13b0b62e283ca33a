// Micro-benchmark (developer tool): what ONE phase boundary of the single-system step costs as a kernel boundary
// inside a replayed hipGraph, and what it would cost as a grid barrier inside one persistent launch.  Geometry of the 64-atom step:
// 64 workgroups x 1024 threads (also 256 x 1024: the 1024-atom step).  Each phase reads a 512-byte record another workgroup wrote in
// the previous phase and writes its own (the dependency a neighbour sweep has), so the barrier carries a real release / acquire.
//   A: P kernels, one per phase, captured into a hipGraph, replayed
//   B: one kernel, P phases separated by a counter barrier (lane 0: release fence, agent-scope atomic add, relaxed sc1 polling with
//      s_sleep, acquire fence; __syncthreads around it) - the "barrier-counter" row of the guide's price list
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/phase_boundary tools/microbench/phase_boundary.hip ; output: us per phase for A and B
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                 \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) {                                                   \
      printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__);   \
      exit(1);                                                                \
    }                                                                         \
  } while (0)

__device__ __forceinline__ void phase_body(float* buf, int nwg, int phase) {
  // read the record of workgroup (b + phase + 1) mod nwg written in the previous phase, write this workgroup's record
  const int b = blockIdx.x, t = threadIdx.x;
  const float* src = buf + (size_t)((phase & 1) * nwg + (b + phase + 1) % nwg) * 128;
  float* dst = buf + (size_t)(((phase + 1) & 1) * nwg + b) * 128;
  if (t < 128) dst[t] = src[t] * 1.0001f + 1.0f;
}

__global__ __launch_bounds__(1024) void k_phase(float* buf, int nwg, int phase) { phase_body(buf, nwg, phase); }

__global__ __launch_bounds__(1024) void k_persistent(float* buf, int nwg, int phases, unsigned* counter, unsigned base) {
  for (int p = 0; p < phases; ++p) {
    phase_body(buf, nwg, p);
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = base + (unsigned)(p + 1) * (unsigned)nwg;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  }
}

int main() {
  const int P = 10, REP = 300;
  float* buf;
  unsigned* counter;
  CK(hipMalloc(&buf, 2 * 1024 * 128 * sizeof(float)));
  CK(hipMemset(buf, 0, 2 * 1024 * 128 * sizeof(float)));
  CK(hipMalloc(&counter, 64));
  CK(hipMemset(counter, 0, 64));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int nwg : {64, 256}) {
    // ---- A: P kernels in a graph
    hipGraph_t graph;
    hipGraphExec_t exec;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int p = 0; p < P; ++p) hipLaunchKernelGGL(k_phase, dim3(nwg), dim3(1024), 0, s, buf, nwg, p);
    CK(hipStreamEndCapture(s, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(exec, s));
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < REP; ++i) CK(hipGraphLaunch(exec, s));
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float msA;
    CK(hipEventElapsedTime(&msA, e0, e1));
    // ---- B: one persistent kernel with P - 1 ... P barriers (also replayed from a graph, to compare like with like)
    unsigned base = 0;
    // the counter is monotonic across launches: each launch adds P * nwg; `base` is passed per launch, so B runs as plain launches
    for (int i = 0; i < 20; ++i) {
      hipLaunchKernelGGL(k_persistent, dim3(nwg), dim3(1024), 0, s, buf, nwg, P, counter, base);
      base += (unsigned)(P * nwg);
    }
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < REP; ++i) {
      hipLaunchKernelGGL(k_persistent, dim3(nwg), dim3(1024), 0, s, buf, nwg, P, counter, base);
      base += (unsigned)(P * nwg);
    }
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float msB;
    CK(hipEventElapsedTime(&msB, e0, e1));
    // ---- C: the persistent kernel with ONE phase (its launch cost, to subtract)
    for (int i = 0; i < 20; ++i) {
      hipLaunchKernelGGL(k_persistent, dim3(nwg), dim3(1024), 0, s, buf, nwg, 1, counter, base);
      base += (unsigned)nwg;
    }
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < REP; ++i) {
      hipLaunchKernelGGL(k_persistent, dim3(nwg), dim3(1024), 0, s, buf, nwg, 1, counter, base);
      base += (unsigned)nwg;
    }
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float msC;
    CK(hipEventElapsedTime(&msC, e0, e1));
    CK(hipMemset(counter, 0, 64));
    printf("{\"workgroups\": %d, \"threads\": 1024, \"phases\": %d, \"graph_of_kernels_us_per_replay\": %.2f, \"us_per_kernel_phase\": %.2f, "
           "\"persistent_us_per_launch\": %.2f, \"persistent_one_phase_us\": %.2f, \"us_per_barrier_phase\": %.2f}\n",
           nwg, P, msA * 1e3 / REP, msA * 1e3 / REP / P, msB * 1e3 / REP, msC * 1e3 / REP, (msB - msC) * 1e3 / REP / (P - 1));
    CK(hipGraphExecDestroy(exec));
    CK(hipGraphDestroy(graph));
  }
  return 0;
}
