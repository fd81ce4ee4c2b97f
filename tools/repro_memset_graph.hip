// Standalone reproducer for the round-2 finding "a hipMemsetAsync node inside a replayed hipGraph loses its ordering
// against the neighbouring kernel nodes when the replay starts on an idle GPU" (ROCm 7.2, gfx950).
// Graph (stream capture, one stream): memset(buf, 0) -> k_accumulate(buf += 1 per thread, float atomics) -> k_copy(out = buf).
// Every replay must leave out[i] == ADDS.  Replays run (a) back to back, (b) after synchronise + an idle gap.
// Build: hipcc --offload-arch=gfx950 -O2 tools/repro_memset_graph.hip -o gpurun_out/repro_memset_graph
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <unistd.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)
constexpr int N = 1 << 20, ADDS = 64;
__global__ void k_accumulate(float* buf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;     // ADDS workgroup-rows add 1.0f into every element
  atomicAdd(&buf[i % N], 1.0f);
}
__global__ void k_copy(const float* buf, float* out) { const int i = blockIdx.x * blockDim.x + threadIdx.x; out[i] = buf[i]; }
__global__ void k_zero(float* buf) { const int i = blockIdx.x * blockDim.x + threadIdx.x; buf[i] = 0.f; }
static int run(bool memset_node, int gap_ms, int replays) {
  float *buf, *out; hipStream_t s; hipGraph_t g; hipGraphExec_t ge;
  CK(hipMalloc(&buf, N * 4)); CK(hipMalloc(&out, N * 4)); CK(hipStreamCreate(&s));
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  if (memset_node) CK(hipMemsetAsync(buf, 0, N * 4, s)); else k_zero<<<N / 256, 256, 0, s>>>(buf);
  k_accumulate<<<N / 256 * ADDS, 256, 0, s>>>(buf);
  k_copy<<<N / 256, 256, 0, s>>>(buf, out);
  CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  std::vector<float> h(N); int bad_replays = 0; double worst = 0;
  for (int r = 0; r < replays; ++r) {
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    CK(hipMemcpy(h.data(), out, N * 4, hipMemcpyDeviceToHost));
    long bad = 0; for (int i = 0; i < N; ++i) { if (h[i] != (float)ADDS) { ++bad; double d = h[i] - ADDS; if (d < 0) d = -d; if (d > worst) worst = d; } }
    bad_replays += bad != 0;
    if (gap_ms) usleep(gap_ms * 1000);
  }
  printf("%-12s idle gap %4d ms: %d of %d replays wrong (worst |error| %.0f)\n", memset_node ? "memset node" : "kernel node", gap_ms, bad_replays, replays, worst);
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipFree(buf)); CK(hipFree(out)); CK(hipStreamDestroy(s));
  return bad_replays ? 1 : 0;
}
// Variant closer to the captured train step: MANY small buffers of odd sizes, each zero-filled by its own memset node
// right before the kernel that accumulates into it, all in one long single-stream graph (the trainer's graph had ~450
// kernel nodes and ~50 memsets of 128 B .. 2 MB: GroupNorm workspaces, bias-gradient rows, the gradient images).
__global__ void k_acc_n(float* buf, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; atomicAdd(&buf[i % n], 1.0f); }
static int run_many(bool memset_node, int gap_ms, int replays) {
  constexpr int K = 64;
  hipStream_t s; hipGraph_t g; hipGraphExec_t ge; CK(hipStreamCreate(&s));
  float* bufs[K]; int ns[K];
  for (int k = 0; k < K; ++k) { ns[k] = 32 + (k * 7919) % 4096 * ((k % 5) ? 1 : 128); ns[k] = (ns[k] + 3) & ~3; CK(hipMalloc(&bufs[k], (size_t)ns[k] * 4)); }
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int k = 0; k < K; ++k) {
    if (memset_node) CK(hipMemsetAsync(bufs[k], 0, (size_t)ns[k] * 4, s));
    else k_zero<<<(ns[k] + 255) / 256, 256, 0, s>>>(bufs[k]);       // may write a few floats past n: the allocation granule covers it
    const int threads = ((ns[k] + 255) / 256) * 256 * 8;
    k_acc_n<<<threads / 256, 256, 0, s>>>(bufs[k], ns[k]);
  }
  CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  int bad_replays = 0; std::vector<float> h;
  for (int r = 0; r < replays; ++r) {
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    long bad = 0;
    for (int k = 0; k < K; ++k) {
      h.resize(ns[k]); CK(hipMemcpy(h.data(), bufs[k], (size_t)ns[k] * 4, hipMemcpyDeviceToHost));
      const int threads = ((ns[k] + 255) / 256) * 256 * 8;
      for (int i = 0; i < ns[k]; ++i) { const float want = (float)(threads / ns[k] + (i < threads % ns[k])); bad += h[i] != want; }
    }
    bad_replays += bad != 0;
    if (gap_ms) usleep(gap_ms * 1000);
  }
  printf("%-12s x%d buffers, idle gap %4d ms: %d of %d replays wrong\n", memset_node ? "memset nodes" : "kernel nodes", K, gap_ms, bad_replays, replays);
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); for (int k = 0; k < K; ++k) CK(hipFree(bufs[k])); CK(hipStreamDestroy(s));
  return bad_replays ? 1 : 0;
}
int main() {
  int rc = 0;
  for (int memset_node = 1; memset_node >= 0; --memset_node)
    for (int gap : {0, 500}) rc |= run_many(memset_node, gap, 10) << (memset_node ? 0 : 4);
  for (int memset_node = 1; memset_node >= 0; --memset_node)
    for (int gap : {0, 50, 500}) rc |= run(memset_node, gap, 12) << (memset_node ? 0 : 4);
  printf("result: memset-node graphs %s, kernel-node graphs %s\n", (rc & 1) ? "WRONG after some replay" : "always right",
         (rc & 16) ? "WRONG after some replay" : "always right");
  return 0;
}
