// common.h — shared device helpers for libmsgm_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/msgm_hip.h"

#define MSGM_WAVE 64

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

static inline int msgm_check_launch() {
  return hipGetLastError() == hipSuccess ? MSGM_OK : MSGM_E_LAUNCH;
}

// Zero-fill as a KERNEL node.  hipMemsetAsync must not be used on this path: inside a captured hipGraph (ROCm 7.2,
// gfx950) a memset node was observed to lose its ordering against the kernel nodes around it when the replay starts
// on an idle GPU (a replay right behind the previous one was fine) — float atomics then accumulated into a buffer that
// had not been cleared yet (tools/debug_race.py; 1e27-size garbage in the embedding gradients after a 0.5 s pause).
static __global__ void __launch_bounds__(256) k_msgm_zero_u32(uint32_t* __restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0u;
}
static inline int msgm_zero_async(void* p, size_t bytes, hipStream_t st) {
  const size_t n = bytes / 4;                              // every buffer on this path is a multiple of 4 bytes
  if (n == 0) return MSGM_OK;
  size_t g = (n + 255) / 256;
  if (g > 1024) g = 1024;
  hipLaunchKernelGGL(k_msgm_zero_u32, dim3((unsigned)g), dim3(256), 0, st, static_cast<uint32_t*>(p), n);
  return msgm_check_launch();
}

// ------------------------------------------------------------------ Philox
// Philox4x32-10 (Salmon et al. 2011).  key = seed, counter = (elem_lo,
// elem_hi, stream, offset_lo) ^ offset_hi folded into the key so that every
// (seed, offset, stream, element-quad) tuple gets its own 128-bit block.
struct Philox4 { uint32_t x, y, z, w; };

__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                  uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
    uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  return Philox4{c0, c1, c2, c3};
}

// Philox state = uint64[4]: {seed, offset, row_base, elem_base}.  The two bases place a rank's shard inside the GLOBAL
// index space of a data-parallel run (row_base = first global row of the shard, elem_base = row_base * n, a multiple of
// 4), so a sharded run draws exactly the numbers the single-GPU run draws for the same rows.  Streams indexed by ROW
// (the per-sample time draw and RNG_STREAM_ROWS) use row_base, every other stream is indexed by element.
enum { RNG_STREAM_T = 0, RNG_STREAM_EPS = 1, RNG_STREAM_V = 2, RNG_STREAM_DW = 3, RNG_STREAM_ROWS = 4, RNG_STREAM_USER = 16 };
__device__ __forceinline__ uint64_t msgm_rng_base(const uint64_t* rng, uint32_t stream) {
  return (stream == RNG_STREAM_T || stream == RNG_STREAM_ROWS) ? rng[2] : rng[3];
}

// One 128-bit block for element-quad `quad` (GLOBAL index) of draw-stream `stream`.
__device__ __forceinline__ Philox4 msgm_philox(const uint64_t* rng, uint64_t extra_offset, uint32_t stream, uint64_t quad) {
  uint64_t seed = rng[0], off = rng[1] + extra_offset;
  return philox4x32_10((uint32_t)quad, (uint32_t)(quad >> 32), stream, (uint32_t)off,
                       (uint32_t)seed, (uint32_t)(seed >> 32) ^ (uint32_t)(off >> 32));
}

__device__ __forceinline__ float u01(uint32_t x) {           // [0,1), 24 bits
  return (float)(x >> 8) * (1.0f / 16777216.0f);
}
__device__ __forceinline__ float u01_open(uint32_t x) {      // (0,1]
  return ((float)(x >> 8) + 1.0f) * (1.0f / 16777216.0f);
}
// Box–Muller: two normals from two words.
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& n0, float& n1) {
  float r = sqrtf(-2.0f * __logf(u01_open(a)));
  float s, c;
  __sincosf(6.283185307179586f * u01(b), &s, &c);
  n0 = r * c; n1 = r * s;
}
__device__ __forceinline__ f32x4 philox_uniform4(const uint64_t* rng, uint64_t extra, uint32_t stream, uint64_t quad) {
  Philox4 p = msgm_philox(rng, extra, stream, quad + (msgm_rng_base(rng, stream) >> 2));
  return f32x4{u01(p.x), u01(p.y), u01(p.z), u01(p.w)};
}
__device__ __forceinline__ f32x4 philox_normal4(const uint64_t* rng, uint64_t extra, uint32_t stream, uint64_t quad) {
  Philox4 p = msgm_philox(rng, extra, stream, quad + (msgm_rng_base(rng, stream) >> 2));
  f32x4 r;
  float a, b, c, d;
  box_muller(p.x, p.y, a, b);
  box_muller(p.z, p.w, c, d);
  r[0] = a; r[1] = b; r[2] = c; r[3] = d;
  return r;
}
// scalar access to element e of a stream (e>>2 selects the quad)
__device__ __forceinline__ float philox_uniform1(const uint64_t* rng, uint64_t extra, uint32_t stream, uint64_t e) {
  e += msgm_rng_base(rng, stream);
  Philox4 p = msgm_philox(rng, extra, stream, e >> 2);
  f32x4 q = f32x4{u01(p.x), u01(p.y), u01(p.z), u01(p.w)};
  int k = (int)(e & 3);
  return k == 0 ? q[0] : k == 1 ? q[1] : k == 2 ? q[2] : q[3];
}
__device__ __forceinline__ float philox_normal1(const uint64_t* rng, uint64_t extra, uint32_t stream, uint64_t e) {
  e += msgm_rng_base(rng, stream);
  Philox4 p = msgm_philox(rng, extra, stream, e >> 2);
  f32x4 q;
  { float a, b, c, d; box_muller(p.x, p.y, a, b); box_muller(p.z, p.w, c, d); q[0] = a; q[1] = b; q[2] = c; q[3] = d; }
  int k = (int)(e & 3);
  return k == 0 ? q[0] : k == 1 ? q[1] : k == 2 ? q[2] : q[3];
}

// ------------------------------------------------------------ SDE schedule
// beta(t) = b0 + (b1-b0) t                                    SDEs.py:72-73
__device__ __forceinline__ float sde_beta(float b0, float b1, float t) { return b0 + (b1 - b0) * t; }
// mean_weight / var                                           SDEs.py:177-181
__device__ __forceinline__ float vp_mean_weight(float b0, float b1, float t) {
  return expf(-0.25f * (t * t) * (b1 - b0) - 0.5f * t * b0);
}
__device__ __forceinline__ float vp_var(float b0, float b1, float t) {
  return 1.0f - expf(-0.5f * (t * t) * (b1 - b0) - t * b0);
}

// -------------------------------------------------------------- reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

static inline int grid_for(int64_t work_items, int block, int max_blocks = 2048) {
  int64_t g = (work_items + block - 1) / block;
  if (g < 1) g = 1;
  if (g > max_blocks) g = max_blocks;
  return (int)g;
}
