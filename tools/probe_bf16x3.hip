// Probe for VERDICT r2 #10 (opt-in experiment, never the headline): can three bf16 pieces per fp32 operand and SIX bf16 MFMA
// products (hh, hm, mh, mm, hl, lh; fp32 accumulate) stand in for an fp32 MFMA on gfx950 — how accurate is it, and how fast can
// a register tile go when its operands come from LDS the way the halo-tile convolution kernels deliver them?
//   x = h + m + l,  h = bf16(x), m = bf16(x - h), l = bf16(x - h - m): 24 mantissa bits in three pieces; the dropped products
//   (ml, lm, ll) are <= 2^-26 relative.  v_mfma_f32_16x16x32_bf16 runs 16 x 16 x 32 in 16 cycles (4 passes) against 32 cycles for the
//   16 x 16 x 4 fp32 instruction: 6 products for K = 32 cost 96 cycles where fp32 needs 8 x 32 = 256 -> 2.67x the MFMA rate.
// Part 1 (accuracy): C = A B, 64 x 64 x K on one wave, against a float64 host product: fp32 MFMA, 6-product split, 3-product
//   split (hh, hm, mh: the "bf16x3" of the literature), each with the fragments laid out as this file assumes — a wrong lane
//   layout shows up as an O(1) error, so the part doubles as the layout check.
// Part 2 (rate): every wave multiplies a 64 x 64 output tile (4 x 4 MFMA tiles) against operands it re-reads from LDS each k-step
//   (ds_read_b128 fragments, no global traffic): the MFMA + operand-delivery ceiling of a halo-tile style kernel, in fp32-EQUIVALENT
//   TFLOP/s (2 M N K per product whatever the instruction mix), for fp32 MFMA, the 6-product and the 3-product split.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe_bf16x3.hip -o tools/probe_bf16x3.bin     Run on the GPU box: tools/probe_bf16x3.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ __bf16 to_bf16(float x) { return (__bf16)x; }             // round to nearest even
struct Split8 { bf16x8 h, m, l; };
// 8 consecutive k-values of one row / column -> the three bf16 fragments
__device__ __forceinline__ Split8 split8(const float* p) {
  Split8 s;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float x = p[j];
    const __bf16 h = to_bf16(x);
    const float r1 = x - (float)h;
    const __bf16 m = to_bf16(r1);
    const __bf16 l = to_bf16(r1 - (float)m);
    s.h[j] = h; s.m[j] = m; s.l[j] = l;
  }
  return s;
}

// ---------------------------------------------------------------- part 1: one wave, C[64][64] = A[64][K] B[K][64] (B given as Bt[64][K])
// Fragment layouts assumed (checked by the result):  16x16x32 bf16: lane l holds A[i = l & 15][k = 8 (l >> 4) + 0..7] and
// B[k = 8 (l >> 4) + 0..7][j = l & 15]; D[i = 4 (l >> 4) + r][j = l & 15].   16x16x4 f32: A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15].
template <int MODE>   // 0 fp32 MFMA, 1 six products, 2 three products
__global__ void __launch_bounds__(64) k_acc(const float* __restrict__ A, const float* __restrict__ Bt, float* __restrict__ C, int K) {
  const int lane = threadIdx.x, il = lane & 15, q = lane >> 4;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  if (MODE == 0) {
    for (int k0 = 0; k0 < K; k0 += 4) {
      float a[4], b[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) { a[t] = A[(size_t)(16 * t + il) * K + k0 + q]; b[t] = Bt[(size_t)(16 * t + il) * K + k0 + q]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  } else {
    for (int k0 = 0; k0 < K; k0 += 32) {
      Split8 a[4], b[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        a[t] = split8(A + (size_t)(16 * t + il) * K + k0 + 8 * q);
        b[t] = split8(Bt + (size_t)(16 * t + il) * K + k0 + 8 * q);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          f32x4 c = acc[i][j];
          if (MODE == 1) {                                   // small terms first
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i].l, b[j].h, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i].h, b[j].l, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i].m, b[j].m, c, 0, 0, 0);
          }
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i].m, b[j].h, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i].h, b[j].m, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i].h, b[j].h, c, 0, 0, 0);
          acc[i][j] = c;
        }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) C[(size_t)(16 * i + 4 * q + r) * 64 + 16 * j + il] = acc[i][j][r];
}

// ---------------------------------------------------------------- part 2: rate with LDS-fed fragments
// LDS holds, per wave, the operands of ONE k-step in fragment order (what a staging pass would have written): fp32: A and B as
// [4 tiles][8 k-quads... ] -> 8 ds_read_b128 each per K = 32; split: [4 tiles][3 pieces] 16-byte fragments -> 12 ds_read_b128 each.
template <int MODE>
__global__ void __launch_bounds__(256) k_rate(float* __restrict__ out, int steps) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  constexpr int PER_WAVE = 2 * 4 * 3 * 64 * 4;                // floats: (A, B) x 4 tiles x 3 pieces x 64 lanes x 16 B
  for (int i = lane; i < PER_WAVE; i += 64) lds[w * PER_WAVE + i] = 1.0f + 1e-3f * (float)((i * 7 + w) % 13);
  __syncthreads();
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  const float* base = lds + w * PER_WAVE;
  for (int s = 0; s < steps; ++s) {
    const float* my = base;
    asm volatile("" : "+v"(my));                           // the fragments are RE-READ from LDS every k-step (not hoisted)
    if (MODE == 0) {
      // K = 32 in fp32: two rounds of (4 + 4 fragments of 16 B = 4 k-values each) and 4 MFMA steps of K = 4 per tile pair
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        f32x4 a[4], b[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          a[t] = *reinterpret_cast<const f32x4*>(my + ((half * 8 + t) * 64 + lane) * 4);
          b[t] = *reinterpret_cast<const f32x4*>(my + ((half * 8 + 4 + t) * 64 + lane) * 4);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][r], b[j][r], acc[i][j], 0, 0, 0);
      }
    } else {
      bf16x8 ah[4], am[4], al[4], bh[4], bm[4], bl[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        ah[t] = *reinterpret_cast<const bf16x8*>(my + ((t * 3 + 0) * 64 + lane) * 4);
        am[t] = *reinterpret_cast<const bf16x8*>(my + ((t * 3 + 1) * 64 + lane) * 4);
        bh[t] = *reinterpret_cast<const bf16x8*>(my + ((12 + t * 3 + 0) * 64 + lane) * 4);
        bm[t] = *reinterpret_cast<const bf16x8*>(my + ((12 + t * 3 + 1) * 64 + lane) * 4);
        if (MODE == 1) {
          al[t] = *reinterpret_cast<const bf16x8*>(my + ((t * 3 + 2) * 64 + lane) * 4);
          bl[t] = *reinterpret_cast<const bf16x8*>(my + ((12 + t * 3 + 2) * 64 + lane) * 4);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          f32x4 c = acc[i][j];
          if (MODE == 1) {
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh[j], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl[j], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[i], bm[j], c, 0, 0, 0);
          }
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[i], bh[j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bm[j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[j], c, 0, 0, 0);
          acc[i][j] = c;
        }
    }
  }
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  if (sum == 12345.678f) out[blockIdx.x * 256 + tid] = sum;      // keep the work alive
}

template <int MODE>
static int accuracy(const char* name, int K, double scale_b) {
  const int M = 64;
  std::vector<float> A((size_t)M * K), Bt((size_t)M * K), C((size_t)M * M);
  srand(1234 + K);
  for (auto& v : A) v = (float)((rand() / (double)RAND_MAX) * 2 - 1);
  for (auto& v : Bt) v = (float)(((rand() / (double)RAND_MAX) * 2 - 1) * scale_b);
  float *dA, *dB, *dC;
  CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, Bt.size() * 4)); CK(hipMalloc(&dC, C.size() * 4));
  CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, Bt.data(), Bt.size() * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL((k_acc<MODE>), dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
  double num = 0, den = 0, worst = 0;
  for (int i = 0; i < M; ++i)
    for (int j = 0; j < M; ++j) {
      double ref = 0, mag = 0;
      for (int k = 0; k < K; ++k) { const double p = (double)A[(size_t)i * K + k] * (double)Bt[(size_t)j * K + k]; ref += p; mag += fabs(p); }
      const double e = C[(size_t)i * M + j] - ref;
      num += e * e; den += ref * ref;
      if (fabs(e) / mag > worst) worst = fabs(e) / mag;      // error relative to sum |a b|: the forward-error yardstick of a dot product
    }
  printf("accuracy %-28s K=%5d: rel-L2 vs float64 %.2e   worst |err| / sum|a b| %.2e\n", name, K, sqrt(num / den), worst);
  CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
  return 0;
}

template <int MODE>
static int rate(const char* name, int wgs_per_cu) {
  int dev = 0; hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, dev));
  const int cus = prop.multiProcessorCount, grid = cus * wgs_per_cu, steps = 2000;
  const size_t lds = (size_t)4 * 2 * 4 * 3 * 64 * 4 * sizeof(float);       // 4 waves x 24 KB
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rate<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  float* out; CK(hipMalloc(&out, (size_t)grid * 256 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k_rate<MODE>), dim3(grid), dim3(256), lds, 0, out, 50);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL((k_rate<MODE>), dim3(grid), dim3(256), lds, 0, out, steps);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  const double flop = 2.0 * 64 * 64 * 32 * (double)steps * 4 * grid;     // per wave and k-step: a 64 x 64 x 32 product
  printf("rate     %-28s %d workgroups/CU: %8.3f ms  %7.1f fp32-equivalent TFLOP/s  (%.2f of the 157.3 fp32-MFMA peak)\n", name, wgs_per_cu, ms,
         flop / ms / 1e9, flop / ms / 1e9 / 157.3);
  CK(hipFree(out));
  return 0;
}

int main() {
  for (int K : {64, 1152, 4608}) {
    if (accuracy<0>("fp32 MFMA 16x16x4", K, 1.0)) return 2;
    if (accuracy<1>("bf16 split, 6 products", K, 1.0)) return 2;
    if (accuracy<2>("bf16 split, 3 products", K, 1.0)) return 2;
  }
  // operands of very different magnitude inside one dot product (weights ~1e-2 against activations ~1)
  if (accuracy<0>("fp32 MFMA, B x 1e-2", 1152, 1e-2)) return 2;
  if (accuracy<1>("6 products, B x 1e-2", 1152, 1e-2)) return 2;
  for (int wg : {1, 2}) {
    if (rate<0>("fp32 MFMA 16x16x4", wg)) return 2;
    if (rate<1>("bf16 split, 6 products", wg)) return 2;
    if (rate<2>("bf16 split, 3 products", wg)) return 2;
  }
  return 0;
}
