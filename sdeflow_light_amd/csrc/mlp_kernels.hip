// mlp_kernels.hip — K5: the MLP score network (NN.py:73-120) as ONE fused
// kernel per use: forward (sampling), forward + Euler–Maruyama update, and the
// whole sliced-score-matching training pass (forward + forward-mode tangent +
// loss + backward), on fp32 MFMA (v_mfma_f32_16x16x4_f32, exact fp32).
//
// Design (gfx950, 4 waves / workgroup, 1 workgroup / CU, persistent grid):
//   * orientation Z[feature][sample] = W[feature][k] . H[k][sample]: the weight
//     is the MFMA A operand, activations the B operand.  The C/D layout then has
//     the sample on lane&15 and 4 consecutive features in the 4 accumulator
//     registers, which is exactly the B-operand layout of the next layer — the
//     only thing that crosses waves is the activation tile, exchanged through LDS
//     in [sample][feature] order with b128 reads/writes.
//   * wave w owns output features [32w, 32w+32) of every layer (2 row tiles).
//     Its weight fragments are streamed from L2 straight into registers one phase
//     ahead (the 128x128 weights never sit in LDS); its slice of dW2/dW3 lives in
//     accumulator registers for the whole kernel.
//   * a tile is 16 samples = 16 primal + 16 tangent columns (forward-mode dual
//     numbers: tangent = J.v), so primal and tangent of one sample share a lane
//     and the Swish first/second derivatives are applied in registers.
//   * per-workgroup gradient slabs are written once at the end and reduced by a
//     second (deterministic) kernel — no float atomics.
#include "common.h"

#define HID 128

// Diagnostic build only (-DMLP_STAMPS): per-phase s_memtime sums of workgroup 0 /
// wave 0, read back with msgm_debug_stamps().  No stamp executes in the shipped .so.
#ifdef MLP_STAMPS
__device__ unsigned long long g_stamps[16];
#define STAMP_DECL unsigned long long st_acc[12] = {0,0,0,0,0,0,0,0,0,0,0,0}; unsigned long long st_prev = __builtin_amdgcn_s_memtime();
#define STAMP(i) do { unsigned long long _t = __builtin_amdgcn_s_memtime(); st_acc[i] += _t - st_prev; st_prev = _t; } while (0)
#define STAMP_FLUSH do { if (blockIdx.x == 0 && threadIdx.x == 0) { for (int _i = 0; _i < 12; ++_i) g_stamps[_i] = st_acc[_i]; } } while (0)
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_FLUSH
#endif
#define ACT_P 132   // LDS pitch of an activation row ([sample][feature])

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ---- LDS carve (floats), per (mode, width) ------------------------------------
// NARROW nets (in_dim <= 16 and d <= 16) use 20-float small rows and 16 padded outputs; the forward / sampler
// modes drop the backward-only buffers — 67 KB instead of 134 KB, so TWO workgroups share a CU there and each
// one's barrier bubbles are filled by the other's MFMAs.
template <int MODE, bool WIDE>
struct MlpLds {
  static constexpr int SMP = WIDE ? 36 : 20;       // pitch of h0 / abar / partial rows
  static constexpr int DP = WIDE ? 32 : 16;        // padded output width
  static constexpr bool TRAIN = MODE == 2;
  static constexpr int X = 0;
  static constexpr int Y = X + 32 * ACT_P;
  static constexpr int Z = Y + 32 * ACT_P;
  static constexpr int U = Z + (TRAIN ? 32 * ACT_P : 0);
  static constexpr int H0 = U + (TRAIN ? 32 * ACT_P : 0);
  static constexpr int ABAR = H0 + 32 * SMP;
  static constexpr int PART = ABAR + (TRAIN ? 32 * SMP : 0);
  static constexpr int W1 = PART + 4 * 32 * SMP;
  static constexpr int W4 = W1 + HID * SMP;
  static constexpr int B1 = W4 + DP * ACT_P;
  static constexpr int B2 = B1 + HID;
  static constexpr int B3 = B2 + HID;
  static constexpr int B4 = B3 + HID;
  static constexpr int DB4 = B4 + 32;
  static constexpr int RED = DB4 + (TRAIN ? 16 * DP : 0);
  static constexpr int FLOATS = RED + 64;
  static constexpr int BYTES = FLOATS * 4;
};

struct MlpArgs {
  msgm_mlp_params_t P;
  const float* y;        // (B,d) inputs (x_t for the sampler)
  const float* t;        // (B) per-sample time, or null => t_scalar
  const float* v;        // (B,d) probe (train)
  const float* u;        // (B,d) optional: cotangent direction of adot, loss_b = adot.u + cst + |a|^2/2 (MSGM); null => SGM closed form
  const float* cst;      // (B) optional per-sample constant of the loss
  float* out;            // forward: a (B,d); EM: x updated in place (== y)
  int64_t B;
  int in_dim, in4, d4;   // in_dim = d + 1 (+1 with premodule); in4 = ceil(in/4); d4 = ceil(d/4)
  float t_scalar;
  // SDE (SGM)
  float b0, b1, T;
  // EM step
  float delta, sqrt_delta, lmbd;
  const float* z; const uint64_t* rng; uint64_t rng_step;
  // train
  float inv_batch;
  float* loss_per; float* slabs;   // slabs: [gridDim.x][n_params + 1] (last = loss sum)
  int64_t n_params;
};

enum { MODE_FWD = 0, MODE_EM = 1, MODE_TRAIN = 2 };

// Swish and its first two derivatives from z.
__device__ __forceinline__ void swish012(float z, float& s0, float& s1, float& s2) {
  float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-z));
  float om = 1.0f - sg;
  s0 = z * sg;
  s1 = sg * (1.0f + z * om);
  s2 = sg * om * (2.0f + z * (1.0f - 2.0f * sg));
}
__device__ __forceinline__ float swish0(float z) { return z * __builtin_amdgcn_rcpf(1.0f + __expf(-z)); }

// A fragment of one 128x128 weight for this wave's 32 output rows, k-group g:
//  !TR: rows = out features of W (forward);  A[r] = W[32w+16it+il][16g+4q+r]
//   TR: rows = in features (dgrad, A = W^T); A[r] = W[16g+4q+r][32w+16it+il]
template <bool TR>
__device__ __forceinline__ f32x4 load_afrag(const float* __restrict__ W, int w, int il, int q, int it, int g) {
  if (!TR) return *reinterpret_cast<const f32x4*>(W + (32 * w + 16 * it + il) * HID + 16 * g + 4 * q);
  const float* p = W + (16 * g + 4 * q) * HID + 32 * w + 16 * it + il;
  return f32x4{p[0], p[HID], p[2 * HID], p[3 * HID]};
}

// The weights are streamed from L2 into registers PF k-groups ahead of use
// (a k-group = 16 MFMAs = 512 cycles of matrix pipe per wave), so only
// (PF+1) x 8 registers hold weights at any time; `pre` carries groups
// 0..PF-1 of the NEXT gemm across the phase boundary.
#ifndef PF
#define PF 2
#endif
#ifndef MLP_HINTS
#define MLP_HINTS 0
#endif
#ifndef MLP_SB
#define MLP_SB 1
#endif
#ifndef MLP_OVL_FWD
#define MLP_OVL_FWD 1
#endif
#ifndef MLP_OVL_BWD
#define MLP_OVL_BWD 1
#endif
struct WPre { f32x4 a[PF]; };     // fragments j = 0..PF-1 of the next gemm (j = it*8 + g: row tile 0 first)

template <bool TR>
__device__ __forceinline__ void prefetch_w(const float* __restrict__ W, int w, int il, int q, WPre& pre) {
#pragma unroll
  for (int j = 0; j < PF; ++j) pre.a[j] = load_afrag<TR>(W, w, il, q, j >> 3, j & 7);
}

// acc[it][ck] += sum_k A[it][k] * buf[ck*16+il][k]   (K = 128, B operand from LDS), row tile 0 first, then
// row tile 1.  Software-pipelined by hand: the LDS reads of step j+1 and the global loads of step j+PF are
// issued ahead of the 8 MFMAs of step j; sched_barrier keeps hipcc from hoisting every load (which spills).
// HOOK: while the matrix pipe works on row tile 1, the VALU applies bias + Swish to the finished row tile 0
// (an MFMA occupies the vector issue port for only 8 of its 32 cycles):
//   HOOK 1 (train): h[0][0] = s(z), h[0][1] = s'(z) zdot;   HOOK 2 (inference): both column halves are primal.
template <bool TR, int HOOK>
__device__ __forceinline__ void gemm128(const float* __restrict__ W, const WPre& pre, const float* buf, int w, int il,
                                        int q, f32x4 (&acc)[2][2], const float* bias_lds, f32x4 (&h)[2][2]) {
  f32x4 ring[PF + 1];
#pragma unroll
  for (int j = 0; j < PF; ++j) ring[j] = pre.a[j];
  const float* bp0 = buf + il * ACT_P + 4 * q;
  const float* bp1 = buf + (16 + il) * ACT_P + 4 * q;
  f32x4 bn0 = *reinterpret_cast<const f32x4*>(bp0), bn1 = *reinterpret_cast<const f32x4*>(bp1);
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int it = j >> 3;
    const f32x4 b0 = bn0, b1 = bn1;
    if (j + PF < 16) ring[(j + PF) % (PF + 1)] = load_afrag<TR>(W, w, il, q, (j + PF) >> 3, (j + PF) & 7);
    if (j + 1 < 16) {
      bn0 = *reinterpret_cast<const f32x4*>(bp0 + 16 * ((j + 1) & 7));
      bn1 = *reinterpret_cast<const f32x4*>(bp1 + 16 * ((j + 1) & 7));
    }
    if (MLP_OVL_FWD && HOOK && j == 8) {        // row tile 0 is complete: bias
      const f32x4 bb = *reinterpret_cast<const f32x4*>(bias_lds + 32 * w + 4 * q);
      acc[0][0] += bb;
      if (HOOK == 2) acc[0][1] += bb;
    }
    if (MLP_OVL_FWD && HOOK && j >= 8 && j < 12) {   // one register per step
      const int r = j - 8;
      if (HOOK == 1) {
        float s0, s1, s2;
        swish012(acc[0][0][r], s0, s1, s2);
        h[0][0][r] = s0; h[0][1][r] = s1 * acc[0][1][r];
      } else {
        h[0][0][r] = swish0(acc[0][0][r]); h[0][1][r] = swish0(acc[0][1][r]);
      }
    }
    const f32x4 a = ring[j % (PF + 1)];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (it == 0) { acc[0][0] = mfma16(a[r], b0[r], acc[0][0]); acc[0][1] = mfma16(a[r], b1[r], acc[0][1]); }
      else { acc[1][0] = mfma16(a[r], b0[r], acc[1][0]); acc[1][1] = mfma16(a[r], b1[r], acc[1][1]); }
    }
    if (MLP_HINTS && HOOK && j >= 8 && j < 12) {
#pragma unroll
      for (int k = 0; k < 4; ++k) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x002, 6, 0); }
    }
    if (MLP_SB) __builtin_amdgcn_sched_barrier(0);
  }
  if (!MLP_OVL_FWD && HOOK) {
    const f32x4 bb = *reinterpret_cast<const f32x4*>(bias_lds + 32 * w + 4 * q);
    acc[0][0] += bb;
    if (HOOK == 2) acc[0][1] += bb;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (HOOK == 1) {
        float s0, s1, s2;
        swish012(acc[0][0][r], s0, s1, s2);
        h[0][0][r] = s0; h[0][1][r] = s1 * acc[0][1][r];
      } else { h[0][0][r] = swish0(acc[0][0][r]); h[0][1][r] = swish0(acc[0][1][r]); }
    }
  }
}

// store this wave's C-layout tile pair to an activation buffer ([sample][feature])
__device__ __forceinline__ void store_act(float* buf, int w, int il, int q, const f32x4 (&h)[2][2]) {
#pragma unroll
  for (int it = 0; it < 2; ++it)
#pragma unroll
    for (int ck = 0; ck < 2; ++ck)
      *reinterpret_cast<f32x4*>(buf + (ck * 16 + il) * ACT_P + 32 * w + 16 * it + 4 * q) = h[it][ck];
}

// dW[it][kt] += sum_n ZB[feat][n] * HB[k][n] over the 32 dual columns.
// A (lane il = feature) and B (lane il = k) are read as b32 from [sample][feature]
// buffers, one (ck,s) step ahead of the MFMAs that consume them.
template <int KT>
__device__ __forceinline__ void wgrad(const float* zb, const float* hb, int hb_pitch, int w, int il, int q,
                                      f32x4 (&dW)[2][KT]) {
  float na0, na1, nb[KT];
  {
    const int n = q;
    na0 = zb[n * ACT_P + 32 * w + il]; na1 = zb[n * ACT_P + 32 * w + 16 + il];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) nb[kt] = hb[n * hb_pitch + 16 * kt + il];
  }
#pragma unroll
  for (int st = 0; st < 8; ++st) {
    const float a0 = na0, a1 = na1;
    float b[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) b[kt] = nb[kt];
    if (st + 1 < 8) {
      const int n = ((st + 1) >> 2) * 16 + 4 * ((st + 1) & 3) + q;
      na0 = zb[n * ACT_P + 32 * w + il]; na1 = zb[n * ACT_P + 32 * w + 16 + il];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) nb[kt] = hb[n * hb_pitch + 16 * kt + il];
    }
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      dW[0][kt] = mfma16(a0, b[kt], dW[0][kt]);
      dW[1][kt] = mfma16(a1, b[kt], dW[1][kt]);
    }
    if (MLP_SB) __builtin_amdgcn_sched_barrier(0);
  }
}

// wgrad of a 128x128 layer with the dual Swish backward of the NEXT pointwise stage hidden under its
// MFMAs: step st of 8 also turns (gP,gT) -> (zbarP, zbarT), in place, for register (it = st>>2, r = st&3).
template <int KT>
__device__ __forceinline__ void wgrad_swish(const float* zbuf, const float* hb, int hb_pitch, int w, int il, int q,
                                            f32x4 (&dW)[2][KT], const f32x4 (&z)[2][2], f32x4 (&g)[2][2]) {
  float na0, na1, nb[KT];
  {
    const int n = q;
    na0 = zbuf[n * ACT_P + 32 * w + il]; na1 = zbuf[n * ACT_P + 32 * w + 16 + il];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) nb[kt] = hb[n * hb_pitch + 16 * kt + il];
  }
#pragma unroll
  for (int st = 0; st < 8; ++st) {
    const float a0 = na0, a1 = na1;
    float b[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) b[kt] = nb[kt];
    if (st + 1 < 8) {
      const int n = ((st + 1) >> 2) * 16 + 4 * ((st + 1) & 3) + q;
      na0 = zbuf[n * ACT_P + 32 * w + il]; na1 = zbuf[n * ACT_P + 32 * w + 16 + il];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) nb[kt] = hb[n * hb_pitch + 16 * kt + il];
    }
    if (MLP_OVL_BWD) {
      const int it = st >> 2, r = st & 3;
      float s0, s1, s2;
      swish012(z[it][0][r], s0, s1, s2);
      const float gt = g[it][1][r];
      g[it][0][r] = g[it][0][r] * s1 + gt * (s2 * z[it][1][r]);     // in place: g becomes (zbarP, zbarT)
      g[it][1][r] = gt * s1;
    }
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      dW[0][kt] = mfma16(a0, b[kt], dW[0][kt]);
      dW[1][kt] = mfma16(a1, b[kt], dW[1][kt]);
    }
    if (MLP_HINTS) {
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x002, 3, 0); }
    }
    if (MLP_SB) __builtin_amdgcn_sched_barrier(0);
  }
  if (!MLP_OVL_BWD) {
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      const int it = st >> 2, r = st & 3;
      float s0, s1, s2;
      swish012(z[it][0][r], s0, s1, s2);
      const float gt = g[it][1][r];
      g[it][0][r] = g[it][0][r] * s1 + gt * (s2 * z[it][1][r]);
      g[it][1][r] = gt * s1;
    }
  }
}

// dual Swish backward in registers: (gP,gT) cotangents of (hP,hT) -> cotangents of (zP,zT)
__device__ __forceinline__ void swish_bwd(const f32x4 (&z)[2][2], const f32x4 (&g)[2][2], f32x4 (&zb)[2][2]) {
#pragma unroll
  for (int it = 0; it < 2; ++it)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float s0, s1, s2;
      swish012(z[it][0][r], s0, s1, s2);
      zb[it][1][r] = g[it][1][r] * s1;
      zb[it][0][r] = g[it][0][r] * s1 + g[it][1][r] * (s2 * z[it][1][r]);
    }
}

template <int MODE, bool WIDE>
__global__ void __launch_bounds__(256, 1) k_mlp(MlpArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int KT1 = WIDE ? 2 : 1;   // k-tiles of the first layer's input (in_dim <= 16 / 32)
  constexpr int OT = WIDE ? 2 : 1;    // o-tiles of the last layer's output (d <= 16 / 32)
  constexpr int SPT = (MODE == MODE_TRAIN) ? 16 : 32;   // samples per tile
  const int tid = threadIdx.x;
  const int w = tid >> 6, lane = tid & 63, il = lane & 15, q = lane >> 4;
  const int d = A.P.d;
  using LO = MlpLds<MODE, WIDE>;
  constexpr int SM_P = LO::SMP, DPAD = LO::DP;
  float* X = lds + LO::X; float* Y = lds + LO::Y; float* Z = lds + LO::Z; float* U = lds + LO::U;
  float* H0 = lds + LO::H0; float* ABAR = lds + LO::ABAR; float* PART = lds + LO::PART;
  float* W1s = lds + LO::W1; float* W4s = lds + LO::W4;
  float* B1s = lds + LO::B1; float* B2s = lds + LO::B2; float* B3s = lds + LO::B3; float* B4s = lds + LO::B4;
  float* DB4 = lds + LO::DB4; float* RED = lds + LO::RED;

  // ---- one-time staging of the small layers and biases -------------------
  for (int i = tid; i < HID * SM_P; i += 256) {
    int r = i / SM_P, c = i - r * SM_P;
    W1s[i] = (c < A.in_dim) ? A.P.W1[r * A.in_dim + c] : 0.f;
  }
  for (int i = tid; i < DPAD * ACT_P; i += 256) {
    int r = i / ACT_P, c = i - r * ACT_P;
    W4s[i] = (r < d && c < HID) ? A.P.W4[r * HID + c] : 0.f;
  }
  for (int i = tid; i < HID; i += 256) { B1s[i] = A.P.b1[i]; B2s[i] = A.P.b2[i]; B3s[i] = A.P.b3[i]; }
  if (tid < 32) B4s[tid] = tid < d ? A.P.b4[tid] : 0.f;
  if (MODE == MODE_TRAIN) {
    for (int i = tid; i < 16 * DPAD; i += 256) DB4[i] = 0.f;
    for (int i = tid; i < 32 * SM_P; i += 256) ABAR[i] = 0.f;
  }
  for (int i = tid; i < 32 * SM_P; i += 256) H0[i] = 0.f;

  // persistent accumulators (train)
  f32x4 dW2[2][8], dW3[2][8], dW1[2][KT1], dW4[2][OT], db1[2], db2[2], db3[2];
  float loss_acc = 0.f;
  if (MODE == MODE_TRAIN) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
#pragma unroll
      for (int k = 0; k < 8; ++k) { dW2[it][k] = f32x4{0, 0, 0, 0}; dW3[it][k] = f32x4{0, 0, 0, 0}; }
#pragma unroll
      for (int k = 0; k < KT1; ++k) dW1[it][k] = f32x4{0, 0, 0, 0};
#pragma unroll
      for (int k = 0; k < OT; ++k) dW4[it][k] = f32x4{0, 0, 0, 0};
      db1[it] = f32x4{0, 0, 0, 0}; db2[it] = f32x4{0, 0, 0, 0}; db3[it] = f32x4{0, 0, 0, 0};
    }
  }

  WPre pre;
  prefetch_w<false>(A.P.W2, w, il, q, pre);
  __syncthreads();

  STAMP_DECL
  const int64_t n_tiles = (A.B + SPT - 1) / SPT;
  const float ca = 1.0f - 0.5f * A.lmbd;

  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t s_base = tile * SPT;
    // ---- phase 0: h0 (and its tangent) -----------------------------------
    // premodule none: h0 = [y, t], h0dot = [v, 0]                     NN.py:113-119
    // NormalizeLogRadius: h0 = [y/r, log r, t], r = |y| + 1e-6          NN.py:56-70
    if (tid < 32 && (MODE != MODE_TRAIN || tid < 16)) {
      const int64_t smp = s_base + tid;
      const bool live = smp < A.B;
      float* hp = H0 + tid * SM_P;
      float* ht = H0 + (16 + tid) * SM_P;   // tangent row (train only)
      float tt = 0.f;
      if (live) {
        tt = A.t ? A.t[smp] : A.t_scalar;
        if (MODE == MODE_EM) tt = A.T - tt;                               // s = T - t  SDEs.py:556-557
      }
      const float* yr = A.y + smp * d;
      const float* vr = (MODE == MODE_TRAIN) ? A.v + smp * d : nullptr;
      if (A.P.premodule == 0) {
        for (int i = 0; i < d; ++i) {
          hp[i] = live ? yr[i] : 0.f;
          if (MODE == MODE_TRAIN) ht[i] = live ? vr[i] : 0.f;
        }
        hp[d] = tt;
        if (MODE == MODE_TRAIN) ht[d] = 0.f;
      } else {
        float ss = 0.f, yv = 0.f;
        for (int i = 0; i < d; ++i) {
          float yi = live ? yr[i] : 1.0f;
          ss += yi * yi;
          if (MODE == MODE_TRAIN) yv += yi * (live ? vr[i] : 0.f);
        }
        const float nr = sqrtf(ss);
        const float r = nr + 1e-6f;
        const float rdot = yv / nr;                                        // d|y| along v
        for (int i = 0; i < d; ++i) {
          float yi = live ? yr[i] : 1.0f;
          hp[i] = yi / r;
          if (MODE == MODE_TRAIN) ht[i] = (live ? vr[i] : 0.f) / r - yi * rdot / (r * r);
        }
        hp[d] = logf(r);
        hp[d + 1] = tt;
        if (MODE == MODE_TRAIN) { ht[d] = rdot / r; ht[d + 1] = 0.f; }
      }
    }
    __syncthreads();
    STAMP(0);

    // ---- phase 1: layer 1 (K = in_dim, weights from LDS) ------------------
    f32x4 z1[2][2], z2[2][2], z3[2][2], h[2][2];
#pragma unroll
    for (int it = 0; it < 2; ++it) { z1[it][0] = f32x4{0, 0, 0, 0}; z1[it][1] = f32x4{0, 0, 0, 0}; }
    for (int s = 0; s < A.in4; ++s) {
      const float a0 = W1s[(32 * w + il) * SM_P + 4 * s + q];
      const float a1 = W1s[(32 * w + 16 + il) * SM_P + 4 * s + q];
      const float bP = H0[il * SM_P + 4 * s + q];
      const float bT = H0[(16 + il) * SM_P + 4 * s + q];
      z1[0][0] = mfma16(a0, bP, z1[0][0]); z1[0][1] = mfma16(a0, bT, z1[0][1]);
      z1[1][0] = mfma16(a1, bP, z1[1][0]); z1[1][1] = mfma16(a1, bT, z1[1][1]);
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const f32x4 bb = *reinterpret_cast<const f32x4*>(B1s + 32 * w + 16 * it + 4 * q);
      z1[it][0] += bb;
      if (MODE != MODE_TRAIN) z1[it][1] += bb;          // second primal half
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (MODE == MODE_TRAIN) {
          float s0, s1, s2; swish012(z1[it][0][r], s0, s1, s2);
          h[it][0][r] = s0; h[it][1][r] = s1 * z1[it][1][r];
        } else { h[it][0][r] = swish0(z1[it][0][r]); h[it][1][r] = swish0(z1[it][1][r]); }
      }
    }
    store_act(X, w, il, q, h);
    __syncthreads();
    STAMP(1);

    // ---- phase 2: layer 2 --------------------------------------------------
#pragma unroll
    for (int it = 0; it < 2; ++it) { z2[it][0] = f32x4{0, 0, 0, 0}; z2[it][1] = f32x4{0, 0, 0, 0}; }
    gemm128<false, (MODE == MODE_TRAIN ? 1 : 2)>(A.P.W2, pre, X, w, il, q, z2, B2s, h);
    prefetch_w<false>(A.P.W3, w, il, q, pre);          // head of the next gemm's weights
    {
      const f32x4 bb = *reinterpret_cast<const f32x4*>(B2s + 32 * w + 16 + 4 * q);
      z2[1][0] += bb;
      if (MODE != MODE_TRAIN) z2[1][1] += bb;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (MODE == MODE_TRAIN) {
          float s0, s1, s2; swish012(z2[1][0][r], s0, s1, s2);
          h[1][0][r] = s0; h[1][1][r] = s1 * z2[1][1][r];
        } else { h[1][0][r] = swish0(z2[1][0][r]); h[1][1][r] = swish0(z2[1][1][r]); }
      }
    }
    store_act(Y, w, il, q, h);
    __syncthreads();
    STAMP(2);

    // ---- phase 3: layer 3, then this wave's K-slice of layer 4 -------------
#pragma unroll
    for (int it = 0; it < 2; ++it) { z3[it][0] = f32x4{0, 0, 0, 0}; z3[it][1] = f32x4{0, 0, 0, 0}; }
    gemm128<false, (MODE == MODE_TRAIN ? 1 : 2)>(A.P.W3, pre, Y, w, il, q, z3, B3s, h);
    if (MODE == MODE_TRAIN) prefetch_w<true>(A.P.W3, w, il, q, pre);   // W3^T for dgrad
    else prefetch_w<false>(A.P.W2, w, il, q, pre);                     // next tile's layer 2
    {
      const f32x4 bb = *reinterpret_cast<const f32x4*>(B3s + 32 * w + 16 + 4 * q);
      z3[1][0] += bb;
      if (MODE != MODE_TRAIN) z3[1][1] += bb;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (MODE == MODE_TRAIN) {
          float s0, s1, s2; swish012(z3[1][0][r], s0, s1, s2);
          h[1][0][r] = s0; h[1][1][r] = s1 * z3[1][1][r];
        } else { h[1][0][r] = swish0(z3[1][0][r]); h[1][1][r] = swish0(z3[1][1][r]); }
      }
    }
    if (MODE == MODE_TRAIN) store_act(Z, w, il, q, h);   // h3 is the dW4 operand later
    {
      // partial[o][col] = sum_{k in this wave's 32 features} W4[o][k] h3[k][col]; B operand = registers
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) {
        f32x4 p0 = {0, 0, 0, 0}, p1 = {0, 0, 0, 0};
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const f32x4 a4 = *reinterpret_cast<const f32x4*>(W4s + (16 * ot + il) * ACT_P + 32 * w + 16 * it + 4 * q);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            p0 = mfma16(a4[r], h[it][0][r], p0);
            p1 = mfma16(a4[r], h[it][1][r], p1);
          }
        }
        *reinterpret_cast<f32x4*>(PART + (w * 32 + il) * SM_P + 16 * ot + 4 * q) = p0;
        *reinterpret_cast<f32x4*>(PART + (w * 32 + 16 + il) * SM_P + 16 * ot + 4 * q) = p1;
      }
    }
    __syncthreads();
    STAMP(3);

    // ---- phase 4: reduce layer-4 partials; outputs / loss ------------------
    if (MODE == MODE_FWD || MODE == MODE_EM) {
      // 32 samples x d outputs
      for (int idx = tid; idx < 32 * d; idx += 256) {
        const int c = idx / d, o = idx - c * d;
        const int64_t smp = s_base + c;
        if (smp < A.B) {
          float a = B4s[o];
#pragma unroll
          for (int ww = 0; ww < 4; ++ww) a += PART[(ww * 32 + c) * SM_P + o];
          const int64_t e = smp * d + o;
          if (MODE == MODE_FWD) {
            A.out[e] = a;
          } else {
            // x += [(1-l/2) sqrt(beta) a + 1/2 beta x] delta + sqrt(1-l) sqrt(beta) dW
            // (SDEs.py:556-561,587-588; sde_scheme.py:82-84,38-40)
            const float s = A.T - A.t_scalar;
            const float beta = sde_beta(A.b0, A.b1, s);
            const float sb = sqrtf(beta);
            const float x = A.y[e];
            const float zz = A.z ? A.z[e] : philox_normal1(A.rng, A.rng_step, RNG_STREAM_DW, (uint64_t)e);
            const float mu = ca * (sb * a) - (-0.5f * beta * x);
            A.out[e] = x + (mu * A.delta + (sqrtf(1.0f - A.lmbd) * sb) * (A.sqrt_delta * zz));
          }
        }
      }
      __syncthreads();   // PART / H0 are rewritten by the next tile
      continue;
    }

    if (MODE == MODE_TRAIN) {
      if (tid < 16) {
        const int64_t smp = s_base + tid;
        const bool live = smp < A.B;
        const float wgt = live ? A.inv_batch : 0.f;
        float beta = 0.f, sb = 0.f;
        if (live && !A.u) { beta = sde_beta(A.b0, A.b1, A.t[smp]); sb = sqrtf(beta); }
        float lj = 0.f;
        for (int o = 0; o < d; ++o) {
          float a = B4s[o], ad = 0.f;
#pragma unroll
          for (int ww = 0; ww < 4; ++ww) { a += PART[(ww * 32 + tid) * SM_P + o]; ad += PART[(ww * 32 + 16 + tid) * SM_P + o]; }
          float adb;
          if (A.u) {
            // general form: loss_b = sum_o adot_o u_o + cst_b + 1/2 a_o^2, u = (d mu/d a)^T v  (MSGM: G(y)^T v)
            const float uo = live ? A.u[smp * d + o] : 0.f;
            lj += ad * uo + 0.5f * a * a;
            adb = uo * wgt;
          } else {
            const float vo = live ? A.v[smp * d + o] : 0.f;
            // SGM: loss_b = sum_o v_o (sqrt(beta) adot_o + 1/2 beta v_o) + 1/2 a_o^2     SDEs.py:631-646
            lj += vo * (sb * ad + 0.5f * beta * vo) + 0.5f * a * a;
            adb = sb * vo * wgt;
          }
          const float ab = a * wgt;
          ABAR[tid * SM_P + o] = ab;
          ABAR[(16 + tid) * SM_P + o] = adb;
          DB4[tid * DPAD + o] += ab;
        }
        if (live && A.u && A.cst) lj += A.cst[smp];
        if (live) { loss_acc += lj; if (A.loss_per) A.loss_per[smp] = lj; }
      }
      __syncthreads();
      STAMP(4);

      // ---- phase 5: layer-4 backward, dW4, Swish' on layer 3 --------------
      f32x4 g[2][2];
#pragma unroll
      for (int it = 0; it < 2; ++it) { g[it][0] = f32x4{0, 0, 0, 0}; g[it][1] = f32x4{0, 0, 0, 0}; }
      for (int s = 0; s < A.d4; ++s) {
        const float a0 = W4s[(4 * s + q) * ACT_P + 32 * w + il];
        const float a1 = W4s[(4 * s + q) * ACT_P + 32 * w + 16 + il];
        const float bP = ABAR[il * SM_P + 4 * s + q];
        const float bT = ABAR[(16 + il) * SM_P + 4 * s + q];
        g[0][0] = mfma16(a0, bP, g[0][0]); g[0][1] = mfma16(a0, bT, g[0][1]);
        g[1][0] = mfma16(a1, bP, g[1][0]); g[1][1] = mfma16(a1, bT, g[1][1]);
      }
      wgrad_swish<OT>(Z, ABAR, SM_P, w, il, q, dW4, z3, g);       // dW4^T[feat][o] += h3 . abar, Swish' hidden
      db3[0] += g[0][0]; db3[1] += g[1][0];
      store_act(U, w, il, q, g);
      __syncthreads();
      STAMP(5);

      // ---- phase 6: dgrad layer 3 (W3^T), dW3, Swish' on layer 2 -----------
#pragma unroll
      for (int it = 0; it < 2; ++it) { g[it][0] = f32x4{0, 0, 0, 0}; g[it][1] = f32x4{0, 0, 0, 0}; }
      gemm128<true, 0>(A.P.W3, pre, U, w, il, q, g, nullptr, h);
      prefetch_w<true>(A.P.W2, w, il, q, pre);           // W2^T for the next dgrad
      wgrad_swish<8>(U, Y, ACT_P, w, il, q, dW3, z2, g);
      db2[0] += g[0][0]; db2[1] += g[1][0];
      store_act(Z, w, il, q, g);                           // h3 is dead after phase 5
      __syncthreads();
      STAMP(6);

      // ---- phase 7: dgrad layer 2 (W2^T), dW2, Swish' on layer 1 -----------
#pragma unroll
      for (int it = 0; it < 2; ++it) { g[it][0] = f32x4{0, 0, 0, 0}; g[it][1] = f32x4{0, 0, 0, 0}; }
      gemm128<true, 0>(A.P.W2, pre, Z, w, il, q, g, nullptr, h);
      prefetch_w<false>(A.P.W2, w, il, q, pre);          // next tile's layer 2
      wgrad_swish<8>(Z, X, ACT_P, w, il, q, dW2, z1, g);
      db1[0] += g[0][0]; db1[1] += g[1][0];
      store_act(U, w, il, q, g);                           // zbar3 is dead after phase 6
      __syncthreads();
      STAMP(7);

      // ---- phase 8: dW1 ------------------------------------------------------
      wgrad<KT1>(U, H0, SM_P, w, il, q, dW1);
      __syncthreads();   // H0/ABAR/PART/X.. are rewritten by the next tile
      STAMP(8);
    }
  }

  // ---- epilogue (train): write this workgroup's gradient slab -------------
  if (MODE == MODE_TRAIN) {
    STAMP(9);
    float* slab = A.slabs + (int64_t)blockIdx.x * (A.n_params + 1);
    const int in_dim = A.in_dim;
    const int64_t oW1 = 0, ob1 = oW1 + (int64_t)HID * in_dim, oW2 = ob1 + HID, ob2 = oW2 + HID * HID,
                  oW3 = ob2 + HID, ob3 = oW3 + HID * HID, oW4 = ob3 + HID, ob4 = oW4 + (int64_t)d * HID;
    // dW2 / dW3: lane (k = 16kt+il, q), reg r -> row 32w+16it+4q+r
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int kt = 0; kt < 8; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 32 * w + 16 * it + 4 * q + r, col = 16 * kt + il;
          slab[oW2 + row * HID + col] = dW2[it][kt][r];
          slab[oW3 + row * HID + col] = dW3[it][kt][r];
        }
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int kt = 0; kt < KT1; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 32 * w + 16 * it + 4 * q + r, col = 16 * kt + il;
          if (col < in_dim) slab[oW1 + (int64_t)row * in_dim + col] = dW1[it][kt][r];
        }
    // dW4 held transposed: lane (o = 16ot+il, q), reg r -> feature 32w+16it+4q+r
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int ot = 0; ot < OT; ++ot)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int feat = 32 * w + 16 * it + 4 * q + r, o = 16 * ot + il;
          if (o < d) slab[oW4 + (int64_t)o * HID + feat] = dW4[it][ot][r];
        }
    // biases: sum the primal cotangents over the 16 sample lanes
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s1 = db1[it][r], s2 = db2[it][r], s3 = db3[it][r];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); s3 += __shfl_xor(s3, o, 64); }
        if (il == 0) {
          const int feat = 32 * w + 16 * it + 4 * q + r;
          slab[ob1 + feat] = s1; slab[ob2 + feat] = s2; slab[ob3 + feat] = s3;
        }
      }
    __syncthreads();
    if (tid < d) {
      float s = 0.f;
      for (int j = 0; j < 16; ++j) s += DB4[j * DPAD + tid];
      slab[ob4 + tid] = s;
    }
    if (tid < 16) RED[tid] = loss_acc;
    __syncthreads();
    if (tid == 0) {
      float s = 0.f;
      for (int j = 0; j < 16; ++j) s += RED[j];
      slab[A.n_params] = s;
    }
    STAMP(10);
    STAMP_FLUSH;
  }
}

// grads[p] = sum_wg slab[wg][p]; element n_params = loss sum (x inv_batch -> mean).
// 64 parameters x 4 slab-groups per block: each thread sums a quarter of the slabs
// (coalesced along p), the four partials meet in LDS — deterministic order.
// ADAM: the same thread then applies the fused Adam update to its parameter
// (single-GPU step: no all-reduce sits between the two), and thread 0 advances the
// Philox offset (nothing after this kernel reads it within the step).
template <bool ADAM>
__global__ void __launch_bounds__(256) k_slab_reduce(const float* __restrict__ slabs, int n_slabs, int64_t stride,
                                                     float* __restrict__ grads, int64_t n_params,
                                                     float* __restrict__ loss_sum, float inv_batch,
                                                     float* __restrict__ prm, float* __restrict__ m, float* __restrict__ v,
                                                     double lr, double b1, double b2, double eps,
                                                     const int64_t* __restrict__ step_dev, uint64_t* rng_advance) {
  __shared__ float part[4][64];
  const int px = threadIdx.x & 63, gy = threadIdx.x >> 6;
  const int64_t p = (int64_t)blockIdx.x * 64 + px;
  float s = 0.f;
  if (p <= n_params)
    for (int g = gy; g < n_slabs; g += 4) s += slabs[(int64_t)g * stride + p];
  part[gy][px] = s;
  __syncthreads();
  if (gy == 0 && p <= n_params) {
    s = (part[0][px] + part[1][px]) + (part[2][px] + part[3][px]);
    if (p < n_params) {
      if (grads) grads[p] = s;
      if (ADAM) {
        const int64_t st = step_dev[0];
        const double bc1 = 1.0 - pow(b1, (double)st), bc2 = 1.0 - pow(b2, (double)st);
        const float step_size = (float)(lr / bc1), bc2_sqrt = (float)sqrt(bc2);
        const float w1 = (float)(1.0 - b1), b2f = (float)b2, w2 = (float)(1.0 - b2);
        const float mm = m[p] + w1 * (s - m[p]);
        const float vv = v[p] * b2f + w2 * (s * s);
        prm[p] = prm[p] - step_size * (mm / (sqrtf(vv) / bc2_sqrt + (float)eps));
        m[p] = mm; v[p] = vv;
      }
    } else if (loss_sum) loss_sum[0] = s * inv_batch;
  }
  if (ADAM && rng_advance && blockIdx.x == 0 && threadIdx.x == 0) rng_advance[1] += 1;
}

// ============================================================ C ABI
static inline hipStream_t S(msgm_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
static const int MLP_MAX_GRID = 256;

template <int MODE, bool WIDE>
static void set_lds_attr() {
  static const int once = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mlp<MODE, WIDE>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, MlpLds<MODE, WIDE>::BYTES);
    return 0;
  }();
  (void)once;
}

template <int MODE>
static int launch_mlp(const MlpArgs& A, int grid, hipStream_t st) {
  const bool wide = A.in_dim > 16 || A.P.d > 16;
  constexpr size_t lds_wide = MlpLds<MODE, true>::BYTES, lds_narrow = MlpLds<MODE, false>::BYTES;
  if (wide) { set_lds_attr<MODE, true>(); hipLaunchKernelGGL((k_mlp<MODE, true>), dim3(grid), dim3(256), lds_wide, st, A); }
  else { set_lds_attr<MODE, false>(); hipLaunchKernelGGL((k_mlp<MODE, false>), dim3(grid), dim3(256), lds_narrow, st, A); }
  return msgm_check_launch();
}

static int fill_common(MlpArgs& A, const msgm_mlp_params_t* P, int64_t B) {
  if (!P || !P->W1 || !P->b1 || !P->W2 || !P->b2 || !P->W3 || !P->b3 || !P->W4 || !P->b4 || B <= 0) return MSGM_E_BADARG;
  if (P->d < 1 || P->d > 30 || (P->premodule != 0 && P->premodule != 1)) return MSGM_E_UNSUPPORTED;
  A.P = *P; A.B = B;
  A.in_dim = P->d + 1 + (P->premodule ? 1 : 0);
  A.in4 = (A.in_dim + 3) / 4; A.d4 = (P->d + 3) / 4;
  return MSGM_OK;
}

extern "C" {

int64_t msgm_mlp_num_params(int32_t d, int32_t premodule) {
  const int64_t in_dim = d + 1 + (premodule ? 1 : 0);
  return HID * in_dim + HID + 2 * ((int64_t)HID * HID + HID) + (int64_t)d * HID + d;
}

size_t msgm_mlp_ssm_workspace(int32_t d, int32_t premodule) {
  return (size_t)MLP_MAX_GRID * (size_t)(msgm_mlp_num_params(d, premodule) + 1) * sizeof(float);
}

int msgm_mlp_forward(const msgm_mlp_params_t* P, const float* y, const float* t, float* a, int64_t B,
                     msgm_stream_t stream) {
  MlpArgs A{};
  int rc = fill_common(A, P, B);
  if (rc) return rc;
  if (!y || !t || !a) return MSGM_E_BADARG;
  A.y = y; A.t = t; A.out = a;
  const int64_t tiles = (B + 31) / 32;
  const int64_t cap = 2 * MLP_MAX_GRID;          // forward modes fit two workgroups per CU
  return launch_mlp<MODE_FWD>(A, (int)(tiles < cap ? tiles : cap), S(stream));
}

int msgm_mlp_em_step(const msgm_mlp_params_t* P, float* x, int64_t B, const msgm_sde_t* sde, float t, float delta,
                     float lmbd, const float* z, const uint64_t* rng, uint64_t rng_step, msgm_stream_t stream) {
  MlpArgs A{};
  int rc = fill_common(A, P, B);
  if (rc) return rc;
  if (!x || !sde || (!z && !rng)) return MSGM_E_BADARG;
  if (sde->kind != MSGM_SDE_SGM) return MSGM_E_UNSUPPORTED;
  A.y = x; A.out = x; A.t = nullptr; A.t_scalar = t;
  A.b0 = sde->beta_min; A.b1 = sde->beta_max; A.T = sde->T;
  A.delta = delta; A.sqrt_delta = (float)sqrt((double)delta); A.lmbd = lmbd;
  A.z = z; A.rng = rng; A.rng_step = rng_step;
  const int64_t tiles = (B + 31) / 32;
  const int64_t cap = 2 * MLP_MAX_GRID;
  return launch_mlp<MODE_EM>(A, (int)(tiles < cap ? tiles : cap), S(stream));
}

int msgm_mlp_ssm_partial(const msgm_mlp_params_t* P, const float* y, const float* t, const float* v, const float* u,
                         const float* cst, int64_t B, const msgm_sde_t* sde, float inv_batch, float* loss_per,
                         void* workspace, size_t workspace_bytes, int32_t* n_slabs, msgm_stream_t stream) {
  MlpArgs A{};
  int rc = fill_common(A, P, B);
  if (rc) return rc;
  if (!y || !t || !v || !sde || !workspace || !n_slabs) return MSGM_E_BADARG;
  if (sde->kind != MSGM_SDE_SGM && !u) return MSGM_E_UNSUPPORTED;      // MSGM needs u = G(y)^T v (msgm_ssm_terms)
  if (workspace_bytes < msgm_mlp_ssm_workspace(P->d, P->premodule)) return MSGM_E_WORKSPACE;
  A.y = y; A.t = t; A.v = v; A.u = u; A.cst = cst;
  A.b0 = sde->beta_min; A.b1 = sde->beta_max; A.T = sde->T;
  A.inv_batch = inv_batch; A.loss_per = loss_per; A.slabs = reinterpret_cast<float*>(workspace);
  A.n_params = msgm_mlp_num_params(P->d, P->premodule);
  const int64_t tiles = (B + 15) / 16;
  const int grid = (int)(tiles < MLP_MAX_GRID ? tiles : MLP_MAX_GRID);
  *n_slabs = grid;
  return launch_mlp<MODE_TRAIN>(A, grid, S(stream));
}

int msgm_mlp_ssm_reduce(int32_t d, int32_t premodule, const void* workspace, int32_t n_slabs, float inv_batch,
                        float* grads, float* loss_sum, msgm_stream_t stream) {
  if (!workspace || !grads || n_slabs < 1 || n_slabs > MLP_MAX_GRID) return MSGM_E_BADARG;
  const int64_t n_params = msgm_mlp_num_params(d, premodule);
  hipLaunchKernelGGL(k_slab_reduce<false>, dim3((unsigned)((n_params + 1 + 63) / 64)), dim3(256), 0, S(stream),
                     reinterpret_cast<const float*>(workspace), n_slabs, n_params + 1, grads, n_params, loss_sum, inv_batch,
                     (float*)nullptr, (float*)nullptr, (float*)nullptr, 0.0, 0.0, 0.0, 0.0, (const int64_t*)nullptr,
                     (uint64_t*)nullptr);
  return msgm_check_launch();
}

int msgm_mlp_ssm_reduce_adam(int32_t d, int32_t premodule, const void* workspace, int32_t n_slabs, float inv_batch,
                             float* grads, float* loss_sum, float* params, float* m, float* v, double lr, double beta1,
                             double beta2, double eps, const int64_t* step_dev, uint64_t* rng_advance,
                             msgm_stream_t stream) {
  if (!workspace || !params || !m || !v || !step_dev || n_slabs < 1 || n_slabs > MLP_MAX_GRID) return MSGM_E_BADARG;
  const int64_t n_params = msgm_mlp_num_params(d, premodule);
  hipLaunchKernelGGL(k_slab_reduce<true>, dim3((unsigned)((n_params + 1 + 63) / 64)), dim3(256), 0, S(stream),
                     reinterpret_cast<const float*>(workspace), n_slabs, n_params + 1, grads, n_params, loss_sum, inv_batch,
                     params, m, v, lr, beta1, beta2, eps, step_dev, rng_advance);
  return msgm_check_launch();
}

int msgm_mlp_ssm_grad(const msgm_mlp_params_t* P, const float* y, const float* t, const float* v, const float* u,
                      const float* cst, int64_t B, const msgm_sde_t* sde, float inv_batch, float* grads, float* loss_per,
                      float* loss_sum, void* workspace, size_t workspace_bytes, msgm_stream_t stream) {
  if (!grads) return MSGM_E_BADARG;
  int32_t n_slabs = 0;
  int rc = msgm_mlp_ssm_partial(P, y, t, v, u, cst, B, sde, inv_batch, loss_per, workspace, workspace_bytes, &n_slabs, stream);
  if (rc) return rc;
  return msgm_mlp_ssm_reduce(P->d, P->premodule, workspace, n_slabs, inv_batch, grads, loss_sum, stream);
}

#ifdef MLP_STAMPS
int msgm_debug_stamps(unsigned long long* out_host) {
  return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 12) == hipSuccess ? 0 : -4;
}
#endif

}  // extern "C"
