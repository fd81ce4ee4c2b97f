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

// Workgroup barrier for LDS hand-offs inside the tile loop.  __syncthreads() also drains vmcnt, i.e. it would wait for
// the weight fragments and the next tile's inputs that are deliberately left in flight across phase boundaries;
// LDS traffic completes in order, so lgkmcnt(0) + s_barrier is all an LDS producer/consumer pair needs.  hipcc's
// own waitcnt insertion still guards the first use of every in-flight load.
#ifndef MLP_FULL_BARRIER
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#else
__device__ __forceinline__ void lds_barrier() { __syncthreads(); }
#endif

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// row pitch of a gradient slab: n_params + 1 (loss) rounded up to 4 floats, so the reduction reads 16-B vectors
__host__ __device__ static inline int64_t slab_stride(int64_t n_params) { return (n_params + 1 + 3) & ~(int64_t)3; }

// ---- LDS carve (floats), per (mode, width) ------------------------------------
// NARROW nets (in_dim <= 16 and d <= 16) use 20-float small rows and 16 padded outputs; the forward / sampler
// modes drop the backward-only buffers — 67 KB instead of 134 KB, so TWO workgroups share a CU there and each
// one's barrier bubbles are filled by the other's MFMAs.
// TINY (inference modes, in_dim <= 4 and d <= 4 — the C1/C2 nets): the small rows shrink to 8 floats, only 4 rows of
// W4 are kept (the MFMA's other 12 output rows are zeros and are fed from registers), so the carve drops from 74 KB
// to 52 KB and THREE workgroups share a CU.
template <int MODE, bool WIDE, int NW, bool TINY = false>
struct MlpLds {
  static_assert(!TINY || (MODE != 2 && !WIDE), "the tiny carve is for the narrow inference modes");
  static constexpr int SMP = WIDE ? 36 : (TINY ? 8 : 20);   // pitch of h0 / abar / W1 rows
  static constexpr int PP = TINY ? 8 : SMP;        // pitch of a layer-4 partial row
  static constexpr int DP = WIDE ? 32 : 16;        // padded output width
  static constexpr int DROWS = TINY ? 4 : DP;      // rows of W4 held in LDS
  static constexpr bool TRAIN = MODE == 2;
  static constexpr int SPT = TRAIN ? 16 : 32;      // samples per tile
  static constexpr int DMAX = WIDE ? 32 : (TINY ? 4 : 16);   // raw rows are packed (pitch d), sized for d <= DMAX
  // one raw input buffer: the tile's y | v | u chunks, t and cst, copied verbatim from global one tile ahead
  static constexpr int RY = 0;
  static constexpr int RV = RY + SPT * DMAX;
  static constexpr int RU = RV + (TRAIN ? SPT * DMAX : 0);
  static constexpr int RT = RU + (TRAIN ? SPT * DMAX : 0);
  static constexpr int RC = RT + 32;
  static constexpr int RAWN = RC + 16;
  static constexpr int X = 0;
  static constexpr int Y = X + 32 * ACT_P;
  static constexpr int Z = Y + 32 * ACT_P;
  static constexpr int U = Z + (TRAIN ? 32 * ACT_P : 0);
  static constexpr int H0 = U + (TRAIN ? 32 * ACT_P : 0);
  static constexpr int ABAR = H0 + 2 * 32 * SMP;   // h0 is double-buffered (built one tile ahead)
  static constexpr int PART = ABAR + (TRAIN ? 32 * SMP : 0);
  static constexpr int W1 = PART + NW * 32 * PP;    // one layer-4 K-slice per wave
  static constexpr int W4 = W1 + HID * SMP;
  static constexpr int B1 = W4 + DROWS * ACT_P;
  static constexpr int B2 = B1 + HID;
  static constexpr int B3 = B2 + HID;
  static constexpr int B4 = B3 + HID;
  static constexpr int DB4 = B4 + 32;
  static constexpr int RED = DB4 + (TRAIN ? 16 * DP : 0);
  static constexpr int RAW = RED + 64;
  static constexpr int FLOATS = RAW + 2 * RAWN;
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
  int n_steps; const float* ts;   // EM loop: n_steps > 1 steps in ONE launch, time of step i = ts[i] (device array)
  // train
  float inv_batch;
  float* loss_per; float* slabs;   // slabs: [gridDim.x][slab_stride], element n_params = loss sum; stride = 4-aligned
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
// Buffer loads (SGPR descriptor + ONE per-lane 32-bit offset + a compile-time scalar offset per fragment) keep the
// address arithmetic out of the vector registers: with flat loads hipcc hoists a 64-bit address per (matrix, k-group)
// out of the persistent tile loop — 32+ VGPRs — and spills.
typedef __amdgpu_buffer_rsrc_t wrsrc_t;
__device__ __forceinline__ wrsrc_t make_wrsrc(const float* W) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, HID * HID * 4, 0x00020000);
}
template <bool TR>
__device__ __forceinline__ f32x4 load_afrag(wrsrc_t W, int fb, int il, int q, int it, int g) {
  if (!TR) {
    const int voff = ((fb + il) * HID + 4 * q) * 4;
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(W, voff, (16 * it * HID + 16 * g) * 4, 0));
  }
  const int voff = (4 * q * HID + fb + il) * 4;
  const int so = (16 * g * HID + 16 * it) * 4;
  f32x4 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(W, voff + r * HID * 4, so, 0));
  return v;
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
#ifndef MLP_STAGGER
#define MLP_STAGGER 1
#endif
struct WPre { f32x4 a[PF]; };     // fragments j = 0..PF-1 of the next gemm (j = it*8 + g: row tile 0 first)

template <bool TR>
__device__ __forceinline__ void prefetch_w(wrsrc_t W, int fb, int il, int q, WPre& pre) {
#pragma unroll
  for (int j = 0; j < PF; ++j) pre.a[j] = load_afrag<TR>(W, fb, il, q, j >> 3, j & 7);
}

// acc[it][ck] += sum_k A[it][k] * buf[ck*16+il][k]   (K = 128, B operand from LDS), row tile 0 first, then
// row tile 1.  Software-pipelined by hand: the LDS reads of step j+1 and the global loads of step j+PF are
// issued ahead of the 8 MFMAs of step j; sched_barrier keeps hipcc from hoisting every load (which spills).
// HOOK: while the matrix pipe works on row tile 1, the VALU applies bias + Swish to the finished row tile 0
// (an MFMA occupies the vector issue port for only 8 of its 32 cycles):
//   HOOK 1 (train): h[0][0] = s(z), h[0][1] = s'(z) zdot;   HOOK 2 (inference): both column halves are primal.
template <bool TR, int HOOK, int IT>
__device__ __forceinline__ void gemm128(wrsrc_t W, const WPre& pre, const float* buf, int fb, int il,
                                        int q, f32x4 (&acc)[IT][2], const float* bias_lds, f32x4 (&h)[IT][2]) {
  constexpr int NJ = 8 * IT;
  constexpr bool OVL = MLP_OVL_FWD && HOOK && IT == 2;   // a second row tile to hide the first one's Swish under
  f32x4 ring[PF + 1];
#pragma unroll
  for (int j = 0; j < PF; ++j) ring[j] = pre.a[j];
  const float* bp0 = buf + il * ACT_P + 4 * q;
  const float* bp1 = buf + (16 + il) * ACT_P + 4 * q;
  f32x4 bn0 = *reinterpret_cast<const f32x4*>(bp0), bn1 = *reinterpret_cast<const f32x4*>(bp1);
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int it = j >> 3;
    const f32x4 b0 = bn0, b1 = bn1;
    if (j + PF < NJ) ring[(j + PF) % (PF + 1)] = load_afrag<TR>(W, fb, il, q, (j + PF) >> 3, (j + PF) & 7);
    if (j + 1 < NJ) {
      bn0 = *reinterpret_cast<const f32x4*>(bp0 + 16 * ((j + 1) & 7));
      bn1 = *reinterpret_cast<const f32x4*>(bp1 + 16 * ((j + 1) & 7));
    }
    if (OVL && j == 8) {        // row tile 0 is complete: bias
      const f32x4 bb = *reinterpret_cast<const f32x4*>(bias_lds + fb + 4 * q);
      acc[0][0] += bb;
      if (HOOK == 2) acc[0][1] += bb;
    }
    if (OVL && j >= 8 && j < 12) {   // one register per step
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
      acc[it][0] = mfma16(a[r], b0[r], acc[it][0]);
      acc[it][1] = mfma16(a[r], b1[r], acc[it][1]);
    }
    if (MLP_HINTS && OVL && j >= 8 && j < 12) {
#pragma unroll
      for (int k = 0; k < 4; ++k) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x002, 6, 0); }
    }
    if (MLP_SB) __builtin_amdgcn_sched_barrier(0);
  }
}
// row tiles whose bias + Swish the caller still has to apply after gemm128<.., HOOK != 0, IT>
template <int IT> struct GemmTail { static constexpr int FIRST = (MLP_OVL_FWD && IT == 2) ? 1 : 0; };

// bias + (dual) Swish of row tiles [first, IT): train -> h = (s(z), s'(z) zdot); inference -> both halves primal
template <bool TRAIN, int IT>
__device__ __forceinline__ void bias_swish(f32x4 (&z)[IT][2], f32x4 (&h)[IT][2], const float* bias_lds, int fb, int q, int first) {
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    if (it < first) continue;
    const f32x4 bb = *reinterpret_cast<const f32x4*>(bias_lds + fb + 16 * it + 4 * q);
    z[it][0] += bb;
    if (!TRAIN) z[it][1] += bb;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (TRAIN) {
        float s0, s1, s2; swish012(z[it][0][r], s0, s1, s2);
        h[it][0][r] = s0; h[it][1][r] = s1 * z[it][1][r];
      } else { h[it][0][r] = swish0(z[it][0][r]); h[it][1][r] = swish0(z[it][1][r]); }
    }
  }
}

// store this wave's C-layout tile pair to an activation buffer ([sample][feature])
template <int IT>
__device__ __forceinline__ void store_act(float* buf, int fb, int il, int q, const f32x4 (&h)[IT][2]) {
#pragma unroll
  for (int it = 0; it < IT; ++it)
#pragma unroll
    for (int ck = 0; ck < 2; ++ck)
      *reinterpret_cast<f32x4*>(buf + (ck * 16 + il) * ACT_P + fb + 16 * it + 4 * q) = h[it][ck];
}

// dW[it][kt] += sum_n ZB[feat][n] * HB[k][n] over the 32 dual columns.
// A (lane il = feature) and B (lane il = k) are read as b32 from [sample][feature]
// buffers, one (ck,s) step ahead of the MFMAs that consume them.
// SWISH: the dual Swish backward of the NEXT pointwise stage is hidden under the MFMAs: step st also turns
// (gP,gT) -> (zbarP, zbarT), in place, for register (it = st>>2, r = st&3).
template <int KT, int IT, bool SWISH>
__device__ __forceinline__ void wgrad_core(const float* zb, const float* hb, int hb_pitch, int fb, int il, int q,
                                           f32x4 (&dW)[IT][KT], const f32x4 (*z)[2], f32x4 (*g)[2]) {
  if (KT * IT <= 4) {
    // thin products (dW1, dW4: one or two MFMAs per step): a one-step-ahead pipeline would expose one LDS round trip
    // per step (~8 x 130 cycles for 8 x 32 cycles of MFMA), so ALL operands are fetched first
    float pa[8][IT], pb[8][KT];
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      const int n = (st >> 2) * 16 + 4 * (st & 3) + q;
#pragma unroll
      for (int it = 0; it < IT; ++it) pa[st][it] = zb[n * ACT_P + fb + 16 * it + il];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) pb[st][kt] = hb[n * hb_pitch + 16 * kt + il];
    }
    if (SWISH) {
#pragma unroll
      for (int st = 0; st < 4 * IT; ++st) {
        const int it = st >> 2, r = st & 3;
        float s0, s1, s2;
        swish012(z[it][0][r], s0, s1, s2);
        const float gt = g[it][1][r];
        g[it][0][r] = g[it][0][r] * s1 + gt * (s2 * z[it][1][r]);
        g[it][1][r] = gt * s1;
      }
    }
#pragma unroll
    for (int st = 0; st < 8; ++st)
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int it = 0; it < IT; ++it) dW[it][kt] = mfma16(pa[st][it], pb[st][kt], dW[it][kt]);
    return;
  }
  float na[IT], nb[KT];
  {
    const int n = q;
#pragma unroll
    for (int it = 0; it < IT; ++it) na[it] = zb[n * ACT_P + fb + 16 * it + il];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) nb[kt] = hb[n * hb_pitch + 16 * kt + il];
  }
#pragma unroll
  for (int st = 0; st < 8; ++st) {
    float a[IT], b[KT];
#pragma unroll
    for (int it = 0; it < IT; ++it) a[it] = na[it];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) b[kt] = nb[kt];
    if (st + 1 < 8) {
      const int n = ((st + 1) >> 2) * 16 + 4 * ((st + 1) & 3) + q;
#pragma unroll
      for (int it = 0; it < IT; ++it) na[it] = zb[n * ACT_P + fb + 16 * it + il];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) nb[kt] = hb[n * hb_pitch + 16 * kt + il];
    }
    if (SWISH && MLP_OVL_BWD && st < 4 * IT) {
      const int it = st >> 2, r = st & 3;
      float s0, s1, s2;
      swish012(z[it][0][r], s0, s1, s2);
      const float gt = g[it][1][r];
      g[it][0][r] = g[it][0][r] * s1 + gt * (s2 * z[it][1][r]);     // in place: g becomes (zbarP, zbarT)
      g[it][1][r] = gt * s1;
    }
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int it = 0; it < IT; ++it) dW[it][kt] = mfma16(a[it], b[kt], dW[it][kt]);
    if (MLP_HINTS && SWISH) {
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x002, 3, 0); }
    }
    if (MLP_SB) __builtin_amdgcn_sched_barrier(0);
  }
  if (SWISH && !MLP_OVL_BWD) {
#pragma unroll
    for (int st = 0; st < 4 * IT; ++st) {
      const int it = st >> 2, r = st & 3;
      float s0, s1, s2;
      swish012(z[it][0][r], s0, s1, s2);
      const float gt = g[it][1][r];
      g[it][0][r] = g[it][0][r] * s1 + gt * (s2 * z[it][1][r]);
      g[it][1][r] = gt * s1;
    }
  }
}
// dual Swish backward in registers, in place: (gP,gT) cotangents of (hP,hT) -> cotangents of (zP,zT)
template <int IT>
__device__ __forceinline__ void swish_bwd_inplace(const f32x4 (&z)[IT][2], f32x4 (&g)[IT][2]) {
#pragma unroll
  for (int it = 0; it < IT; ++it)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float s0, s1, s2;
      swish012(z[it][0][r], s0, s1, s2);
      const float gt = g[it][1][r];
      g[it][0][r] = g[it][0][r] * s1 + gt * (s2 * z[it][1][r]);
      g[it][1][r] = gt * s1;
    }
}

template <int KT, int IT>
__device__ __forceinline__ void wgrad(const float* zb, const float* hb, int hb_pitch, int fb, int il, int q, f32x4 (&dW)[IT][KT]) {
  wgrad_core<KT, IT, false>(zb, hb, hb_pitch, fb, il, q, dW, nullptr, nullptr);
}
template <int KT, int IT>
__device__ __forceinline__ void wgrad_swish(const float* zbuf, const float* hb, int hb_pitch, int fb, int il, int q,
                                            f32x4 (&dW)[IT][KT], const f32x4 (&z)[IT][2], f32x4 (&g)[IT][2]) {
  wgrad_core<KT, IT, true>(zbuf, hb, hb_pitch, fb, il, q, dW, z, g);
}

template <int MODE, bool WIDE, int NW, bool TINY = false>
__global__ void __launch_bounds__(64 * NW, 1) k_mlp(MlpArgs A) {
  constexpr int NT = 64 * NW;         // threads
  constexpr int IT = 8 / NW;          // 16-feature row tiles per wave: 4 waves x 2 or 8 waves x 1
  constexpr int FW = 16 * IT;         // features owned by a wave
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int KT1 = WIDE ? 2 : 1;   // k-tiles of the first layer's input (in_dim <= 16 / 32)
  constexpr int OT = WIDE ? 2 : 1;    // o-tiles of the last layer's output (d <= 16 / 32)
  constexpr int SPT = (MODE == MODE_TRAIN) ? 16 : 32;   // samples per tile
  const int tid = threadIdx.x;
  const int w = tid >> 6, lane = tid & 63, il = lane & 15, q = lane >> 4;
  const int d = A.P.d;
  const int fb = FW * w;              // this wave's first feature
  using LO = MlpLds<MODE, WIDE, NW, TINY>;
  constexpr int SM_P = LO::SMP, DPAD = LO::DP, PP = LO::PP, DROWS = LO::DROWS;
  float* X = lds + LO::X; float* Y = lds + LO::Y; float* Z = lds + LO::Z; float* U = lds + LO::U;
  float* H0 = lds + LO::H0; float* ABAR = lds + LO::ABAR; float* PART = lds + LO::PART;
  float* W1s = lds + LO::W1; float* W4s = lds + LO::W4;
  float* B1s = lds + LO::B1; float* B2s = lds + LO::B2; float* B3s = lds + LO::B3; float* B4s = lds + LO::B4;
  float* DB4 = lds + LO::DB4; float* RED = lds + LO::RED;

  // persistent accumulators (train)
  f32x4 dW2[IT][8], dW3[IT][8], dW1[IT][KT1], dW4[IT][OT], db1[IT], db2[IT], db3[IT];
  float loss_acc = 0.f;
  if (MODE == MODE_TRAIN) {
#pragma unroll
    for (int it = 0; it < IT; ++it) {
#pragma unroll
      for (int k = 0; k < 8; ++k) { dW2[it][k] = f32x4{0, 0, 0, 0}; dW3[it][k] = f32x4{0, 0, 0, 0}; }
#pragma unroll
      for (int k = 0; k < KT1; ++k) dW1[it][k] = f32x4{0, 0, 0, 0};
#pragma unroll
      for (int k = 0; k < OT; ++k) dW4[it][k] = f32x4{0, 0, 0, 0};
      db1[it] = f32x4{0, 0, 0, 0}; db2[it] = f32x4{0, 0, 0, 0}; db3[it] = f32x4{0, 0, 0, 0};
    }
  }

  // ---- input pipeline: the tile's y / v / u / t / cst chunks (contiguous in global memory) are fetched into
  // registers ONE TILE AHEAD (raw_issue), parked in LDS a phase later (raw_commit) and turned into the layer-1
  // operand h0 at the end of the previous tile (build_h0) — no global-load latency sits on the tile's critical path.
  constexpr int NR = (SPT * LO::DMAX + NT - 1) / NT;
  float* RAW0 = lds + LO::RAW;
  float pf_y[NR], pf_v[NR], pf_u[NR], pf_t = 0.f, pf_c = 0.f;
  const int64_t etot = A.B * d;
  auto raw_issue = [&](int64_t tl, int step = 0) {
    const int64_t e0 = tl * SPT * d;                 // tl past the last tile => e0 >= etot => zeros
    const int cnt = SPT * d;
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const int e = tid + NT * k;
      const bool ok = e < cnt && e0 + e < etot;
      // EM loop: x was rewritten by THIS workgroup a sweep ago — device-scope load so no stale L1 line is served
      if (MODE == MODE_EM) pf_y[k] = ok ? __hip_atomic_load(A.y + e0 + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
      else pf_y[k] = ok ? A.y[e0 + e] : 0.f;
      if (MODE == MODE_TRAIN) {
        pf_v[k] = ok ? A.v[e0 + e] : 0.f;
        pf_u[k] = (ok && A.u) ? A.u[e0 + e] : 0.f;
      }
    }
    if (tid < SPT) {
      const int64_t smp = tl * SPT + tid;
      pf_t = smp < A.B ? (A.t ? A.t[smp] : ((MODE == MODE_EM && A.ts) ? A.ts[step] : A.t_scalar)) : 0.f;
      if (MODE == MODE_TRAIN) pf_c = (smp < A.B && A.u && A.cst) ? A.cst[smp] : 0.f;
    }
  };
  auto raw_commit = [&](float* R) {
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const int e = tid + NT * k;
      if (e < SPT * LO::DMAX) {
        R[LO::RY + e] = pf_y[k];
        if (MODE == MODE_TRAIN) { R[LO::RV + e] = pf_v[k]; R[LO::RU + e] = pf_u[k]; }
      }
    }
    if (tid < SPT) { R[LO::RT + tid] = pf_t; if (MODE == MODE_TRAIN) R[LO::RC + tid] = pf_c; }
  };
  // h0 (and its tangent) of tile tl from its raw buffer; run by SPT threads, one per sample row l.
  // premodule none: h0 = [y, t], h0dot = [v, 0]                     NN.py:113-119
  // NormalizeLogRadius: h0 = [y/r, log r, t], r = |y| + 1e-6          NN.py:56-70
  auto build_h0 = [&](float* H0b, const float* R, int64_t tl, int l) {   // l = row (sample) of the tile
    const int64_t smp = tl * SPT + l;
    const bool live = smp < A.B;
    float* hp = H0b + l * SM_P;
    float* ht = H0b + (16 + l) * SM_P;   // tangent row (train only)
    float tt = 0.f;
    if (live) {
      tt = R[LO::RT + l];
      if (MODE == MODE_EM) tt = A.T - tt;                                 // s = T - t  SDEs.py:556-557
    }
    const float* yr = R + LO::RY + l * d;
    const float* vr = R + LO::RV + l * d;
    if (A.P.premodule == 0) {
      for (int i = 0; i < d; ++i) {
        hp[i] = yr[i];
        if (MODE == MODE_TRAIN) ht[i] = vr[i];
      }
      hp[d] = tt;
      if (MODE == MODE_TRAIN) ht[d] = 0.f;
    } else {
      float ss = 0.f, yv = 0.f;
      for (int i = 0; i < d; ++i) {
        const float yi = live ? yr[i] : 1.0f;
        ss += yi * yi;
        if (MODE == MODE_TRAIN) yv += yi * vr[i];
      }
      const float nr = sqrtf(ss);
      const float r = nr + 1e-6f;
      const float rdot = yv / nr;                                          // d|y| along v
      for (int i = 0; i < d; ++i) {
        const float yi = live ? yr[i] : 1.0f;
        hp[i] = yi / r;
        if (MODE == MODE_TRAIN) ht[i] = vr[i] / r - yi * rdot / (r * r);
      }
      hp[d] = logf(r);
      hp[d + 1] = tt;
      if (MODE == MODE_TRAIN) { ht[d] = rdot / r; ht[d + 1] = 0.f; }
    }
  };

  // ---- prologue.  The first tile's inputs and the first weight fragments go out first; the one-time staging of
  // the small layers and biases is written as fixed-trip loops with all global loads ahead of the LDS stores, so its
  // latency is ONE L2 round trip, not one per element (at the C2 sampler size a workgroup only has 4 tiles to
  // amortise this over).
  raw_issue(blockIdx.x);
  const wrsrc_t R2 = make_wrsrc(A.P.W2), R3 = make_wrsrc(A.P.W3);
  WPre pre;
  prefetch_w<false>(R2, fb, il, q, pre);
  {
    constexpr int N1 = (HID * SM_P + NT - 1) / NT, N4 = (DROWS * ACT_P + NT - 1) / NT;
    float v1[N1], v4[N4], vb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < N1; ++k) {
      const int i = tid + NT * k, r = i / SM_P, c = i - r * SM_P;
      v1[k] = (i < HID * SM_P && c < A.in_dim) ? A.P.W1[r * A.in_dim + c] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < N4; ++k) {
      const int i = tid + NT * k, r = i / ACT_P, c = i - r * ACT_P;
      v4[k] = (i < DROWS * ACT_P && r < d && c < HID) ? A.P.W4[r * HID + c] : 0.f;
    }
    if (tid < HID) { vb[0] = A.P.b1[tid]; vb[1] = A.P.b2[tid]; vb[2] = A.P.b3[tid]; }
    if (tid < d) vb[3] = A.P.b4[tid];
#pragma unroll
    for (int k = 0; k < N1; ++k) { const int i = tid + NT * k; if (i < HID * SM_P) W1s[i] = v1[k]; }
#pragma unroll
    for (int k = 0; k < N4; ++k) { const int i = tid + NT * k; if (i < DROWS * ACT_P) W4s[i] = v4[k]; }
    if (tid < HID) { B1s[tid] = vb[0]; B2s[tid] = vb[1]; B3s[tid] = vb[2]; }
    if (tid < 32) B4s[tid] = vb[3];
    if (MODE == MODE_TRAIN) {
      for (int i = tid; i < 16 * DPAD; i += NT) DB4[i] = 0.f;
      for (int i = tid; i < 32 * SM_P; i += NT) ABAR[i] = 0.f;
    }
    for (int i = tid; i < 2 * 32 * SM_P; i += NT) H0[i] = 0.f;
  }
  raw_commit(RAW0);
  __syncthreads();
  if (tid < SPT) build_h0(H0, RAW0, blockIdx.x, tid);
  __syncthreads();

  // layer 1 (K = in_dim, weights from LDS) of the tile whose h0 sits in H0b: z1 stays in registers (train: needed
  // again by the Swish backward), h1 goes to `dst`.  It runs one tile AHEAD — in the prologue for the first tile, then
  // at the tail of every tile for the next one, into the activation buffer that tile no longer needs — so the tile
  // loop starts directly with the layer-2 gemm and has no layer-1 phase / barrier of its own.
  constexpr bool TRN = MODE == MODE_TRAIN;
  f32x4 z1[IT][2];
  auto layer1 = [&](const float* H0b, float* dst) {
    f32x4 h1[IT][2];
#pragma unroll
    for (int it = 0; it < IT; ++it) { z1[it][0] = f32x4{0, 0, 0, 0}; z1[it][1] = f32x4{0, 0, 0, 0}; }
    for (int s = 0; s < A.in4; ++s) {
      const float bP = H0b[il * SM_P + 4 * s + q];
      const float bT = H0b[(16 + il) * SM_P + 4 * s + q];
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        const float a = W1s[(fb + 16 * it + il) * SM_P + 4 * s + q];
        z1[it][0] = mfma16(a, bP, z1[it][0]); z1[it][1] = mfma16(a, bT, z1[it][1]);
      }
    }
    bias_swish<TRN, IT>(z1, h1, B1s, fb, q, 0);
    store_act<IT>(dst, fb, il, q, h1);
  };
  layer1(H0, X);
  __syncthreads();

  STAMP_DECL
  const int64_t n_tiles = (A.B + SPT - 1) / SPT;
  const float ca = 1.0f - 0.5f * A.lmbd;
  int cur = 0;      // parity of the tile: selects the h0 / raw buffers and which of X / Y holds h1
  // Waves w and w + NW/2 share a SIMD and would otherwise walk through the backward phases in lockstep (both in their
  // LDS-latency head, both in their VALU tail).  The second one runs its weight-gradient product BEFORE its dgrad, so
  // one wave's heads and tails fall under the other's MFMAs.
  const bool late = MLP_STAGGER && NW == 8 && w >= NW / 2;

  // Work items are (step, tile) with the tile index fastest; training / forward / single-step EM have one step.
  // EM loop (n_steps > 1): rows never interact (sde_scheme.py:82-86), so a workgroup takes its tiles through ALL
  // steps in one launch — weights and biases are staged once, x only travels to L2 and back.  The host guarantees
  // >= 2 tiles per workgroup, so the item prefetched one ahead is always a tile whose previous step is complete.
  const int n_steps = (MODE == MODE_EM && A.n_steps > 1) ? A.n_steps : 1;
  int step = 0;
  for (int64_t tile = blockIdx.x; tile < n_tiles;) {
    const int64_t s_base = tile * SPT;
    int64_t ntile = tile + gridDim.x;           // the next item
    int nstep = step;
    if (ntile >= n_tiles) { ++nstep; ntile = nstep < n_steps ? (int64_t)blockIdx.x : n_tiles; }
    const float* H0c = H0 + cur * 32 * SM_P;
    float* H0n = H0 + (cur ^ 1) * 32 * SM_P;
    const float* Rc = RAW0 + cur * LO::RAWN;
    float* Rn = RAW0 + (cur ^ 1) * LO::RAWN;
    float* Xc = cur ? Y : X;          // h1 of this tile (written one tile ahead)
    float* Yc = cur ? X : Y;          // h2 of this tile; after phase 6 (train) / 3 (inference): h1 of the next tile
    f32x4 z2[IT][2], z3[IT][2], h[IT][2];
    cur ^= 1;

    // ---- phase 2: layer 2 --------------------------------------------------
#pragma unroll
    for (int it = 0; it < IT; ++it) { z2[it][0] = f32x4{0, 0, 0, 0}; z2[it][1] = f32x4{0, 0, 0, 0}; }
    if (MODE != MODE_TRAIN) raw_issue(ntile, nstep);
    gemm128<false, (TRN ? 1 : 2), IT>(R2, pre, Xc, fb, il, q, z2, B2s, h);
    prefetch_w<false>(R3, fb, il, q, pre);         // head of the next gemm's weights
    bias_swish<TRN, IT>(z2, h, B2s, fb, q, GemmTail<IT>::FIRST);
    store_act<IT>(Yc, fb, il, q, h);
    if (MODE != MODE_TRAIN) raw_commit(Rn);
    lds_barrier();
    STAMP(2);

    // ---- phase 3: layer 3, then this wave's K-slice of layer 4 -------------
#pragma unroll
    for (int it = 0; it < IT; ++it) { z3[it][0] = f32x4{0, 0, 0, 0}; z3[it][1] = f32x4{0, 0, 0, 0}; }
    if (MODE != MODE_TRAIN) { if (tid < SPT) build_h0(H0n, Rn, ntile, tid); }   // next item's layer-1 operand
    else raw_issue(ntile);                                   // train: next tile's inputs, in flight during the gemm
    gemm128<false, (TRN ? 1 : 2), IT>(R3, pre, Yc, fb, il, q, z3, B3s, h);
    if (MODE == MODE_TRAIN) prefetch_w<true>(R3, fb, il, q, pre);  // W3^T for dgrad
    else prefetch_w<false>(R2, fb, il, q, pre);                    // next tile's layer 2
    bias_swish<TRN, IT>(z3, h, B3s, fb, q, GemmTail<IT>::FIRST);
    if (MODE == MODE_TRAIN) store_act<IT>(Z, fb, il, q, h);   // h3 is the dW4 operand later
    {
      // partial[o][col] = sum_{k in this wave's FW features} W4[o][k] h3[k][col]; B operand = registers
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) {
        f32x4 p0 = {0, 0, 0, 0}, p1 = {0, 0, 0, 0};
#pragma unroll
        for (int it = 0; it < IT; ++it) {
          f32x4 a4 = {0.f, 0.f, 0.f, 0.f};                  // output rows >= DROWS are zero padding
          if (16 * ot + il < DROWS) a4 = *reinterpret_cast<const f32x4*>(W4s + (16 * ot + il) * ACT_P + fb + 16 * it + 4 * q);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            p0 = mfma16(a4[r], h[it][0][r], p0);
            p1 = mfma16(a4[r], h[it][1][r], p1);
          }
        }
        if (16 * ot + 4 * q < PP) {                         // (tiny carve: only the first 8 output columns exist)
          *reinterpret_cast<f32x4*>(PART + (w * 32 + il) * PP + 16 * ot + 4 * q) = p0;
          *reinterpret_cast<f32x4*>(PART + (w * 32 + 16 + il) * PP + 16 * ot + 4 * q) = p1;
        }
      }
    }
    if (MODE == MODE_TRAIN) raw_commit(Rn);
    lds_barrier();
    STAMP(3);

    // ---- phase 4: reduce layer-4 partials; outputs / loss ------------------
    if (MODE == MODE_FWD || MODE == MODE_EM) {
      layer1(H0n, Yc);                                         // next tile's h1 (h2 is dead)
      // 32 samples x d outputs
      for (int idx = tid; idx < 32 * d; idx += NT) {
        const int c = idx / d, o = idx - c * d;
        const int64_t smp = s_base + c;
        if (smp < A.B) {
          float a = B4s[o];
#pragma unroll
          for (int ww = 0; ww < NW; ++ww) a += PART[(ww * 32 + c) * PP + o];
          const int64_t e = smp * d + o;
          if (MODE == MODE_FWD) {
            A.out[e] = a;
          } else {
            // x += [(1-l/2) sqrt(beta) a + 1/2 beta x] delta + sqrt(1-l) sqrt(beta) dW
            // (SDEs.py:556-561,587-588; sde_scheme.py:82-84,38-40)
            const float s = A.T - Rc[LO::RT + c];               // this item's time (ts[step] in the EM loop)
            const float beta = sde_beta(A.b0, A.b1, s);
            const float sb = sqrtf(beta);
            const float x = Rc[LO::RY + idx];
            const float zz = A.z ? A.z[e] : philox_normal1(A.rng, A.rng_step + (uint64_t)step, RNG_STREAM_DW, (uint64_t)e);
            const float mu = ca * (sb * a) - (-0.5f * beta * x);
            A.out[e] = x + (mu * A.delta + (sqrtf(1.0f - A.lmbd) * sb) * (A.sqrt_delta * zz));
          }
        }
      }
      if (n_steps > 1) __syncthreads();   // + vmcnt(0): this tile's new x must have left the CU before it is re-read
      else lds_barrier();                 // PART / H0 are rewritten by the next tile
      tile = ntile; step = nstep;
      continue;
    }

    if (MODE == MODE_TRAIN) {
      if (tid < 256) {
        // thread (c = sample, ol = output): a, adot from the K-slices; the sample's loss terms meet by shuffle
        const int c = tid >> 4, ol = tid & 15;
        const int64_t smp = s_base + c;
        const bool live = smp < A.B;
        const float wgt = live ? A.inv_batch : 0.f;
        float beta = 0.f, sb = 0.f;
        if (!A.u) { beta = sde_beta(A.b0, A.b1, Rc[LO::RT + c]); sb = sqrtf(beta); }
        float lj = 0.f;
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) {
          const int o = 16 * ot + ol;
          const bool oo = o < d;
          float a = B4s[o], ad = 0.f;
#pragma unroll
          for (int ww = 0; ww < NW; ++ww) { a += PART[(ww * 32 + c) * PP + o]; ad += PART[(ww * 32 + 16 + c) * PP + o]; }
          float adb;
          if (A.u) {
            // general form: loss_b = sum_o adot_o u_o + cst_b + 1/2 a_o^2, u = (d mu/d a)^T v  (MSGM: G(y)^T v)
            const float uo = oo ? Rc[LO::RU + c * d + o] : 0.f;
            lj += ad * uo + 0.5f * a * a;
            adb = uo * wgt;
          } else {
            const float vo = oo ? Rc[LO::RV + c * d + o] : 0.f;
            // SGM: loss_b = sum_o v_o (sqrt(beta) adot_o + 1/2 beta v_o) + 1/2 a_o^2     SDEs.py:631-646
            lj += vo * (sb * ad + 0.5f * beta * vo) + 0.5f * a * a;
            adb = sb * vo * wgt;
          }
          const float ab = a * wgt;
          ABAR[c * SM_P + o] = ab;
          ABAR[(16 + c) * SM_P + o] = adb;
          DB4[c * DPAD + o] += ab;
        }
#pragma unroll
        for (int m = 8; m > 0; m >>= 1) lj += __shfl_xor(lj, m, 64);
        if (ol == 0 && live) {
          if (A.u && A.cst) lj += Rc[LO::RC + c];
          loss_acc += lj;
          if (A.loss_per) A.loss_per[smp] = lj;
        }
      }
      // next tile's layer-1 operand, by a wave that has no part in the loss when there are eight
      if (tid >= NT - 64 && tid < NT - 64 + 16) build_h0(H0n, Rn, ntile, tid - (NT - 64));
      lds_barrier();
      STAMP(4);

      // ---- phase 5: layer-4 backward, dW4, Swish' on layer 3 --------------
      f32x4 g[IT][2];
#pragma unroll
      for (int it = 0; it < IT; ++it) { g[it][0] = f32x4{0, 0, 0, 0}; g[it][1] = f32x4{0, 0, 0, 0}; }
      for (int s = 0; s < A.d4; ++s) {
        const float bP = ABAR[il * SM_P + 4 * s + q];
        const float bT = ABAR[(16 + il) * SM_P + 4 * s + q];
#pragma unroll
        for (int it = 0; it < IT; ++it) {
          const float a = W4s[(4 * s + q) * ACT_P + fb + 16 * it + il];
          g[it][0] = mfma16(a, bP, g[it][0]); g[it][1] = mfma16(a, bT, g[it][1]);
        }
      }
      wgrad_swish<OT, IT>(Z, ABAR, SM_P, fb, il, q, dW4, z3, g);  // dW4^T[feat][o] += h3 . abar, Swish' hidden
#pragma unroll
      for (int it = 0; it < IT; ++it) db3[it] += g[it][0];
      store_act<IT>(U, fb, il, q, g);
      lds_barrier();
      STAMP(5);

      // ---- phase 6: dgrad layer 3 (W3^T), dW3, Swish' on layer 2 -----------
#pragma unroll
      for (int it = 0; it < IT; ++it) { g[it][0] = f32x4{0, 0, 0, 0}; g[it][1] = f32x4{0, 0, 0, 0}; }
      if (!late) {
        gemm128<true, 0, IT>(R3, pre, U, fb, il, q, g, nullptr, h);
        prefetch_w<true>(R2, fb, il, q, pre);        // W2^T for the next dgrad
        wgrad_swish<8, IT>(U, Yc, ACT_P, fb, il, q, dW3, z2, g);
      } else {                                         // the SIMD's other wave: weight gradient first (it only needs
        wgrad<8, IT>(U, Yc, ACT_P, fb, il, q, dW3);    // this wave's zbar3 columns), dgrad second, Swish' last
        gemm128<true, 0, IT>(R3, pre, U, fb, il, q, g, nullptr, h);
        prefetch_w<true>(R2, fb, il, q, pre);
        swish_bwd_inplace<IT>(z2, g);
      }
#pragma unroll
      for (int it = 0; it < IT; ++it) db2[it] += g[it][0];
      store_act<IT>(Z, fb, il, q, g);                      // h3 is dead after phase 5
      lds_barrier();
      STAMP(6);

      // ---- phase 7: dgrad layer 2 (W2^T), dW2, Swish' on layer 1 -----------
#pragma unroll
      for (int it = 0; it < IT; ++it) { g[it][0] = f32x4{0, 0, 0, 0}; g[it][1] = f32x4{0, 0, 0, 0}; }
      if (!late) {
        gemm128<true, 0, IT>(R2, pre, Z, fb, il, q, g, nullptr, h);
        prefetch_w<false>(R2, fb, il, q, pre);       // next tile's layer 2
        wgrad_swish<8, IT>(Z, Xc, ACT_P, fb, il, q, dW2, z1, g);
      } else {
        wgrad<8, IT>(Z, Xc, ACT_P, fb, il, q, dW2);
        gemm128<true, 0, IT>(R2, pre, Z, fb, il, q, g, nullptr, h);
        prefetch_w<false>(R2, fb, il, q, pre);
        swish_bwd_inplace<IT>(z1, g);
      }
#pragma unroll
      for (int it = 0; it < IT; ++it) db1[it] += g[it][0];
      store_act<IT>(U, fb, il, q, g);                      // zbar3 is dead after phase 6
      STAMP(7);
      // dW1 reads only THIS wave's columns of zbar1 (just written by this wave) and h0: no barrier in between
      wgrad<KT1, IT>(U, H0c, SM_P, fb, il, q, dW1);
      layer1(H0n, Yc);                                     // next tile's h1 (h2 is dead since phase 6)
      lds_barrier();     // ABAR/PART/.. are rewritten by the next tile
      STAMP(8);
    }
    tile = ntile; step = nstep;
  }

  // ---- epilogue (train): write this workgroup's gradient slab -------------
  if (MODE == MODE_TRAIN) {
    STAMP(9);
    float* slab = A.slabs + (int64_t)blockIdx.x * slab_stride(A.n_params);
    const int in_dim = A.in_dim;
    const int64_t oW1 = 0, ob1 = oW1 + (int64_t)HID * in_dim, oW2 = ob1 + HID, ob2 = oW2 + HID * HID,
                  oW3 = ob2 + HID, ob3 = oW3 + HID * HID, oW4 = ob3 + HID, ob4 = oW4 + (int64_t)d * HID;
    // dW2 / dW3: lane (k = 16kt+il, q), reg r -> row 32w+16it+4q+r
#pragma unroll
    for (int it = 0; it < IT; ++it)
#pragma unroll
      for (int kt = 0; kt < 8; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = fb + 16 * it + 4 * q + r, col = 16 * kt + il;
          slab[oW2 + row * HID + col] = dW2[it][kt][r];
          slab[oW3 + row * HID + col] = dW3[it][kt][r];
        }
#pragma unroll
    for (int it = 0; it < IT; ++it)
#pragma unroll
      for (int kt = 0; kt < KT1; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = fb + 16 * it + 4 * q + r, col = 16 * kt + il;
          if (col < in_dim) slab[oW1 + (int64_t)row * in_dim + col] = dW1[it][kt][r];
        }
    // dW4 held transposed: lane (o = 16ot+il, q), reg r -> feature 32w+16it+4q+r
#pragma unroll
    for (int it = 0; it < IT; ++it)
#pragma unroll
      for (int ot = 0; ot < OT; ++ot)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int feat = fb + 16 * it + 4 * q + r, o = 16 * ot + il;
          if (o < d) slab[oW4 + (int64_t)o * HID + feat] = dW4[it][ot][r];
        }
    // biases: sum the primal cotangents over the 16 sample lanes
#pragma unroll
    for (int it = 0; it < IT; ++it)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s1 = db1[it][r], s2 = db2[it][r], s3 = db3[it][r];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); s3 += __shfl_xor(s3, o, 64); }
        if (il == 0) {
          const int feat = fb + 16 * it + 4 * q + r;
          slab[ob1 + feat] = s1; slab[ob2 + feat] = s2; slab[ob3 + feat] = s3;
        }
      }
    __syncthreads();
    if (tid < d) {
      float s = 0.f;
      for (int j = 0; j < 16; ++j) s += DB4[j * DPAD + tid];
      slab[ob4 + tid] = s;
    }
    if (tid < 256 && (tid & 15) == 0) RED[tid >> 4] = loss_acc;
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
// 64 parameters (16 quads) x 16 slab groups per block, 16-B loads, fixed summation order (see the kernel).
// ADAM: one thread per parameter then applies the fused Adam update
// (single-GPU step: no all-reduce sits between the two), and thread 0 advances the
// Philox offset (nothing after this kernel reads it within the step).
template <bool ADAM>
__global__ void __launch_bounds__(256) k_slab_reduce(const float* __restrict__ slabs, int n_slabs, int64_t stride,
                                                     float* __restrict__ grads, int64_t n_params,
                                                     float* __restrict__ loss_sum, float inv_batch,
                                                     float* __restrict__ prm, float* __restrict__ m, float* __restrict__ v,
                                                     double lr, double b1, double b2, double eps,
                                                     const int64_t* __restrict__ step_dev, uint64_t* rng_advance) {
  // 64 parameters per block = 16 quads x 16 slab groups: thread (quad, group) sums every 16th slab with 16-B loads
  // (16 independent loads in flight), the 16 partial quads meet in LDS in a fixed order — deterministic.
  __shared__ f32x4 part[16][16];
  const int q4 = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int64_t p0 = (int64_t)blockIdx.x * 64 + 4 * q4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (p0 <= n_params)
    for (int g = grp; g < n_slabs; g += 16) acc += *reinterpret_cast<const f32x4*>(slabs + (int64_t)g * stride + p0);
  part[grp][q4] = acc;
  __syncthreads();
  if (threadIdx.x < 64) {
    const int px = threadIdx.x;
    const int64_t p = (int64_t)blockIdx.x * 64 + px;
    if (p <= n_params) {
      float s = 0.f;
#pragma unroll
      for (int g = 0; g < 16; ++g) s += part[g][px >> 2][px & 3];
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
  }
  if (ADAM && rng_advance && blockIdx.x == 0 && threadIdx.x == 0) rng_advance[1] += 1;
}

// ============================================================ C ABI
static inline hipStream_t S(msgm_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
static const int MLP_MAX_GRID = 256;

// Waves per workgroup.  Training at narrow width runs 8 waves x 16 features: two waves per SIMD, so one wave's
// Swish / LDS / barrier latencies are covered by the other's MFMAs (each holds half of the dW accumulators).  The
// wide carve (d > 15) has no LDS left for eight layer-4 K-slices and stays at 4 waves x 32 features.
#ifndef MLP_NW_TRAIN
#define MLP_NW_TRAIN 8
#endif
#ifndef MLP_NW_FWD
#define MLP_NW_FWD 4
#endif
template <int MODE, bool WIDE>
struct MlpCfg { static constexpr int NW = WIDE ? 4 : (MODE == MODE_TRAIN ? MLP_NW_TRAIN : MLP_NW_FWD); };

template <int MODE, bool WIDE, bool TINY>
static void set_lds_attr() {
  static const int once = [] {
    constexpr int NW = MlpCfg<MODE, WIDE>::NW;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mlp<MODE, WIDE, NW, TINY>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, MlpLds<MODE, WIDE, NW, TINY>::BYTES);
    return 0;
  }();
  (void)once;
}

// workgroups that fit one CU with this carve (inference modes: 2, or 3 with the tiny carve)
static bool mlp_tiny(const MlpArgs& A) { return A.in_dim <= 4 && A.P.d <= 4 && !getenv("MSGM_NO_TINY_CARVE"); }

template <int MODE>
static int launch_mlp(const MlpArgs& A, int grid, hipStream_t st) {
  const bool wide = A.in_dim > 16 || A.P.d > 16;
  constexpr int nw_w = MlpCfg<MODE, true>::NW, nw_n = MlpCfg<MODE, false>::NW;
  constexpr size_t lds_wide = MlpLds<MODE, true, nw_w>::BYTES, lds_narrow = MlpLds<MODE, false, nw_n>::BYTES;
  if (wide) { set_lds_attr<MODE, true, false>(); hipLaunchKernelGGL((k_mlp<MODE, true, nw_w, false>), dim3(grid), dim3(64 * nw_w), lds_wide, st, A); }
  else if (MODE != MODE_TRAIN && mlp_tiny(A)) {
    constexpr bool T = MODE != MODE_TRAIN;     // never instantiated for training
    constexpr size_t lds_tiny = MlpLds<MODE, false, nw_n, T>::BYTES;
    set_lds_attr<MODE, false, T>();
    hipLaunchKernelGGL((k_mlp<MODE, false, nw_n, T>), dim3(grid), dim3(64 * nw_n), lds_tiny, st, A);
  }
  else { set_lds_attr<MODE, false, false>(); hipLaunchKernelGGL((k_mlp<MODE, false, nw_n, false>), dim3(grid), dim3(64 * nw_n), lds_narrow, st, A); }
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
  return (size_t)MLP_MAX_GRID * (size_t)slab_stride(msgm_mlp_num_params(d, premodule)) * sizeof(float);
}

int msgm_mlp_forward(const msgm_mlp_params_t* P, const float* y, const float* t, float* a, int64_t B,
                     msgm_stream_t stream) {
  MlpArgs A{};
  int rc = fill_common(A, P, B);
  if (rc) return rc;
  if (!y || !t || !a) return MSGM_E_BADARG;
  A.y = y; A.t = t; A.out = a;
  const int64_t tiles = (B + 31) / 32;
  const int64_t cap = (mlp_tiny(A) ? 3 : 2) * MLP_MAX_GRID;   // forward modes: two (tiny carve: three) workgroups per CU
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
  const int64_t cap = (mlp_tiny(A) ? 3 : 2) * MLP_MAX_GRID;
  return launch_mlp<MODE_EM>(A, (int)(tiles < cap ? tiles : cap), S(stream));
}

int msgm_mlp_em_loop(const msgm_mlp_params_t* P, float* x, int64_t B, const msgm_sde_t* sde, const float* ts, int32_t n_steps,
                     float delta, float lmbd, const uint64_t* rng, uint64_t rng_step0, msgm_stream_t stream) {
  MlpArgs A{};
  int rc = fill_common(A, P, B);
  if (rc) return rc;
  if (!x || !sde || !ts || !rng || n_steps < 1) return MSGM_E_BADARG;
  if (sde->kind != MSGM_SDE_SGM) return MSGM_E_UNSUPPORTED;
  const int64_t tiles = (B + 31) / 32;
  if (tiles < 2) return MSGM_E_UNSUPPORTED;              // a workgroup needs two tiles to prefetch one ahead
  A.y = x; A.out = x; A.t = nullptr; A.t_scalar = 0.f;
  A.b0 = sde->beta_min; A.b1 = sde->beta_max; A.T = sde->T;
  A.delta = delta; A.sqrt_delta = (float)sqrt((double)delta); A.lmbd = lmbd;
  A.z = nullptr; A.rng = rng; A.rng_step = rng_step0;
  A.n_steps = n_steps; A.ts = ts;
  const int64_t cap = (mlp_tiny(A) ? 3 : 2) * MLP_MAX_GRID;
  const int64_t grid = tiles / 2 < cap ? tiles / 2 : cap;
  return launch_mlp<MODE_EM>(A, (int)grid, S(stream));
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
                     reinterpret_cast<const float*>(workspace), n_slabs, slab_stride(n_params), grads, n_params, loss_sum, inv_batch,
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
                     reinterpret_cast<const float*>(workspace), n_slabs, slab_stride(n_params), grads, n_params, loss_sum, inv_batch,
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
