// unet2d_kernels.hip — the 2-D U-Net's non-convolution ops on gfx950:
//   K7  GroupNorm(min(C,32) groups) [+ SiLU] on dual numbers, forward and backward
//       (model/nn_utils.py:39-46,107-114; model/unet.py:140-143,152-155,214,443-444)
//   K8  single-head attention pieces: batched fp32-MFMA GEMM + dual softmax rows
//       (model/unet.py:236-250)
//   K10 sinusoidal timestep embedding (model/nn_utils.py:130-148)
//   K9  layout glue: flat <-> channels-last image (NNUnet.py:19-77), 2x2 block sum
//       (adjoint of the nearest-2x upsample folded into the conv gather)
// Activations are channels-last [N][P][C]; the forward-mode tangent is the second
// half of the batch (rows n >= Bp).
#include "common.h"
#include <stdlib.h>
typedef double f64x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x4 mfma16u(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ void silu012u(float z, float& s0, float& s1, float& s2) {
  const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-z));
  const float om = 1.0f - sg;
  s0 = z * sg;
  s1 = sg * (1.0f + z * om);
  s2 = sg * om * (2.0f + z * (1.0f - 2.0f * sg));
}

// ------------------------------------------------------------------ GroupNorm dual
// Two fully parallel launches per direction (a per-sample workgroup would leave the
// chip idle at 32 samples/GPU):
//   1. k_gn_*_reduce: grid (sample, pixel chunk); threads run along channels (coalesced,
//      256/C pixel lanes), fp32 partials -> LDS -> ONE double per (sample, chunk, group, moment)
//      stored in that chunk's own slot — no atomics: the slots of a sample are added in chunk
//      order by the finalise kernels, so the result is bitwise reproducible.  Moments are raw
//      (sum x, sum x^2, sum xdot, sum x xdot) but combined in double, so the variance does not
//      suffer from E[x^2]-mu^2 cancellation.
//   2. k_gn_*_apply: grid-stride elementwise pass that rebuilds {mean, inv_std,
//      mean(xdot), a = mean(xhat xdot)} from the moments.
// acc[b][chunk < GN_SLOTS][g][8] doubles: forward uses moments 0..3, backward 0..4; the per-channel
// parameter gradients of the backward go to per-(sample, chunk) slots too and are summed in a fixed order.
#define GN_SLOTS 32
struct GnArgs {
  const float* x; const float* gamma; const float* beta;
  float* out; double* acc; float* stats;
  int P, C, G, Bp, dual, silu, chunk;   // chunk: pixels per workgroup (a multiple of sub in the reduce kernels)
  float eps;
  // backward
  const float* gout; float* gx; float* dgamma; float* dbeta;
  // input = channel concatenation of two tensors (x: C0 channels, x1: C - C0; both (primal | tangent) stacked when
  // dual): the decoder's cat([h, skip]) (model/unet.py:514) is never materialised.  out / gout stay ONE tensor of C
  // channels; the backward writes the input cotangent into gx (C0 channels) and gx1 (C - C0).
  const float* x1; int C0;
  float* gx1;
  int sub;      // reduce kernels: pixels per fp32 partial sum — fixed per sample, whatever the batch size
  int nch;      // pixel chunks (= slots) per sample of the reduce kernel that filled acc
  double* accf; // backward: the five moments summed over the slots, [Bp][G][8]
  float* pslots;// backward: dgamma | dbeta partials, [2][Bp * GN_SLOTS][C]
  const float* resid;   // backward apply: added to gx (skip-branch cotangent), may alias gx
  const float* resid2;  // a second addend (the cotangent that reached this tensor through the U-Net's skip stack)
};

// Reduction tail of the reduce kernels: every thread holds V double partials per moment (channels V*cv .. V*cv+V-1 of pixel
// lane pl; per-thread fp32 sums over one sub-chunk are widened and from there on everything is added in double).  ALL moments
// go to LDS images [moment][pl][C] at once; after ONE barrier one thread per (moment, group) sums its channels over the pixel
// lanes in a fixed order and STORES the chunk's value in its slot.
template <bool VEC>
__global__ void __launch_bounds__(256) k_gn_fwd_reduce(GnArgs A) {
  constexpr int V = VEC ? 4 : 1;
  __shared__ double red[4 * 1024];       // all moments at once: ONE barrier in the tail instead of two per moment
  const int tid = threadIdx.x, b = blockIdx.x;
  const int C = A.C, P = A.P, G = A.G, cpg = C / G;
  const int CV = C / V, PL = 256 / CV;
  const bool live = tid < CV * PL;
  const int cv = live ? tid % CV : 0, pl = live ? tid / CV : 0;
  // channel V*cv of the (possibly concatenated) input lives in x (pitch C0) or in x1 (pitch C - C0)
  const int C0 = A.x1 ? A.C0 : C;
  const bool second = V * cv >= C0;
  const int pitch = second ? C - C0 : C0, coff = second ? V * cv - C0 : V * cv;
  const float* xp = (second ? A.x1 : A.x) + (size_t)b * P * pitch + coff;
  const float* xt = (second ? A.x1 : A.x) + (size_t)(b + A.Bp) * P * pitch + coff;
  const int p0 = blockIdx.y * A.chunk, p1 = min(p0 + A.chunk, P);
  double d0[V], d1[V], d2[V], d3[V];
#pragma unroll
  for (int k = 0; k < V; ++k) { d0[k] = 0.0; d1[k] = 0.0; d2[k] = 0.0; d3[k] = 0.0; }
  if (live)
   for (int ps = p0; ps < p1; ps += max(A.sub, 1)) {
    const int pe = min(ps + max(A.sub, 1), p1);
    float s0[V], s1[V], s2[V], s3[V];
#pragma unroll
    for (int k = 0; k < V; ++k) { s0[k] = 0.f; s1[k] = 0.f; s2[k] = 0.f; s3[k] = 0.f; }
    for (int p = ps + pl; p < pe; p += PL) {
      float xv[V], dv[V];
      if (VEC) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(xp + (size_t)p * pitch);
#pragma unroll
        for (int k = 0; k < V; ++k) xv[k] = t[k];
        if (A.dual) {
          const f32x4 u = *reinterpret_cast<const f32x4*>(xt + (size_t)p * pitch);
#pragma unroll
          for (int k = 0; k < V; ++k) dv[k] = u[k];
        }
      } else {
        xv[0] = xp[(size_t)p * pitch];
        if (A.dual) dv[0] = xt[(size_t)p * pitch];
      }
#pragma unroll
      for (int k = 0; k < V; ++k) {
        s0[k] += xv[k]; s1[k] += xv[k] * xv[k];
        if (A.dual) { s2[k] += dv[k]; s3[k] += xv[k] * dv[k]; }
      }
    }
#pragma unroll
    for (int k = 0; k < V; ++k) { d0[k] += (double)s0[k]; d1[k] += (double)s1[k]; d2[k] += (double)s2[k]; d3[k] += (double)s3[k]; }
   }
  double* dst = A.acc + ((size_t)b * GN_SLOTS + blockIdx.y) * G * 8;
  const int nm = A.dual ? 4 : 2;
  if (live) {
#pragma unroll
    for (int k = 0; k < V; ++k) {
      const int e = pl * C + V * cv + k;
      red[e] = d0[k]; red[1024 + e] = d1[k];
      if (A.dual) { red[2048 + e] = d2[k]; red[3072 + e] = d3[k]; }
    }
  }
  __syncthreads();
  // one thread per (moment, group): the pixel lanes in order, the group's channels inside
  for (int i = tid; i < nm * G; i += 256) {
    const int m = i / G, g = i - m * G;
    const double* r = red + m * 1024 + g * cpg;
    double t = 0.0;
    for (int p = 0; p < PL; ++p)
      for (int cc = 0; cc < cpg; ++cc) t += r[p * C + cc];
    dst[(size_t)g * 8 + m] = t;
  }
}

// moments of (sample b, group g): the chunk slots added in chunk order.  The loads of up to 4 slots are issued before the
// first add (the slots are independent addresses; a plain loop waited for every slot's round trip in turn).
__device__ __forceinline__ void gn_sum_slots(const GnArgs& A, int b, int g, int nm, double (&a8)[8]) {
#pragma unroll
  for (int m = 0; m < 8; ++m) a8[m] = 0.0;
  const double* base = A.acc + ((size_t)b * GN_SLOTS * A.G + g) * 8;
  const size_t sstride = (size_t)A.G * 8;
  for (int ch0 = 0; ch0 < A.nch; ch0 += 4) {
    double v[4][5];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ch = min(ch0 + j, A.nch - 1);                // clamped: every load unconditional
      const double* p = base + (size_t)ch * sstride;
      const f64x2 q0 = *reinterpret_cast<const f64x2*>(p), q1 = *reinterpret_cast<const f64x2*>(p + 2);
      v[j][0] = q0[0]; v[j][1] = q0[1]; v[j][2] = q1[0]; v[j][3] = q1[1];
      v[j][4] = nm > 4 ? p[4] : 0.0;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (ch0 + j < A.nch) {
#pragma unroll
        for (int m = 0; m < 5; ++m) if (m < nm) a8[m] += v[j][m];
      }
  }
}

__device__ __forceinline__ void gn_stats(const double* a8, double cnt, float eps, float& mu, float& inv, float& md, float& a) {
  const double m = a8[0] / cnt;
  double var = a8[1] / cnt - m * m;
  if (var < 0) var = 0;
  const double iv = 1.0 / sqrt(var + (double)eps);
  const double d = a8[2] / cnt;
  mu = (float)m; inv = (float)iv; md = (float)d;
  a = (float)(iv * (a8[3] / cnt - m * d));
}

// GroupNorm as a per-(sample, channel) affine map y = a x + b (a = gamma/sigma, b = beta - mean a) for a consumer that
// applies it while reading x (msgm_conv_forward_fused).  One block per sample.
__global__ void __launch_bounds__(256) k_gn_affine(GnArgs A, float* __restrict__ scale, float* __restrict__ shift) {
  const int b = blockIdx.x, c = threadIdx.x, cpg = A.C / A.G;
  const double cnt = (double)A.P * cpg;
  if (c < A.C) {
    float mu, inv, md, a;
    double a8[8];
    gn_sum_slots(A, b, c / cpg, 2, a8);
    gn_stats(a8, cnt, A.eps, mu, inv, md, a);
    const float sc = inv * A.gamma[c];
    scale[(size_t)b * A.C + c] = sc;
    shift[(size_t)b * A.C + c] = A.beta[c] - mu * sc;
  }
}

// Elementwise pass.  VEC: grid (pixel chunk, sample); a thread owns 4 consecutive channels (its statistics and
// affine parameters are loaded once) and walks over pixels — 16-B loads/stores, consecutive threads on consecutive
// addresses.  !VEC (C % 4 != 0): one thread per channel, same walk.
// element offset of (sample b, pixel p, channel c) of the possibly two-source input: `pitch` / `coff` select the source
// of this thread's channel vector (a vector never straddles the sources: C0 % 4 == 0), `half` is the primal -> tangent
// offset of that source
struct GnSrc { const float* base; long half; int pitch; };
__device__ __forceinline__ GnSrc gn_src(const GnArgs& A, int c) {
  const int C0 = A.x1 ? A.C0 : A.C;
  const bool second = c >= C0;
  const int pitch = second ? A.C - C0 : C0;
  return GnSrc{(second ? A.x1 : A.x) + (second ? c - C0 : c), (long)A.Bp * A.P * pitch, pitch};
}

template <bool VEC>
__global__ void __launch_bounds__(256) k_gn_fwd_apply(GnArgs A) {
  constexpr int V = VEC ? 4 : 1;
  __shared__ f32x4 sst[64];
  const int C = A.C, P = A.P, G = A.G, cpg = C / G;
  const int CV = C / V, PL = 256 / CV;
  const int tid = threadIdx.x, b = blockIdx.y;
  // {mean, inv_std, mean(xdot), a} of this sample's groups, rebuilt by every workgroup from the chunk slots the reduce
  // kernel left (thread g: the slots of group g in chunk order, in double — the arithmetic of the former one-thread-per-
  // (sample, group) finalise launch, which cost 5-10 us per GroupNorm at the 32-row shard for a few KB of work).
  // The sample's first workgroup keeps them for the backward.
  if (tid < G) {
    float mu, inv, md, a;
    double a8[8];
    gn_sum_slots(A, b, tid, A.dual ? 4 : 2, a8);
    gn_stats(a8, (double)P * cpg, A.eps, mu, inv, md, a);
    sst[tid] = f32x4{mu, inv, md, a};
    if (A.stats && blockIdx.x == 0) *reinterpret_cast<f32x4*>(A.stats + ((size_t)b * G + tid) * 4) = f32x4{mu, inv, md, a};
  }
  __syncthreads();
  if (tid >= CV * PL) return;
  const int cv = tid % CV, pl = tid / CV;
  const long tot = (long)A.Bp * P * C;
  float mu[V], inv[V], md[V], a[V], ga[V], be[V];
#pragma unroll
  for (int k = 0; k < V; ++k) {
    const int c = V * cv + k;
    const f32x4 st = sst[c / cpg];
    mu[k] = st[0]; inv[k] = st[1]; md[k] = st[2]; a[k] = st[3];
    ga[k] = A.gamma[c]; be[k] = A.beta[c];
  }
  const int p0 = blockIdx.x * A.chunk, p1 = min(p0 + A.chunk, P);
  const GnSrc sx = gn_src(A, V * cv);
  for (int p = p0 + pl; p < p1; p += PL) {
    const long e = ((long)b * P + p) * C + V * cv;
    const long ex = ((long)b * P + p) * sx.pitch;
    float x[V], xd[V], y[V], yd[V];
    if (VEC) {
      const f32x4 xv = *reinterpret_cast<const f32x4*>(sx.base + ex);
#pragma unroll
      for (int k = 0; k < V; ++k) x[k] = xv[k];
      if (A.dual) {
        const f32x4 dv = *reinterpret_cast<const f32x4*>(sx.base + ex + sx.half);
#pragma unroll
        for (int k = 0; k < V; ++k) xd[k] = dv[k];
      }
    } else {
      x[0] = sx.base[ex];
      if (A.dual) xd[0] = sx.base[ex + sx.half];
    }
#pragma unroll
    for (int k = 0; k < V; ++k) {
      const float xh = (x[k] - mu[k]) * inv[k];
      y[k] = ga[k] * xh + be[k];
      yd[k] = A.dual ? ga[k] * (inv[k] * ((xd[k] - md[k]) - xh * a[k])) : 0.f;
      if (A.silu) {
        float z0, z1, z2;
        silu012u(y[k], z0, z1, z2);
        y[k] = z0; yd[k] = z1 * yd[k];
      }
    }
    if (VEC) {
      f32x4 o, od;
#pragma unroll
      for (int k = 0; k < V; ++k) { o[k] = y[k]; od[k] = yd[k]; }
      *reinterpret_cast<f32x4*>(A.out + e) = o;
      if (A.dual) *reinterpret_cast<f32x4*>(A.out + e + tot) = od;
    } else {
      A.out[e] = y[0];
      if (A.dual) A.out[e + tot] = yd[0];
    }
  }
}

// Backward over the dual pair.  With X = gamma*zbar, W = gamma*zdotbar (cotangents of xhat and of
// its tangent what = inv (xdot_c - xhat a)):
//   xdotbar = inv (W - mean W - xhat p),                      p = mean(xhat W)
//   xbar    = inv (X - mean X - xhat mean(xhat X)) - inv (c xhat + a xdotbar + p what),  c = mean(W what)
// (derived by hand, checked against autograd in tests/test_unet2d_gpu.py).
template <bool VEC>
__global__ void __launch_bounds__(256) k_gn_bwd_reduce(GnArgs A) {
  constexpr int V = VEC ? 4 : 1;
  __shared__ double redd[5 * 1024];      // the five moments and (below) both parameter partials: ONE barrier in the tail
  __shared__ float red[2 * 1024];
  const int tid = threadIdx.x, b = blockIdx.x;
  const int C = A.C, P = A.P, G = A.G, cpg = C / G;
  const int CV = C / V, PL = 256 / CV;
  const bool live = tid < CV * PL;
  const int cv = live ? tid % CV : 0, pl = live ? tid / CV : 0;
  const long tot = (long)A.Bp * P * C;
  float mu[V], inv[V], md[V], a[V], ga[V], be[V];
#pragma unroll
  for (int k = 0; k < V; ++k) {
    const int c = V * cv + k;
    const f32x4 st = *reinterpret_cast<const f32x4*>(A.stats + ((size_t)b * G + c / cpg) * 4);
    mu[k] = st[0]; inv[k] = st[1]; md[k] = st[2]; a[k] = st[3];
    ga[k] = A.gamma[c]; be[k] = A.beta[c];
  }
  const int p0 = blockIdx.y * A.chunk, p1 = min(p0 + A.chunk, P);
  const GnSrc sx = gn_src(A, V * cv);
  double dX[V], dXx[V], dW[V], dWx[V], dWw[V];
  float dga[V], dbe[V];
#pragma unroll
  for (int k = 0; k < V; ++k) { dX[k] = dXx[k] = dW[k] = dWx[k] = dWw[k] = 0.0; dga[k] = dbe[k] = 0.f; }
  if (live)
   for (int ps = p0; ps < p1; ps += max(A.sub, 1)) {
    const int pe = min(ps + max(A.sub, 1), p1);
    float sX[V], sXx[V], sW[V], sWx[V], sWw[V];
#pragma unroll
    for (int k = 0; k < V; ++k) { sX[k] = sXx[k] = sW[k] = sWx[k] = sWw[k] = 0.f; }
    for (int p = ps + pl; p < pe; p += PL) {
      const long e = ((long)b * P + p) * C + V * cv;
      const long ex = ((long)b * P + p) * sx.pitch;
      float x[V], xd[V], zb[V], zdb[V];
      if (VEC) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(sx.base + ex), v1 = *reinterpret_cast<const f32x4*>(sx.base + ex + sx.half);
        const f32x4 v2 = *reinterpret_cast<const f32x4*>(A.gout + e), v3 = *reinterpret_cast<const f32x4*>(A.gout + e + tot);
#pragma unroll
        for (int k = 0; k < V; ++k) { x[k] = v0[k]; xd[k] = v1[k]; zb[k] = v2[k]; zdb[k] = v3[k]; }
      } else {
        x[0] = sx.base[ex]; xd[0] = sx.base[ex + sx.half]; zb[0] = A.gout[e]; zdb[0] = A.gout[e + tot];
      }
#pragma unroll
      for (int k = 0; k < V; ++k) {
        const float xh = (x[k] - mu[k]) * inv[k];
        const float wh = inv[k] * ((xd[k] - md[k]) - xh * a[k]);
        float zbk = zb[k], zdbk = zdb[k];
        if (A.silu) {
          float z0, z1, z2;
          silu012u(ga[k] * xh + be[k], z0, z1, z2);
          const float yd = ga[k] * wh;
          const float nzb = zbk * z1 + zdbk * (z2 * yd);
          zdbk = zdbk * z1;
          zbk = nzb;
        }
        const float X = ga[k] * zbk, W = ga[k] * zdbk;
        sX[k] += X; sXx[k] += X * xh; sW[k] += W; sWx[k] += W * xh; sWw[k] += W * wh;
        dga[k] += zbk * xh + zdbk * wh; dbe[k] += zbk;
      }
    }
#pragma unroll
    for (int k = 0; k < V; ++k) {
      dX[k] += (double)sX[k]; dXx[k] += (double)sXx[k]; dW[k] += (double)sW[k]; dWx[k] += (double)sWx[k]; dWw[k] += (double)sWw[k];
    }
   }
  double* dst = A.acc + ((size_t)b * GN_SLOTS + blockIdx.y) * G * 8;
  const size_t slot = (size_t)b * gridDim.y + blockIdx.y, nslots = (size_t)gridDim.x * gridDim.y;
  if (live) {
#pragma unroll
    for (int k = 0; k < V; ++k) {
      const int e = pl * C + V * cv + k;
      redd[e] = dX[k]; redd[1024 + e] = dXx[k]; redd[2048 + e] = dW[k]; redd[3072 + e] = dWx[k]; redd[4096 + e] = dWw[k];
      red[e] = dga[k]; red[1024 + e] = dbe[k];
    }
  }
  __syncthreads();
  // one thread per (moment, group): the pixel lanes in order, the group's channels inside
  for (int i = tid; i < 5 * G; i += 256) {
    const int m = i / G, g = i - m * G;
    const double* r = redd + m * 1024 + g * cpg;
    double t = 0.0;
    for (int p = 0; p < PL; ++p)
      for (int cc = 0; cc < cpg; ++cc) t += r[p * C + cc];
    dst[(size_t)g * 8 + m] = t;
  }
  // per-channel parameter gradients: the pixel lanes in order, one value per channel into this (sample, chunk)'s slot
  for (int i = tid; i < 2 * C; i += 256) {
    const int which = i / C, c = i - which * C;
    float t = 0.f;
    for (int p = 0; p < PL; ++p) t += red[which * 1024 + p * C + c];
    A.pslots[((size_t)which * nslots + slot) * C + c] = t;
  }
}

// dgamma[c] += sum over the (sample, chunk) slots, dbeta alike (blockIdx.y = 0 / 1): 32 channels x 32 slot slices per
// workgroup (1024 threads: the grid is only C/32 x 2 workgroups, so the parallelism has to come from inside), every
// slice walks its slots in order and the 32 slices are added in order — no atomics, same bits every run.
__global__ void __launch_bounds__(1024) k_gn_param_reduce(const float* __restrict__ pslots, size_t nslots, int C,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta, int nbc) {
  __shared__ float red[32][32];
  const int bid = blockIdx.x, bx = bid % nbc, by = bid / nbc;
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int c = bx * 32 + cl;
  const float* p = pslots + (size_t)by * nslots * C;
  float t = 0.f;
  if (c < C)
    for (size_t s = sl; s < nslots; s += 32) t += p[s * C + c];
  red[sl][cl] = t;
  __syncthreads();
  if (sl == 0 && c < C) {
    float r = red[0][cl];
#pragma unroll
    for (int k = 1; k < 32; ++k) r += red[k][cl];
    float* dst = by == 0 ? dgamma : dbeta;
    dst[c] += r;
  }
}

template <bool VEC>
__global__ void __launch_bounds__(256) k_gn_bwd_apply(GnArgs A) {
  constexpr int V = VEC ? 4 : 1;
  const int C = A.C, P = A.P, G = A.G, cpg = C / G;
  const int CV = C / V, PL = 256 / CV;
  const int tid = threadIdx.x, b = blockIdx.y;
  // the five backward moments of this sample's groups: every workgroup adds the reduce kernel's chunk slots itself (thread
  // g, chunk order, double) instead of a finalise launch in between
  __shared__ float sm5[64][5];
  if (tid < G) {
    double a8[8];
    gn_sum_slots(A, b, tid, 5, a8);
#pragma unroll
    for (int m = 0; m < 5; ++m) sm5[tid][m] = (float)a8[m];
  }
  __syncthreads();
  if (tid >= CV * PL) return;
  const int cv = tid % CV, pl = tid / CV;
  const long tot = (long)A.Bp * P * C;
  const float rc = 1.0f / ((float)P * (float)cpg);
  float mu[V], inv[V], md[V], a[V], ga[V], be[V], mX[V], mXx[V], mW[V], pp[V], cc[V];
#pragma unroll
  for (int k = 0; k < V; ++k) {
    const int c = V * cv + k, g = c / cpg;
    const f32x4 st = *reinterpret_cast<const f32x4*>(A.stats + ((size_t)b * G + g) * 4);
    mu[k] = st[0]; inv[k] = st[1]; md[k] = st[2]; a[k] = st[3];
    mX[k] = sm5[g][0] * rc; mXx[k] = sm5[g][1] * rc; mW[k] = sm5[g][2] * rc; pp[k] = sm5[g][3] * rc;
    cc[k] = sm5[g][4] * rc;
    ga[k] = A.gamma[c]; be[k] = A.beta[c];
  }
  const int p0 = blockIdx.x * A.chunk, p1 = min(p0 + A.chunk, P);
  const GnSrc sx = gn_src(A, V * cv);
  // destination of the input cotangent: the source's own tensor when the input is a concatenation
  float* gbase = A.x1 ? ((V * cv >= A.C0) ? A.gx1 + (V * cv - A.C0) : A.gx + V * cv) : A.gx + V * cv;
  const float* rbase = A.resid ? A.resid + V * cv : nullptr;            // single-source only (host checks)
  const float* rbase2 = A.resid2 ? A.resid2 + V * cv : nullptr;
  for (int p = p0 + pl; p < p1; p += PL) {
    const long e = ((long)b * P + p) * C + V * cv;
    const long ex = ((long)b * P + p) * sx.pitch;
    float x[V], xd[V], zb[V], zdb[V], xb[V], xdb[V];
    if (VEC) {
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(sx.base + ex), v1 = *reinterpret_cast<const f32x4*>(sx.base + ex + sx.half);
      const f32x4 v2 = *reinterpret_cast<const f32x4*>(A.gout + e), v3 = *reinterpret_cast<const f32x4*>(A.gout + e + tot);
#pragma unroll
      for (int k = 0; k < V; ++k) { x[k] = v0[k]; xd[k] = v1[k]; zb[k] = v2[k]; zdb[k] = v3[k]; }
    } else {
      x[0] = sx.base[ex]; xd[0] = sx.base[ex + sx.half]; zb[0] = A.gout[e]; zdb[0] = A.gout[e + tot];
    }
#pragma unroll
    for (int k = 0; k < V; ++k) {
      const float xh = (x[k] - mu[k]) * inv[k];
      const float wh = inv[k] * ((xd[k] - md[k]) - xh * a[k]);
      float zbk = zb[k], zdbk = zdb[k];
      if (A.silu) {
        float z0, z1, z2;
        silu012u(ga[k] * xh + be[k], z0, z1, z2);
        const float yd = ga[k] * wh;
        const float nzb = zbk * z1 + zdbk * (z2 * yd);
        zdbk = zdbk * z1;
        zbk = nzb;
      }
      const float X = ga[k] * zbk, W = ga[k] * zdbk;
      xdb[k] = inv[k] * (W - mW[k] - xh * pp[k]);
      xb[k] = inv[k] * (X - mX[k] - xh * mXx[k]) - inv[k] * (cc[k] * xh + a[k] * xdb[k] + pp[k] * wh);
    }
    if (VEC) {
      f32x4 o, od;
#pragma unroll
      for (int k = 0; k < V; ++k) { o[k] = xb[k]; od[k] = xdb[k]; }
      if (rbase) { o += *reinterpret_cast<const f32x4*>(rbase + ex); od += *reinterpret_cast<const f32x4*>(rbase + ex + sx.half); }
      if (rbase2) { o += *reinterpret_cast<const f32x4*>(rbase2 + ex); od += *reinterpret_cast<const f32x4*>(rbase2 + ex + sx.half); }
      *reinterpret_cast<f32x4*>(gbase + ex) = o;
      *reinterpret_cast<f32x4*>(gbase + ex + sx.half) = od;
    } else {
      gbase[ex] = xb[0] + (rbase ? rbase[ex] : 0.f) + (rbase2 ? rbase2[ex] : 0.f);
      gbase[ex + sx.half] = xdb[0] + (rbase ? rbase[ex + sx.half] : 0.f) + (rbase2 ? rbase2[ex + sx.half] : 0.f);
    }
  }
}

// ------------------------------------------------------------------ batched GEMM
// C[b](i,j) (+)= alpha * ( sum_k A[b](i,k) B[b](k,j)  [+ sum_k A2[b](i,k) B2[b](k,j)] ) with element strides.
// The optional second operand pair (same strides) fuses the two-term products of the dual attention
// (Wdot = qdot k^T + q kdot^T, adot = Pdot v + P vdot, and their adjoints) into one pass over the output.
// One wave computes a (16 MT) x (16 NT) tile, 4 waves a 2x2 arrangement.  K runs in groups of 16 with the
// permuted mapping k = 16g + 4q + r (lane quarter q, MFMA step r): an operand that is contiguous along K is
// read with one 16-B load per group (AVEC / BVEC), otherwise with 4 coalesced dword loads.  The host swaps the
// operand roles so that the output's contiguous dimension is the MFMA row index: the 4 accumulator registers of
// a lane are then 4 consecutive addresses and the tile is stored with 16-B stores (CVEC).
struct BmmArgs {
  const float* A; const float* B; const float* A2; const float* B2; float* C;
  int M, N, K, batch;
  long sAb, sAi, sAk, sBb, sBk, sBj, sCb, sCi, sCj;
  float alpha; int accumulate;
  // DUAL: a second output C3 = alpha * A3 . B that shares the (big) B operand of the first pair — the dual-number
  // attention always needs P v next to Pdot v + P vdot (and the three adjoint pairs alike), so the (T,T) operand is
  // streamed once for both.  A3 has A's strides, C3 has C's.
  const float* A3; float* C3;
};
template <bool VEC>
__device__ __forceinline__ f32x4 bmm_frag(const float* p, long sk, bool valid) {
  if (!valid) return f32x4{0, 0, 0, 0};
  if (VEC) return *reinterpret_cast<const f32x4*>(p);
  return f32x4{p[0], p[sk], p[2 * sk], p[3 * sk]};
}
template <int MT, int NT, bool AVEC, bool BVEC, bool DUAL = false>
__global__ void __launch_bounds__(256) k_bmm(BmmArgs P) {
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, il = lane & 15, q = lane >> 4;
  const int tiles_n = (P.N + 32 * NT - 1) / (32 * NT);
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
  const int i0 = tm * 32 * MT + (w >> 1) * 16 * MT, j0 = tn * 32 * NT + (w & 1) * 16 * NT;
  if (i0 >= P.M || j0 >= P.N) return;
  f32x4 acc[MT][NT], acc3[DUAL ? MT : 1][DUAL ? NT : 1];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) { acc[m][n] = f32x4{0, 0, 0, 0}; if (DUAL) acc3[m][n] = f32x4{0, 0, 0, 0}; }
  bool vi[MT], vj[NT];
#pragma unroll
  for (int m = 0; m < MT; ++m) vi[m] = i0 + 16 * m + il < P.M;
#pragma unroll
  for (int n = 0; n < NT; ++n) vj[n] = j0 + 16 * n + il < P.N;
  const int ngroups = P.K >> 4;                     // host guarantees K % 16 == 0
  for (int pair = 0; pair < 2; ++pair) {
    const float* Ab = (pair ? P.A2 : P.A);
    const float* Bb = (pair ? P.B2 : P.B);
    if (!Ab) break;
    Ab += (size_t)blockIdx.y * P.sAb;
    Bb += (size_t)blockIdx.y * P.sBb;
    const float* ap[MT];
    const float* bp[NT];
#pragma unroll
    for (int m = 0; m < MT; ++m) ap[m] = Ab + (size_t)(vi[m] ? i0 + 16 * m + il : 0) * P.sAi + (size_t)(4 * q) * P.sAk;
#pragma unroll
    for (int n = 0; n < NT; ++n) bp[n] = Bb + (size_t)(vj[n] ? j0 + 16 * n + il : 0) * P.sBj + (size_t)(4 * q) * P.sBk;
    const size_t stepA = (size_t)16 * P.sAk, stepB = (size_t)16 * P.sBk;
    const bool third = DUAL && pair == 0;                 // A3 rides along with the first pair's B operand
    const float* ap3[MT];
    f32x4 na[MT], nb[NT], na3[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) na[m] = bmm_frag<AVEC>(ap[m], P.sAk, vi[m]);
#pragma unroll
    for (int n = 0; n < NT; ++n) nb[n] = bmm_frag<BVEC>(bp[n], P.sBk, vj[n]);
    if (third) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        ap3[m] = P.A3 + (size_t)blockIdx.y * P.sAb + (size_t)(vi[m] ? i0 + 16 * m + il : 0) * P.sAi + (size_t)(4 * q) * P.sAk;
        na3[m] = bmm_frag<AVEC>(ap3[m], P.sAk, vi[m]);
      }
    }
    for (int g = 0; g < ngroups; ++g) {
      f32x4 a[MT], b[NT], a3[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) { a[m] = na[m]; ap[m] += stepA; if (third) { a3[m] = na3[m]; ap3[m] += stepA; } }
#pragma unroll
      for (int n = 0; n < NT; ++n) { b[n] = nb[n]; bp[n] += stepB; }
      if (g + 1 < ngroups) {
#pragma unroll
        for (int m = 0; m < MT; ++m) na[m] = bmm_frag<AVEC>(ap[m], P.sAk, vi[m]);
#pragma unroll
        for (int n = 0; n < NT; ++n) nb[n] = bmm_frag<BVEC>(bp[n], P.sBk, vj[n]);
        if (third) {
#pragma unroll
          for (int m = 0; m < MT; ++m) na3[m] = bmm_frag<AVEC>(ap3[m], P.sAk, vi[m]);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            acc[m][n] = mfma16u(a[m][r], b[n][r], acc[m][n]);
            if (third) acc3[m][n] = mfma16u(a3[m][r], b[n][r], acc3[m][n]);
          }
    }
  }
#pragma unroll
  for (int which = 0; which < (DUAL ? 2 : 1); ++which) {
  float* Cb = (which ? P.C3 : P.C) + (size_t)blockIdx.y * P.sCb;
  const bool cvec = P.sCi == 1 && (P.sCj & 3) == 0 && (P.sCb & 3) == 0 && ((reinterpret_cast<uintptr_t>(which ? P.C3 : P.C) & 15) == 0);
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int j = j0 + 16 * n + il;
      if (j >= P.N) continue;
      const int i = i0 + 16 * m + 4 * q;
      if (i >= P.M) continue;
      f32x4 v = P.alpha * (which ? acc3[DUAL ? m : 0][DUAL ? n : 0] : acc[m][n]);
      if (cvec && i + 3 < P.M) {
        float* cp = Cb + (size_t)j * P.sCj + i;
        if (P.accumulate) v += *reinterpret_cast<const f32x4*>(cp);
        *reinterpret_cast<f32x4*>(cp) = v;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (i + r >= P.M) continue;
          float* cp = Cb + (size_t)(i + r) * P.sCi + (size_t)j * P.sCj;
          *cp = P.accumulate ? *cp + v[r] : v[r];
        }
      }
    }
  }
}

// ------------------------------------------------------------------ batched GEMM, LDS-staged
// The same contract as k_bmm (BmmArgs), for operands with ONE unit stride each: a workgroup owns a 64 x 64 output tile
// and walks K in blocks of 32.  Per block the A rows [64][32], the B columns [64][32] (and the A3 rows when DUAL) are
// brought in with coalesced 16-byte global loads — along k when k is the contiguous index (KV), along the row index when
// the operand is stored transposed (RV: the float4 is scattered into four LDS rows) — into a canonical [row][k] LDS image
// (pitch 36 floats), double-buffered, with the next block's loads in flight during this block's MFMAs.  Every wave then
// reads its fragments with ds_read_b128, like k_conv_tile.  k_bmm (register-direct, no LDS) stays as the fallback for
// operands without a unit stride / unaligned bases.
#define BL_P 36
enum { BL_KV = 0, BL_RV = 1 };
template <int MODE>
__device__ __forceinline__ void bl_load(f32x4 (&dst)[2], const float* base, long sRow, long sK, int row0, int rmax, int k0, int tid) {
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    f32x4 v = {0, 0, 0, 0};
    if (MODE == BL_KV) {
      const int row = (tid >> 3) + 32 * ps, k4 = tid & 7;
      if (row0 + row < rmax) v = *reinterpret_cast<const f32x4*>(base + (size_t)(row0 + row) * sRow + k0 + 4 * k4);
    } else {
      // lanes run along k (32 consecutive k of one row quad): the transposed LDS store below then hits 32 distinct banks per
      // half wave (with lanes along the rows it was an 8-way conflict: PMC LDSBankConflict 27-37 % in these kernels)
      const int kk = tid & 31, r4 = (tid >> 5) + 8 * ps;
      if (row0 + 4 * r4 < rmax) v = *reinterpret_cast<const f32x4*>(base + (size_t)(k0 + kk) * sK + row0 + 4 * r4);
    }
    dst[ps] = v;
  }
}
template <int MODE>
__device__ __forceinline__ void bl_store(float* L, const f32x4 (&src)[2], int tid) {
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    if (MODE == BL_KV) {
      const int row = (tid >> 3) + 32 * ps, k4 = tid & 7;
      *reinterpret_cast<f32x4*>(L + row * BL_P + 4 * k4) = src[ps];
    } else {
      const int kk = tid & 31, r4 = (tid >> 5) + 8 * ps;
#pragma unroll
      for (int r = 0; r < 4; ++r) L[(4 * r4 + r) * BL_P + kk] = src[ps][r];
    }
  }
}

template <int AMODE, int BMODE, bool DUAL>
__global__ void __launch_bounds__(256) k_bmm_lds(BmmArgs P) {
  __shared__ __attribute__((aligned(16))) float lds[2][(DUAL ? 3 : 2) * 64 * BL_P];
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, il = lane & 15, q = lane >> 4;
  const int tiles_n = (P.N + 63) / 64;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
  const int i0 = tm * 64, j0 = tn * 64;
  const int wi = (w >> 1) * 32, wj = (w & 1) * 32;
  f32x4 acc[2][2], acc3[DUAL ? 2 : 1][DUAL ? 2 : 1];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n) { acc[m][n] = f32x4{0, 0, 0, 0}; if (DUAL) acc3[m][n] = f32x4{0, 0, 0, 0}; }
  const int nkb = P.K >> 5;                               // host guarantees K % 32 == 0
  const int npairs = P.A2 ? 2 : 1;
  const int total = nkb * npairs;                         // the second operand pair simply continues the K loop
  f32x4 ra[2], rb[2], ra3[2];
  auto gload = [&](int it) __attribute__((always_inline)) {
    const int pair = it / nkb, k0 = (it - pair * nkb) * 32;
    const float* Ab = (pair ? P.A2 : P.A) + (size_t)blockIdx.y * P.sAb;
    const float* Bb = (pair ? P.B2 : P.B) + (size_t)blockIdx.y * P.sBb;
    bl_load<AMODE>(ra, Ab, P.sAi, P.sAk, i0, P.M, k0, tid);
    bl_load<BMODE>(rb, Bb, P.sBj, P.sBk, j0, P.N, k0, tid);
    if (DUAL) {
      if (pair == 0) bl_load<AMODE>(ra3, P.A3 + (size_t)blockIdx.y * P.sAb, P.sAi, P.sAk, i0, P.M, k0, tid);
    }
  };
  auto lstore = [&](int buf, int it) __attribute__((always_inline)) {
    float* L = lds[buf];
    bl_store<AMODE>(L, ra, tid);
    bl_store<BMODE>(L + 64 * BL_P, rb, tid);
    if (DUAL) {
      if (it < nkb) bl_store<AMODE>(L + 128 * BL_P, ra3, tid);
    }
  };
  gload(0);
  lstore(0, 0);
  __syncthreads();
  for (int it = 0; it < total; ++it) {
    const int cur = it & 1;
    if (it + 1 < total) gload(it + 1);
    const float* LA = lds[cur];
    const float* LB = LA + 64 * BL_P;
    const float* LA3 = LA + 128 * BL_P;
    const bool third = DUAL && it < nkb;                  // A3 rides along with the first pair's B operand
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      f32x4 a[2], b[2], a3[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        a[m] = *reinterpret_cast<const f32x4*>(LA + (wi + 16 * m + il) * BL_P + 16 * g + 4 * q);
        if (third) a3[m] = *reinterpret_cast<const f32x4*>(LA3 + (wi + 16 * m + il) * BL_P + 16 * g + 4 * q);
      }
#pragma unroll
      for (int n = 0; n < 2; ++n) b[n] = *reinterpret_cast<const f32x4*>(LB + (wj + 16 * n + il) * BL_P + 16 * g + 4 * q);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n) {
            acc[m][n] = mfma16u(a[m][r], b[n][r], acc[m][n]);
            if (third) acc3[m][n] = mfma16u(a3[m][r], b[n][r], acc3[m][n]);
          }
    }
    if (it + 1 < total) lstore(cur ^ 1, it + 1);          // the other buffer: its readers finished before the last barrier
    __syncthreads();
  }
#pragma unroll
  for (int which = 0; which < (DUAL ? 2 : 1); ++which) {
    float* Cb = (which ? P.C3 : P.C) + (size_t)blockIdx.y * P.sCb;
    const bool cvec = P.sCi == 1 && (P.sCj & 3) == 0 && (P.sCb & 3) == 0 && ((reinterpret_cast<uintptr_t>(which ? P.C3 : P.C) & 15) == 0);
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const int j = j0 + wj + 16 * n + il;
        if (j >= P.N) continue;
        const int i = i0 + wi + 16 * m + 4 * q;
        if (i >= P.M) continue;
        f32x4 v = P.alpha * (which ? acc3[DUAL ? m : 0][DUAL ? n : 0] : acc[m][n]);
        if (cvec && i + 3 < P.M) {
          float* cp = Cb + (size_t)j * P.sCj + i;
          if (P.accumulate) v += *reinterpret_cast<const f32x4*>(cp);
          *reinterpret_cast<f32x4*>(cp) = v;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (i + r >= P.M) continue;
            float* cp = Cb + (size_t)(i + r) * P.sCi + (size_t)j * P.sCj;
            *cp = P.accumulate ? *cp + v[r] : v[r];
          }
        }
      }
  }
}

// generic-K fallback (K % 16 != 0): scalar, one k per lane quarter
__global__ void __launch_bounds__(256) k_bmm_slow(BmmArgs P) {
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, il = lane & 15, q = lane >> 4;
  const int tiles_n = (P.N + 31) / 32;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
  const int i0 = tm * 32 + (w >> 1) * 16, j0 = tn * 32 + (w & 1) * 16;
  if (i0 >= P.M || j0 >= P.N) return;
  f32x4 acc = {0, 0, 0, 0};
  const bool vi = i0 + il < P.M, vj = j0 + il < P.N;
  for (int pair = 0; pair < 2; ++pair) {
    const float* Ab = (pair ? P.A2 : P.A);
    const float* Bb = (pair ? P.B2 : P.B);
    if (!Ab) break;
    Ab += (size_t)blockIdx.y * P.sAb + (size_t)(vi ? i0 + il : 0) * P.sAi;
    Bb += (size_t)blockIdx.y * P.sBb + (size_t)(vj ? j0 + il : 0) * P.sBj;
    for (int k0 = 0; k0 < P.K; k0 += 4) {
      const int k = k0 + q;
      const float a = (vi && k < P.K) ? Ab[(size_t)k * P.sAk] : 0.f;
      const float b = (vj && k < P.K) ? Bb[(size_t)k * P.sBk] : 0.f;
      acc = mfma16u(a, b, acc);
    }
  }
  float* Cb = P.C + (size_t)blockIdx.y * P.sCb;
  const int j = j0 + il;
  if (j < P.N)
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + 4 * q + r;
      if (i >= P.M) continue;
      float* cp = Cb + (size_t)i * P.sCi + (size_t)j * P.sCj;
      const float v = P.alpha * acc[r];
      *cp = P.accumulate ? *cp + v : v;
    }
}

// ------------------------------------------------------------------ dual softmax rows
// forward: S (primal logits, in place -> P); Wd (tangent logits, kept); Pd <- P (Wd - sum P Wd)
__global__ void __launch_bounds__(256) k_softmax_dual_fwd(float* __restrict__ S, const float* __restrict__ Wd,
                                                          float* __restrict__ Pd, long rows, int T, int dual) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float* s = S + row * T;
  float mx = -3.0e38f;
  for (int j = lane; j < T; j += 64) mx = fmaxf(mx, s[j]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float sum = 0.f;
  for (int j = lane; j < T; j += 64) { const float e = expf(s[j] - mx); s[j] = e; sum += e; }
  sum = wave_sum(sum);
  const float rs = 1.0f / sum;
  float r = 0.f;
  for (int j = lane; j < T; j += 64) {
    const float p = s[j] * rs;
    s[j] = p;
    if (dual) r += p * Wd[row * T + j];
  }
  if (dual) {
    r = wave_sum(r);
    for (int j = lane; j < T; j += 64) Pd[row * T + j] = s[j] * (Wd[row * T + j] - r);
  }
}
// backward (in place): Pb <- Wbar, Pdb <- Wdbar given P, Wd
__global__ void __launch_bounds__(256) k_softmax_dual_bwd(const float* __restrict__ Pm, const float* __restrict__ Wd,
                                                          float* __restrict__ Pb, float* __restrict__ Pdb, long rows, int T) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* p = Pm + row * T;
  const float* wd = Wd + row * T;
  float* pb = Pb + row * T;
  float* pdb = Pdb + row * T;
  float r = 0.f, m = 0.f;
  for (int j = lane; j < T; j += 64) { r += p[j] * wd[j]; m += p[j] * pdb[j]; }
  r = wave_sum(r); m = wave_sum(m);
  float s2 = 0.f;
  for (int j = lane; j < T; j += 64) {
    const float pt = pb[j] + pdb[j] * (wd[j] - r) - wd[j] * m;
    pb[j] = pt;
    s2 += p[j] * pt;
  }
  s2 = wave_sum(s2);
  for (int j = lane; j < T; j += 64) {
    pb[j] = p[j] * (pb[j] - s2);
    pdb[j] = p[j] * (pdb[j] - m);
  }
}

// ------------------------------------------------------------------ embedding / layout glue
// emb[b][j] = cos(t_b f_j), emb[b][half + j] = sin(t_b f_j), f_j = exp(-ln(max_period) j / half)
__global__ void k_timestep_embedding(const float* __restrict__ t, float* __restrict__ emb, int B, int dim, float max_period) {
  const int half = dim / 2;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < B * dim; e += gridDim.x * blockDim.x) {
    const int b = e / dim, j = e - b * dim;
    float v = 0.f;
    if (j < 2 * half) {
      const int jj = j < half ? j : j - half;
      const float f = expf(-logf(max_period) * (float)jj / (float)half);
      const float arg = t[b] * f;
      v = j < half ? cosf(arg) : sinf(arg);
    }
    emb[e] = v;
  }
}
// dual-number version for an argument that depends on the input (log-radius conditioning, NNUnet.py:101-105):
// rows b < Bp primal, rows Bp + b hold the tangent tdot:  d/dt cos(t f) = -f sin(t f) tdot, d/dt sin = f cos tdot
__global__ void k_timestep_embedding_dual(const float* __restrict__ t, float* __restrict__ emb, int Bp, int dim, float max_period) {
  const int half = dim / 2;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < Bp * dim; e += gridDim.x * blockDim.x) {
    const int b = e / dim, j = e - b * dim;
    float v = 0.f, vd = 0.f;
    if (j < 2 * half) {
      const int jj = j < half ? j : j - half;
      const float f = expf(-logf(max_period) * (float)jj / (float)half);
      const float arg = t[b] * f, td = t[Bp + b];
      const float c = cosf(arg), s = sinf(arg);
      v = j < half ? c : s;
      vd = j < half ? -f * s * td : f * c * td;
    }
    emb[e] = v;
    emb[(size_t)Bp * dim + e] = vd;
  }
}

// NormalizeLogRadius on dual numbers (NN.py:56-70 followed by the x sqrt(n) rescale of NNUnet.py:205 /
// NNUnet1D.py:134): r = |x| + eps; out = scale x / r; logr = log r; tangent: rdot = x.xdot/|x|,
// outdot = scale (xdot / r - x rdot / r^2), logr_dot = rdot / r.   One wave per row.
__global__ void __launch_bounds__(256) k_normalize_dual(const float* __restrict__ x, float* __restrict__ out,
                                                        float* __restrict__ logr, int Bp, int n, int dual, float scale,
                                                        float eps) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= Bp) return;
  const float* xp = x + (size_t)b * n;
  const float* xt = x + (size_t)(b + Bp) * n;
  float ss = 0.f, sd = 0.f;
  for (int i = lane; i < n; i += 64) { const float v = xp[i]; ss += v * v; if (dual) sd += v * xt[i]; }
  ss = wave_sum(ss);
  if (dual) sd = wave_sum(sd);
  const float nr = sqrtf(ss), r = nr + eps;
  const float rdot = dual ? sd / nr : 0.f;
  for (int i = lane; i < n; i += 64) {
    const float v = xp[i];
    out[(size_t)b * n + i] = scale * (v / r);
    if (dual) out[(size_t)(b + Bp) * n + i] = scale * (xt[i] / r - v * rdot / (r * r));
  }
  if (lane == 0) { logr[b] = logf(r); if (dual) logr[Bp + b] = rdot / r; }
}

// flat (B, C*H*W) channel-major, per-channel 'C' (h*W+w) or 'F' (w*H+h) order  <->  [B][H][W][C], with scale
__global__ void k_flat_to_cl(const float* __restrict__ flat, float* __restrict__ img, int B, int C, int H, int W, int forder,
                             float scale) {
  const long tot = (long)B * H * W * C;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < tot; e += (long)gridDim.x * blockDim.x) {
    const int c = (int)(e % C);
    const int w = (int)((e / C) % W);
    const int h = (int)((e / ((long)C * W)) % H);
    const long b = e / ((long)C * W * H);
    const long idx = forder ? (long)w * H + h : (long)h * W + w;
    img[e] = flat[(b * C + c) * (long)H * W + idx] * scale;
  }
}
__global__ void k_cl_to_flat(const float* __restrict__ img, float* __restrict__ flat, int B, int C, int H, int W, int forder,
                             float scale) {
  const long tot = (long)B * H * W * C;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < tot; e += (long)gridDim.x * blockDim.x) {
    const int c = (int)(e % C);
    const int w = (int)((e / C) % W);
    const int h = (int)((e / ((long)C * W)) % H);
    const long b = e / ((long)C * W * H);
    const long idx = forder ? (long)w * H + h : (long)h * W + w;
    flat[(b * C + c) * (long)H * W + idx] = img[e] * scale;
  }
}
// out[n][h][w][c] = sum of the 2x2 block of in[n][2h+dh][2w+dw][c]
__global__ void k_sum2x2(const float* __restrict__ in, float* __restrict__ out, int N, int H, int W, int C) {
  const long tot = (long)N * H * W * C;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < tot; e += (long)gridDim.x * blockDim.x) {
    const int c = (int)(e % C);
    const int w = (int)((e / C) % W);
    const int h = (int)((e / ((long)C * W)) % H);
    const long n = e / ((long)C * W * H);
    const float* p = in + ((n * 2 * H + 2 * h) * (long)(2 * W) + 2 * w) * C + c;
    out[e] = (p[0] + p[C]) + (p[(long)2 * W * C] + p[(long)2 * W * C + C]);
  }
}

// ------------------------------------------------------------------ embedding-projection bank
// Every ResBlock projects the SAME time embedding through its own Linear(4*mc -> co) (emb_layers, model/unet.py:145-151,
// applied at :176-180).  Upstream that is 22 tiny GEMMs forward and 22 x (wgrad + bias sum + dgrad) backward per step —
// 4-16 us launches that are pure latency at the 32-row shard.  The bank runs them as ONE launch per direction straight on
// the PyTorch-layout parameters (no packed images): job j = one ResBlock; a workgroup owns 32 output channels of one job.
//   forward : out_j[r][c]  = (r < n_bias ? b_j[c] : 0) + sum_k semb[r][k] W_j[c][k]
//   wgrad   : dW_j[c][k]   = sum_r dout_j[r][c] semb[r][k]          (rows in order: deterministic)
//             db_j[c]      = sum_{r < n_bias} dout_j[r][c]          (also written to db2_j: the conv bias that is added at
//                                                                    the same place has the same gradient)
//   dgrad   : dsemb[r][k]  = sum_j sum_c dout_j[r][c] W_j[c][k]     (jobs and channels in order)
// K <= 512 and a multiple of 4; out_j / dout_j are [rows][co] contiguous per job.
#define EB_ROWS 64
template <int MODE>    // 0 forward, 1 wgrad
__global__ void __launch_bounds__(256) k_emb_bank(const msgm_emb_job_t* __restrict__ jobs, int n_jobs, const float* __restrict__ semb,
                                                  int R, int K, int n_bias) {
  extern __shared__ float lds[];
  const int KP = K + 4;                              // pitch: rows 16-byte aligned, consecutive rows on different banks
  float* sS = lds;                                   // [EB_ROWS][KP]   semb rows of the current row tile
  float* sW = lds + EB_ROWS * KP;                    // forward: [32][KP] weight rows; wgrad: [EB_ROWS][32] dout tile
  int lo = 0, hi = n_jobs - 1;
  const int blk = blockIdx.x;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (jobs[mid].block_begin <= blk) lo = mid; else hi = mid - 1; }
  const msgm_emb_job_t J = jobs[lo];
  const int c0 = (blk - J.block_begin) * 32, tid = threadIdx.x;
  const int K4 = K >> 2;
  if (MODE == 0) {
    for (int i = tid; i < 32 * K4; i += 256) {       // weight rows c0..c0+31, coalesced 16-byte loads
      const int c = i / K4, k4 = i - c * K4;
      f32x4 w = f32x4{0.f, 0.f, 0.f, 0.f};
      if (c0 + c < J.co) w = *reinterpret_cast<const f32x4*>(J.W + (size_t)(c0 + c) * K + 4 * k4);
      *reinterpret_cast<f32x4*>(sW + c * KP + 4 * k4) = w;
    }
    const int cl = tid & 31, rg = tid >> 5;
    const bool okc = c0 + cl < J.co;
    const float bias = (okc && J.b) ? J.b[c0 + cl] : 0.f;
    for (int r0 = blockIdx.y * EB_ROWS; r0 < R; r0 += gridDim.y * EB_ROWS) {
      __syncthreads();
      for (int i = tid; i < EB_ROWS * K4; i += 256) {
        const int r = i / K4, k4 = i - r * K4;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (r0 + r < R) v = *reinterpret_cast<const f32x4*>(semb + (size_t)(r0 + r) * K + 4 * k4);
        *reinterpret_cast<f32x4*>(sS + r * KP + 4 * k4) = v;
      }
      __syncthreads();
      float acc[EB_ROWS / 8];
#pragma unroll
      for (int j = 0; j < EB_ROWS / 8; ++j) acc[j] = 0.f;
      for (int k4 = 0; k4 < K4; ++k4) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(sW + cl * KP + 4 * k4);
#pragma unroll
        for (int j = 0; j < EB_ROWS / 8; ++j) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(sS + (rg + 8 * j) * KP + 4 * k4);
          acc[j] = fmaf(v[0], w[0], acc[j]); acc[j] = fmaf(v[1], w[1], acc[j]);
          acc[j] = fmaf(v[2], w[2], acc[j]); acc[j] = fmaf(v[3], w[3], acc[j]);
        }
      }
#pragma unroll
      for (int j = 0; j < EB_ROWS / 8; ++j) {
        const int r = r0 + rg + 8 * j;
        if (okc && r < R) J.out[(size_t)r * J.co + c0 + cl] = acc[j] + (r < n_bias ? bias : 0.f);
      }
    }
  } else {
    // thread: 4 consecutive k (k4 = tid % K4 ... strided over K4) x 4 channels (cg = group of 4 of the 32)
    const int kt = tid & 31, cg = tid >> 5;          // kt walks k4 = kt, kt + 32, ...; K4 <= 128 -> at most 4 rounds
    float acc[4][4][4];                              // [round][channel][k]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][b][c] = 0.f;
    float bsum = 0.f;                                // threads 0..31: bias gradient of channel c0 + tid
    for (int r0 = 0; r0 < R; r0 += EB_ROWS) {
      __syncthreads();
      for (int i = tid; i < EB_ROWS * K4; i += 256) {
        const int r = i / K4, k4 = i - r * K4;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (r0 + r < R) v = *reinterpret_cast<const f32x4*>(semb + (size_t)(r0 + r) * K + 4 * k4);
        *reinterpret_cast<f32x4*>(sS + r * KP + 4 * k4) = v;
      }
      for (int i = tid; i < EB_ROWS * 32; i += 256) {
        const int r = i >> 5, c = i & 31;
        sW[r * 36 + c] = (r0 + r < R && c0 + c < J.co) ? J.dout[(size_t)(r0 + r) * J.co + c0 + c] : 0.f;
      }
      __syncthreads();
      const int rend = min(EB_ROWS, R - r0);
      for (int r = 0; r < rend; ++r) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(sW + r * 36 + 4 * cg);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const int k4 = kt + 32 * a;
          if (k4 < K4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(sS + r * KP + 4 * k4);
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
              for (int c = 0; c < 4; ++c) acc[a][b][c] = fmaf(g[b], v[c], acc[a][b][c]);
          }
        }
      }
      if (tid < 32) {
        const int rb = min(rend, n_bias - r0);
        for (int r = 0; r < rb; ++r) bsum += sW[r * 36 + tid];
      }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int k4 = kt + 32 * a;
      if (k4 < K4) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int c = c0 + 4 * cg + b;
          if (c < J.co) *reinterpret_cast<f32x4*>(J.dW + (size_t)c * K + 4 * k4) = f32x4{acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]};
        }
      }
    }
    if (tid < 32 && c0 + tid < J.co) {
      if (J.db) J.db[c0 + tid] = bsum;
      if (J.db2) J.db2[c0 + tid] = bsum;
    }
  }
}

// dsemb[r][k] = sum over jobs and their channels, in a fixed order.  Grid (row tiles of 8, k tiles of 32); thread (k, cs):
// channel slice cs of 8 takes the channels c = cs (mod 8) of every 64-channel chunk, the 8 slices meet through LDS in slice
// order.  (One thread per k walking all ~1400 channels alone was 125 us at 32 rows: a serial chain of dependent loads.)
__global__ void __launch_bounds__(256) k_emb_bank_dgrad(const msgm_emb_job_t* __restrict__ jobs, int n_jobs, float* __restrict__ dsemb,
                                                        int R, int K) {
  __shared__ float sG[8][68];
  __shared__ float red[8][8][33];
  const int tid = threadIdx.x, kl = tid & 31, cs = tid >> 5;
  const int k = blockIdx.y * 32 + kl, r0 = blockIdx.x * 8;
  const bool okk = k < K;
  float acc[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) acc[r] = 0.f;
  for (int j = 0; j < n_jobs; ++j) {
    const msgm_emb_job_t J = jobs[j];
    for (int cb = 0; cb < J.co; cb += 64) {
      __syncthreads();
      for (int i = tid; i < 8 * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        sG[r][c] = (r0 + r < R && cb + c < J.co) ? J.dout[(size_t)(r0 + r) * J.co + cb + c] : 0.f;
      }
      __syncthreads();
      float w[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {                      // this slice's 8 channels of the chunk: loads first, then the FMAs
        const int c = cb + cs + 8 * i;
        w[i] = (okk && c < J.co) ? J.W[(size_t)c * K + k] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 8; ++r) acc[r] = fmaf(sG[r][cs + 8 * i], w[i], acc[r]);
    }
  }
#pragma unroll
  for (int r = 0; r < 8; ++r) red[cs][r][kl] = acc[r];
  __syncthreads();
  {                                                      // thread (kl, r = cs): the 8 slices of row r in slice order
    const int r = cs;
    float t = red[0][r][kl];
#pragma unroll
    for (int sl = 1; sl < 8; ++sl) t += red[sl][r][kl];
    if (okk && r0 + r < R) dsemb[(size_t)(r0 + r) * K + k] = t;
  }
}

// ============================================================ C ABI
static inline hipStream_t S(msgm_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

extern "C" {

static int gn_chunks(int Bp, int P, int* chunk, int* sub) {
  // fp32 partial sums run over the SAME sub-chunks of a sample whatever the batch size (32 per sample, >= 64 pixels
  // each) and are combined in double: a row's statistics — bit for bit — do not depend on how many other rows share
  // the launch, so a row's result is the same in a 32-row shard and in the 256-row batch (tests/test_fullsize_gpu.py).
  // How many consecutive sub-chunks one workgroup takes is free (the test pairs of test_fullsize_gpu.py hold bit for bit;
  // across very different batch sizes the double-precision regrouping could move a float by an ulp).
  int c = (P + 31) / 32;
  if (c < 64) c = 64;
  if (c > P) c = P;
  *sub = c;
  const int nsub = (P + c - 1) / c;
  // ~512 workgroups in all: every workgroup ends with a fixed tail (LDS group sums of up to 5 moments + the parameter
  // partials), so fewer, fatter workgroups win — measured 1024 -> 512: B = 32 step 23.5 -> 23.1 ms, B = 256 129.7 -> 128.8 ms
  static const int target = getenv("MSGM_GN_WGS") ? atoi(getenv("MSGM_GN_WGS")) : 512;
  int want = (target + Bp - 1) / Bp;            // workgroups per sample
  if (want < 1) want = 1;
  int m = nsub / want;
  if (m < 1) m = 1;
  *chunk = m * c;
  return (P + m * c - 1) / (m * c);
}

// the elementwise passes have no summation order to protect: enough (sample, chunk) workgroups to fill 256 CUs a few
// times over, >= 64 pixels per chunk
static int gn_chunks_apply(int Bp, int P, int* chunk) {
  int n = (1024 + Bp - 1) / Bp;
  if (n < 1) n = 1;
  int c = (P + n - 1) / n;
  if (c < 64) c = 64;
  if (c > P) c = P;
  *chunk = c;
  return (P + c - 1) / c;
}

// workspace: acc[Bp][GN_SLOTS][G][8] doubles (per-chunk moment slots) | accf[Bp][G][8] doubles (backward moments summed)
// | [Bp][G][4] floats (finalised forward statistics) | [2][Bp*GN_SLOTS][256] floats (dgamma / dbeta slots, C <= 256).
// Nothing in it has to be zero on entry: every slot that is read was written by the same call.
static inline size_t gn_acc_bytes(int32_t Bp, int32_t G) { return (size_t)Bp * GN_SLOTS * (size_t)G * 8 * sizeof(double); }
static inline size_t gn_accf_bytes(int32_t Bp, int32_t G) { return (size_t)Bp * (size_t)G * 8 * sizeof(double); }
static inline size_t gn_stats_bytes(int32_t Bp, int32_t G) { return (size_t)Bp * (size_t)G * 4 * sizeof(float); }
size_t msgm_groupnorm_workspace(int32_t Bp, int32_t G) {
  return gn_acc_bytes(Bp, G) + gn_accf_bytes(Bp, G) + gn_stats_bytes(Bp, G) + (size_t)2 * Bp * GN_SLOTS * 256 * sizeof(float);
}

static int gn_forward_impl(const float* x, int32_t C0, const float* x1, int32_t C, const float* gamma, const float* beta, float* out,
                           float* stats, int32_t Bp, int32_t P, int32_t G, int32_t dual, int32_t silu, float eps, void* workspace,
                           size_t workspace_bytes, msgm_stream_t stream) {
  if (!x || !gamma || !beta || !out || !workspace || Bp <= 0 || P <= 0 || C <= 0 || G <= 0) return MSGM_E_BADARG;
  if (C % G || C > 256 || G > 64 || (x1 && (C0 % 4 || (C - C0) % 4 || C0 <= 0 || C0 >= C))) return MSGM_E_UNSUPPORTED;
  if (workspace_bytes < msgm_groupnorm_workspace(Bp, G)) return MSGM_E_WORKSPACE;
  GnArgs A{x, gamma, beta, out, reinterpret_cast<double*>(workspace), stats, P, C, G, Bp, dual, silu, 0, eps,
           nullptr, nullptr, nullptr, nullptr, x1, C0};
  const int nch = gn_chunks(Bp, P, &A.chunk, &A.sub);
  if (nch > GN_SLOTS) return MSGM_E_UNSUPPORTED;
  A.nch = nch;
  // two launches: chunk moments -> slots, then the apply pass (every workgroup finalises its sample's statistics itself)
  if (C % 4 == 0) hipLaunchKernelGGL(k_gn_fwd_reduce<true>, dim3(Bp, nch), dim3(256), 0, S(stream), A);
  else hipLaunchKernelGGL(k_gn_fwd_reduce<false>, dim3(Bp, nch), dim3(256), 0, S(stream), A);
  const int nap = gn_chunks_apply(Bp, P, &A.chunk);
  if (C % 4 == 0) hipLaunchKernelGGL(k_gn_fwd_apply<true>, dim3(nap, Bp), dim3(256), 0, S(stream), A);
  else hipLaunchKernelGGL(k_gn_fwd_apply<false>, dim3(nap, Bp), dim3(256), 0, S(stream), A);
  return msgm_check_launch();
}

int msgm_groupnorm_dual_forward(const float* x, const float* gamma, const float* beta, float* out, float* stats, int32_t Bp,
                                int32_t P, int32_t C, int32_t G, int32_t dual, int32_t silu, float eps, void* workspace,
                                size_t workspace_bytes, msgm_stream_t stream) {
  return gn_forward_impl(x, C, nullptr, C, gamma, beta, out, stats, Bp, P, G, dual, silu, eps, workspace, workspace_bytes, stream);
}

int msgm_groupnorm_dual_forward2(const float* x0, int32_t C0, const float* x1, int32_t C1, const float* gamma, const float* beta,
                                 float* out, float* stats, int32_t Bp, int32_t P, int32_t G, int32_t dual, int32_t silu, float eps,
                                 void* workspace, size_t workspace_bytes, msgm_stream_t stream) {
  if (!x1 || C1 <= 0) return MSGM_E_BADARG;
  return gn_forward_impl(x0, C0, x1, C0 + C1, gamma, beta, out, stats, Bp, P, G, dual, silu, eps, workspace, workspace_bytes, stream);
}

int msgm_groupnorm_affine(const float* x0, int32_t C0, const float* x1, int32_t C1, const float* gamma, const float* beta,
                          float* scale, float* shift, int32_t Bp, int32_t P, int32_t G, float eps, void* workspace,
                          size_t workspace_bytes, msgm_stream_t stream) {
  if (!x0 || !gamma || !beta || !scale || !shift || !workspace || Bp <= 0 || P <= 0 || C0 <= 0 || G <= 0 || (x1 && C1 <= 0))
    return MSGM_E_BADARG;
  const int C = C0 + (x1 ? C1 : 0);
  if (C % G || C > 256 || G > 64 || (x1 && (C0 % 4 || C1 % 4))) return MSGM_E_UNSUPPORTED;
  if (workspace_bytes < msgm_groupnorm_workspace(Bp, G)) return MSGM_E_WORKSPACE;
  GnArgs A{x0, gamma, beta, nullptr, reinterpret_cast<double*>(workspace), nullptr, P, C, G, Bp, 0, 0, 0, eps,
           nullptr, nullptr, nullptr, nullptr, x1, C0};
  const int nch = gn_chunks(Bp, P, &A.chunk, &A.sub);
  if (nch > GN_SLOTS) return MSGM_E_UNSUPPORTED;
  A.nch = nch;
  if (C % 4 == 0 && C0 % 4 == 0) hipLaunchKernelGGL(k_gn_fwd_reduce<true>, dim3(Bp, nch), dim3(256), 0, S(stream), A);
  else hipLaunchKernelGGL(k_gn_fwd_reduce<false>, dim3(Bp, nch), dim3(256), 0, S(stream), A);
  hipLaunchKernelGGL(k_gn_affine, dim3(Bp), dim3(256), 0, S(stream), A, scale, shift);
  return msgm_check_launch();
}

// The same affine map from the per-channel partial sums a producing convolution left behind (msgm_conv_fuse_t.chanstats:
// cs[n][slot][{sum, sum of squares}][Csrc], fp32 sums over one wave's 64..256 pixels): thread c adds the slots of its
// channel in slot order in double, the channels of a group are combined in double — no pass over the tensor itself.
// One block per sample; two sources = the decoder's cat([h, skip]) (each source has its own producer and slot count).
__global__ void __launch_bounds__(256) k_gn_affine_cs(const float* __restrict__ cs0, int S0, int C0, const float* __restrict__ cs1,
                                                      int S1, int C1, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, int P, int G, float eps,
                                                      float* __restrict__ scale, float* __restrict__ shift) {
  __shared__ double red[2][256];
  const int b = blockIdx.x, c = threadIdx.x, C = C0 + C1, cpg = C / G;
  if (c < C) {
    const bool second = c >= C0;
    const int Cs = second ? C1 : C0, S = second ? S1 : S0, cc = second ? c - C0 : c;
    const float* p = (second ? cs1 : cs0) + (size_t)b * S * 2 * Cs + cc;
    double s = 0.0, ss = 0.0;
    for (int k = 0; k < S; ++k) { s += (double)p[(size_t)(2 * k) * Cs]; ss += (double)p[(size_t)(2 * k + 1) * Cs]; }
    red[0][c] = s; red[1][c] = ss;
  }
  __syncthreads();
  if (c < C) {
    const int g0 = (c / cpg) * cpg;
    double a8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int k = 0; k < cpg; ++k) { a8[0] += red[0][g0 + k]; a8[1] += red[1][g0 + k]; }
    float mu, inv, md, a;
    gn_stats(a8, (double)P * cpg, eps, mu, inv, md, a);
    const float sc = inv * gamma[c];
    scale[(size_t)b * C + c] = sc;
    shift[(size_t)b * C + c] = beta[c] - mu * sc;
  }
}

int msgm_groupnorm_affine_chanstats(const float* cs0, int32_t S0, int32_t C0, const float* cs1, int32_t S1, int32_t C1,
                                    const float* gamma, const float* beta, float* scale, float* shift, int32_t Bp, int32_t P,
                                    int32_t G, float eps, msgm_stream_t stream) {
  if (!cs0 || !gamma || !beta || !scale || !shift || Bp <= 0 || P <= 0 || C0 <= 0 || S0 <= 0 || G <= 0) return MSGM_E_BADARG;
  if (cs1 ? (C1 <= 0 || S1 <= 0) : (C1 != 0)) return MSGM_E_BADARG;
  const int C = C0 + C1;
  if (C % G || C > 256 || G > 64) return MSGM_E_UNSUPPORTED;
  hipLaunchKernelGGL(k_gn_affine_cs, dim3(Bp), dim3(256), 0, S(stream), cs0, S0, C0, cs1, S1, C1, gamma, beta, P, G, eps, scale,
                     shift);
  return msgm_check_launch();
}

// pslots_ext != nullptr: the per-(sample, chunk) dgamma | dbeta partials go to the CALLER's buffer and their slot-ordered
// sums are described in jobs_out[0..1] instead of being launched — a backward pass batches the parameter reductions of all
// its GroupNorms (and its convolutions' weight gradients) into ONE msgm_slot_reduce_batched launch.
static int gn_backward_impl(const float* x, int32_t C0, const float* x1, int32_t C, const float* gamma, const float* beta,
                            const float* stats, const float* gout, float* gx, float* gx1, float* dgamma, float* dbeta, int32_t Bp,
                            int32_t P, int32_t G, int32_t silu, float eps, const float* residual, void* workspace,
                            size_t workspace_bytes, msgm_stream_t stream, float* pslots_ext = nullptr, size_t pslots_bytes = 0,
                            msgm_reduce_job_t* jobs_out = nullptr, int32_t* n_jobs_out = nullptr, const float* residual2 = nullptr) {
  if (!x || !gamma || !beta || !stats || !gout || !gx || !dgamma || !dbeta || !workspace || Bp <= 0 || P <= 0 || C <= 0 || G <= 0)
    return MSGM_E_BADARG;
  if (C % G || C > 256 || G > 64) return MSGM_E_UNSUPPORTED;
  if (x1 && (!gx1 || residual || residual2 || C0 % 4 || (C - C0) % 4 || C0 <= 0 || C0 >= C)) return MSGM_E_UNSUPPORTED;
  if (workspace_bytes < msgm_groupnorm_workspace(Bp, G)) return MSGM_E_WORKSPACE;
  GnArgs A{x, gamma, beta, nullptr, reinterpret_cast<double*>(workspace), const_cast<float*>(stats), P, C, G, Bp, 1, silu, 0,
           eps, gout, gx, dgamma, dbeta, x1, C0, gx1};
  const int nch = gn_chunks(Bp, P, &A.chunk, &A.sub);
  if (nch > GN_SLOTS) return MSGM_E_UNSUPPORTED;
  A.nch = nch;
  char* wsb = reinterpret_cast<char*>(workspace);
  A.accf = nullptr;
  A.pslots = reinterpret_cast<float*>(wsb + gn_acc_bytes(Bp, G) + gn_accf_bytes(Bp, G) + gn_stats_bytes(Bp, G));
  const size_t nslots = (size_t)Bp * nch;
  if (pslots_ext) {
    if (!jobs_out || !n_jobs_out || pslots_bytes < 2 * nslots * C * sizeof(float)) return MSGM_E_WORKSPACE;
    A.pslots = pslots_ext;
  }
  A.resid = residual;
  A.resid2 = residual2;
  if (C % 4 == 0) hipLaunchKernelGGL(k_gn_bwd_reduce<true>, dim3(Bp, nch), dim3(256), 0, S(stream), A);
  else hipLaunchKernelGGL(k_gn_bwd_reduce<false>, dim3(Bp, nch), dim3(256), 0, S(stream), A);
  if (pslots_ext) {
    for (int w = 0; w < 2; ++w)
      jobs_out[w] = msgm_reduce_job_t{A.pslots + (size_t)w * nslots * C, w == 0 ? dgamma : dbeta, nullptr, (int64_t)C, (int64_t)C, 0, 0,
                                      (int32_t)nslots, 1, 0, 0, 0, 0, 1, 0};
    *n_jobs_out = 2;
  } else {
    const int nbc = (C + 31) / 32;
    hipLaunchKernelGGL(k_gn_param_reduce, dim3(2 * nbc), dim3(1024), 0, S(stream), (const float*)A.pslots, nslots, C, dgamma, dbeta,
                       nbc);
  }
  const int nap = gn_chunks_apply(Bp, P, &A.chunk);
  if (C % 4 == 0) hipLaunchKernelGGL(k_gn_bwd_apply<true>, dim3(nap, Bp), dim3(256), 0, S(stream), A);
  else hipLaunchKernelGGL(k_gn_bwd_apply<false>, dim3(nap, Bp), dim3(256), 0, S(stream), A);
  return msgm_check_launch();
}

size_t msgm_groupnorm_param_slots_bytes(int32_t Bp, int32_t P, int32_t C) {
  if (Bp <= 0 || P <= 0 || C <= 0) return 0;
  int chunk, sub;
  const int nch = gn_chunks(Bp, P, &chunk, &sub);
  return (size_t)2 * Bp * nch * C * sizeof(float);
}

int msgm_groupnorm_dual_backward_slots(const float* x0, int32_t C0, const float* x1, int32_t C1, const float* gamma,
                                       const float* beta, const float* stats, const float* gout, float* gx0, float* gx1,
                                       float* dgamma, float* dbeta, int32_t Bp, int32_t P, int32_t G, int32_t silu, float eps,
                                       const float* residual, const float* residual2, void* workspace, size_t workspace_bytes,
                                       float* pslots, size_t pslots_bytes, msgm_reduce_job_t* jobs_out, int32_t* n_jobs_out,
                                       msgm_stream_t stream) {
  if (!pslots || !jobs_out || !n_jobs_out || (x1 ? C1 <= 0 : C1 != 0)) return MSGM_E_BADARG;
  *n_jobs_out = 0;
  return gn_backward_impl(x0, C0, x1, C0 + C1, gamma, beta, stats, gout, gx0, gx1, dgamma, dbeta, Bp, P, G, silu, eps, residual,
                          workspace, workspace_bytes, stream, pslots, pslots_bytes, jobs_out, n_jobs_out, residual2);
}

int msgm_groupnorm_dual_backward(const float* x, const float* gamma, const float* beta, const float* stats, const float* gout,
                                 float* gx, float* dgamma, float* dbeta, int32_t Bp, int32_t P, int32_t C, int32_t G,
                                 int32_t silu, float eps, const float* residual, void* workspace, size_t workspace_bytes,
                                 msgm_stream_t stream) {
  return gn_backward_impl(x, C, nullptr, C, gamma, beta, stats, gout, gx, nullptr, dgamma, dbeta, Bp, P, G, silu, eps, residual,
                          workspace, workspace_bytes, stream);
}

int msgm_groupnorm_dual_backward2(const float* x0, int32_t C0, const float* x1, int32_t C1, const float* gamma, const float* beta,
                                  const float* stats, const float* gout, float* gx0, float* gx1, float* dgamma, float* dbeta,
                                  int32_t Bp, int32_t P, int32_t G, int32_t silu, float eps, void* workspace,
                                  size_t workspace_bytes, msgm_stream_t stream) {
  if (!x1 || C1 <= 0) return MSGM_E_BADARG;
  return gn_backward_impl(x0, C0, x1, C0 + C1, gamma, beta, stats, gout, gx0, gx1, dgamma, dbeta, Bp, P, G, silu, eps, nullptr,
                          workspace, workspace_bytes, stream);
}

int msgm_bmm(const float* A, const float* B, const float* A2, const float* B2, float* C, int32_t M, int32_t N, int32_t K,
             int32_t batch, int64_t sAb, int64_t sAi, int64_t sAk, int64_t sBb, int64_t sBk, int64_t sBj, int64_t sCb,
             int64_t sCi, int64_t sCj, float alpha, int32_t accumulate, msgm_stream_t stream) {
  return msgm_bmm_dual(A, B, A2, B2, nullptr, C, nullptr, M, N, K, batch, sAb, sAi, sAk, sBb, sBk, sBj, sCb, sCi, sCj, alpha,
                       accumulate, stream);
}

int msgm_bmm_dual(const float* A, const float* B, const float* A2, const float* B2, const float* B3, float* C, float* C3,
                  int32_t M, int32_t N, int32_t K, int32_t batch, int64_t sAb, int64_t sAi, int64_t sAk, int64_t sBb,
                  int64_t sBk, int64_t sBj, int64_t sCb, int64_t sCi, int64_t sCj, float alpha, int32_t accumulate,
                  msgm_stream_t stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || batch <= 0 || ((A2 == nullptr) != (B2 == nullptr))) return MSGM_E_BADARG;
  if ((B3 == nullptr) != (C3 == nullptr)) return MSGM_E_BADARG;
  // the second output shares the A operand: C3 = alpha A.B3 — built for the case where the host swap below turns A
  // into the kernel's B role (outputs contiguous along j, as every attention product here), K % 16 == 0
  if (B3 && (!(sCj == 1 && sCi != 1) || K % 16)) return MSGM_E_UNSUPPORTED;
  BmmArgs P{A, B, A2, B2, C, M, N, K, batch, sAb, sAi, sAk, sBb, sBk, sBj, sCb, sCi, sCj, alpha, accumulate, B3, C3};
  if (sCj == 1 && sCi != 1) {
    // C^T = B^T A^T: make the output's contiguous dimension the MFMA row index (16-B stores)
    P.A = B; P.B = A; P.A2 = B2; P.B2 = A2;
    P.M = N; P.N = M;
    P.sAb = sBb; P.sAi = sBj; P.sAk = sBk;
    P.sBb = sAb; P.sBk = sAk; P.sBj = sAi;
    P.sCi = sCj; P.sCj = sCi;
  }
  if (K % 16) {
    dim3 grid((unsigned)(((P.M + 31) / 32) * ((P.N + 31) / 32)), (unsigned)batch);
    hipLaunchKernelGGL(k_bmm_slow, grid, dim3(256), 0, S(stream), P);
    return msgm_check_launch();
  }
  auto al16 = [](const float* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  // LDS-staged kernel: every operand has a unit stride (along k or along its row index), 16-byte aligned bases and
  // strides, K a multiple of 32, M and N multiples of 4
  {
    static const bool no_lds = getenv("MSGM_BMM_NO_LDS") != nullptr;   // diagnostic A/B
    auto mode = [&](long sRow, long sK, long sB) { return (sK == 1 && sRow % 4 == 0 && sB % 4 == 0) ? (int)BL_KV
                                                        : (sRow == 1 && sK % 4 == 0 && sB % 4 == 0) ? (int)BL_RV : -1; };
    const int am = mode(P.sAi, P.sAk, P.sAb), bm = mode(P.sBj, P.sBk, P.sBb);
    if (!no_lds && am >= 0 && bm >= 0 && P.K % 32 == 0 && P.M % 4 == 0 && P.N % 4 == 0 && al16(P.A) && al16(P.A2) && al16(P.B) &&
        al16(P.B2) && al16(P.A3)) {
      dim3 grid((unsigned)(((P.M + 63) / 64) * ((P.N + 63) / 64)), (unsigned)batch);
#define BL_GO(AM, BM)                                                                                              \
  do {                                                                                                             \
    if (P.A3) hipLaunchKernelGGL((k_bmm_lds<AM, BM, true>), grid, dim3(256), 0, S(stream), P);                      \
    else hipLaunchKernelGGL((k_bmm_lds<AM, BM, false>), grid, dim3(256), 0, S(stream), P);                          \
  } while (0)
      if (am == BL_KV && bm == BL_KV) BL_GO(BL_KV, BL_KV);
      else if (am == BL_KV) BL_GO(BL_KV, BL_RV);
      else if (bm == BL_KV) BL_GO(BL_RV, BL_KV);
      else BL_GO(BL_RV, BL_RV);
#undef BL_GO
      return msgm_check_launch();
    }
  }
  const bool avec = P.sAk == 1 && (P.sAi % 4 == 0) && (P.sAb % 4 == 0) && al16(P.A) && al16(P.A2);
  const bool bvec = P.sBk == 1 && (P.sBj % 4 == 0) && (P.sBb % 4 == 0) && al16(P.B) && al16(P.B2);
  const int tiles = ((P.M + 63) / 64) * ((P.N + 63) / 64);
  dim3 grid((unsigned)tiles, (unsigned)batch);
  if (P.A3) {
    const bool avec3 = avec && al16(P.A3);
    if (avec3 && bvec) hipLaunchKernelGGL((k_bmm<2, 2, true, true, true>), grid, dim3(256), 0, S(stream), P);
    else if (avec3) hipLaunchKernelGGL((k_bmm<2, 2, true, false, true>), grid, dim3(256), 0, S(stream), P);
    else if (bvec) hipLaunchKernelGGL((k_bmm<2, 2, false, true, true>), grid, dim3(256), 0, S(stream), P);
    else hipLaunchKernelGGL((k_bmm<2, 2, false, false, true>), grid, dim3(256), 0, S(stream), P);
    return msgm_check_launch();
  }
  if (avec && bvec) hipLaunchKernelGGL((k_bmm<2, 2, true, true>), grid, dim3(256), 0, S(stream), P);
  else if (avec) hipLaunchKernelGGL((k_bmm<2, 2, true, false>), grid, dim3(256), 0, S(stream), P);
  else if (bvec) hipLaunchKernelGGL((k_bmm<2, 2, false, true>), grid, dim3(256), 0, S(stream), P);
  else hipLaunchKernelGGL((k_bmm<2, 2, false, false>), grid, dim3(256), 0, S(stream), P);
  return msgm_check_launch();
}

int msgm_softmax_dual_forward(float* Sm, const float* Wd, float* Pd, int64_t rows, int32_t T, int32_t dual, msgm_stream_t stream) {
  if (!Sm || rows <= 0 || T <= 0 || (dual && (!Wd || !Pd))) return MSGM_E_BADARG;
  hipLaunchKernelGGL(k_softmax_dual_fwd, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, S(stream), Sm, Wd, Pd, (long)rows, T, dual);
  return msgm_check_launch();
}

int msgm_softmax_dual_backward(const float* Pm, const float* Wd, float* Pb, float* Pdb, int64_t rows, int32_t T,
                               msgm_stream_t stream) {
  if (!Pm || !Wd || !Pb || !Pdb || rows <= 0 || T <= 0) return MSGM_E_BADARG;
  hipLaunchKernelGGL(k_softmax_dual_bwd, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, S(stream), Pm, Wd, Pb, Pdb, (long)rows, T);
  return msgm_check_launch();
}

int msgm_timestep_embedding(const float* t, float* emb, int32_t B, int32_t dim, float max_period, msgm_stream_t stream) {
  if (!t || !emb || B <= 0 || dim <= 0) return MSGM_E_BADARG;
  hipLaunchKernelGGL(k_timestep_embedding, dim3(grid_for((int64_t)B * dim, 256)), dim3(256), 0, S(stream), t, emb, B, dim, max_period);
  return msgm_check_launch();
}

int msgm_timestep_embedding_dual(const float* t, float* emb, int32_t Bp, int32_t dim, float max_period, msgm_stream_t stream) {
  if (!t || !emb || Bp <= 0 || dim <= 0) return MSGM_E_BADARG;
  hipLaunchKernelGGL(k_timestep_embedding_dual, dim3(grid_for((int64_t)Bp * dim, 256)), dim3(256), 0, S(stream), t, emb, Bp, dim, max_period);
  return msgm_check_launch();
}

int msgm_normalize_dual(const float* x, float* out, float* logr, int32_t Bp, int32_t n, int32_t dual, float scale, float eps,
                        msgm_stream_t stream) {
  if (!x || !out || !logr || Bp <= 0 || n <= 0) return MSGM_E_BADARG;
  hipLaunchKernelGGL(k_normalize_dual, dim3((Bp + 3) / 4), dim3(256), 0, S(stream), x, out, logr, Bp, n, dual, scale, eps);
  return msgm_check_launch();
}

int msgm_flat_to_image(const float* flat, float* img, int32_t B, int32_t C, int32_t H, int32_t W, int32_t forder, float scale,
                       msgm_stream_t stream) {
  if (!flat || !img || B <= 0 || C <= 0 || H <= 0 || W <= 0) return MSGM_E_BADARG;
  hipLaunchKernelGGL(k_flat_to_cl, dim3(grid_for((int64_t)B * C * H * W, 256)), dim3(256), 0, S(stream), flat, img, B, C, H, W, forder, scale);
  return msgm_check_launch();
}

int msgm_image_to_flat(const float* img, float* flat, int32_t B, int32_t C, int32_t H, int32_t W, int32_t forder, float scale,
                       msgm_stream_t stream) {
  if (!flat || !img || B <= 0 || C <= 0 || H <= 0 || W <= 0) return MSGM_E_BADARG;
  hipLaunchKernelGGL(k_cl_to_flat, dim3(grid_for((int64_t)B * C * H * W, 256)), dim3(256), 0, S(stream), img, flat, B, C, H, W, forder, scale);
  return msgm_check_launch();
}

int msgm_sum2x2(const float* in, float* out, int32_t N, int32_t H, int32_t W, int32_t C, msgm_stream_t stream) {
  if (!in || !out || N <= 0 || H <= 0 || W <= 0 || C <= 0) return MSGM_E_BADARG;
  hipLaunchKernelGGL(k_sum2x2, dim3(grid_for((int64_t)N * H * W * C, 256)), dim3(256), 0, S(stream), in, out, N, H, W, C);
  return msgm_check_launch();
}

int msgm_emb_bank_forward(const msgm_emb_job_t* jobs_dev, int32_t n_jobs, int32_t total_blocks, const float* semb, int32_t R,
                          int32_t K, int32_t n_bias, msgm_stream_t stream) {
  if (!jobs_dev || !semb || n_jobs <= 0 || total_blocks <= 0 || R <= 0 || K <= 0 || n_bias < 0) return MSGM_E_BADARG;
  if (K % 4 || K > 512) return MSGM_E_UNSUPPORTED;
  const size_t lds = (size_t)(EB_ROWS + 32) * (K + 4) * sizeof(float);
  static const int once = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_emb_bank<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_emb_bank<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return 0;
  }();
  (void)once;
  int ry = (R + EB_ROWS - 1) / EB_ROWS;
  if (ry > 8) ry = 8;
  hipLaunchKernelGGL(k_emb_bank<0>, dim3((unsigned)total_blocks, (unsigned)ry), dim3(256), lds, S(stream), jobs_dev, n_jobs, semb, R, K,
                     n_bias);
  return msgm_check_launch();
}

int msgm_emb_bank_backward(const msgm_emb_job_t* jobs_dev, int32_t n_jobs, int32_t total_blocks, const float* semb, float* dsemb,
                           int32_t R, int32_t K, int32_t n_bias, msgm_stream_t stream) {
  if (!jobs_dev || !semb || !dsemb || n_jobs <= 0 || total_blocks <= 0 || R <= 0 || K <= 0 || n_bias < 0) return MSGM_E_BADARG;
  if (K % 4 || K > 512) return MSGM_E_UNSUPPORTED;
  size_t lds = (size_t)EB_ROWS * (K + 4) * sizeof(float) + (size_t)EB_ROWS * 36 * sizeof(float);
  const size_t lds_f = (size_t)(EB_ROWS + 32) * (K + 4) * sizeof(float);
  if (lds < lds_f) lds = lds_f;                      // one carve formula for both modes (sW starts at EB_ROWS * KP)
  static const int once = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_emb_bank<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return 0;
  }();
  (void)once;
  hipLaunchKernelGGL(k_emb_bank<1>, dim3((unsigned)total_blocks), dim3(256), lds, S(stream), jobs_dev, n_jobs, semb, R, K, n_bias);
  hipLaunchKernelGGL(k_emb_bank_dgrad, dim3((unsigned)((R + 7) / 8), (unsigned)((K + 31) / 32)), dim3(256), 0, S(stream), jobs_dev,
                     n_jobs, dsemb, R, K);
  return msgm_check_launch();
}

}  // extern "C"
