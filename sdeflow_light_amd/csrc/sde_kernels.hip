// sde_kernels.hip — HBM-bound elementwise / row-reduced kernels of the SDE
// hot path (K1 perturb_vp, K2/K3/K4 integrator stages, K13 Adam, helpers).
// gfx950 only.  Reference lines are cited per kernel (relative to
// /root/reference).
#include "common.h"

// ============================================================ RNG helpers
__global__ void k_rng_advance(uint64_t* rng, uint64_t n) { rng[1] += n; }
__global__ void k_counter_inc(int64_t* c) { c[0] += 1; }

template <bool NORMAL>
__global__ void k_fill(float* __restrict__ out, int64_t n, const uint64_t* __restrict__ rng, uint32_t stream) {
  int64_t nq = (n + 3) >> 2;
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < nq; q += (int64_t)gridDim.x * blockDim.x) {
    f32x4 r = NORMAL ? philox_normal4(rng, 0, stream, q) : philox_uniform4(rng, 0, stream, q);
    int64_t e = q << 2;
    if (e + 3 < n && ((reinterpret_cast<uintptr_t>(out) & 15) == 0)) {
      *reinterpret_cast<f32x4*>(out + e) = r;
    } else {
      for (int k = 0; k < 4 && e + k < n; ++k) out[e + k] = r[k];
    }
  }
}

// Device-side clock of a graph-replayed sampler step: i = *step; t_dev = ts[i]; s_out[b] = T - t (the
// reverse-time argument of the score net, SDEs.py:556-557).  The step counter itself is bumped by
// msgm_counter_inc at the END of the step so every kernel of the step reads the same i.
__global__ void k_time_tick(const float* __restrict__ ts, const int64_t* __restrict__ step, int64_t n_ts, float T,
                            float* __restrict__ t_dev, float* __restrict__ s_out, int64_t B, float t_add) {
  int64_t i = step[0];
  if (i >= n_ts) i = n_ts - 1;
  // stage time of Heun / RK4: fp32 t + fp32 offset, as upstream forms t + delta/2 and t + delta on a float32 tensor
  // (sde_scheme.py:150,233-247); t_add = 0 for Euler-Maruyama and for the first stage
  const float t = t_add != 0.f ? __fadd_rn(ts[i], t_add) : ts[i];
  if (blockIdx.x == 0 && threadIdx.x == 0) t_dev[0] = t;
  for (int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; b < B; b += (int64_t)gridDim.x * blockDim.x) s_out[b] = T - t;
}

// ============================================================ K1 perturb_vp
// t_b = clamp(u_b T) (SDEs.py:688-693), y = eps*sqrt(var(t)) + mean_weight(t)*x0
// (SDEs.py:139-142).  One thread per element-quad of the flat (B*d) tensor.
__global__ void k_perturb_vp(const float* __restrict__ x0, float* __restrict__ y, float* __restrict__ t_out,
                             float* __restrict__ eps_out, int64_t B, int64_t d, float b0, float b1, float T,
                             float t_eps, const float* __restrict__ u, const float* __restrict__ eps,
                             const uint64_t* __restrict__ rng, const float* __restrict__ t_given) {
  const int64_t n = B * d;
  const int64_t nq = (n + 3) >> 2;
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < nq; q += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e0 = q << 2;
    f32x4 ez;
    if (eps == nullptr) ez = philox_normal4(rng, 0, RNG_STREAM_EPS, q);
    // the row's time (one Philox block + two exp) is evaluated once per row the quad touches, not once per element
    int64_t b_prev = -1;
    float t = 0.f, mw = 0.f, sd = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t e = e0 + k;
      if (e >= n) break;
      const int64_t b = e / d;
      if (b != b_prev) {
        if (t_given) {                                    // SGMsde.sample(t, y0): t used as given, no clamp (SDEs.py:134-146)
          t = t_given[b];
        } else {
          const float ub = u ? u[b] : philox_uniform1(rng, 0, RNG_STREAM_T, (uint64_t)b);
          t = ub * T;
          const float m = (t <= t_eps) ? 1.0f : 0.0f;    // mask arithmetic as upstream
          t = m * t_eps + (1.0f - m) * t;
        }
        mw = vp_mean_weight(b0, b1, t);
        sd = sqrtf(vp_var(b0, b1, t));
        b_prev = b;
      }
      float ee = eps ? eps[e] : ez[k];
      y[e] = ee * sd + mw * x0[e];
      if (eps_out) eps_out[e] = ee;
      if (t_out && e - b * d == 0) t_out[b] = t;
    }
  }
}

// K1 + probe in one launch for the training step: y,t as k_perturb_vp, v = Rademacher
// (SDEs.py:514-515,637-638), and (thread 0) the optimizer step counter tick.
__global__ void k_ssm_prep(const float* __restrict__ x0, float* __restrict__ y, float* __restrict__ t_out,
                           float* __restrict__ v, int64_t B, int64_t d, float b0, float b1, float T, float t_eps,
                           const uint64_t* __restrict__ rng, int64_t* step_ctr) {
  const int64_t n = B * d;
  const int64_t nq = (n + 3) >> 2;
  if (step_ctr && blockIdx.x == 0 && threadIdx.x == 0) step_ctr[0] += 1;
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < nq; q += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e0 = q << 2;
    const f32x4 ez = philox_normal4(rng, 0, RNG_STREAM_EPS, q);
    const f32x4 uv = philox_uniform4(rng, 0, RNG_STREAM_V, q);
    int64_t b_prev = -1;
    float t = 0.f, mw = 0.f, sd = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t e = e0 + k;
      if (e >= n) break;
      const int64_t b = e / d;
      if (b != b_prev) {                                 // once per row the quad touches
        t = philox_uniform1(rng, 0, RNG_STREAM_T, (uint64_t)b) * T;
        const float m = (t <= t_eps) ? 1.0f : 0.0f;
        t = m * t_eps + (1.0f - m) * t;
        mw = vp_mean_weight(b0, b1, t);
        sd = sqrtf(vp_var(b0, b1, t));
        b_prev = b;
      }
      y[e] = ez[k] * sd + mw * x0[e];
      v[e] = (uv[k] >= 0.5f) ? 1.0f : -1.0f;
      if (e - b * d == 0) t_out[b] = t;
    }
  }
}

// bit-exact stop index                                        SDEs.py:89-101
__global__ void k_forward_step_index(const float* __restrict__ t, int32_t* __restrict__ k, int64_t B, int32_t nsf, float T) {
  for (int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; b < B; b += (int64_t)gridDim.x * blockDim.x) {
    float tb = t[b];
    float f = __fdiv_rn(__fmul_rn((float)nsf, tb), T);   // (nsf*t)/T in fp32, no contraction
    int32_t kk = (int32_t)truncf(f);
    if (tb >= T) kk = nsf;
    k[b] = kk;
  }
}

__global__ void k_rademacher(float* __restrict__ v, int64_t n, const float* __restrict__ u, const uint64_t* __restrict__ rng) {
  const int64_t nq = (n + 3) >> 2;
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < nq; q += (int64_t)gridDim.x * blockDim.x) {
    f32x4 r;
    if (!u) r = philox_uniform4(rng, 0, RNG_STREAM_V, q);
    for (int k = 0; k < 4; ++k) {
      int64_t e = (q << 2) + k;
      if (e >= n) break;
      float uu = u ? u[e] : r[k];
      v[e] = (uu >= 0.5f) ? 1.0f : -1.0f;                 // 2*[u>=.5]-1   SDEs.py:515
    }
  }
}

// Sum over the GS lanes that share a row.  GS <= 64: a lane group inside a wave (xor tree).  GS = 256: the whole workgroup
// owns the row (long rows: n >= 2048 — 12288-element images have too few rows per launch to fill the chip with 64-lane
// groups: 32 rows were 8 workgroups and 67-82 us per launch); wave sums meet through LDS in a fixed order.  The group size
// depends on n only, never on the batch: a row's sum is the same bits in every batch size.
template <int GS>
__device__ __forceinline__ float group_sum(float v) {
  if constexpr (GS <= 64) {
#pragma unroll
    for (int o = GS >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, GS);
    return v;
  } else {
    __shared__ float red[4];
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const float r = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return r;
  }
}

// ============================================================ integrator stage
struct StageArgs {
  float* out; const float* base; float c_out;
  const float* x; const float* a; const float* dW; const float* z; float sqrt_delta;
  const uint64_t* rng; uint64_t rng_step; float* dW_out;
  int64_t B; int64_t n;
  int kind, proc, strato;
  float b0, b1, T, t, delta, lmbd;
  const float* G; const float* L_G;
  const float* norm0;
  float* inc_out;              // optional: the bare increment
  const float* delta_rows;     // optional per-row step length (MSGM short-time rows, SDEs.py:112-117)
  float t_frac;                // with delta_rows: t_b = t + t_frac * delta_b
  const float* t_dev;          // optional device scalar overriding t   (graph replay: time lives on the device)
  const int64_t* step_dev;     // optional device scalar overriding rng_step
};

__device__ __forceinline__ uint64_t stage_step(const StageArgs& A) { return A.step_dev ? (uint64_t)A.step_dev[0] : A.rng_step; }
__device__ __forceinline__ float stage_time(const StageArgs& A) { return A.t_dev ? A.t_dev[0] : A.t; }
__device__ __forceinline__ float stage_noise(const StageArgs& A, int64_t e, float sqrt_delta) {
  if (A.dW) return A.dW[e];
  float zz = A.z ? A.z[e] : philox_normal1(A.rng, stage_step(A), RNG_STREAM_DW, (uint64_t)e);
  return sqrt_delta * zz;                                  // delta**0.5 * randn   sde_scheme.py:84
}

// SGM (diagonal) flat kernel, no norm correction: 12-16 B/element.
// Reverse: mu = (1-l/2) sqrt(beta) a + 1/2 beta x (SDEs.py:560-561,183-194 with
// divSigma = 0); sigma = sqrt(1-l) sqrt(beta).  Forward: mu = -1/2 beta x.
__global__ void k_stage_diag_flat(StageArgs A) {
  const int64_t n = A.B * A.n;
  const int64_t nq = (n + 3) >> 2;
  const float tnow = stage_time(A);
  const float s = A.proc == MSGM_PROC_REVERSE ? A.T - tnow : tnow;
  const float beta = sde_beta(A.b0, A.b1, s);
  const float sb = sqrtf(beta);
  const float ca = (1.0f - 0.5f * A.lmbd);
  const float sig = (A.proc == MSGM_PROC_REVERSE ? sqrtf(1.0f - A.lmbd) : 1.0f) * sb;
  const bool vec = (n & 3) == 0 && ((reinterpret_cast<uintptr_t>(A.x) | reinterpret_cast<uintptr_t>(A.out) |
                                     reinterpret_cast<uintptr_t>(A.a) | reinterpret_cast<uintptr_t>(A.dW) |
                                     reinterpret_cast<uintptr_t>(A.z) | reinterpret_cast<uintptr_t>(A.base) |
                                     reinterpret_cast<uintptr_t>(A.dW_out) | reinterpret_cast<uintptr_t>(A.inc_out)) & 15) == 0;
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < nq; q += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e0 = q << 2;
    f32x4 xv, av = {0, 0, 0, 0}, wv, bv = {0, 0, 0, 0};
    if (vec) {
      xv = *reinterpret_cast<const f32x4*>(A.x + e0);
      if (A.proc == MSGM_PROC_REVERSE) av = *reinterpret_cast<const f32x4*>(A.a + e0);
      if (A.base) bv = (A.base == A.x) ? xv : *reinterpret_cast<const f32x4*>(A.base + e0);
      if (A.dW) wv = *reinterpret_cast<const f32x4*>(A.dW + e0);
      else {
        f32x4 zz = A.z ? *reinterpret_cast<const f32x4*>(A.z + e0) : philox_normal4(A.rng, stage_step(A), RNG_STREAM_DW, q);
        wv = A.sqrt_delta * zz;
      }
    } else {
      for (int k = 0; k < 4; ++k) {
        int64_t e = e0 + k;
        if (e < n) {
          xv[k] = A.x[e];
          if (A.proc == MSGM_PROC_REVERSE) av[k] = A.a[e];
          if (A.base) bv[k] = A.base[e];
          wv[k] = stage_noise(A, e, A.sqrt_delta);
        }
      }
    }
    f32x4 o, iv;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float mu;
      if (A.proc == MSGM_PROC_REVERSE) mu = ca * (sb * av[k]) - (-0.5f * beta * xv[k]);
      else mu = -0.5f * beta * xv[k];
      iv[k] = mu * A.delta + sig * wv[k];
      o[k] = bv[k] + A.c_out * iv[k];
    }
    if (vec) {
      *reinterpret_cast<f32x4*>(A.out + e0) = o;
      if (A.dW_out) *reinterpret_cast<f32x4*>(A.dW_out + e0) = wv;
      if (A.inc_out) *reinterpret_cast<f32x4*>(A.inc_out + e0) = iv;
    } else {
      for (int k = 0; k < 4; ++k) {
        int64_t e = e0 + k;
        if (e < n) { A.out[e] = o[k]; if (A.dW_out) A.dW_out[e] = wv[k]; if (A.inc_out) A.inc_out[e] = iv[k]; }
      }
    }
  }
}

// Multiplicative SDE, SPARSE rotation tensor, no norm correction (the RK4 stages of the forward perturbation and of the
// Stratonovich samplers): the update is a 3-point circular stencil along the row,
//   dx_i = c sqrt(beta) (x_{i+1} w_i - x_{i-1} w_{i-1})        (SDEs.py:369-399,425-430; sde_scheme.py:27-32),
// so it runs as a FLAT float4 kernel like the additive one — 12-16 B / element — instead of one 64-lane group per row
// (256 rows of 12288 elements were 309 us per stage = 0.12 TB/s).  n % 4 == 0: a quad never straddles two rows.
// Same expressions, in the same order, as k_stage_rows (bit-identical results).
__global__ void __launch_bounds__(256) k_stage_sparse_flat(StageArgs A) {
  const int64_t n = A.n, nq = (A.B * n) >> 2;
  const float l = A.lmbd;
  const float sig_scale = (A.proc == MSGM_PROC_REVERSE) ? sqrtf(1.0f - l) : 1.0f;
  const float cV = 0.5f * sqrtf(2.0f);
  const float tnow = stage_time(A);
  const int lane = threadIdx.x & 63;
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < nq; q += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e0 = q << 2, b = e0 / n, i0 = e0 - b * n;
    const float delta = A.delta_rows ? A.delta_rows[b] : A.delta;
    const float sqd = A.delta_rows ? sqrtf(delta) : A.sqrt_delta;
    const float tb = A.delta_rows ? tnow + A.t_frac * delta : tnow;
    const float s = A.proc == MSGM_PROC_REVERSE ? A.T - tb : tb;
    const float beta = sde_beta(A.b0, A.b1, s);
    const float sb = sqrtf(beta);
    const int64_t row = b * n;
    const int64_t em = row + (i0 == 0 ? n - 1 : i0 - 1), ep = row + (i0 + 4 == n ? 0 : i0 + 4);
    const f32x4 xv = *reinterpret_cast<const f32x4*>(A.x + e0);
    const float xm0 = A.x[em], xp3 = A.x[ep];
    f32x4 av = {0, 0, 0, 0}, bv = {0, 0, 0, 0}, wv;
    float am0 = 0.f, wm0;
    if (A.a) { av = *reinterpret_cast<const f32x4*>(A.a + e0); am0 = A.a[em]; }
    if (A.base) bv = (A.base == A.x) ? xv : *reinterpret_cast<const f32x4*>(A.base + e0);
    if (A.dW) { wv = *reinterpret_cast<const f32x4*>(A.dW + e0); wm0 = A.dW[em]; }
    else if (A.z) { wv = sqd * *reinterpret_cast<const f32x4*>(A.z + e0); wm0 = sqd * A.z[em]; }
    else {
      const f32x4 zz = philox_normal4(A.rng, stage_step(A), RNG_STREAM_DW, (uint64_t)q);
      wv = sqd * zz;
      // the previous element's draw sits in the neighbouring lane's quad (same row unless this quad opens the row)
      const float zprev = __shfl_up(zz[3], 1, 64);
      wm0 = sqd * ((lane == 0 || i0 == 0) ? philox_normal1(A.rng, stage_step(A), RNG_STREAM_DW, (uint64_t)em) : zprev);
    }
    f32x4 o, iv;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float xi = xv[k];
      const float xp = k < 3 ? xv[k + 1] : xp3, xm = k > 0 ? xv[k - 1] : xm0;
      const float w_i = wv[k], w_m = k > 0 ? wv[k - 1] : wm0;
      const float f = 0.5f * beta * xi, div = 2.0f * f;
      const float gw = (cV * (sb * xp)) * w_i + (-cV * (sb * xm)) * w_m;
      float ga = 0.f;
      if (A.a) ga = (cV * (sb * xp)) * av[k] + (-cV * (sb * xm)) * (k > 0 ? av[k - 1] : am0);
      float mu;
      if (A.proc == MSGM_PROC_REVERSE) {
        mu = (1.0f - 0.5f * l) * ga - f + (1.0f - l) * div;
        if (A.strato) mu = mu - 0.5f * (1.0f - l) * div;
      } else {
        mu = 0.f;
        if (!A.strato) mu = mu + 0.5f * div;
      }
      iv[k] = mu * delta + sig_scale * gw;
      o[k] = bv[k] + A.c_out * iv[k];
    }
    *reinterpret_cast<f32x4*>(A.out + e0) = o;
    if (A.dW_out) *reinterpret_cast<f32x4*>(A.dW_out + e0) = wv;
    if (A.inc_out) *reinterpret_cast<f32x4*>(A.inc_out + e0) = iv;
  }
}

// x + (k1 + 2k2 + 2k3 + k4)/6 without norm correction: flat float4 pass (24 B / element)
__global__ void __launch_bounds__(256) k_rk4_combine_flat(float* __restrict__ out, const float* __restrict__ x,
                                                          const float* __restrict__ k1, const float* __restrict__ k2,
                                                          const float* __restrict__ k3, const float* __restrict__ k4, int64_t nq) {
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < nq; q += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = q << 2;
    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + e), a = *reinterpret_cast<const f32x4*>(k1 + e);
    const f32x4 b = *reinterpret_cast<const f32x4*>(k2 + e), c = *reinterpret_cast<const f32x4*>(k3 + e);
    const f32x4 d = *reinterpret_cast<const f32x4*>(k4 + e);
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = xv[k] + (a[k] + 2.0f * b[k] + 2.0f * c[k] + d[k]) / 6.0f;
    *reinterpret_cast<f32x4*>(out + e) = o;
  }
}

// Row kernel: a group of GS lanes owns one row; handles all three layouts and
// the optional norm correction (x <- x * norm0/||x||, sde_scheme.py:85-86).
//  sparse: dx_i = c sb (x_{i+1} w_i - x_{i-1} w_{i-1}), c = sqrt(2)/2, circular
//          (SDEs.py:369-399,425-430; sde_scheme.py:27-32).
//  dense : dx_i = sum_{j,k} G[i,j,k] (sb x_j) w_k (SDEs.py:432; sde_scheme.py:36)
//          and f = L_G (beta x) (SDEs.py:415) — (B,n,n) is never materialised.
template <int GS>
__global__ void k_stage_rows(StageArgs A) {
  const int lane = threadIdx.x & (GS - 1);
  const int64_t groups_per_block = blockDim.x / GS;
  const int64_t gid = blockIdx.x * groups_per_block + threadIdx.x / GS;
  const int64_t gstride = (int64_t)gridDim.x * groups_per_block;
  const int64_t n = A.n;
  const float l = A.lmbd;
  const float sig_scale = (A.proc == MSGM_PROC_REVERSE) ? sqrtf(1.0f - l) : 1.0f;
  const float cV = 0.5f * sqrtf(2.0f);
  // rows are padded to a multiple of the group count so that shuffles stay convergent
  const int64_t rows_pad = ((A.B + gstride - 1) / gstride) * gstride;
  for (int64_t b = gid; b < rows_pad; b += gstride) {
    const bool live = b < A.B;
    const float* xr = A.x + b * n;
    const float* ar = A.a ? A.a + b * n : nullptr;
    float ss = 0.f;
    if (live) {
      const float delta = A.delta_rows ? A.delta_rows[b] : A.delta;
      const float sqd = A.delta_rows ? sqrtf(delta) : A.sqrt_delta;
      const float tnow = stage_time(A);
      const float tb = A.delta_rows ? tnow + A.t_frac * delta : tnow;
      const float s = A.proc == MSGM_PROC_REVERSE ? A.T - tb : tb;
      const float beta = sde_beta(A.b0, A.b1, s);
      const float sb = sqrtf(beta);
      for (int64_t i = lane; i < n; i += GS) {
        const int64_t e = b * n + i;
        float xi = xr[i];
        float w_i = stage_noise(A, e, sqd);
        if (A.dW_out) A.dW_out[e] = w_i;
        float ga = 0.f, gw = 0.f, f = 0.f, div = 0.f, fs = 0.f;
        if (A.kind == MSGM_SDE_SGM) {
          f = -0.5f * beta * xi; fs = f; div = 0.f;
          if (ar) ga = sb * ar[i];
          gw = sb * w_i;
        } else if (A.kind == MSGM_SDE_MSGM_SPARSE) {
          const int64_t ip = (i + 1 == n) ? 0 : i + 1, im = (i == 0) ? n - 1 : i - 1;
          const float xp = xr[ip], xm = xr[im];
          const float w_m = stage_noise(A, b * n + im, sqd);
          f = 0.5f * beta * xi; fs = 0.f; div = 2.0f * f;
          // entries (I=i,J=i+1,K=i,V=+c) and (I=i,J=i-1,K=i-1,V=-c)
          gw = (cV * (sb * xp)) * w_i + (-cV * (sb * xm)) * w_m;
          if (ar) ga = (cV * (sb * xp)) * ar[i] + (-cV * (sb * xm)) * ar[im];
        } else {  // dense
          float accw = 0.f, acca = 0.f, accf = 0.f;
          const float* Gi = A.G + i * n * n;
          for (int64_t j = 0; j < n; ++j) {
            const float yj = sb * xr[j];
            accf += A.L_G[i * n + j] * (beta * xr[j]);
            float rw = 0.f, ra = 0.f;
            for (int64_t k = 0; k < n; ++k) {
              const float g = Gi[j * n + k];
              rw += g * stage_noise(A, b * n + k, sqd);
              if (ar) ra += g * ar[k];
            }
            accw += yj * rw; acca += yj * ra;
          }
          f = accf; fs = 0.f; div = 2.0f * f; gw = accw; ga = acca;
        }
        float mu;
        if (A.proc == MSGM_PROC_REVERSE) {
          mu = (1.0f - 0.5f * l) * ga - f + (1.0f - l) * div;          // SDEs.py:560-561
          if (A.strato) mu = mu - 0.5f * (1.0f - l) * div;             // SDEs.py:583-584
        } else {
          mu = fs;                                                      // SDEs.py:42-43
          if (!A.strato) mu = mu + 0.5f * div;                          // SDEs.py:38-39
        }
        float inc = mu * delta + sig_scale * gw;                        // sde_scheme.py:40
        float o = (A.base ? A.base[e] : 0.f) + A.c_out * inc;
        A.out[e] = o;
        if (A.inc_out) A.inc_out[e] = inc;
        ss += o * o;
      }
    }
    if (A.norm0) {
ss = group_sum<GS>(ss);
      if (live) {
        const float scale = A.norm0[b] / sqrtf(ss);
        for (int64_t i = lane; i < n; i += GS) A.out[b * n + i] *= scale;   // own writes, same lane
      }
    }
  }
}

// x + (k1 + 2k2 + 2k3 + k4)/6 with optional norm correction   sde_scheme.py:250-253
template <int GS>
__global__ void k_rk4_combine(float* __restrict__ out, const float* __restrict__ x, const float* __restrict__ k1,
                              const float* __restrict__ k2, const float* __restrict__ k3, const float* __restrict__ k4,
                              int64_t B, int64_t n, const float* __restrict__ norm0) {
  const int lane = threadIdx.x & (GS - 1);
  const int64_t gpb = blockDim.x / GS;
  const int64_t gid = blockIdx.x * gpb + threadIdx.x / GS;
  const int64_t gstride = (int64_t)gridDim.x * gpb;
  const int64_t rows_pad = ((B + gstride - 1) / gstride) * gstride;
  for (int64_t b = gid; b < rows_pad; b += gstride) {
    const bool live = b < B;
    float ss = 0.f;
    if (live)
      for (int64_t i = lane; i < n; i += GS) {
        const int64_t e = b * n + i;
        float o = x[e] + (k1[e] + 2.0f * k2[e] + 2.0f * k3[e] + k4[e]) / 6.0f;
        out[e] = o; ss += o * o;
      }
    if (norm0) {
ss = group_sum<GS>(ss);
      if (live) {
        const float sc = norm0[b] / sqrtf(ss);
        for (int64_t i = lane; i < n; i += GS) out[b * n + i] *= sc;
      }
    }
  }
}

// K12: per-sample SSM loss and the cotangents of (a, adot) for the SGM (diagonal) case.
// out is the score net's (primal | tangent) output [2B][n]: a = out[:B], adot = J_a v = out[B:].
//   loss_b = sum_i v_i (sqrt(beta) adot_i + 1/2 beta v_i) + 1/2 a_i^2        SDEs.py:631-646
//   g[:B] = a * w, g[B:] = sqrt(beta) v * w   (w = 1/global batch: gradient of the mean)
template <int GS>
__global__ void k_ssm_loss_diag(const float* __restrict__ out, const float* __restrict__ v, const float* __restrict__ t,
                                float* __restrict__ per, float* __restrict__ g, int64_t B, int64_t n, float b0, float b1,
                                float w) {
  const int lane = threadIdx.x & (GS - 1);
  const int64_t gpb = blockDim.x / GS;
  const int64_t gid = blockIdx.x * gpb + threadIdx.x / GS;
  const int64_t gstride = (int64_t)gridDim.x * gpb;
  const int64_t rows_pad = ((B + gstride - 1) / gstride) * gstride;
  for (int64_t b = gid; b < rows_pad; b += gstride) {
    float acc = 0.f;
    if (b < B) {
      const float beta = sde_beta(b0, b1, t[b]);
      const float sb = sqrtf(beta);
      for (int64_t i = lane; i < n; i += GS) {
        const float a = out[b * n + i], ad = out[(B + b) * n + i], vi = v[b * n + i];
        acc += vi * (sb * ad + 0.5f * beta * vi) + 0.5f * a * a;
        g[b * n + i] = a * w;
        g[(B + b) * n + i] = sb * vi * w;
      }
    }
acc = group_sum<GS>(acc);
    if (b < B && lane == 0) per[b] = acc;
  }
}

// u = (d mu_to_div / d a)^T v and the a-independent constant of the SSM loss, for the three SDE families:
//   mu_to_div = G(t,y) a - f + 1/2 divSigma (SDEs.py:560-561,631-632); along v its derivative is
//   G(y) adot + [G(v) a] + d(-f + 1/2 divSigma).v, so loss_b = adot.u + cst + 1/2|a|^2 with
//   SGM  : u = sqrt(beta) v,                    cst = 1/2 beta |v|^2
//   MSGM : u_k = sqrt(beta) sum_ij G_ijk y_j v_i, cst = 0   (-f + 1/2 divSigma == 0; v^T G(v) a == 0 since
//          every G[:,:,k] is skew-symmetric — upstream carries that term as rounding noise only)
//   sparse: u_k = c sqrt(beta) (v_k y_{k+1} - v_{k+1} y_k)   (entries of SDEs.py:369-383)
template <int GS>
__global__ void k_ssm_terms(const float* __restrict__ y, const float* __restrict__ v, const float* __restrict__ t,
                            float* __restrict__ u, float* __restrict__ cst, int64_t B, int64_t n, int kind, float b0,
                            float b1, const float* __restrict__ G) {
  const int lane = threadIdx.x & (GS - 1);
  const int64_t gpb = blockDim.x / GS;
  const int64_t gid = blockIdx.x * gpb + threadIdx.x / GS;
  const int64_t gstride = (int64_t)gridDim.x * gpb;
  const int64_t rows_pad = ((B + gstride - 1) / gstride) * gstride;
  const float cV = 0.5f * sqrtf(2.0f);
  for (int64_t b = gid; b < rows_pad; b += gstride) {
    float acc = 0.f;
    if (b < B) {
      const float beta = sde_beta(b0, b1, t[b]);
      const float sb = sqrtf(beta);
      const float* yr = y + b * n;
      const float* vr = v + b * n;
      for (int64_t k = lane; k < n; k += GS) {
        float uk;
        if (kind == MSGM_SDE_SGM) {
          uk = sb * vr[k];
          acc += 0.5f * beta * vr[k] * vr[k];
        } else if (kind == MSGM_SDE_MSGM_SPARSE) {
          const int64_t kp = (k + 1 == n) ? 0 : k + 1;
          uk = (cV * sb) * (vr[k] * yr[kp] - vr[kp] * yr[k]);
        } else {
          float s = 0.f;
          for (int64_t i = 0; i < n; ++i) {
            float r = 0.f;
            for (int64_t j = 0; j < n; ++j) r += G[(i * n + j) * n + k] * yr[j];
            s += r * vr[i];
          }
          uk = sb * s;
        }
        u[b * n + k] = uk;
      }
    }
acc = group_sum<GS>(acc);
    if (b < B && lane == 0) cst[b] = acc;
  }
}

// generic K12: per[b] = adot.u + cst + 1/2|a|^2 ; g[:B] = a w ; g[B:] = u w      (out = [a ; adot] stacked)
template <int GS>
__global__ void k_ssm_loss_generic(const float* __restrict__ out, const float* __restrict__ u, const float* __restrict__ cst,
                                   float* __restrict__ per, float* __restrict__ g, int64_t B, int64_t n, float w) {
  const int lane = threadIdx.x & (GS - 1);
  const int64_t gpb = blockDim.x / GS;
  const int64_t gid = blockIdx.x * gpb + threadIdx.x / GS;
  const int64_t gstride = (int64_t)gridDim.x * gpb;
  const int64_t rows_pad = ((B + gstride - 1) / gstride) * gstride;
  for (int64_t b = gid; b < rows_pad; b += gstride) {
    float acc = 0.f;
    if (b < B)
      for (int64_t i = lane; i < n; i += GS) {
        const float a = out[b * n + i], ad = out[(B + b) * n + i], ui = u[b * n + i];
        acc += ad * ui + 0.5f * a * a;
        g[b * n + i] = a * w;
        g[(B + b) * n + i] = ui * w;
      }
acc = group_sum<GS>(acc);
    if (b < B && lane == 0) per[b] = acc + cst[b];
  }
}

// out = c0*a + c1*b + c2*c (b, c optional) — glue for Heun / RK4 stage points
__global__ void k_lincomb(float* __restrict__ out, const float* __restrict__ a, float c0, const float* __restrict__ b,
                          float c1, const float* __restrict__ c, float c2, int64_t n) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    float v = c0 * a[e];
    if (b) v += c1 * b[e];
    if (c) v += c2 * c[e];
    out[e] = v;
  }
}

template <int GS>
__global__ void k_row_norm(const float* __restrict__ x, float* __restrict__ out, int64_t B, int64_t n) {
  const int lane = threadIdx.x & (GS - 1);
  const int64_t gpb = blockDim.x / GS;
  const int64_t gid = blockIdx.x * gpb + threadIdx.x / GS;
  const int64_t gstride = (int64_t)gridDim.x * gpb;
  const int64_t rows_pad = ((B + gstride - 1) / gstride) * gstride;
  for (int64_t b = gid; b < rows_pad; b += gstride) {
    float ss = 0.f;
    if (b < B) for (int64_t i = lane; i < n; i += GS) { float v = x[b * n + i]; ss += v * v; }
ss = group_sum<GS>(ss);
    if (b < B && lane == 0) out[b] = sqrtf(ss);
  }
}

__global__ void k_keep_rows(float* __restrict__ kept, const float* __restrict__ x, const int32_t* __restrict__ stop,
                            int32_t index, int64_t B, int64_t n) {
  const int64_t tot = B * n;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < tot; e += (int64_t)gridDim.x * blockDim.x) {
    if (stop[e / n] == index) kept[e] = x[e];
  }
}

// ============================================================ K13 Adam
// torch.optim.Adam (defaults, no weight decay / amsgrad): MSGM_higherDim.py:792.
// Mirrors torch's single-tensor update: m.lerp_(g, 1-b1); v = v*b2 + (1-b2) g g;
// denom = sqrt(v)/sqrt(bc2) + eps; p += -(lr/bc1) * (m/denom), with the bias
// corrections formed in double precision as the Python scalars upstream are.
__global__ void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                       int64_t n, double lr, double b1, double b2, double eps, float gscale, int64_t step,
                       const int64_t* __restrict__ step_dev) {
  const int64_t st = step_dev ? step_dev[0] : step;
  const double bc1 = 1.0 - pow(b1, (double)st);
  const double bc2 = 1.0 - pow(b2, (double)st);
  const float step_size = (float)(lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  const float w1 = (float)(1.0 - b1), b2f = (float)b2, w2 = (float)(1.0 - b2), epsf = (float)eps;
  const int64_t nq = (n + 3) >> 2;
  const bool vec = (n & 3) == 0 && ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) |
                                     reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0;
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < nq; q += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e0 = q << 2;
    if (vec) {
      f32x4 pv = *reinterpret_cast<f32x4*>(p + e0), gv = *reinterpret_cast<const f32x4*>(g + e0);
      f32x4 mv = *reinterpret_cast<f32x4*>(m + e0), vv = *reinterpret_cast<f32x4*>(v + e0);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float gg = gv[k] * gscale;
        mv[k] = mv[k] + w1 * (gg - mv[k]);
        vv[k] = vv[k] * b2f + w2 * (gg * gg);
        float denom = sqrtf(vv[k]) / bc2_sqrt + epsf;
        pv[k] = pv[k] - step_size * (mv[k] / denom);
      }
      *reinterpret_cast<f32x4*>(p + e0) = pv; *reinterpret_cast<f32x4*>(m + e0) = mv; *reinterpret_cast<f32x4*>(v + e0) = vv;
    } else {
      for (int k = 0; k < 4; ++k) {
        int64_t e = e0 + k;
        if (e >= n) break;
        float gg = g[e] * gscale;
        float mm = m[e] + w1 * (gg - m[e]);
        float vv = v[e] * b2f + w2 * (gg * gg);
        float denom = sqrtf(vv) / bc2_sqrt + epsf;
        p[e] = p[e] - step_size * (mm / denom); m[e] = mm; v[e] = vv;
      }
    }
  }
}

// ============================================================ C ABI
static inline hipStream_t S(msgm_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

template <typename F>
static int launch_rows(int64_t B, int64_t n, F&& f) {
  // lanes per row: next power of two >= min(n,64), at least 2
  int gs = 2;
  while (gs < 64 && gs < n) gs <<= 1;
  if (n >= 2048) gs = 256;             // long rows: the whole workgroup owns a row (group_sum<256>); a function of n only
  const int block = 256;
  int64_t groups = (int64_t)block / gs;
  int grid = (int)((B + groups - 1) / groups);
  if (grid > 4096) grid = 4096;
  if (grid < 1) grid = 1;
  f(gs, grid, block);
  return msgm_check_launch();
}

extern "C" {

int msgm_version(void) { return 100; }

const char* msgm_error_string(int code) {
  switch (code) {
    case MSGM_OK: return "ok";
    case MSGM_E_BADARG: return "bad argument (null pointer, non-positive size or forbidden aliasing)";
    case MSGM_E_UNSUPPORTED: return "unsupported shape/configuration for this kernel";
    case MSGM_E_WORKSPACE: return "workspace too small";
    case MSGM_E_LAUNCH: return "HIP launch error";
    default: return "unknown error";
  }
}

int msgm_rng_advance(uint64_t* rng, uint64_t n, msgm_stream_t stream) {
  if (!rng) return MSGM_E_BADARG;
  hipLaunchKernelGGL(k_rng_advance, dim3(1), dim3(1), 0, S(stream), rng, n);
  return msgm_check_launch();
}

int msgm_time_tick(const float* ts, const int64_t* step, int64_t n_ts, float T, float* t_dev, float* s_out, int64_t B,
                   msgm_stream_t stream) {
  if (!ts || !step || !t_dev || !s_out || n_ts <= 0 || B <= 0) return MSGM_E_BADARG;
  hipLaunchKernelGGL(k_time_tick, dim3(grid_for(B, 256, 256)), dim3(256), 0, S(stream), ts, step, n_ts, T, t_dev, s_out, B, 0.f);
  return msgm_check_launch();
}

int msgm_time_tick_stage(const float* ts, const int64_t* step, int64_t n_ts, float T, float t_add, float* t_dev, float* s_out,
                         int64_t B, msgm_stream_t stream) {
  if (!ts || !step || !t_dev || !s_out || n_ts <= 0 || B <= 0) return MSGM_E_BADARG;
  hipLaunchKernelGGL(k_time_tick, dim3(grid_for(B, 256, 256)), dim3(256), 0, S(stream), ts, step, n_ts, T, t_dev, s_out, B, t_add);
  return msgm_check_launch();
}

int msgm_counter_inc(int64_t* ctr, msgm_stream_t stream) {
  if (!ctr) return MSGM_E_BADARG;
  hipLaunchKernelGGL(k_counter_inc, dim3(1), dim3(1), 0, S(stream), ctr);
  return msgm_check_launch();
}

int msgm_fill_uniform(float* out, int64_t n, const uint64_t* rng, uint32_t stream_id, msgm_stream_t stream) {
  if (!out || !rng || n <= 0) return MSGM_E_BADARG;
  hipLaunchKernelGGL(k_fill<false>, dim3(grid_for((n + 3) / 4, 256)), dim3(256), 0, S(stream), out, n, rng, stream_id);
  return msgm_check_launch();
}

int msgm_fill_normal(float* out, int64_t n, const uint64_t* rng, uint32_t stream_id, msgm_stream_t stream) {
  if (!out || !rng || n <= 0) return MSGM_E_BADARG;
  hipLaunchKernelGGL(k_fill<true>, dim3(grid_for((n + 3) / 4, 256)), dim3(256), 0, S(stream), out, n, rng, stream_id);
  return msgm_check_launch();
}

int msgm_perturb_vp(const float* x0, float* y, float* t_out, float* eps_out, int64_t B, int64_t d,
                    const msgm_sde_t* sde, const float* u, const float* eps, const uint64_t* rng,
                    msgm_stream_t stream) {
  if (!x0 || !y || !t_out || !sde || B <= 0 || d <= 0) return MSGM_E_BADARG;
  if ((!u || !eps) && !rng) return MSGM_E_BADARG;
  if (sde->kind != MSGM_SDE_SGM) return MSGM_E_UNSUPPORTED;   // MSGM has no closed form (SDEs.py:434-436)
  hipLaunchKernelGGL(k_perturb_vp, dim3(grid_for((B * d + 3) / 4, 256)), dim3(256), 0, S(stream), x0, y, t_out, eps_out,
                     B, d, sde->beta_min, sde->beta_max, sde->T, sde->t_epsilon, u, eps, rng, (const float*)nullptr);
  return msgm_check_launch();
}

int msgm_perturb_vp_at(const float* x0, float* y, float* eps_out, int64_t B, int64_t d, const msgm_sde_t* sde, const float* t,
                       const float* eps, const uint64_t* rng, msgm_stream_t stream) {
  if (!x0 || !y || !t || !sde || B <= 0 || d <= 0 || (!eps && !rng)) return MSGM_E_BADARG;
  if (sde->kind != MSGM_SDE_SGM) return MSGM_E_UNSUPPORTED;
  hipLaunchKernelGGL(k_perturb_vp, dim3(grid_for((B * d + 3) / 4, 256)), dim3(256), 0, S(stream), x0, y, (float*)nullptr,
                     eps_out, B, d, sde->beta_min, sde->beta_max, sde->T, sde->t_epsilon, (const float*)nullptr, eps, rng, t);
  return msgm_check_launch();
}

int msgm_ssm_prep(const float* x0, float* y, float* t_out, float* v, int64_t B, int64_t d, const msgm_sde_t* sde,
                  const uint64_t* rng, int64_t* step_ctr, msgm_stream_t stream) {
  if (!x0 || !y || !t_out || !v || !sde || !rng || B <= 0 || d <= 0) return MSGM_E_BADARG;
  if (sde->kind != MSGM_SDE_SGM) return MSGM_E_UNSUPPORTED;
  hipLaunchKernelGGL(k_ssm_prep, dim3(grid_for((B * d + 3) / 4, 256)), dim3(256), 0, S(stream), x0, y, t_out, v, B, d,
                     sde->beta_min, sde->beta_max, sde->T, sde->t_epsilon, rng, step_ctr);
  return msgm_check_launch();
}

int msgm_forward_step_index(const float* t, int32_t* k, int64_t B, int32_t nsf, float T, msgm_stream_t stream) {
  if (!t || !k || B <= 0 || nsf <= 0) return MSGM_E_BADARG;
  hipLaunchKernelGGL(k_forward_step_index, dim3(grid_for(B, 256)), dim3(256), 0, S(stream), t, k, B, nsf, T);
  return msgm_check_launch();
}

int msgm_rademacher(float* v, int64_t n, const float* u, const uint64_t* rng, msgm_stream_t stream) {
  if (!v || n <= 0 || (!u && !rng)) return MSGM_E_BADARG;
  hipLaunchKernelGGL(k_rademacher, dim3(grid_for((n + 3) / 4, 256)), dim3(256), 0, S(stream), v, n, u, rng);
  return msgm_check_launch();
}

int msgm_sde_stage(float* out, const float* base, float c_out, const float* x, const float* a, const float* dW,
                   const float* z, float sqrt_delta, const uint64_t* rng, uint64_t rng_step, float* dW_out,
                   float* inc_out, int64_t B, int64_t n, const msgm_sde_t* sde, int32_t proc, int32_t strato, float t,
                   float delta, float lmbd, const float* norm0, const float* delta_rows, float t_frac,
                   const float* t_dev, const int64_t* step_dev, msgm_stream_t stream) {
  if (!out || !x || !sde || B <= 0 || n <= 0) return MSGM_E_BADARG;
  if (!dW && !z && !rng) return MSGM_E_BADARG;
  if (proc == MSGM_PROC_REVERSE && !a) return MSGM_E_BADARG;
  if (proc != MSGM_PROC_REVERSE && proc != MSGM_PROC_FORWARD) return MSGM_E_BADARG;
  if (sde->kind != MSGM_SDE_SGM && (out == x)) return MSGM_E_BADARG;      // stencil / contraction reads neighbours
  if (sde->kind == MSGM_SDE_MSGM_DENSE && (!sde->G || !sde->L_G)) return MSGM_E_BADARG;
  if (sde->kind == MSGM_SDE_MSGM_DENSE && n > 64) return MSGM_E_UNSUPPORTED;
  if (sde->kind < 0 || sde->kind > 2) return MSGM_E_BADARG;
  StageArgs A{out, base, c_out, x, a, dW, z, sqrt_delta, rng, rng_step, dW_out, B, n, sde->kind, proc, strato,
              sde->beta_min, sde->beta_max, sde->T, t, delta, lmbd, sde->G, sde->L_G, norm0, inc_out, delta_rows, t_frac, t_dev, step_dev};
  if (sde->kind == MSGM_SDE_SGM && !norm0 && !delta_rows) {
    hipLaunchKernelGGL(k_stage_diag_flat, dim3(grid_for((B * n + 3) / 4, 256)), dim3(256), 0, S(stream), A);
    return msgm_check_launch();
  }
  {
    const uintptr_t al = reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(a) |
                         reinterpret_cast<uintptr_t>(dW) | reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(base) |
                         reinterpret_cast<uintptr_t>(dW_out) | reinterpret_cast<uintptr_t>(inc_out);
    if (sde->kind == MSGM_SDE_MSGM_SPARSE && !norm0 && n % 4 == 0 && n >= 8 && (al & 15) == 0 && out != x) {
      hipLaunchKernelGGL(k_stage_sparse_flat, dim3(grid_for((B * n) / 4, 256)), dim3(256), 0, S(stream), A);
      return msgm_check_launch();
    }
  }
  return launch_rows(B, n, [&](int gs, int grid, int block) {
    switch (gs) {
      case 2: hipLaunchKernelGGL(k_stage_rows<2>, dim3(grid), dim3(block), 0, S(stream), A); break;
      case 4: hipLaunchKernelGGL(k_stage_rows<4>, dim3(grid), dim3(block), 0, S(stream), A); break;
      case 8: hipLaunchKernelGGL(k_stage_rows<8>, dim3(grid), dim3(block), 0, S(stream), A); break;
      case 16: hipLaunchKernelGGL(k_stage_rows<16>, dim3(grid), dim3(block), 0, S(stream), A); break;
      case 32: hipLaunchKernelGGL(k_stage_rows<32>, dim3(grid), dim3(block), 0, S(stream), A); break;
      case 256: hipLaunchKernelGGL(k_stage_rows<256>, dim3(grid), dim3(block), 0, S(stream), A); break;
      default: hipLaunchKernelGGL(k_stage_rows<64>, dim3(grid), dim3(block), 0, S(stream), A); break;
    }
  });
}

int msgm_rk4_combine(float* out, const float* x, const float* k1, const float* k2, const float* k3, const float* k4,
                     int64_t B, int64_t n, const float* norm0, msgm_stream_t stream) {
  if (!out || !x || !k1 || !k2 || !k3 || !k4 || B <= 0 || n <= 0) return MSGM_E_BADARG;
  {
    const uintptr_t al = reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(k1) |
                         reinterpret_cast<uintptr_t>(k2) | reinterpret_cast<uintptr_t>(k3) | reinterpret_cast<uintptr_t>(k4);
    if (!norm0 && (B * n) % 4 == 0 && (al & 15) == 0) {
      hipLaunchKernelGGL(k_rk4_combine_flat, dim3(grid_for((B * n) / 4, 256)), dim3(256), 0, S(stream), out, x, k1, k2, k3, k4, (B * n) / 4);
      return msgm_check_launch();
    }
  }
  return launch_rows(B, n, [&](int gs, int grid, int block) {
    switch (gs) {
      case 2: hipLaunchKernelGGL(k_rk4_combine<2>, dim3(grid), dim3(block), 0, S(stream), out, x, k1, k2, k3, k4, B, n, norm0); break;
      case 4: hipLaunchKernelGGL(k_rk4_combine<4>, dim3(grid), dim3(block), 0, S(stream), out, x, k1, k2, k3, k4, B, n, norm0); break;
      case 8: hipLaunchKernelGGL(k_rk4_combine<8>, dim3(grid), dim3(block), 0, S(stream), out, x, k1, k2, k3, k4, B, n, norm0); break;
      case 16: hipLaunchKernelGGL(k_rk4_combine<16>, dim3(grid), dim3(block), 0, S(stream), out, x, k1, k2, k3, k4, B, n, norm0); break;
      case 32: hipLaunchKernelGGL(k_rk4_combine<32>, dim3(grid), dim3(block), 0, S(stream), out, x, k1, k2, k3, k4, B, n, norm0); break;
      case 256: hipLaunchKernelGGL(k_rk4_combine<256>, dim3(grid), dim3(block), 0, S(stream), out, x, k1, k2, k3, k4, B, n, norm0); break;
      default: hipLaunchKernelGGL(k_rk4_combine<64>, dim3(grid), dim3(block), 0, S(stream), out, x, k1, k2, k3, k4, B, n, norm0); break;
    }
  });
}

int msgm_ssm_loss_diag(const float* out, const float* v, const float* t, float* per, float* g, int64_t B, int64_t n,
                       const msgm_sde_t* sde, float inv_batch, msgm_stream_t stream) {
  if (!out || !v || !t || !per || !g || !sde || B <= 0 || n <= 0) return MSGM_E_BADARG;
  if (sde->kind != MSGM_SDE_SGM) return MSGM_E_UNSUPPORTED;
  return launch_rows(B, n, [&](int gs, int grid, int block) {
    switch (gs) {
      case 2: hipLaunchKernelGGL(k_ssm_loss_diag<2>, dim3(grid), dim3(block), 0, S(stream), out, v, t, per, g, B, n, sde->beta_min, sde->beta_max, inv_batch); break;
      case 4: hipLaunchKernelGGL(k_ssm_loss_diag<4>, dim3(grid), dim3(block), 0, S(stream), out, v, t, per, g, B, n, sde->beta_min, sde->beta_max, inv_batch); break;
      case 8: hipLaunchKernelGGL(k_ssm_loss_diag<8>, dim3(grid), dim3(block), 0, S(stream), out, v, t, per, g, B, n, sde->beta_min, sde->beta_max, inv_batch); break;
      case 16: hipLaunchKernelGGL(k_ssm_loss_diag<16>, dim3(grid), dim3(block), 0, S(stream), out, v, t, per, g, B, n, sde->beta_min, sde->beta_max, inv_batch); break;
      case 32: hipLaunchKernelGGL(k_ssm_loss_diag<32>, dim3(grid), dim3(block), 0, S(stream), out, v, t, per, g, B, n, sde->beta_min, sde->beta_max, inv_batch); break;
      case 256: hipLaunchKernelGGL(k_ssm_loss_diag<256>, dim3(grid), dim3(block), 0, S(stream), out, v, t, per, g, B, n, sde->beta_min, sde->beta_max, inv_batch); break;
      default: hipLaunchKernelGGL(k_ssm_loss_diag<64>, dim3(grid), dim3(block), 0, S(stream), out, v, t, per, g, B, n, sde->beta_min, sde->beta_max, inv_batch); break;
    }
  });
}

#define ROWS_DISPATCH(KERNEL, ...)                                                                              \
  switch (gs) {                                                                                                  \
    case 2: hipLaunchKernelGGL(KERNEL<2>, dim3(grid), dim3(block), 0, S(stream), __VA_ARGS__); break;           \
    case 4: hipLaunchKernelGGL(KERNEL<4>, dim3(grid), dim3(block), 0, S(stream), __VA_ARGS__); break;           \
    case 8: hipLaunchKernelGGL(KERNEL<8>, dim3(grid), dim3(block), 0, S(stream), __VA_ARGS__); break;           \
    case 16: hipLaunchKernelGGL(KERNEL<16>, dim3(grid), dim3(block), 0, S(stream), __VA_ARGS__); break;         \
    case 32: hipLaunchKernelGGL(KERNEL<32>, dim3(grid), dim3(block), 0, S(stream), __VA_ARGS__); break;         \
    case 256: hipLaunchKernelGGL(KERNEL<256>, dim3(grid), dim3(block), 0, S(stream), __VA_ARGS__); break;       \
    default: hipLaunchKernelGGL(KERNEL<64>, dim3(grid), dim3(block), 0, S(stream), __VA_ARGS__); break;         \
  }

int msgm_ssm_terms(const float* y, const float* v, const float* t, float* u, float* cst, int64_t B, int64_t n,
                   const msgm_sde_t* sde, msgm_stream_t stream) {
  if (!y || !v || !t || !u || !cst || !sde || B <= 0 || n <= 0) return MSGM_E_BADARG;
  if (sde->kind < 0 || sde->kind > 2) return MSGM_E_BADARG;
  if (sde->kind == MSGM_SDE_MSGM_DENSE && (!sde->G || n > 64)) return sde->G ? MSGM_E_UNSUPPORTED : MSGM_E_BADARG;
  return launch_rows(B, n, [&](int gs, int grid, int block) {
    ROWS_DISPATCH(k_ssm_terms, y, v, t, u, cst, B, n, sde->kind, sde->beta_min, sde->beta_max, sde->G)
  });
}

int msgm_ssm_loss(const float* out, const float* u, const float* cst, float* per, float* g, int64_t B, int64_t n,
                  float inv_batch, msgm_stream_t stream) {
  if (!out || !u || !cst || !per || !g || B <= 0 || n <= 0) return MSGM_E_BADARG;
  return launch_rows(B, n, [&](int gs, int grid, int block) {
    ROWS_DISPATCH(k_ssm_loss_generic, out, u, cst, per, g, B, n, inv_batch)
  });
}

int msgm_lincomb(float* out, const float* a, float c0, const float* b, float c1, const float* c, float c2, int64_t n,
                 msgm_stream_t stream) {
  if (!out || !a || n <= 0) return MSGM_E_BADARG;
  hipLaunchKernelGGL(k_lincomb, dim3(grid_for(n, 256)), dim3(256), 0, S(stream), out, a, c0, b, c1, c, c2, n);
  return msgm_check_launch();
}

int msgm_row_norm(const float* x, float* out, int64_t B, int64_t n, msgm_stream_t stream) {
  if (!x || !out || B <= 0 || n <= 0) return MSGM_E_BADARG;
  return launch_rows(B, n, [&](int gs, int grid, int block) {
    switch (gs) {
      case 2: hipLaunchKernelGGL(k_row_norm<2>, dim3(grid), dim3(block), 0, S(stream), x, out, B, n); break;
      case 4: hipLaunchKernelGGL(k_row_norm<4>, dim3(grid), dim3(block), 0, S(stream), x, out, B, n); break;
      case 8: hipLaunchKernelGGL(k_row_norm<8>, dim3(grid), dim3(block), 0, S(stream), x, out, B, n); break;
      case 16: hipLaunchKernelGGL(k_row_norm<16>, dim3(grid), dim3(block), 0, S(stream), x, out, B, n); break;
      case 32: hipLaunchKernelGGL(k_row_norm<32>, dim3(grid), dim3(block), 0, S(stream), x, out, B, n); break;
      case 256: hipLaunchKernelGGL(k_row_norm<256>, dim3(grid), dim3(block), 0, S(stream), x, out, B, n); break;
      default: hipLaunchKernelGGL(k_row_norm<64>, dim3(grid), dim3(block), 0, S(stream), x, out, B, n); break;
    }
  });
}

int msgm_keep_rows(float* kept, const float* x, const int32_t* stop, int32_t index, int64_t B, int64_t n,
                   msgm_stream_t stream) {
  if (!kept || !x || !stop || B <= 0 || n <= 0) return MSGM_E_BADARG;
  hipLaunchKernelGGL(k_keep_rows, dim3(grid_for(B * n, 256)), dim3(256), 0, S(stream), kept, x, stop, index, B, n);
  return msgm_check_launch();
}

int msgm_adam_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                   double eps, float gscale, int64_t step, const int64_t* step_dev, msgm_stream_t stream) {
  if (!p || !g || !m || !v || n <= 0) return MSGM_E_BADARG;
  if (!step_dev && step < 1) return MSGM_E_BADARG;
  hipLaunchKernelGGL(k_adam, dim3(grid_for((n + 3) / 4, 256)), dim3(256), 0, S(stream), p, g, m, v, n, lr, beta1, beta2,
                     eps, gscale, step, step_dev);
  return msgm_check_launch();
}

}  // extern "C"
