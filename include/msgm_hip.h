/*
 * msgm_hip.h — C ABI of libmsgm_hip.so: the MI355X (gfx950) hot path of
 * sdeflow-light / MSGM (score-matching train step + reverse-SDE sampling).
 *
 * The reference (vressegu/sdeflow-light) is pure Python/PyTorch and has no
 * FFI; each entry point below replaces the *sequence of eager ATen ops* the
 * cited reference lines launch.  The reference-side binding a maintainer
 * would add is a ctypes stub — see INTEGRATION.md.
 *
 * Conventions (all entry points):
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless
 *     the name ends in _host; all float tensors are contiguous fp32;
 *   - returns MSGM_OK (0) or a negative MSGM_E* code; never throws, never
 *     allocates, never synchronises; work is enqueued on `stream` only, so
 *     every call is hipGraph-capturable;
 *   - caller owns all buffers, including workspaces (sizes from *_workspace);
 *   - re-entrant / thread-safe: no mutable globals.
 *
 * Randomness: kernels that draw noise take either an explicit noise buffer
 * (parity mode: the oracle is fed the same numbers) or, when that pointer is
 * NULL, a device-resident Philox state `rng` = {seed, offset, row_base,
 * elem_base} (uint64[4]).  Draws are counter-based: value = philox(seed,
 * offset + stream_id, GLOBAL element), so they do not depend on the launch
 * geometry; row_base / elem_base (= row_base * n, a multiple of 4) are the
 * first global row / element of this rank's shard in a data-parallel run, so a
 * sharded run draws what the single-GPU run draws for the same rows (streams
 * 0 and 4 are indexed by row, all others by element).  msgm_rng_advance bumps
 * `offset` on the stream (graph-safe).
 */
#ifndef MSGM_HIP_H
#define MSGM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* msgm_stream_t;   /* hipStream_t */

enum {
  MSGM_OK = 0,
  MSGM_E_BADARG = -1,       /* null pointer / non-positive size                  */
  MSGM_E_UNSUPPORTED = -2,  /* shape outside what the kernel was built for       */
  MSGM_E_WORKSPACE = -3,    /* workspace too small                               */
  MSGM_E_LAUNCH = -4        /* hipGetLastError() != hipSuccess after the launch  */
};

/* SDE family of the base process (reference: SDEs.py:161-215 SGMsde,
 * :221-509 MSGMsde dense / sparse tensor). */
enum { MSGM_SDE_SGM = 0, MSGM_SDE_MSGM_SPARSE = 1, MSGM_SDE_MSGM_DENSE = 2 };

/* Which process the integrator stage evaluates (SDEs.py:30-47 forward_SDE,
 * :538-588 PluginReverseSDE). */
enum { MSGM_PROC_REVERSE = 0, MSGM_PROC_FORWARD = 1 };

typedef struct {
  int32_t kind;        /* MSGM_SDE_*                                              */
  float beta_min;      /* SDEs.py:58-59                                           */
  float beta_max;
  float T;             /* horizon (SDEs.py:57)                                    */
  float t_epsilon;     /* SDEs.py:60                                              */
  const float* G;      /* dense (n,n,n) tensor, MSGM_SDE_MSGM_DENSE only          */
  const float* L_G;    /* dense (n,n) Ito correction (SDEs.py:246)                */
} msgm_sde_t;

int msgm_version(void);
const char* msgm_error_string(int code);

/* ---- RNG state ---------------------------------------------------------- */
/* rng[1] += n (one thread).  Replaces the implicit advance of torch's global
 * generator after each randn/rand call. */
int msgm_rng_advance(uint64_t* rng, uint64_t n, msgm_stream_t stream);

/* Fill with U[0,1) / N(0,1) from the Philox stream (used by tests to obtain
 * the exact numbers a fused kernel will draw, and by latent_sample). */
int msgm_fill_uniform(float* out, int64_t n, const uint64_t* rng, uint32_t stream_id, msgm_stream_t stream);
int msgm_fill_normal(float* out, int64_t n, const uint64_t* rng, uint32_t stream_id, msgm_stream_t stream);

/* ---- K1: forward-diffusion perturbation (SGM closed form) ---------------- */
/* Replaces PluginReverseSDE.sample_t (SDEs.py:684-693) + SDE.sample_Song_et_al
 * (SDEs.py:134-146) + mean_weight/var (SDEs.py:177-181):
 *   t_b = clamp(u_b*T), y = mean_weight(t) x0 + sqrt(var(t)) eps.
 * u (B) and eps (B*d) are read when non-NULL, else drawn from rng streams
 * 0 (u) and 1 (eps).  t_out (B) always written; eps_out optional. */
int msgm_perturb_vp(const float* x0, float* y, float* t_out, float* eps_out,
                    int64_t B, int64_t d, const msgm_sde_t* sde,
                    const float* u, const float* eps, const uint64_t* rng,
                    msgm_stream_t stream);
/* The same closed form with the times GIVEN: y = mean_weight(t_b) x0 + sqrt(var(t_b)) eps, t (B) used exactly as passed
 * (no u*T round trip, no clamp) — SGMsde.sample(t, y0) / sample_Song_et_al (SDEs.py:134-146,196-199). */
int msgm_perturb_vp_at(const float* x0, float* y, float* eps_out, int64_t B, int64_t d, const msgm_sde_t* sde,
                       const float* t, const float* eps, const uint64_t* rng, msgm_stream_t stream);

/* Training-step prologue in ONE launch: msgm_perturb_vp + msgm_rademacher with
 * in-kernel Philox draws (streams 0,1,2), plus `*step_ctr += 1` (optimizer step
 * count, may be NULL).  Draws are identical to the two separate kernels. */
int msgm_ssm_prep(const float* x0, float* y, float* t_out, float* v, int64_t B, int64_t d,
                  const msgm_sde_t* sde, const uint64_t* rng, int64_t* step_ctr, msgm_stream_t stream);

/* Bit-exact per-row stop index k = trunc((nsf*t)/T) (int32), rows with t>=T
 * forced to nsf.  Replaces SDEs.py:89-101. */
int msgm_forward_step_index(const float* t, int32_t* k, int64_t B, int32_t nsf, float T,
                            msgm_stream_t stream);

/* Rademacher probe v = 2[u>=1/2]-1 (SDEs.py:514-515); u NULL => rng stream 2. */
int msgm_rademacher(float* v, int64_t n, const float* u, const uint64_t* rng, msgm_stream_t stream);

/* ---- K2/K3/K4: one integrator stage -------------------------------------- */
/* inc = drift(t, x) * delta + sigma(t, x) . dW   (sde_scheme.py:18-40 EMstep
 * applied to sde.mu / sde.mu_Strato and sde.sigma, SDEs.py:30-47,556-588),
 * for the three diffusion layouts (diagonal SGM, sparse 3-point stencil,
 * dense G contraction without materialising (B,n,n)).
 *   proc    MSGM_PROC_REVERSE: drift = (1-l/2) G(s,x)a - f(s,x) + (1-l) divS(s,x),
 *           s = T - t, sigma = sqrt(1-l) g(s,x); `a` = score-net output (B,n).
 *           MSGM_PROC_FORWARD: drift = f_strato (+ 1/2 divS if !strato), sigma = g(t,x).
 *   strato  0: Ito drift (EM); 1: Stratonovich drift (Heun / RK4).
 *   dW      Wiener increment (B,n) (already scaled by sqrt(delta)); when NULL,
 *           dW = sqrt_delta * z with z read from `z` or drawn (rng stream 3).
 *   out = base + c_out * inc  (base may be NULL => out = c_out * inc; out may
 *           alias base or x).
 *   norm0   optional (B): after the update rescale each row of `out` to this
 *           norm (norm_correction, sde_scheme.py:85-86).
 *   dW_out / inc_out  optional (B,n): the Wiener increment actually used (RK4
 *           shares one dW across its stages, sde_scheme.py:227) / the bare inc.
 *   t_dev / step_dev  optional device scalars overriding `t` / `rng_step` (graph replay).
 *   delta_rows  optional (B): per-row step length; then t_b = t + t_frac*delta_b
 *           and dW_b = sqrt(delta_b) z_b (the one-step RK4 of SDEs.py:112-117
 *           for rows whose stop index is 0).
 * t is a scalar shared by the batch (the integrators fill a (B,1) tensor with
 * one value, sde_scheme.py:81). */
int msgm_sde_stage(float* out, const float* base, float c_out,
                   const float* x, const float* a, const float* dW, const float* z,
                   float sqrt_delta, const uint64_t* rng, uint64_t rng_step,
                   float* dW_out, float* inc_out,
                   int64_t B, int64_t n, const msgm_sde_t* sde, int32_t proc, int32_t strato,
                   float t, float delta, float lmbd, const float* norm0,
                   const float* delta_rows, float t_frac,
                   const float* t_dev, const int64_t* step_dev,
                   msgm_stream_t stream);

/* Device-side clock for a hipGraph-replayed sampler step: t_dev[0] = ts[*step],
 * s_out[b] = T - t (score-net time argument).  With t_dev / step_dev passed to
 * msgm_sde_stage (overriding `t` / `rng_step`) one captured step can be
 * replayed for every i; msgm_counter_inc(step) closes the step. */
int msgm_time_tick(const float* ts, const int64_t* step, int64_t n_ts, float T, float* t_dev, float* s_out,
                   int64_t B, msgm_stream_t stream);
/* The same clock for the later stages of Heun / RK4 (sde_scheme.py:150,233-247): t_dev[0] = ts[*step] + t_add in
 * fp32 (t_add = delta/2 or delta as a float32, exactly as upstream adds them to a float32 tensor). */
int msgm_time_tick_stage(const float* ts, const int64_t* step, int64_t n_ts, float T, float t_add, float* t_dev,
                         float* s_out, int64_t B, msgm_stream_t stream);

/* K12: SSM loss reduction for a score net evaluated on a (primal | tangent)
 * stacked batch, SGM case.  `out` is [2B][n] (a = out[:B], adot = J_a v = out[B:]):
 *   per[b] = sum_i v_i (sqrt(beta_b) adot_i + 1/2 beta_b v_i) + 1/2 a_i^2   (SDEs.py:631-646)
 * and the cotangents of mean_b(per) * (1/inv_batch scaling) w.r.t. (a, adot):
 *   g[:B] = a * inv_batch, g[B:] = sqrt(beta) v * inv_batch. */
int msgm_ssm_loss_diag(const float* out, const float* v, const float* t, float* per, float* g,
                       int64_t B, int64_t n, const msgm_sde_t* sde, float inv_batch, msgm_stream_t stream);

/* u = (d mu_to_div / d a)^T v and the a-independent constant of the SSM loss
 * (SDEs.py:560-561,631-642) for the three SDE families — with them
 *   loss_b = adot.u + cst + 1/2 |a|^2       (adot = J_a v):
 *   SGM: u = sqrt(beta) v, cst = 1/2 beta |v|^2;
 *   MSGM: u_k = sqrt(beta) sum_ij G_ijk y_j v_i (sparse: c sqrt(beta) (v_k y_{k+1} - v_{k+1} y_k)), cst = 0. */
int msgm_ssm_terms(const float* y, const float* v, const float* t, float* u, float* cst, int64_t B, int64_t n,
                   const msgm_sde_t* sde, msgm_stream_t stream);
/* K12, general form: per[b] = adot.u + cst + 1/2|a|^2; g = [a ; u] * inv_batch, with
 * `out` the (primal | tangent) stacked net output [2B][n]. */
int msgm_ssm_loss(const float* out, const float* u, const float* cst, float* per, float* g, int64_t B, int64_t n,
                  float inv_batch, msgm_stream_t stream);

/* out = c0*a + c1*b + c2*c (b, c may be NULL): stage points of Heun / RK4
 * (x + K/2, sde_scheme.py:148,234,240,246). */
int msgm_lincomb(float* out, const float* a, float c0, const float* b, float c1,
                 const float* c, float c2, int64_t n, msgm_stream_t stream);

/* out = x + (k1 + 2 k2 + 2 k3 + k4)/6 with optional norm correction
 * (sde_scheme.py:250-253). */
int msgm_rk4_combine(float* out, const float* x, const float* k1, const float* k2,
                     const float* k3, const float* k4, int64_t B, int64_t n,
                     const float* norm0, msgm_stream_t stream);

/* Row L2 norms (B,n)->(B) (torch.norm(x,dim=1), sde_scheme.py:65). */
int msgm_row_norm(const float* x, float* out, int64_t B, int64_t n, msgm_stream_t stream);

/* Masked capture: rows with stop[b]==index copy x[b,:] -> kept[b,:]
 * (samplesToKeep branch, sde_scheme.py:89-92, device-resident). */
int msgm_keep_rows(float* kept, const float* x, const int32_t* stop, int32_t index,
                   int64_t B, int64_t n, msgm_stream_t stream);

/* ---- K13: Adam on a flat bucket ------------------------------------------ */
/* torch.optim.Adam defaults (MSGM_higherDim.py:792): p,m,v updated in place
 * from g * gscale; step count (1-based, after increment) read from
 * *step_dev when non-NULL else `step`.  Hyper-parameters are doubles because
 * upstream they are Python floats and the bias corrections are formed in
 * double before being rounded to fp32. */
int msgm_adam_step(float* p, const float* g, float* m, float* v, int64_t n,
                   double lr, double beta1, double beta2, double eps, float gscale,
                   int64_t step, const int64_t* step_dev, msgm_stream_t stream);

/* *ctr += 1 (one thread) — device-side step counter for graph replays. */
int msgm_counter_inc(int64_t* ctr, msgm_stream_t stream);

/* ---- K5: fused MLP score net (NN.py:73-120) ------------------------------ */
/* Parameters are the reference state_dict tensors, PyTorch layout:
 *   W1 (128, in), b1 (128), W2 (128,128), b2, W3 (128,128), b3, W4 (d,128), b4 (d)
 * with in = d + 1 (premodule None) or d + 2 (NormalizeLogRadius: [x^, log r, t]).
 * Supported: hidden 128, d <= 30. */
typedef struct {
  const float* W1; const float* b1;
  const float* W2; const float* b2;
  const float* W3; const float* b3;
  const float* W4; const float* b4;
  int32_t d;            /* input_dim == output_dim                              */
  int32_t premodule;    /* 0 none, 1 NormalizeLogRadius (NN.py:56-70)            */
} msgm_mlp_params_t;

/* a = MLP(y, t): y (B,d), t (B) -> a (B,d).  One kernel: all four layers
 * chained through fp32 MFMA accumulators, activations never leave the CU. */
int msgm_mlp_forward(const msgm_mlp_params_t* P, const float* y, const float* t, float* a,
                     int64_t B, msgm_stream_t stream);

/* Fused sampler step for SGM + MLP: a = MLP(x, T - t) then the EM update of
 * msgm_sde_stage (diag layout) in the same kernel.  x updated in place. */
int msgm_mlp_em_step(const msgm_mlp_params_t* P, float* x, int64_t B, const msgm_sde_t* sde,
                     float t, float delta, float lmbd, const float* z, const uint64_t* rng,
                     uint64_t rng_step, msgm_stream_t stream);

/* The WHOLE Euler-Maruyama loop of euler_maruyama_sampler (sde_scheme.py:43-99, keep_all_samples=False, SGM,
 * no norm correction) in ONE launch: step i uses t = ts[i] (device array of n_steps+1 values,
 * linspace(0,1,N+1)*T as upstream computes it, sde_scheme.py:59) and Philox step rng_step0 + i, i.e. exactly the
 * numbers n_steps successive msgm_mlp_em_step calls draw.  Rows never interact, so each workgroup takes its rows
 * through all steps; the small layers are staged once.  Needs B > 32 (MSGM_E_UNSUPPORTED otherwise). */
int msgm_mlp_em_loop(const msgm_mlp_params_t* P, float* x, int64_t B, const msgm_sde_t* sde, const float* ts, int32_t n_steps,
                     float delta, float lmbd, const uint64_t* rng, uint64_t rng_step0, msgm_stream_t stream);

/* Fused SSM training pass for SGM + MLP (replaces SDEs.py:607-646 +
 * loss.mean().backward(), MSGM_higherDim.py:807-808):
 *   per sample: a = MLP(y,t), adot = J_a v (forward-mode), loss_b =
 *   sqrt(beta) v.adot + 1/2 beta |v|^2 + 1/2 |a|^2, then the backward pass of
 *   mean_b(loss_b) w.r.t. every parameter, all inside one persistent kernel.
 * u (B,d), cst (B): optional general form of the loss from msgm_ssm_terms
 * (loss_b = adot.u + cst + 1/2|a|^2) — required for the MSGM families, NULL
 * selects the SGM closed form above.
 * Inputs y,t,v are (B,d),(B),(B,d).  Outputs: grads (flat, layout =
 * [W1,b1,W2,b2,W3,b3,W4,b4], n_params floats), loss_per (B, optional),
 * loss_sum (1 float, optional).  inv_batch scales the mean (1/global batch).
 * workspace: msgm_mlp_ssm_workspace(d, premodule) bytes (per-workgroup partial
 * gradient slabs, reduced deterministically by a second kernel). */
size_t msgm_mlp_ssm_workspace(int32_t d, int32_t premodule);
int64_t msgm_mlp_num_params(int32_t d, int32_t premodule);
int msgm_mlp_ssm_grad(const msgm_mlp_params_t* P, const float* y, const float* t, const float* v,
                      const float* u, const float* cst,
                      int64_t B, const msgm_sde_t* sde, float inv_batch,
                      float* grads, float* loss_per, float* loss_sum,
                      void* workspace, size_t workspace_bytes, msgm_stream_t stream);

/* The two halves of msgm_mlp_ssm_grad, exposed so the dominant kernel can be
 * timed / profiled on its own: the persistent fused kernel (writes *n_slabs
 * per-workgroup gradient slabs into the workspace; n_slabs is a HOST int) and
 * the deterministic slab reduction. */
int msgm_mlp_ssm_partial(const msgm_mlp_params_t* P, const float* y, const float* t, const float* v,
                         const float* u, const float* cst,
                         int64_t B, const msgm_sde_t* sde, float inv_batch, float* loss_per,
                         void* workspace, size_t workspace_bytes, int32_t* n_slabs_host,
                         msgm_stream_t stream);
int msgm_mlp_ssm_reduce(int32_t d, int32_t premodule, const void* workspace, int32_t n_slabs,
                        float inv_batch, float* grads, float* loss_sum, msgm_stream_t stream);

/* ---- K6/K11: convolutions as implicit GEMMs (U-Net score nets) ------------- */
/* Activations are channels-last [N][H][W][C] (1-D: H = 1).  The forward-mode
 * tangent rides as the second half of the batch (rows n >= n_bias get no bias).
 * One geometry struct describes forward, dgrad and wgrad of Conv1d/Conv2d
 * (NNUnet1D.py:17-20,84; model/unet.py:58,92,143,157,356) and ConvTranspose1d
 * (NNUnet1D.py:98):
 *   mode 0: input index i = o*stride + k - pad        (strided convolution)
 *   mode 1: i = (o + pad - k)/stride when divisible   (transposed convolution;
 *           also the dgrad of mode 0, and vice versa)
 *   ups 1 : the stored input is nearest-upsampled 2x on the fly (Upsample,
 *           model/unet.py:60-73). */
typedef struct {
  int32_t N, Hi, Wi, Ho, Wo;
  int32_t KH, KW, strideH, padH, strideW, padW;   /* 1-D: H = 1, KH = 1, strideH = 1, padH = 0 */
  int32_t mode, ups;
} msgm_conv_geom_t;

/* out[m][co] (+)= sum_tap sum_c in[src(m,tap)][c] Wp[tap][co][c] (+ bias[co] for n < n_bias, + samp_bias[n][co] for n < n_samp:
 * the per-sample embedding bias; n_samp = N when the embedding itself carries a tangent).
 * Up to two inputs are concatenated along channels without materialising the
 * concat (src1 may be NULL).  Wp is the packed weight [taps][CoutP][Ktot]
 * (msgm_pack_weight; CoutP multiple of 16, each source's channels padded to 16).
 * A Linear layer is the 1x1 case with H = W = 1. */
int msgm_conv_forward(const msgm_conv_geom_t* geom, const float* src0, int32_t C0, const float* src1, int32_t C1,
                      const float* Wp, int32_t Cout, int32_t CoutP, int32_t Ktot, const float* bias,
                      const float* samp_bias, int32_t n_bias, int32_t n_samp, float* out, int32_t accumulate,
                      msgm_stream_t stream);

/* The same convolution with optional work folded in:
 *  - residual [N][Ho][Wo][Cout]: added in the epilogue (ResBlock `h + x`, attention `x + proj(...)`, model/unet.py:
 *    187,232) — saves one read-modify-write pass over the output;
 *  - in_scale / in_shift [N][C0+C1] (+ in_act = 1: SiLU): the convolution reads act(a x + b) instead of x, i.e.
 *    GroupNorm(+SiLU) (model/unet.py:140-143,152-155,214) is applied while the input tile is staged and the
 *    normalised tensor is never written.  Only the halo-tile kernel (stride-1 "same" shapes) can do this:
 *    msgm_conv_input_transform_supported() tells; otherwise the call returns MSGM_E_UNSUPPORTED.
 * fuse may be NULL (= msgm_conv_forward). */
typedef struct {
  const float* residual;
  const float* in_scale;
  const float* in_shift;
  int32_t in_act;
  int32_t reserved;
  /* structurally-zero weight blocks the halo-tile kernel may skip (ignored by the other kernels, whose packed
   * weights hold the zeros): bit t = tap t present; 0 = all taps.  tapmask_in[i]: i-th 32-channel input chunk
   * (flattened over the sources); tapmask_out[y]: y-th output-channel block of 32 (CoutP % 64 != 0) or 64. */
  uint16_t tapmask_in[16];
  uint16_t tapmask_out[8];
  /* optional by-product for the GroupNorm that reads this convolution's output next (model/unet.py:140-143,152-155,214:
   * every GroupNorm of the U-Net reads a convolution output): per-channel partial sums of the FINAL output values
   * (after bias / accumulate / residual), chanstats[N][S][2][Cout] floats = {sum, sum of squares} per (sample, slot),
   * S = msgm_conv_chanstats_slots(...) > 0 (one slot per (tile, wave) of the kernel's tiling; 0 = this convolution
   * cannot produce them -> MSGM_E_UNSUPPORTED when set).  msgm_groupnorm_affine_chanstats() turns them into the
   * (scale, shift) a consuming convolution applies — the separate statistics pass over the tensor disappears. */
  float* chanstats;
} msgm_conv_fuse_t;
int msgm_conv_input_transform_supported(const msgm_conv_geom_t* geom, int32_t C0, int32_t C1, int32_t CoutP);
int32_t msgm_conv_chanstats_slots(const msgm_conv_geom_t* geom, int32_t C0, int32_t C1, int32_t Cout, int32_t CoutP);
/* 1 if this forward convolution is the U-Net's output convolution shape (3x3 "same", 32 input channels, <= 4 output
 * channels; model/unet.py:442-446) that the vector-ALU kernel serves — it also accepts the in_scale / in_shift input
 * transform, although msgm_conv_input_transform_supported() (which does not see Cout) says no for CoutP = 16. */
int msgm_conv_small_cout_supported(const msgm_conv_geom_t* geom, int32_t C0, int32_t C1, int32_t Cout);
int msgm_conv_forward_fused(const msgm_conv_geom_t* geom, const float* src0, int32_t C0, const float* src1, int32_t C1,
                            const float* Wp, int32_t Cout, int32_t CoutP, int32_t Ktot, const float* bias,
                            const float* samp_bias, int32_t n_bias, int32_t n_samp, float* out, int32_t accumulate,
                            const msgm_conv_fuse_t* fuse, msgm_stream_t stream);

/* Winograd F(2x2, 3x3) FORWARD of a stride-1, pad-1 3x3 convolution (mode 0; Ho, Wo multiples of 16; C0, C1 multiples
 * of 16; CoutP a multiple of 32) for the reverse-SDE SAMPLER, whose step is 55 % 3x3 convolutions (sde_scheme.py:82 ->
 * model/unet.py:140-158): 16 instead of 36 multiplications per (co, ci) and 2x2 outputs, all in fp32 (the result differs
 * from msgm_conv_forward by rounding only).  Same arguments and fused options (second source, folded 2x upsample,
 * bias / per-sample bias, accumulate, residual, GroupNorm(+SiLU) input transform) as msgm_conv_forward_fused, except
 * that the weight image WpW is [16][CoutP][Ktot]: the transformed kernels G g G^T written by
 * msgm_wino_pack_weights_batched (same job table as msgm_pack_weights_batched, taps = 9).  No tangent-specific code: tangent
 * rows are batch rows.  r3: the training step uses it too, for the forward and — with the image of the flipped, transposed
 * kernels (a pack job with tap stride -1 starting at tap 8) — for the dgrad; weight gradients keep their own kernels.  fuse->chanstats: [N][(Ho/16) (Wo/16) 4][2][Cout] (Cout % 4 == 0). */
int msgm_conv_wino_supported(const msgm_conv_geom_t* geom, int32_t C0, int32_t C1, int32_t CoutP);
int msgm_conv_forward_wino(const msgm_conv_geom_t* geom, const float* src0, int32_t C0, const float* src1, int32_t C1,
                           const float* WpW, int32_t Cout, int32_t CoutP, int32_t Ktot, const float* bias,
                           const float* samp_bias, int32_t n_bias, int32_t n_samp, float* out, int32_t accumulate,
                           const msgm_conv_fuse_t* fuse, msgm_stream_t stream);

/* dWp[tap][co][koff + c] += sum_m gy[m][co] in[src(m,tap)][c] (float atomics across
 * position chunks; zero dWp first).  One call per concatenated source.
 * dbias (may be NULL): dbias[co] += sum over the primal rows n < n_bias and all pixels of gy — the bias gradient
 * (torch: conv backward's grad_bias) as a by-product of the tiles the kernel stages anyway; zero it first.
 * tapmask_c32 / tapmask_co32 (HOST arrays, may be NULL): per 32-channel block of the input / output channels, bit t =
 * tap t of that block is a real weight (0 = all); blocks that are structurally zero are not computed (the tile
 * kernel only — their dWp entries are then left untouched). */
int msgm_conv_wgrad(const msgm_conv_geom_t* geom, const float* gy, const float* src, int32_t C, int32_t koff,
                    float* dWp, int32_t Cout, int32_t CoutP, int32_t Ktot, float* dbias, int32_t n_bias,
                    const uint16_t* tapmask_c32, const uint16_t* tapmask_co32, msgm_stream_t stream);
/* The same gradients WITHOUT float atomics (bitwise reproducible from run to run): every workgroup column stores its
 * partial [taps][CoutP][C] block (+ bias partials) into its own slab of the workspace and a second kernel adds the
 * slabs in slot order; `dWp` / `dbias` are accumulated into exactly as above.  Workspace bytes from
 * msgm_conv_wgrad_workspace (n_bias = 0 when dbias is NULL); ~37 MB per call at the C4 shapes, i.e. ~20 us of HBM
 * time — the trainers use this entry by default. */
size_t msgm_conv_wgrad_workspace(const msgm_conv_geom_t* geom, int32_t C, int32_t Cout, int32_t CoutP, int32_t n_bias);
int msgm_conv_wgrad_det(const msgm_conv_geom_t* geom, const float* gy, const float* src, int32_t C, int32_t koff,
                        float* dWp, int32_t Cout, int32_t CoutP, int32_t Ktot, float* dbias, int32_t n_bias,
                        const uint16_t* tapmask_c32, const uint16_t* tapmask_co32, void* workspace, size_t workspace_bytes,
                        msgm_stream_t stream);

/* Deferred form of msgm_conv_wgrad_det: launches the producing kernel(s) only and describes the slot reduction(s) it would
 * have launched in jobs_out[0 .. *n_jobs_out) (host memory, room for 2) — the workspace must stay untouched until the
 * caller has run them, all jobs of a backward pass in ONE launch, with msgm_slot_reduce_batched (the weight / bias
 * gradients of a U-Net step are ~140 reductions of a few microseconds each).  The caller uploads the job table after
 * filling block_begin with the running sum of ceil(n_elem / 32) + ceil(n_elem2 / 32); total_blocks = that sum. */
typedef struct {
  const float* part;      /* [nslots][stride] partials */
  float* out;             /* weight image (Ktot > 0: element e = (row)*C + c goes to row*Ktot + koff + c) or plain vector */
  float* out2;            /* optional second range [n_elem, n_elem + n_elem2) of the same slots, identity-mapped (bias) */
  int64_t stride, n_elem, n_elem2, block_begin;
  int32_t nslots, C, Ktot, koff, rowsP, rows, accumulate, reserved;
} msgm_reduce_job_t;
int msgm_conv_wgrad_slabs(const msgm_conv_geom_t* geom, const float* gy, const float* src, int32_t C, int32_t koff,
                          float* dWp, int32_t Cout, int32_t CoutP, int32_t Ktot, float* dbias, int32_t n_bias,
                          const uint16_t* tapmask_c32, const uint16_t* tapmask_co32, void* workspace, size_t workspace_bytes,
                          msgm_reduce_job_t* jobs_out, int32_t* n_jobs_out, msgm_stream_t stream);
int msgm_slot_reduce_batched(const msgm_reduce_job_t* jobs_dev, int32_t n_jobs, int64_t total_blocks, msgm_stream_t stream);

/* Wp[t][r][kp_off + c] = W[r*sr + (col_off + c)*sc + t*st] for r < rows, c < ncols
 * (strides in elements: any of the PyTorch layouts (Cout,Cin,k), (Cin,Cout,k) and
 * their transposes for dgrad); msgm_unpack_weight is the inverse for gradients. */
int msgm_pack_weight(const float* W, float* Wp, int32_t rows, int32_t ncols, int32_t col_off, int32_t taps,
                     int64_t sr, int64_t sc, int64_t st, int32_t rowsP, int32_t Ktot, int32_t kp_off,
                     msgm_stream_t stream);
int msgm_unpack_weight(float* dW, const float* dWp, int32_t rows, int32_t ncols, int32_t col_off, int32_t taps,
                       int64_t sr, int64_t sc, int64_t st, int32_t rowsP, int32_t Ktot, int32_t kp_off,
                       int32_t accumulate, msgm_stream_t stream);

/* One launch for every (un)pack job of a network.  `jobs` is a DEVICE array; job j copies
 * Wp[t][r][kp_off+c] <- W[r*sr + (col_off+c)*sc + t*st] (unpack != 0: the other way, overwriting W, or adding
 * atomically when the job's `reserved` field is 1).  The
 * table is static as long as the parameter buckets and packed images do not move. */
typedef struct {
  float* W; float* Wp;
  int64_t sr, sc, st;
  int32_t rows, ncols, col_off, taps, rowsP, Ktot, kp_off, reserved;
} msgm_pack_job_t;
int msgm_pack_weights_batched(const msgm_pack_job_t* jobs, int32_t n_jobs, int32_t unpack, msgm_stream_t stream);
/* OPT-IN EXPERIMENT (never the default path; MSGM_SAMPLER_BF16X3=1 on the host side): the same 3x3 stride-1 pad-1 forward
 * convolution of the sampler (sde_scheme.py:82 -> model/unet.py:140-158; Ho, Wo multiples of 16; C0, C1, CoutP multiples of
 * 32) with its matrix work in bf16-SPLIT arithmetic: every fp32 operand as three bf16 pieces (24 mantissa bits), six
 * v_mfma_f32_16x16x32_bf16 products per fp32 product, fp32 accumulate — as accurate as the fp32 MFMA (tools/probe_bf16x3.hip).
 * Same arguments and fused options as msgm_conv_forward_wino (no tap masks); Wb = [3][9][CoutP][Ktot] bf16 written by
 * msgm_b6_split_weights from the packed fp32 image Wp ([9][CoutP][Ktot], n_elem = 9 CoutP Ktot; Wb needs 3 n_elem bf16 + 64
 * bytes).  fuse->chanstats: [N][(Ho/16) (Wo/16) 4][2][Cout] (Cout % 4 == 0). */
int msgm_conv_b6_supported(const msgm_conv_geom_t* geom, int32_t C0, int32_t C1, int32_t CoutP);
int msgm_b6_split_weights(const float* Wp, void* Wb, int64_t n_elem, msgm_stream_t stream);
int msgm_conv_forward_b6(const msgm_conv_geom_t* geom, const float* src0, int32_t C0, const float* src1, int32_t C1,
                         const void* Wb, int32_t Cout, int32_t CoutP, int32_t Ktot, const float* bias,
                         const float* samp_bias, int32_t n_bias, int32_t n_samp, float* out, int32_t accumulate,
                         const msgm_conv_fuse_t* fuse, msgm_stream_t stream);

/* The Winograd images of msgm_conv_forward_wino: same job table (taps = 9), Wp = [16][rowsP][Ktot]. */
int msgm_wino_pack_weights_batched(const msgm_pack_job_t* jobs, int32_t n_jobs, msgm_stream_t stream);

/* Dual-number activations on a (primal | tangent) stacked tensor of 2*half
 * elements: act 0 = exact-erf GELU (NNUnet1D.py:18), 1 = SiLU (nn_utils.py:44-46).
 * forward: hP = f(zP), hT = f'(zP) zT (dual = 0: primal only, `half` elements).
 * backward (in place on g): gP <- gP f' + gT f'' zT, gT <- gT f'. */
int msgm_act_dual_forward(int32_t act, const float* z, float* h, int64_t half, int32_t dual, msgm_stream_t stream);
int msgm_act_dual_backward(int32_t act, const float* z, float* g, int64_t half, msgm_stream_t stream);

/* S[n][c] = sum_pos x[n][pos][c]; out[n][c] = x[n][pos][c]; x[n][pos][c] += sgn*E[n][c]. */
int msgm_colsum(const float* x, float* S, int32_t N, int32_t P, int32_t C, msgm_stream_t stream);
/* msgm_colsum without float atomics: per-chunk partials in the workspace, added in chunk order. */
size_t msgm_colsum_workspace(int32_t N, int32_t P, int32_t C);
int msgm_colsum_det(const float* x, float* S, int32_t N, int32_t P, int32_t C, void* workspace, size_t workspace_bytes,
                    msgm_stream_t stream);
int msgm_gather_row(const float* x, float* out, int32_t N, int32_t P, int32_t C, int32_t pos, msgm_stream_t stream);
int msgm_add_row(float* x, const float* E, int32_t N, int32_t P, int32_t C, int32_t pos, float sgn, msgm_stream_t stream);

/* ---- K7/K8/K9/K10: the 2-D U-Net's non-convolution ops ---------------------- */
/* GroupNorm(G groups, affine) [+ SiLU] on a (primal | tangent) stacked tensor
 * x [N][P][C] channels-last, N = 2*Bp when dual (model/nn_utils.py:39-46,107-114;
 * used at model/unet.py:140-143,152-155,214,443-444).  Tangent:
 *   ydot = gamma (xdot - mean(xdot) - xhat mean(xhat xdot)) / sigma.
 * Each direction is fully parallel launches (moment reduction over (sample,
 * pixel-chunk) workgroups into per-chunk SLOTS of `workspace`, summed in chunk
 * order by a finalise kernel, then an elementwise apply pass), so 32 samples/GPU
 * still fill the chip and the result is bitwise reproducible (no atomics).
 * stats [Bp][G][4] = {mean, 1/sigma, mean(xdot), mean(xhat xdot)} is written by
 * forward (may be NULL when no backward follows) and read by backward, which
 * recomputes xhat / SiLU from x, adds to dgamma / dbeta (slot-ordered sums) and
 * writes the input cotangents (primal | tangent) to gx (may alias gout);
 * `residual` (may be NULL, same shape as gx, may alias gx) is added to them — the
 * skip branch of a ResBlock / attention block (`h + x`, model/unet.py:187,232)
 * without a separate axpy pass.
 * The workspace needs no initialisation: every slot that is read was written by the same call. */
size_t msgm_groupnorm_workspace(int32_t Bp, int32_t G);   /* bytes: double moment accumulators + float statistics */
int msgm_groupnorm_dual_forward(const float* x, const float* gamma, const float* beta, float* out, float* stats,
                                int32_t Bp, int32_t P, int32_t C, int32_t G, int32_t dual, int32_t silu, float eps,
                                void* workspace, size_t workspace_bytes, msgm_stream_t stream);
int msgm_groupnorm_dual_backward(const float* x, const float* gamma, const float* beta, const float* stats,
                                 const float* gout, float* gx, float* dgamma, float* dbeta, int32_t Bp, int32_t P,
                                 int32_t C, int32_t G, int32_t silu, float eps, const float* residual, void* workspace,
                                 size_t workspace_bytes, msgm_stream_t stream);
/* The same pair for an input that is the channel concatenation of TWO tensors (x0: C0 channels, x1: C1; the decoder's
 * cat([h, skip]), model/unet.py:514) without materialising the concatenation: forward reads both and writes ONE
 * normalised tensor of C0 + C1 channels; backward reads both and writes the input cotangent into gx0 / gx1 (each shaped
 * like its source).  C0 % 4 == 0 and C1 % 4 == 0. */
int msgm_groupnorm_dual_forward2(const float* x0, int32_t C0, const float* x1, int32_t C1, const float* gamma,
                                 const float* beta, float* out, float* stats, int32_t Bp, int32_t P, int32_t G, int32_t dual,
                                 int32_t silu, float eps, void* workspace, size_t workspace_bytes, msgm_stream_t stream);
int msgm_groupnorm_dual_backward2(const float* x0, int32_t C0, const float* x1, int32_t C1, const float* gamma,
                                  const float* beta, const float* stats, const float* gout, float* gx0, float* gx1,
                                  float* dgamma, float* dbeta, int32_t Bp, int32_t P, int32_t G, int32_t silu, float eps,
                                  void* workspace, size_t workspace_bytes, msgm_stream_t stream);
/* The backward of either form (x1 == NULL, C1 == 0: one source) with the PARAMETER gradients left as slots: the
 * per-(sample, chunk) partial sums of dgamma | dbeta are written to `pslots` (msgm_groupnorm_param_slots_bytes() bytes, caller
 * owned, must stay valid until the reduction ran) and their slot-ordered sums into dgamma / dbeta (accumulating) are described
 * in jobs_out[0..1] for msgm_slot_reduce_batched — a backward pass runs ONE reduction launch for all its GroupNorms and
 * convolutions instead of one per layer (model/nn_utils.py:107-114 under autograd: two reductions per layer). */
size_t msgm_groupnorm_param_slots_bytes(int32_t Bp, int32_t P, int32_t C);
int msgm_groupnorm_dual_backward_slots(const float* x0, int32_t C0, const float* x1, int32_t C1, const float* gamma,
                                       const float* beta, const float* stats, const float* gout, float* gx0, float* gx1,
                                       float* dgamma, float* dbeta, int32_t Bp, int32_t P, int32_t G, int32_t silu, float eps,
                                       const float* residual, const float* residual2, void* workspace, size_t workspace_bytes,
                                       float* pslots, size_t pslots_bytes, msgm_reduce_job_t* jobs_out, int32_t* n_jobs_out,
                                       msgm_stream_t stream);
/* residual2 (may be NULL; one source only): a SECOND tensor added to the input cotangent in the same apply pass — the cotangent
 * that reached this tensor through the U-Net's skip stack (model/unet.py:514: the encoder activations feed the decoder as
 * well), which was a separate `dh += skip` pass over the tensor for each of the nine encoder blocks. */

/* GroupNorm statistics only, returned as the per-(sample, channel) affine map y = scale x + shift
 * (scale = gamma/sigma, shift = beta - mean scale; [Bp][C0+C1] each) for a consumer that applies it while reading
 * x (msgm_conv_forward_fused) — the normalised tensor is never written.  x1 (may be NULL) is a second tensor whose
 * C1 channels are concatenated after x0's C0 (decoder ResBlocks normalise cat([h, skip]), model/unet.py:514), so the
 * concatenation need not be materialised either.  No tangent (sampler path).  Workspace as above. */
int msgm_groupnorm_affine(const float* x0, int32_t C0, const float* x1, int32_t C1, const float* gamma, const float* beta,
                          float* scale, float* shift, int32_t Bp, int32_t P, int32_t G, float eps, void* workspace,
                          size_t workspace_bytes, msgm_stream_t stream);

/* The same affine map WITHOUT reading the tensor: from the per-channel partial sums the producing convolution(s) wrote
 * (msgm_conv_fuse_t.chanstats; cs0 = [Bp][S0][2][C0], cs1 (may be NULL) = [Bp][S1][2][C1] for the second, concatenated
 * source).  Slots are added in slot order in double, so the result does not depend on the batch size or launch shape.
 * P = pixels per sample.  No workspace. */
int msgm_groupnorm_affine_chanstats(const float* cs0, int32_t S0, int32_t C0, const float* cs1, int32_t S1, int32_t C1,
                                    const float* gamma, const float* beta, float* scale, float* shift, int32_t Bp, int32_t P,
                                    int32_t G, float eps, msgm_stream_t stream);

/* Batched fp32-MFMA GEMM with element strides:
 *   C[b](i,j) (+)= alpha ( sum_k A[b](i,k) B[b](k,j) + sum_k A2[b](i,k) B2[b](k,j) )
 * (QK^T, PV and their adjoints in QKVAttention, model/unet.py:236-250).  The
 * optional second pair (A2, B2: same strides, may be NULL) fuses the two-term
 * products of the dual-number attention into one pass over the output. */
int msgm_bmm(const float* A, const float* B, const float* A2, const float* B2, float* C,
             int32_t M, int32_t N, int32_t K, int32_t batch,
             int64_t sAb, int64_t sAi, int64_t sAk, int64_t sBb, int64_t sBk, int64_t sBj,
             int64_t sCb, int64_t sCi, int64_t sCj, float alpha, int32_t accumulate, msgm_stream_t stream);

/* The same with a SECOND output that shares the A operand: C3 = alpha A.B3 (B3 with B's strides, C3 with C's).
 * The dual-number attention always needs such a pair — P v next to Pdot v + P vdot, and the three adjoint pairs of
 * the backward alike — so the (T,T) operand is streamed from HBM once instead of twice.  Built for outputs that are
 * contiguous along j (sCj == 1) and K % 16 == 0; MSGM_E_UNSUPPORTED otherwise.  B3 / C3 NULL = msgm_bmm. */
int msgm_bmm_dual(const float* A, const float* B, const float* A2, const float* B2, const float* B3, float* C, float* C3,
                  int32_t M, int32_t N, int32_t K, int32_t batch,
                  int64_t sAb, int64_t sAi, int64_t sAk, int64_t sBb, int64_t sBk, int64_t sBj,
                  int64_t sCb, int64_t sCi, int64_t sCj, float alpha, int32_t accumulate, msgm_stream_t stream);

/* Row softmax on dual numbers (model/unet.py:249): S (primal logits) is
 * overwritten by P = softmax(S); with dual, Wd holds the tangent logits (kept
 * for backward) and Pd receives Pdot = P (Wd - sum_j P Wd).  Backward, in place:
 * (Pb, Pdb) = cotangents of (P, Pdot) -> cotangents of (S, Wd). */
int msgm_softmax_dual_forward(float* S, const float* Wd, float* Pd, int64_t rows, int32_t T, int32_t dual,
                              msgm_stream_t stream);
int msgm_softmax_dual_backward(const float* P, const float* Wd, float* Pb, float* Pdb, int64_t rows, int32_t T,
                               msgm_stream_t stream);

/* Fused single-head self-attention FORWARD without tangent (the sampler's use of
 * QKVAttention.forward, model/unet.py:236-250): out[n][t][:] = softmax_s(scale q_t.k_s) v_s with
 * qkv [N][T][3C] channels-last (q | k | v slices) and out [N][T][C]; scale = ch^-1/2 (the two ch^-1/4
 * factors of unet.py:245-248).  The (T,T) probabilities are never written to memory (online softmax).
 * msgm_attention_supported(T, C) != 0 iff the shape is built (C in {32,64,128}, T a multiple of 64);
 * otherwise msgm_attention_forward returns MSGM_E_UNSUPPORTED and the caller composes msgm_bmm +
 * msgm_softmax_dual_forward + msgm_bmm (what the training path always does: its backward needs P). */
int msgm_attention_supported(int32_t T, int32_t C);
int msgm_attention_forward(const float* qkv, float* out, int64_t N, int32_t T, int32_t C, float scale, msgm_stream_t stream);

/* Fused single-head self-attention on DUAL numbers for the TRAINING path — forward and backward of
 * QKVAttention.forward (model/unet.py:236-250) under the forward-mode SSM step (SDEs.py:616-646: primal +
 * tangent forward, one first-order backward over the pair), with no (B,T,T) tensor in memory:
 *   forward:  S = scale q k^T, Sd = scale (qd k^T + q kd^T), P = softmax(S), Pd = P o (Sd - rowsum(P o Sd)),
 *             att = [P v ; Pd v + P vd]; stats [2][Bp*T] receives per query the log-sum-exp of S and
 *             rbar = rowsum(P o Sd) — all the backward needs besides q, k, v and att.
 *   backward: given datt = [obar ; odbar], recomputes S / Sd tile by tile and writes dqkv = [qbar|kbar|vbar ;
 *             qdbar|kdbar|vdbar] (every element written exactly once; deterministic: the query-side partial
 *             sums of each 64-key block go to workspace slabs that are added in block order, no float atomics).
 * qkv / dqkv [2Bp][T][3C] channels-last (rows [0,Bp) primal, [Bp,2Bp) tangent), att / datt [2Bp][T][C].
 * Built for C in {32, 64} and T a multiple of 64 (msgm_attention_dual_supported); other shapes return
 * MSGM_E_UNSUPPORTED and the caller composes msgm_bmm / msgm_softmax_dual_* instead. */
int msgm_attention_dual_supported(int32_t T, int32_t C);
size_t msgm_attention_dual_workspace(int64_t Bp, int32_t T, int32_t C);     /* bytes, for the backward */
int msgm_attention_dual_forward(const float* qkv, float* att, float* stats, int64_t Bp, int32_t T, int32_t C, float scale,
                                msgm_stream_t stream);
int msgm_attention_dual_backward(const float* qkv, const float* att, const float* datt, const float* stats, float* dqkv,
                                 int64_t Bp, int32_t T, int32_t C, float scale, void* workspace, size_t workspace_bytes,
                                 msgm_stream_t stream);

/* ---- reporting metric next to the hot path -------------------- */
/* RBF kernel of compute_kernel / compute_mmd (quantitative_comparison.py:22-46):
 * k(x_i, y_j) = exp(-sum_d (x_i - y_j)^2 / d^2) for x [Nx][d], y [Ny][d].  K (may be NULL) receives the
 * [Nx][Ny] matrix; sum (may be NULL; a device double, zeroed by this call) receives sum_ij k — the MMD is
 * sxx/Nx^2 + syy/Ny^2 - 2 sxy/(Nx Ny) without the (Nx, Ny, d) broadcast tensor of the reference. */
int msgm_rbf_kernel(const float* x, const float* y, int64_t Nx, int64_t Ny, int32_t d, float* K, double* sum,
                    msgm_stream_t stream);

/* [cos(t f_j), sin(t f_j)], f_j = exp(-ln(max_period) j/half) (model/nn_utils.py:130-148). */
int msgm_timestep_embedding(const float* t, float* emb, int32_t B, int32_t dim, float max_period, msgm_stream_t stream);

/* The same on dual numbers when the argument depends on the input (log-radius
 * conditioning, NNUnet.py:101-105): t = [t ; tdot] (2*Bp), emb = [emb ; embdot]. */
int msgm_timestep_embedding_dual(const float* t, float* emb, int32_t Bp, int32_t dim, float max_period,
                                 msgm_stream_t stream);
/* NormalizeLogRadius premodule (NN.py:56-70) + the sqrt(n) rescale (NNUnet.py:205,
 * NNUnet1D.py:134) on a (primal | tangent) stacked (2*Bp, n) input:
 * out = scale x/(|x|+eps) (and its tangent), logr = [log r ; rdot/r]. */
int msgm_normalize_dual(const float* x, float* out, float* logr, int32_t Bp, int32_t n, int32_t dual, float scale,
                        float eps, msgm_stream_t stream);

/* flat (B, C*H*W) [channel-major; per channel 'C' (h*W+w) or 'F' (w*H+h) order]
 * <-> channels-last image [B][H][W][C], times `scale` (the /5, x5 of NNUnet.py:19-77). */
int msgm_flat_to_image(const float* flat, float* img, int32_t B, int32_t C, int32_t H, int32_t W, int32_t forder,
                       float scale, msgm_stream_t stream);
int msgm_image_to_flat(const float* img, float* flat, int32_t B, int32_t C, int32_t H, int32_t W, int32_t forder,
                       float scale, msgm_stream_t stream);
/* Embedding-projection bank: the emb_layers Linear(4*mc -> co) of EVERY ResBlock (model/unet.py:145-151, applied at :176-180)
 * in one launch per direction, straight on the PyTorch-layout parameters.  job j = one ResBlock; `block_begin` = index of the
 * job's first 32-channel workgroup (jobs sorted, total_blocks = sum of ceil(co/32)).
 *   forward : out_j[r][c] = (r < n_bias ? b_j[c] : 0) + sum_k semb[r][k] W_j[c][k]            (out_j: [R][co])
 *   backward: dW_j[c][k] = sum_r dout_j[r][c] semb[r][k];  db_j[c] = db2_j[c] = sum_{r<n_bias} dout_j[r][c]  (db / db2 may be
 *             NULL; db2 = the conv bias added at the same place);  dsemb[r][k] = sum_j sum_c dout_j[r][c] W_j[c][k]  (written,
 *             not accumulated).  Sums run in a fixed order: bitwise reproducible.  K % 4 == 0, K <= 512. */
typedef struct {
  const float* W;      /* [co][K]  (PyTorch Linear.weight)              */
  const float* b;      /* [co] or NULL                                   */
  float* out;          /* forward output  [R][co]                        */
  const float* dout;   /* backward input  [R][co]                        */
  float* dW;           /* [co][K]                                        */
  float* db;           /* [co] or NULL                                   */
  float* db2;          /* [co] or NULL                                   */
  int32_t co;
  int32_t block_begin;
} msgm_emb_job_t;
int msgm_emb_bank_forward(const msgm_emb_job_t* jobs_dev, int32_t n_jobs, int32_t total_blocks, const float* semb, int32_t R,
                          int32_t K, int32_t n_bias, msgm_stream_t stream);
int msgm_emb_bank_backward(const msgm_emb_job_t* jobs_dev, int32_t n_jobs, int32_t total_blocks, const float* semb, float* dsemb,
                           int32_t R, int32_t K, int32_t n_bias, msgm_stream_t stream);

/* out[n][h][w][c] = sum of the 2x2 block in[n][2h..2h+1][2w..2w+1][c]: adjoint of the
 * nearest-2x upsample (model/unet.py:67). */
int msgm_sum2x2(const float* in, float* out, int32_t N, int32_t H, int32_t W, int32_t C, msgm_stream_t stream);

/* Slab reduction fused with the Adam update of msgm_adam_step (single-GPU step:
 * nothing sits between them) and, when rng_advance != NULL, rng_advance[1] += 1.
 * grads may be NULL. */
int msgm_mlp_ssm_reduce_adam(int32_t d, int32_t premodule, const void* workspace, int32_t n_slabs,
                             float inv_batch, float* grads, float* loss_sum, float* params, float* m,
                             float* v, double lr, double beta1, double beta2, double eps,
                             const int64_t* step_dev, uint64_t* rng_advance, msgm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MSGM_HIP_H */
