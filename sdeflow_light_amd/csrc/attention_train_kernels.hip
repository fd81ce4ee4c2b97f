// attention_train_kernels.hip — K8 for the TRAINING path: single-head self-attention on dual numbers
// (QKVAttention, model/unet.py:236-250, under the forward-mode SSM step) WITHOUT materialising any (B,T,T) tensor.
//
//   S = s2 q k^T,  Sd = s2 (qd k^T + q kd^T),  P = softmax(S),  Pd = P o (Sd - r),  r_i = sum_j P_ij Sd_ij
//   o = P v,       od = Pd v + P vd
//
// FORWARD (k_attn_dual_fwd) is an online-softmax sweep over key blocks that carries, per query, the running max m,
// the sums l = sum e^{S-m}, r~ = sum e^{S-m} Sd and the accumulators  A1 = sum e^{S-m} v,  A2 = sum e^{S-m}(Sd v + vd);
// then o = A1/l, od = A2/l - (r~/l) o.  Six T^2 C products per block — exactly the six the materialised form runs
// (S, 2 for Sd, P v, Pd v, P vd) — and no (T,T) traffic.  It keeps L = m + log l and rbar = r~/l per query.
//
// BACKWARD (k_attn_dual_bwd) recomputes S, Sd from q, k (3 products) next to the 12 adjoint products.  With the
// cotangents g = obar, gd = odbar and the row scalars  c_i = gd_i.o_i,  delta_i = g_i.o_i + gd_i.od_i - rbar_i c_i
// (k_attn_dual_delta; derivation in DESIGN.md §4c):
//   Pbar = g v^T + gd vd^T,   D = gd v^T
//   Sdbar = P o (D - c),      Sbar = P o (Pbar + D o (Sd - rbar) - c Sd - delta)
//   vbar = P^T g + Pd^T gd,   vdbar = P^T gd
//   kbar = s2 (Sbar^T q + Sdbar^T qd),  kdbar = s2 Sdbar^T q
//   qbar = s2 (Sbar k + Sdbar kd),      qdbar = s2 Sdbar k
// One workgroup owns 64 keys of one sample (kbar.. vdbar accumulate in registers over all queries); the query-side
// gradients of each (query block, key block) pair leave the chip as per-key-block SLABS that k_attn_dq_reduce sums in a
// fixed order — no float atomics, bitwise reproducible.
//
// Layouts: qkv [2Bp][T][3C] channels-last (q | k | v channel slices; rows [0,Bp) primal, [Bp,2Bp) tangent),
// att / datt [2Bp][T][C], dqkv as qkv.  All MFMA work is v_mfma_f32_16x16x4_f32 (exact fp32); every operand reaches the
// matrix cores from LDS tiles that were filled with coalesced 16-byte global loads.  exp / log are the accurate expf /
// logf, not the 2-ulp hardware approximations (3 % of the forward kernel's time): the backward multiplies P by
// differences that cancel (Pbar - delta ...), so the probabilities are kept at fp32 accuracy.
#include "common.h"

__device__ __forceinline__ f32x4 mfma16t(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
static inline hipStream_t S(msgm_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// =============================================================================================== forward
// Transposed formulation (as the sampler kernel): S^T[key][query] = K Q^T, O^T[c][query] += V^T P^T, so a lane owns ONE
// query (lane&15): running max / sums are per-lane scalars, and e^{S-m} in the C/D layout IS the B operand of the
// second product (MFMA step r of key tile kt contracts key 16kt + 4(lane>>4) + r on both operands).
template <int CT, int QT, int KB, int NW = 4>   // C = 16*CT channels; QT tiles of 16 queries per wave; KB keys per LDS block;
__global__ void __launch_bounds__(64 * NW) k_attn_dual_fwd(const float* __restrict__ qkv, float* __restrict__ att,   // NW waves
                                                        float* __restrict__ lse, float* __restrict__ rbar, int T, int nqb,
                                                        int64_t Bp, float scale) {
  constexpr int C = 16 * CT, KP = C + 4, LD = 3 * C, KT = KB / 16, NT = 64 * NW;
  constexpr int NV = (KB * C / 4) / NT;                   // float4 per thread per matrix and key block
  static_assert(NV >= 1 && (KB * C / 4) % NT == 0, "key block does not divide over the workgroup's threads");
  extern __shared__ __attribute__((aligned(16))) float atd_lds[];
  float* Ks = atd_lds;
  float* Kd = Ks + KB * KP;
  float* Vs = Kd + KB * KP;
  float* Vd = Vs + KB * KP;
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, il = lane & 15, q = lane >> 4;
  // XCD-aware block order: workgroup ids are dealt round-robin to the 8 XCDs (each with its own L2); the nqb query
  // blocks of one sample all stream that sample's K / V tiles, so they get ids 8 apart (same XCD, back to back) and
  // the re-reads hit that L2.  Samples beyond the last multiple of 8 keep the plain order.
  int smp, qb;
  {
    const int full = (int)(Bp / 8) * 8 * nqb, b = blockIdx.x;
    if (b < full) { const int loc = b >> 3; smp = (loc / nqb) * 8 + (b & 7); qb = loc % nqb; }
    else { const int r = b - full; smp = (int)(Bp / 8) * 8 + r / nqb; qb = r % nqb; }
  }
  const float* bp = qkv + (size_t)smp * T * LD;            // primal rows of this sample
  const float* bt = bp + (size_t)Bp * T * LD;              // tangent rows
  const int q0 = (qb * NW + w) * 16 * QT;                  // first query of this wave

  f32x4 qf[QT][CT], qd[QT][CT], o[QT][CT], od[QT][CT];
  float m[QT], l[QT], rr[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
#pragma unroll
    for (int g = 0; g < CT; ++g) {
      const size_t off = (size_t)(q0 + 16 * qt + il) * LD + 16 * g + 4 * q;
      qf[qt][g] = *reinterpret_cast<const f32x4*>(bp + off);
      qd[qt][g] = *reinterpret_cast<const f32x4*>(bt + off);
      o[qt][g] = f32x4{0, 0, 0, 0};
      od[qt][g] = f32x4{0, 0, 0, 0};
    }
    m[qt] = -INFINITY; l[qt] = 0.f; rr[qt] = 0.f;
  }

  f32x4 pk[NV], pkd[NV], pv[NV], pvd[NV];
  auto gload = [&](int kb) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + NT * i, key = idx / (C / 4), c4 = idx - key * (C / 4);
      const size_t off = (size_t)(kb * KB + key) * LD + 4 * c4;
      pk[i] = *reinterpret_cast<const f32x4*>(bp + off + C);
      pv[i] = *reinterpret_cast<const f32x4*>(bp + off + 2 * C);
      pkd[i] = *reinterpret_cast<const f32x4*>(bt + off + C);
      pvd[i] = *reinterpret_cast<const f32x4*>(bt + off + 2 * C);
    }
  };
  gload(0);
  const int nkb = T / KB;
  for (int kb = 0; kb < nkb; ++kb) {
    __syncthreads();                                       // the previous block's readers are done
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + NT * i, key = idx / (C / 4), c4 = idx - key * (C / 4);
      *reinterpret_cast<f32x4*>(Ks + key * KP + 4 * c4) = pk[i];
      *reinterpret_cast<f32x4*>(Kd + key * KP + 4 * c4) = pkd[i];
      *reinterpret_cast<f32x4*>(Vs + key * KP + 4 * c4) = pv[i];
      *reinterpret_cast<f32x4*>(Vd + key * KP + 4 * c4) = pvd[i];
    }
    __syncthreads();
    if (kb + 1 < nkb) gload(kb + 1);

    // ---- S^T = K Q^T and Sd^T = K Qd^T + Kd Q^T for the KT key tiles of the block
    f32x4 s[QT][KT], sd[QT][KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) { s[qt][kt] = f32x4{0, 0, 0, 0}; sd[qt][kt] = f32x4{0, 0, 0, 0}; }
#pragma unroll
      for (int g = 0; g < CT; ++g) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(Ks + (16 * kt + il) * KP + 16 * g + 4 * q);
        const f32x4 ad = *reinterpret_cast<const f32x4*>(Kd + (16 * kt + il) * KP + 16 * g + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int qt = 0; qt < QT; ++qt) {
            s[qt][kt] = mfma16t(a[r], qf[qt][g][r], s[qt][kt]);
            sd[qt][kt] = mfma16t(a[r], qd[qt][g][r], sd[qt][kt]);
            sd[qt][kt] = mfma16t(ad[r], qf[qt][g][r], sd[qt][kt]);
          }
      }
    }
    // ---- online softmax on dual numbers: this lane's query, its 4*KT keys of the block (+ the other key groups by shuffle)
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      float mb = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s[qt][kt][r] *= scale; mb = fmaxf(mb, s[qt][kt][r]); }
      mb = fmaxf(mb, __shfl_xor(mb, 16, 64));
      mb = fmaxf(mb, __shfl_xor(mb, 32, 64));
      const float mn = fmaxf(m[qt], mb);
      const float alpha = expf(m[qt] - mn);
      m[qt] = mn;
      float ls = 0.f, lr = 0.f;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = expf(s[qt][kt][r] - mn);
          const float wv = p * (sd[qt][kt][r] * scale);
          s[qt][kt][r] = p; sd[qt][kt][r] = wv;
          ls += p; lr += wv;
        }
      l[qt] = l[qt] * alpha + ls;                          // per key group; the four groups meet at the end
      rr[qt] = rr[qt] * alpha + lr;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) { o[qt][ct] *= alpha; od[qt][ct] *= alpha; }
    }
    // ---- A1^T += V^T E^T ;  A2^T += V^T (E o Sd)^T + Vd^T E^T
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        float a[4], ad[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          a[r] = Vs[(16 * kt + 4 * q + r) * KP + 16 * ct + il];
          ad[r] = Vd[(16 * kt + 4 * q + r) * KP + 16 * ct + il];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int qt = 0; qt < QT; ++qt) {
            o[qt][ct] = mfma16t(a[r], s[qt][kt][r], o[qt][ct]);
            od[qt][ct] = mfma16t(a[r], sd[qt][kt][r], od[qt][ct]);
            od[qt][ct] = mfma16t(ad[r], s[qt][kt][r], od[qt][ct]);
          }
      }
  }
  // ---- normalise and store: lane (query il, q) holds channels 16ct + 4q + r
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    float lt = l[qt], rt = rr[qt];
    lt += __shfl_xor(lt, 16, 64); lt += __shfl_xor(lt, 32, 64);
    rt += __shfl_xor(rt, 16, 64); rt += __shfl_xor(rt, 32, 64);
    const float inv = 1.0f / lt, rb = rt * inv;
    const size_t row = (size_t)smp * T + q0 + 16 * qt + il;
    float* orow = att + row * C;
    float* drow = att + ((size_t)Bp * T + row) * C;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const f32x4 ov = o[qt][ct] * inv;
      *reinterpret_cast<f32x4*>(orow + 16 * ct + 4 * q) = ov;
      *reinterpret_cast<f32x4*>(drow + 16 * ct + 4 * q) = od[qt][ct] * inv - ov * rb;
    }
    if (q == 0) { lse[row] = m[qt] + logf(lt); rbar[row] = rb; }
  }
}

// =============================================================================================== backward: row scalars
// c_i = gd_i . o_i ;  delta_i = g_i . o_i + gd_i . od_i - rbar_i c_i       (16 lanes per row, float4 each per 64 ch.)
__global__ void __launch_bounds__(256) k_attn_dual_delta(const float* __restrict__ att, const float* __restrict__ datt,
                                                          const float* __restrict__ rbar, float* __restrict__ cc,
                                                          float* __restrict__ de, int64_t rows, int C) {
  const int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int sl = threadIdx.x & 15;
  float c = 0.f, d = 0.f;
  if (row < rows) {
    const float* o = att + row * C;
    const float* g = datt + row * C;
    const float* od = att + (rows + row) * C;
    const float* gd = datt + (rows + row) * C;
    for (int c4 = sl; c4 < C / 4; c4 += 16) {
      const f32x4 ov = *reinterpret_cast<const f32x4*>(o + 4 * c4), gv = *reinterpret_cast<const f32x4*>(g + 4 * c4);
      const f32x4 odv = *reinterpret_cast<const f32x4*>(od + 4 * c4), gdv = *reinterpret_cast<const f32x4*>(gd + 4 * c4);
#pragma unroll
      for (int r = 0; r < 4; ++r) { c += gdv[r] * ov[r]; d += gv[r] * ov[r] + gdv[r] * odv[r]; }
    }
  }
#pragma unroll
  for (int s = 8; s > 0; s >>= 1) { c += __shfl_xor(c, s, 16); d += __shfl_xor(d, s, 16); }
  if (row < rows && sl == 0) { cc[row] = c; de[row] = d - rbar[row] * c; }
}

// =============================================================================================== backward: main kernel
// Workgroup = 2*KG waves = KB = 16*KG keys of one sample; wave (qs = w&1, kg = w>>1) works on the 16-query sub-tile qs of the
// current 32-query block and on keys [16kg, 16kg+16).  Per 32-query block:
//   phase 1  tiles S, Sd, Pbar, D (24 MFMA per 16 channels), the elementwise adjoints, the key-side products (24 per 16
//            channels; the tile registers are their B operand), Sbar / Sdbar parked in LDS;
//   phase 2  query-side products from the parked tiles: the 2*CT (query sub-tile, 16-channel tile) pairs are dealt to the
//            waves (wave (qs, kg) takes channel tiles kg, kg + KG, ...) -> one slab row per (key block, query).
// KG = 4 (64 keys, 8 waves) for C <= 64; KG = 2 (32 keys, 4 waves) for C = 128, where the four resident key-side tiles of
// 64 keys alone would be 135 KB of LDS (the T = 256 blocks of the C4 network, model/unet.py:236-250 at (ch, T) = (128, 256)).
// Key-block SEQUENCING (the query-gradient slabs): at T = 1024 a sample has 16 key blocks, i.e. 16 slabs of [2Bp][T][C] that
// k_attn_dq_reduce read back (3.4 ms of the C4 step, HBM-bound).  The launcher may run the kernel kseq times over every
// kseq-th key block (launch ks takes key blocks kgi kseq + ks): launches ks > 0 (ACC) ADD their partial sums to the slab
// rows of their group instead of writing rows of their own — one 16-byte load per store, requested before the MFMAs whose
// result it joins — so the reduction reads nkg = T / (KB kseq) slabs.  Same fixed summation order every run.  (An in-kernel
// loop over the key blocks of a group was tried first: it needs ~37 more registers than the 246 the kernel has, spilled, and
// ran the whole step 1 ms SLOWER at kseq = 1; it also exposed a hipcc hazard bug — no wait states between an MFMA and a global
// store of its result when a branch sits in between — which a compile-time ACC cannot hit.)
template <int CT, int KG, bool ACC = false>
__global__ void __launch_bounds__(128 * KG) k_attn_dual_bwd(const float* __restrict__ qkv, const float* __restrict__ datt,
                                                        const float* __restrict__ stats /* [2][Bp*T]: lse | rbar */,
                                                        const float* __restrict__ ccde /* [2][Bp*T]: c | delta */,
                                                        float* __restrict__ dqkv, float* __restrict__ slab, int T, int nkg,
                                                        int kseq, int ks, int64_t Bp, float scale) {
  constexpr int C = 16 * CT, KP = C + 4, LD = 3 * C, KB = 16 * KG, DP = KB + 4, QB = 32, NT = 128 * KG;
  constexpr int NK = (KB * C / 4) / NT;                   // float4 per thread per resident matrix
  constexpr int NQ = (QB * C / 4 + NT - 1) / NT;          // float4 per thread per streamed matrix and query block
  constexpr bool QFULL = (QB * C / 4) % NT == 0;          // C = 32: only half of the threads stage a float4
  static_assert(NK >= 1 && (KB * C / 4) % NT == 0, "resident tiles do not divide over the workgroup");
  extern __shared__ __attribute__((aligned(16))) float atb_lds[];
  float* Ks = atb_lds;                                    // [KB][KP] x4, resident
  float* Kd = Ks + KB * KP;
  float* Vs = Kd + KB * KP;
  float* Vd = Vs + KB * KP;
  float* Qs = Vd + KB * KP;                               // [32][KP] x4, per query block
  float* Qd = Qs + QB * KP;
  float* Gs = Qd + QB * KP;
  float* Gd = Gs + QB * KP;
  float* St = Gd + QB * KP;                               // [4][32] row scalars
  float* dS0 = St + 4 * QB;                               // 2 x ([32][DP] Sbar | [32][DP] Sdbar): double-buffered, so the
                                                          // next block's phase 1 may start while a slow wave is still in phase 2
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, il = lane & 15, q = lane >> 4;
  const int qs = w & 1, kg = w >> 1;
  int smp, kgi;                                            // XCD-aware order (see the forward kernel): the key groups
  {                                                        // of one sample stream the same q / g rows
    const int full = (int)(Bp / 8) * 8 * nkg, b = blockIdx.x;
    if (b < full) { const int loc = b >> 3; smp = (loc / nkg) * 8 + (b & 7); kgi = loc % nkg; }
    else { const int r = b - full; smp = (int)(Bp / 8) * 8 + r / nkg; kgi = r % nkg; }
  }
  const int kb = kgi * kseq + ks;
  const size_t half_qkv = (size_t)Bp * T * LD, half_att = (size_t)Bp * T * C;
  const float* bp = qkv + (size_t)smp * T * LD;
  const float* gp = datt + (size_t)smp * T * C;
  const size_t srow0 = (size_t)smp * T;

  // ---- resident K, Kd, V, Vd tiles of this key block
#pragma unroll
  for (int i = 0; i < NK; ++i) {
    const int idx = tid + NT * i, key = idx / (C / 4), c4 = idx - key * (C / 4);
    const size_t off = (size_t)(kb * KB + key) * LD + 4 * c4;
    *reinterpret_cast<f32x4*>(Ks + key * KP + 4 * c4) = *reinterpret_cast<const f32x4*>(bp + off + C);
    *reinterpret_cast<f32x4*>(Vs + key * KP + 4 * c4) = *reinterpret_cast<const f32x4*>(bp + off + 2 * C);
    *reinterpret_cast<f32x4*>(Kd + key * KP + 4 * c4) = *reinterpret_cast<const f32x4*>(bp + half_qkv + off + C);
    *reinterpret_cast<f32x4*>(Vd + key * KP + 4 * c4) = *reinterpret_cast<const f32x4*>(bp + half_qkv + off + 2 * C);
  }

  f32x4 dk[CT], dkd[CT], dv[CT], dvd[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) dk[ct] = dkd[ct] = dv[ct] = dvd[ct] = f32x4{0, 0, 0, 0};

  f32x4 pq[NQ], pqd[NQ], pg[NQ], pgd[NQ];
  float pst = 0.f;
  auto gload = [&](int qb) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int idx = tid + NT * i, qr = idx / (C / 4), c4 = idx - qr * (C / 4);
      if (!QFULL && idx >= QB * C / 4) break;
      const size_t row = (size_t)(qb * QB + qr);
      pq[i] = *reinterpret_cast<const f32x4*>(bp + row * LD + 4 * c4);
      pqd[i] = *reinterpret_cast<const f32x4*>(bp + half_qkv + row * LD + 4 * c4);
      pg[i] = *reinterpret_cast<const f32x4*>(gp + row * C + 4 * c4);
      pgd[i] = *reinterpret_cast<const f32x4*>(gp + half_att + row * C + 4 * c4);
    }
    if (tid < 4 * QB) {
      const int which = tid >> 5, qr = tid & 31;
      const size_t o = (size_t)(which & 1) * (size_t)Bp * T + srow0 + (size_t)qb * QB + qr;
      pst = which < 2 ? stats[o] : ccde[o];
    }
  };
  gload(0);
  const int nqb = T / QB;
  for (int qb = 0; qb < nqb; ++qb) {
    float* dS = dS0 + (qb & 1) * (2 * QB * DP);
    float* dSd = dS + QB * DP;
    // no barrier here: every wave has passed [C] of the previous block, i.e. finished READING Q / Qd / G / Gd and the row
    // scalars; phase 2 of the previous block only reads its own dS buffer and the resident K tiles
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int idx = tid + NT * i, qr = idx / (C / 4), c4 = idx - qr * (C / 4);
      if (!QFULL && idx >= QB * C / 4) break;
      *reinterpret_cast<f32x4*>(Qs + qr * KP + 4 * c4) = pq[i];
      *reinterpret_cast<f32x4*>(Qd + qr * KP + 4 * c4) = pqd[i];
      *reinterpret_cast<f32x4*>(Gs + qr * KP + 4 * c4) = pg[i];
      *reinterpret_cast<f32x4*>(Gd + qr * KP + 4 * c4) = pgd[i];
    }
    if (tid < 4 * QB) St[tid] = pst;
    __syncthreads();                                       // [B] staged block visible
    if (qb + 1 < nqb) gload(qb + 1);

    // ---- phase 1a: S, Sd, Pbar, D for (queries 16qs.., keys 16kg..); D layout: lane (key il, queries 4q+r)
    f32x4 s = {0, 0, 0, 0}, sd = {0, 0, 0, 0}, pb = {0, 0, 0, 0}, dd = {0, 0, 0, 0};
#pragma unroll
    for (int g = 0; g < CT; ++g) {
      const int ao = (16 * qs + il) * KP + 16 * g + 4 * q, bo = (16 * kg + il) * KP + 16 * g + 4 * q;
      const f32x4 aQ = *reinterpret_cast<const f32x4*>(Qs + ao), aQd = *reinterpret_cast<const f32x4*>(Qd + ao);
      const f32x4 aG = *reinterpret_cast<const f32x4*>(Gs + ao), aGd = *reinterpret_cast<const f32x4*>(Gd + ao);
      const f32x4 bK = *reinterpret_cast<const f32x4*>(Ks + bo), bKd = *reinterpret_cast<const f32x4*>(Kd + bo);
      const f32x4 bV = *reinterpret_cast<const f32x4*>(Vs + bo), bVd = *reinterpret_cast<const f32x4*>(Vd + bo);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s = mfma16t(aQ[r], bK[r], s);
        sd = mfma16t(aQd[r], bK[r], sd);
        pb = mfma16t(aG[r], bV[r], pb);
        dd = mfma16t(aGd[r], bV[r], dd);
        sd = mfma16t(aQ[r], bKd[r], sd);
        pb = mfma16t(aGd[r], bVd[r], pb);
      }
      __builtin_amdgcn_sched_barrier(0);                   // keep one channel group of fragments in flight, not all
    }
    // ---- phase 1b: elementwise adjoints (row scalars of queries 16qs + 4q + r)
    const f32x4 L4 = *reinterpret_cast<const f32x4*>(St + 16 * qs + 4 * q);
    const f32x4 R4 = *reinterpret_cast<const f32x4*>(St + QB + 16 * qs + 4 * q);
    const f32x4 C4 = *reinterpret_cast<const f32x4*>(St + 2 * QB + 16 * qs + 4 * q);
    const f32x4 D4 = *reinterpret_cast<const f32x4*>(St + 3 * QB + 16 * qs + 4 * q);
    f32x4 p, pd, ds, dsd;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float sv = s[r] * scale, sdv = sd[r] * scale;
      const float pv = expf(sv - L4[r]), e = sdv - R4[r];
      p[r] = pv;
      pd[r] = pv * e;
      dsd[r] = pv * (dd[r] - C4[r]);
      ds[r] = pv * (pb[r] + dd[r] * e - C4[r] * sdv - D4[r]);
      dS[(16 * qs + 4 * q + r) * DP + 16 * kg + il] = ds[r];
      dSd[(16 * qs + 4 * q + r) * DP + 16 * kg + il] = dsd[r];
    }
    // ---- phase 1c: key-side products, contraction over the 16 queries of the sub-tile (B operand = tile registers)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      float aG[4], aGd[4], aQ[4], aQd[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = (16 * qs + 4 * q + r) * KP + 16 * ct + il;
        aG[r] = Gs[o]; aGd[r] = Gd[o]; aQ[r] = Qs[o]; aQd[r] = Qd[o];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        dv[ct] = mfma16t(aG[r], p[r], dv[ct]);
        dvd[ct] = mfma16t(aGd[r], p[r], dvd[ct]);
        dk[ct] = mfma16t(aQ[r], ds[r], dk[ct]);
        dkd[ct] = mfma16t(aQ[r], dsd[r], dkd[ct]);
        dv[ct] = mfma16t(aGd[r], pd[r], dv[ct]);
        dk[ct] = mfma16t(aQd[r], dsd[r], dk[ct]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();                                       // [C] Sbar / Sdbar of the whole 32 x KB block are parked
    // ---- phase 2: qbar^T[c][query] += K^T Sbar^T + Kd^T Sdbar^T ; qdbar^T += K^T Sdbar^T   (wave (qs, kg): tiles ct = kg, kg+KG, ..)
#pragma unroll
    for (int ct = kg; ct < CT; ct += KG) {
      const size_t row = (size_t)qb * QB + 16 * qs + il;
      float* sp = slab + (((size_t)smp * nkg + kgi) * T + row) * C + 16 * ct + 4 * q;
      float* sdp = slab + (((size_t)(Bp + smp) * nkg + kgi) * T + row) * C + 16 * ct + 4 * q;
      f32x4 dq = {0, 0, 0, 0}, dqd = {0, 0, 0, 0}, pdq = {0, 0, 0, 0}, pdqd = {0, 0, 0, 0};
      if (ACC) {          // compile-time: what the earlier launches left for this key group — requested BEFORE the MFMAs it joins
        pdq = *reinterpret_cast<const f32x4*>(sp);
        pdqd = *reinterpret_cast<const f32x4*>(sdp);
      }
#pragma unroll
      for (int kt = 0; kt < KG; ++kt) {
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(dS + (16 * qs + il) * DP + 16 * kt + 4 * q);
        const f32x4 b2 = *reinterpret_cast<const f32x4*>(dSd + (16 * qs + il) * DP + 16 * kt + 4 * q);
        float a1[4], a2[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          a1[r] = Ks[(16 * kt + 4 * q + r) * KP + 16 * ct + il];
          a2[r] = Kd[(16 * kt + 4 * q + r) * KP + 16 * ct + il];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dq = mfma16t(a1[r], b1[r], dq);
          dqd = mfma16t(a1[r], b2[r], dqd);
          dq = mfma16t(a2[r], b2[r], dq);
        }
      }
#ifdef ATT_EXP_NODQ           // diagnostic (WRONG results): what the query-gradient slab stores cost inside the loop
      if (dq[0] == 12345.678f) { *reinterpret_cast<f32x4*>(sp) = dq; *reinterpret_cast<f32x4*>(sdp) = dqd; }
#else
      if (ACC) { dq += pdq; dqd += pdqd; }
      *reinterpret_cast<f32x4*>(sp) = dq;
      *reinterpret_cast<f32x4*>(sdp) = dqd;
#endif
    }
  }
  // ---- the two query halves of each key group meet through LDS; scaled key / value gradients leave the chip
  __syncthreads();
  float* scratch = atb_lds;                                // [kg][4*CT][64 lanes] float4
  if (qs == 1) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      f32x4* base = reinterpret_cast<f32x4*>(scratch) + ((size_t)kg * 4 * CT + 4 * ct) * 64 + lane;
      base[0] = dk[ct]; base[64] = dkd[ct]; base[128] = dv[ct]; base[192] = dvd[ct];
    }
  }
  __syncthreads();
  if (qs == 0) {
    const size_t row = (size_t)smp * T + kb * KB + 16 * kg + il;     // key of this lane
    float* kp_ = dqkv + row * LD;
    float* kt_ = dqkv + half_qkv + row * LD;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const f32x4* base = reinterpret_cast<const f32x4*>(scratch) + ((size_t)kg * 4 * CT + 4 * ct) * 64 + lane;
      *reinterpret_cast<f32x4*>(kp_ + C + 16 * ct + 4 * q) = (dk[ct] + base[0]) * scale;
      *reinterpret_cast<f32x4*>(kt_ + C + 16 * ct + 4 * q) = (dkd[ct] + base[64]) * scale;
      *reinterpret_cast<f32x4*>(kp_ + 2 * C + 16 * ct + 4 * q) = dv[ct] + base[128];
      *reinterpret_cast<f32x4*>(kt_ + 2 * C + 16 * ct + 4 * q) = dvd[ct] + base[192];
    }
  }
}

// qbar / qdbar = s2 * sum over key blocks of the slabs, in block order (deterministic); written into the q slice of dqkv
__global__ void __launch_bounds__(256) k_attn_dq_reduce(const float* __restrict__ slab, float* __restrict__ dqkv, int64_t N,
                                                         int T, int C, int nkb, float scale) {
  const int64_t total = N * T * (C / 4);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(i % (C / 4));
    const int64_t row = i / (C / 4);                       // n*T + query
    const int64_t n = row / T, qr = row - n * T;
    const float* p = slab + ((size_t)n * nkb * T + qr) * C + 4 * c4;
    f32x4 acc = *reinterpret_cast<const f32x4*>(p);
    for (int kb = 1; kb < nkb; ++kb) acc += *reinterpret_cast<const f32x4*>(p + (size_t)kb * T * C);
    *reinterpret_cast<f32x4*>(dqkv + (size_t)row * 3 * C + 4 * c4) = acc * scale;
  }
}

// =============================================================================================== launchers
template <int CT, int QT, int KB, int NW>
static int launch_fwd(const float* qkv, float* att, float* stats, int64_t Bp, int T, float scale, hipStream_t st) {
  constexpr int C = 16 * CT;
  constexpr size_t lds = (size_t)4 * KB * (C + 4) * sizeof(float);
  static const int once = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attn_dual_fwd<CT, QT, KB, NW>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    return 0;
  }();
  (void)once;
  const int nqb = T / (16 * NW * QT);
  hipLaunchKernelGGL((k_attn_dual_fwd<CT, QT, KB, NW>), dim3((unsigned)(Bp * nqb)), dim3(64 * NW), lds, st, qkv, att, stats,
                     stats + Bp * T, T, nqb, Bp, scale);
  return msgm_check_launch();
}

// keys per backward workgroup
static inline int attn_bwd_keys(int C) { return C == 128 ? 32 : 64; }

template <int CT, int KG>
static int launch_bwd(const float* qkv, const float* att, const float* datt, const float* stats, float* dqkv, int64_t Bp, int T,
                      float scale, float* ws, hipStream_t st) {
  constexpr int C = 16 * CT, KP = C + 4, KB = 16 * KG;
  constexpr size_t lds_main = ((size_t)4 * KB * KP + 4 * 32 * KP + 4 * 32 + 4 * 32 * (KB + 4)) * sizeof(float);
  constexpr size_t lds_epi = (size_t)KG * 4 * CT * 64 * 4 * sizeof(float);        // the cross-wave sum of the key-side gradients
  constexpr size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
  static_assert(lds <= 160 * 1024, "backward tiles exceed the CU's LDS");
  static const int once = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attn_dual_bwd<CT, KG, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attn_dual_bwd<CT, KG, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    return 0;
  }();
  (void)once;
  const int64_t rows = Bp * T;
  const int nkb = T / KB;
  // launches over the key blocks: two while each launch still has two full rounds of resident workgroups (two per CU).
  // Measured at B = 256 (rocprofv3, 7 steps): one launch 147.2 + 29.1 ms in the two backward kernels + 24.0 ms of reduce;
  // two launches 154.1 + 32.4 + 10.5; four 157.5 + 34.4 + 5.3 — the ACC launches run 7 % (C = 64) to 20 % (C = 128) longer
  // than the plain ones (their slab reads are not free under the MFMAs), so two and four both end 0.5 ms per step ahead.
  static const int kseq_x = getenv("MSGM_ATTN_KSEQ") ? atoi(getenv("MSGM_ATTN_KSEQ")) : 0;      // diagnostic override
  int kseq = 1;
  for (int k = 2; k > 1; k >>= 1)
    if (nkb % k == 0 && Bp * (int64_t)(nkb / k) >= 1024) { kseq = k; break; }
  if (kseq_x > 0 && nkb % kseq_x == 0) kseq = kseq_x;
  const int nkg = nkb / kseq;
  float* cc = ws;
  float* de = ws + rows;
  float* slab = ws + 2 * rows;
  hipLaunchKernelGGL(k_attn_dual_delta, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, st, att, datt, stats + rows, cc, de,
                     rows, C);
  hipLaunchKernelGGL((k_attn_dual_bwd<CT, KG, false>), dim3((unsigned)(Bp * nkg)), dim3(128 * KG), lds, st, qkv, datt, stats, cc, dqkv, slab,
                     T, nkg, kseq, 0, Bp, scale);
  for (int ks = 1; ks < kseq; ++ks)
    hipLaunchKernelGGL((k_attn_dual_bwd<CT, KG, true>), dim3((unsigned)(Bp * nkg)), dim3(128 * KG), lds, st, qkv, datt, stats, cc, dqkv, slab,
                       T, nkg, kseq, ks, Bp, scale);
  const int64_t work = 2 * Bp * T * (C / 4);
  hipLaunchKernelGGL(k_attn_dq_reduce, dim3((unsigned)grid_for(work, 256, 16384)), dim3(256), 0, st, slab, dqkv, 2 * Bp, T, C,
                     nkg, scale);
  return msgm_check_launch();
}

extern "C" {

// C in {32, 64}: T a multiple of 64; C = 128 (the 16x16 / 8x8 attention of the 2-D U-Net): T a multiple of 32
int msgm_attention_dual_supported(int32_t T, int32_t C) {
  if (C == 128) return T >= 32 && T % 32 == 0;
  return (C == 32 || C == 64) && T >= 64 && T % 64 == 0;
}

size_t msgm_attention_dual_workspace(int64_t Bp, int32_t T, int32_t C) {
  if (!msgm_attention_dual_supported(T, C) || Bp <= 0) return 0;
  return ((size_t)2 * Bp * T + (size_t)2 * Bp * (T / attn_bwd_keys(C)) * T * C) * sizeof(float);
}

int msgm_attention_dual_forward(const float* qkv, float* att, float* stats, int64_t Bp, int32_t T, int32_t C, float scale,
                                msgm_stream_t stream) {
  if (!qkv || !att || !stats || Bp <= 0 || T <= 0 || C <= 0) return MSGM_E_BADARG;
  if (!msgm_attention_dual_supported(T, C) || Bp * (int64_t)(T / 32) > 0x7fffffffLL) return MSGM_E_UNSUPPORTED;
  if (C == 128) {
    // 64-query workgroups (4 waves) while they fill the chip twice over, else 32-query workgroups (2 waves): at the 32-row
    // shard the T = 256 blocks are only 128 workgroups of 64 queries
    if (T % 64 == 0 && Bp * (int64_t)(T / 64) >= 512) return launch_fwd<8, 1, 16, 4>(qkv, att, stats, Bp, T, scale, S(stream));
    return launch_fwd<8, 1, 16, 2>(qkv, att, stats, Bp, T, scale, S(stream));
  }
  static const bool qt2 = getenv("MSGM_ATTN_DUAL_QT2") != nullptr;   // diagnostic A/B
  const bool two = qt2 && T % 128 == 0;
  if (C == 32) return launch_fwd<2, 1, 64, 4>(qkv, att, stats, Bp, T, scale, S(stream));
  return two ? launch_fwd<4, 2, 32, 4>(qkv, att, stats, Bp, T, scale, S(stream)) : launch_fwd<4, 1, 32, 4>(qkv, att, stats, Bp, T, scale, S(stream));
}

int msgm_attention_dual_backward(const float* qkv, const float* att, const float* datt, const float* stats, float* dqkv,
                                 int64_t Bp, int32_t T, int32_t C, float scale, void* workspace, size_t workspace_bytes,
                                 msgm_stream_t stream) {
  if (!qkv || !att || !datt || !stats || !dqkv || !workspace || Bp <= 0 || T <= 0 || C <= 0) return MSGM_E_BADARG;
  if (!msgm_attention_dual_supported(T, C) || Bp * (int64_t)(T / 32) > 0x7fffffffLL) return MSGM_E_UNSUPPORTED;
  if (workspace_bytes < msgm_attention_dual_workspace(Bp, T, C)) return MSGM_E_WORKSPACE;
  float* ws = static_cast<float*>(workspace);
  if (C == 128) return launch_bwd<8, 2>(qkv, att, datt, stats, dqkv, Bp, T, scale, ws, S(stream));
  if (C == 32) return launch_bwd<2, 4>(qkv, att, datt, stats, dqkv, Bp, T, scale, ws, S(stream));
  return launch_bwd<4, 4>(qkv, att, datt, stats, dqkv, Bp, T, scale, ws, S(stream));
}

}  // extern "C"
