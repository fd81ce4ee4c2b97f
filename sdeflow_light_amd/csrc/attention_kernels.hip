// attention_kernels.hip — K8b: fused single-head self-attention forward for the SAMPLER path
// (QKVAttention, model/unet.py:236-250: softmax((q ch^-1/4)(k ch^-1/4)^T) v, fp32) without materialising the
// (B,T,T) probabilities.  The training path keeps the three-kernel form (bmm / dual softmax / bmm) because its
// backward needs P and the tangent logits; here there is no tangent and nothing to keep.
//
// Layout: qkv [N][T][3C] channels-last (q | k | v channel slices, as msgm_conv_forward writes them),
// out [N][T][C].  One workgroup = 4 waves = 64*QT queries of one sample; keys/values stream through LDS in blocks
// of 64 (register-prefetched one block ahead).  Everything is computed TRANSPOSED so that the query sits on lane&15:
//   S^T[key][query] = K[key][:] . Q[query][:]      A = K fragment (LDS, b128), B = Q fragment (registers, resident)
//   O^T[c][query]  += V^T[c][key] . P^T[key][query]  A = V fragment (LDS, b32),  B = the S^T accumulator itself
// In the C/D layout of v_mfma_f32_16x16x4_f32 a lane holds 4 consecutive keys of ITS query, so the running max / sum
// of the online softmax are per-lane scalars (+ two xor-shuffles across the four key groups), the rescale of O is a
// per-lane multiply, and exp(S - m) is directly the B operand of the second product (same k-permutation trick as
// the MLP kernel: MFMA step r of key tile kt contracts key 16kt + 4q + r on both operands).
#include "common.h"

__device__ __forceinline__ f32x4 mfma16a(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

template <int CT, int QT>   // C = 16*CT channels; QT tiles of 16 queries per wave
__global__ void __launch_bounds__(256) k_attn_fwd(const float* __restrict__ qkv, float* __restrict__ out, int T, int nqb, float scale) {
  constexpr int C = 16 * CT, KP = C + 4, LD = 3 * C;
  constexpr int NV = (64 * C / 4) / 256;                  // float4 per thread per matrix and key block
  extern __shared__ __attribute__((aligned(16))) float at_lds[];
  float* Ks = at_lds;
  float* Vs = at_lds + 64 * KP;
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, il = lane & 15, q = lane >> 4;
  const int smp = blockIdx.x / nqb, qb = blockIdx.x - smp * nqb;   // sample, query block of 64*QT
  const float* base = qkv + (size_t)smp * T * LD;
  const int q0 = (qb * 4 + w) * 16 * QT;                  // first query of this wave

  f32x4 qf[QT][CT], o[QT][CT];
  float m[QT], l[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
#pragma unroll
    for (int g = 0; g < CT; ++g) {
      qf[qt][g] = *reinterpret_cast<const f32x4*>(base + (size_t)(q0 + 16 * qt + il) * LD + 16 * g + 4 * q);
      o[qt][g] = f32x4{0, 0, 0, 0};
    }
    m[qt] = -INFINITY; l[qt] = 0.f;
  }

  f32x4 pk[NV], pv[NV];
  auto gload = [&](int kb) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + 256 * i, key = idx / (C / 4), c4 = idx - key * (C / 4);
      const float* p = base + (size_t)(kb * 64 + key) * LD + 4 * c4;
      pk[i] = *reinterpret_cast<const f32x4*>(p + C);
      pv[i] = *reinterpret_cast<const f32x4*>(p + 2 * C);
    }
  };
  gload(0);
  const int nkb = T / 64;
  for (int kb = 0; kb < nkb; ++kb) {
    __syncthreads();                                       // the previous block's readers are done
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + 256 * i, key = idx / (C / 4), c4 = idx - key * (C / 4);
      *reinterpret_cast<f32x4*>(Ks + key * KP + 4 * c4) = pk[i];
      *reinterpret_cast<f32x4*>(Vs + key * KP + 4 * c4) = pv[i];
    }
    __syncthreads();
    if (kb + 1 < nkb) gload(kb + 1);

    // ---- S^T = K Q^T for the 4 key tiles of the block
    f32x4 s[QT][4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) s[qt][kt] = f32x4{0, 0, 0, 0};
#pragma unroll
      for (int g = 0; g < CT; ++g) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(Ks + (16 * kt + il) * KP + 16 * g + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int qt = 0; qt < QT; ++qt) s[qt][kt] = mfma16a(a[r], qf[qt][g][r], s[qt][kt]);
      }
    }
    // ---- online softmax: this lane's query, its 16 keys of the block (+ the other three key groups by shuffle)
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      float mb = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s[qt][kt][r] *= scale; mb = fmaxf(mb, s[qt][kt][r]); }
      mb = fmaxf(mb, __shfl_xor(mb, 16, 64));
      mb = fmaxf(mb, __shfl_xor(mb, 32, 64));
      const float mn = fmaxf(m[qt], mb);
      const float alpha = __expf(m[qt] - mn);
      m[qt] = mn;
      float ls = 0.f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float p = __expf(s[qt][kt][r] - mn); s[qt][kt][r] = p; ls += p; }
      l[qt] = l[qt] * alpha + ls;                          // per key group; the four groups meet at the end
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) o[qt][ct] *= alpha;
    }
    // ---- O^T += V^T P^T
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        float a[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] = Vs[(16 * kt + 4 * q + r) * KP + 16 * ct + il];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int qt = 0; qt < QT; ++qt) o[qt][ct] = mfma16a(a[r], s[qt][kt][r], o[qt][ct]);
      }
  }
  // ---- normalise and store: lane (query il, q) holds channels 16ct + 4q + r
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    float lt = l[qt];
    lt += __shfl_xor(lt, 16, 64);
    lt += __shfl_xor(lt, 32, 64);
    const float inv = 1.0f / lt;
    float* orow = out + ((size_t)smp * T + q0 + 16 * qt + il) * C;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) *reinterpret_cast<f32x4*>(orow + 16 * ct + 4 * q) = o[qt][ct] * inv;
  }
}

static inline hipStream_t S(msgm_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

template <int CT, int QT>
static int launch_attn(const float* qkv, float* out, int64_t N, int T, float scale, hipStream_t st) {
  constexpr int C = 16 * CT;
  constexpr size_t lds = (size_t)2 * 64 * (C + 4) * sizeof(float);
  static const int once = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attn_fwd<CT, QT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    return 0;
  }();
  (void)once;
  const int nqb = T / (64 * QT);
  hipLaunchKernelGGL((k_attn_fwd<CT, QT>), dim3((unsigned)(N * nqb)), dim3(256), lds, st, qkv, out, T, nqb, scale);
  return msgm_check_launch();
}

extern "C" {

int msgm_attention_supported(int32_t T, int32_t C) {
  return (C == 32 || C == 64 || C == 128) && T >= 64 && T % 64 == 0;
}

int msgm_attention_forward(const float* qkv, float* out, int64_t N, int32_t T, int32_t C, float scale, msgm_stream_t stream) {
  if (!qkv || !out || N <= 0 || T <= 0 || C <= 0) return MSGM_E_BADARG;
  if (!msgm_attention_supported(T, C) || N * (int64_t)(T / 64) > 0x7fffffffLL) return MSGM_E_UNSUPPORTED;
  static const bool force1 = getenv("MSGM_ATTN_QT1") != nullptr;   // diagnostic A/B
  // 128 queries per workgroup when T allows — except at C = 128, where the second query tile costs the second
  // resident wave per SIMD (357 registers): 89 vs 102 TFLOP/s at T = 256
  const bool two = T % 128 == 0 && C < 128 && !force1;
  if (C == 32) return two ? launch_attn<2, 2>(qkv, out, N, T, scale, S(stream)) : launch_attn<2, 1>(qkv, out, N, T, scale, S(stream));
  if (C == 64) return two ? launch_attn<4, 2>(qkv, out, N, T, scale, S(stream)) : launch_attn<4, 1>(qkv, out, N, T, scale, S(stream));
  return two ? launch_attn<8, 2>(qkv, out, N, T, scale, S(stream)) : launch_attn<8, 1>(qkv, out, N, T, scale, S(stream));
}

}  // extern "C"
