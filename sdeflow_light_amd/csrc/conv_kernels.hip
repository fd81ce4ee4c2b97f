// conv_kernels.hip — K6/K11: convolutions of the U-Net score nets as implicit
// GEMMs on fp32 MFMA (v_mfma_f32_16x16x4_f32), channels-last activations.
//
//   forward / dgrad : out[m][co] = sum_tap sum_c  in[src(m,tap)][c] * Wp[tap][co][c]
//   wgrad           : dWp[tap][co][c] += sum_m gy[m][co] * in[src(m,tap)][c]
//
// m runs over output positions (n, oh, ow) (1-D: H = 1).  src() is either the
// gather of a strided convolution (mode 0: i = o*s + k - p) or of a transposed
// one (mode 1: i = (o + p - k)/s when divisible) — which also is the dgrad of the
// other.  Optional nearest-2x upsampling of the input is folded into the gather
// (model/unet.py:60-73), and up to two inputs are concatenated along channels
// without materialising the concat (NNUnet1D.py:175, model/unet.py:514).
// Weight is the MFMA A operand, activations the B operand; a lane loads 4
// consecutive channels (16 B) of one position, so the C/D layout writes 4
// consecutive output channels of one position: everything stays channels-last.
// No LDS: fragments stream L2 -> registers through a 3-deep software pipeline;
// occupancy (<= 128 VGPR) hides the rest.  The forward-mode tangent is simply the
// second half of the batch (n >= n_bias): convolutions are linear, only the bias
// distinguishes the halves.
#include "common.h"
#include <type_traits>
#include <stdlib.h>

#define CONV_MAX_SRC 2

__device__ __forceinline__ f32x4 mfma16c(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

struct ConvGeom {
  int N, Hi, Wi, Ho, Wo;      // Hi/Wi: stored input size (before the optional 2x upsample)
  int KH, KW, strideH, padH, strideW, padW;
  int mode;                   // 0 conv gather, 1 transposed gather
  int ups;                    // 1: input is nearest-upsampled 2x on the fly
};

struct ConvArgs {
  ConvGeom g;
  const float* src[CONV_MAX_SRC];
  int C[CONV_MAX_SRC];        // channels of each source
  int koff[CONV_MAX_SRC];     // offset of each source in the packed K axis (multiple of 16)
  int nsrc;
  const float* Wp;            // [taps][CoutP][Ktot], zero padded
  int Cout, CoutP, Ktot;
  const float* bias;          // [Cout] or null          (rows n < n_bias only)
  const float* samp_bias;     // [n_samp][Cout] or null   (rows n < n_samp only)
  int n_bias, n_samp;
  float* out;                 // [N][Ho][Wo][Cout]
  int accumulate;             // out += result (fused residual / skip add)
  const float* residual;      // [N][Ho][Wo][Cout] or null: added in the epilogue (ResBlock / attention skip path)
  // optional transform of the INPUT while it is staged (halo-tile kernel only): the conv reads act(a x + b) with
  // per-(sample, input channel) a, b — GroupNorm (+ SiLU) folded into the consuming convolution
  const float* in_scale;      // [N][Ctot] or null, Ctot = sum of the sources' channels
  const float* in_shift;      // [N][Ctot]
  int in_act;                 // 0 none, 1 SiLU
  // structurally-zero weight blocks (halo-tile kernel): bit t of a mask = tap t is present; 0 = all taps.
  // tapmask_in[i]: per 32-channel input chunk (flattened over the sources), tapmask_out[y]: per output-channel block
  unsigned short tapmask_in[16];
  unsigned short tapmask_out[8];
  // optional by-product (k_conv_tile, k_conv1x1): per-channel partial sums of the FINAL output values (after bias /
  // accumulate / residual) for the GroupNorm that reads this output next — cstat[n][slot < cs_S][{sum, sum of squares}][Cout],
  // one slot per (tile, wave) of a sample; the consumer adds the slots in slot order (k_gn_affine_cs)
  float* cstat;
  int cs_S;
};

// Sum over the 16 lanes of a DPP row (the 16 pixel lanes of one channel quad), fixed butterfly: every lane ends with it.
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));  // row_mirror
  return v;
}
// A wave's (sum, sum of squares) of 4 consecutive output channels over its pixels -> slot `slot` of sample n.
__device__ __forceinline__ void cstat_store(const ConvArgs& A, int n, int slot, int co, f32x4 s, f32x4 ss, int il) {
#pragma unroll
  for (int r = 0; r < 4; ++r) { s[r] = row16_sum(s[r]); ss[r] = row16_sum(ss[r]); }
  if (il == 0) {
    float* cp = A.cstat + (((size_t)n * A.cs_S + slot) * 2) * A.Cout + co;
    *reinterpret_cast<f32x4*>(cp) = s;
    *reinterpret_cast<f32x4*>(cp + A.Cout) = ss;
  }
}


// input coordinate of output coordinate o for tap k; returns false when the tap falls outside
__device__ __forceinline__ bool src_coord(const ConvGeom& g, int o, int k, int in_size_up, int stride, int pad, int& i) {
  int v;
  if (g.mode == 0) {
    v = o * stride + k - pad;
  } else {
    const int t = o + pad - k;
    if (t < 0) return false;
    if (stride == 2) { if (t & 1) return false; v = t >> 1; }
    else if (stride == 1) v = t;
    else { if (t % stride) return false; v = t / stride; }
  }
  if (v < 0 || v >= in_size_up) return false;
  i = g.ups ? (v >> 1) : v;
  return true;
}

// FAST: every source has C % 16 == 0 (no channel predicate, pure 16-B loads).  The inner loop is kept lean on
// purpose — with 32 MFMAs (1024 matrix-pipe cycles) per iteration the vector ALU must stay well below that:
// pointers advance by constants, the 3-deep fragment pipeline is unrolled by 3 (no register rotation), and all
// index arithmetic happens once per (tap, source) segment.
template <int MT, int NT, bool FAST>
__global__ void __launch_bounds__(256) k_conv_gemm(ConvArgs A) {
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, il = lane & 15, q = lane >> 4;
  const ConvGeom g = A.g;
  const int HoWo = g.Ho * g.Wo;
  const int Mtot = g.N * HoWo;
  const int m0 = (blockIdx.x * 4 + w) * (NT * 16);
  if (m0 >= Mtot) return;                       // no barriers in this kernel
  const int co0 = blockIdx.y * (MT * 16);
  const int Hup = g.ups ? 2 * g.Hi : g.Hi, Wup = g.ups ? 2 * g.Wi : g.Wi;

  int pn[NT], poh[NT], pow_[NT];
  bool pin[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    int m = m0 + 16 * nt + il;
    pin[nt] = m < Mtot;
    if (!pin[nt]) m = Mtot - 1;
    pn[nt] = m / HoWo;
    const int r = m - pn[nt] * HoWo;
    poh[nt] = r / g.Wo;
    pow_[nt] = r - poh[nt] * g.Wo;
  }
  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0, 0, 0, 0};

  const int taps = g.KH * g.KW;
  int groups[CONV_MAX_SRC];
  int per_tap = 0;
#pragma unroll
  for (int s = 0; s < CONV_MAX_SRC; ++s) { groups[s] = s < A.nsrc ? (A.C[s] + 15) >> 4 : 0; per_tap += groups[s]; }
  const int total = taps * per_tap;

  // ---- loader state (runs two iterations ahead of the MFMAs) ----------------
  int l_tap = 0, l_s = 0, g_left = 0, cb = 4 * q, Cseg = 0;
  bool done = total == 0;
  const float* ap = nullptr;
  const float* bptr[NT];
  bool bval[NT];
  const size_t a_mt_stride = (size_t)16 * A.Ktot;
  const float* a_lane = A.Wp + (size_t)(co0 + il) * A.Ktot + 4 * q;
  auto begin_segment = [&]() {
    const int kh = l_tap / g.KW, kw = l_tap - kh * g.KW;
    Cseg = A.C[l_s];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      int ih = 0, iw = 0;
      const bool ok = pin[nt] && src_coord(g, poh[nt], kh, Hup, g.strideH, g.padH, ih) &&
                      src_coord(g, pow_[nt], kw, Wup, g.strideW, g.padW, iw);
      bval[nt] = ok;
      bptr[nt] = A.src[l_s] + ((size_t)(pn[nt] * g.Hi + ih) * g.Wi + iw) * Cseg + 4 * q;
    }
    ap = a_lane + (size_t)l_tap * A.CoutP * A.Ktot + A.koff[l_s];
    g_left = groups[l_s];
    cb = 4 * q;
  };
  if (!done) begin_segment();
  auto load = [&](f32x4 (&fa)[MT], f32x4 (&fb)[NT]) {
    if (!done) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) fa[mt] = *reinterpret_cast<const f32x4*>(ap + mt * a_mt_stride);
      if (FAST) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) fb[nt] = bval[nt] ? *reinterpret_cast<const f32x4*>(bptr[nt]) : f32x4{0, 0, 0, 0};
      } else if ((Cseg & 3) == 0) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          fb[nt] = (bval[nt] && cb < Cseg) ? *reinterpret_cast<const f32x4*>(bptr[nt]) : f32x4{0, 0, 0, 0};
      } else {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          f32x4 v = {0, 0, 0, 0};
          if (bval[nt]) {
            if (cb + 0 < Cseg) v[0] = bptr[nt][0];
            if (cb + 1 < Cseg) v[1] = bptr[nt][1];
            if (cb + 2 < Cseg) v[2] = bptr[nt][2];
            if (cb + 3 < Cseg) v[3] = bptr[nt][3];
          }
          fb[nt] = v;
        }
      }
      ap += 16; cb += 16;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bptr[nt] += 16;
      if (--g_left == 0) {
        if (++l_s == A.nsrc) { l_s = 0; ++l_tap; }
        if (l_tap < taps) begin_segment(); else done = true;
      }
    } else {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) fa[mt] = f32x4{0, 0, 0, 0};
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) fb[nt] = f32x4{0, 0, 0, 0};
    }
  };
  auto mma = [&](const f32x4 (&fa)[MT], const f32x4 (&fb)[NT]) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma16c(fa[mt][r], fb[nt][r], acc[mt][nt]);
  };

  f32x4 a0[MT], b0[NT], a1[MT], b1[NT], a2[MT], b2[NT];
  load(a0, b0);
  load(a1, b1);
  for (int it = 0; it < total; it += 3) {       // iterations past `total` multiply zero fragments
    load(a2, b2); mma(a0, b0);
    load(a0, b0); mma(a1, b1);
    load(a1, b1); mma(a2, b2);
  }

  // ---- epilogue: lane (position il of tile nt, q) holds channels co0+16mt+4q+r
  f32x4 cs[MT], css[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) { cs[mt] = f32x4{0.f, 0.f, 0.f, 0.f}; css[mt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    if (!pin[nt]) continue;
    const int m = m0 + 16 * nt + il;
    const bool primal = pn[nt] < A.n_bias;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int co = co0 + 16 * mt + 4 * q;
      if (co >= A.Cout) continue;
      f32x4 v = acc[mt][nt];
      float* op = A.out + (size_t)m * A.Cout + co;
      const bool full = (co + 3 < A.Cout) && ((A.Cout & 3) == 0);
      if (primal && A.bias) {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (co + r < A.Cout) v[r] += A.bias[co + r];
      }
      if (A.samp_bias && pn[nt] < A.n_samp) {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (co + r < A.Cout) v[r] += A.samp_bias[(size_t)pn[nt] * A.Cout + co + r];
      }
      if (full) {
        if (A.accumulate) v += *reinterpret_cast<const f32x4*>(op);
        if (A.residual) v += *reinterpret_cast<const f32x4*>(A.residual + (op - A.out));
        *reinterpret_cast<f32x4*>(op) = v;
        cs[mt] += v; css[mt] += v * v;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (co + r < A.Cout) op[r] = (A.accumulate ? op[r] + v[r] : v[r]) + (A.residual ? A.residual[(op - A.out) + r] : 0.f);
      }
    }
  }
  if (A.cstat) {       // host: Ho*Wo % (16 NT) == 0 and Cout % 4 == 0, so the wave's pixels are one slot of one sample
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int co = co0 + 16 * mt + 4 * q;
      if (co < A.Cout) cstat_store(A, m0 / HoWo, (m0 % HoWo) / (16 * NT), co, cs[mt], css[mt], il);
    }
  }
}

// ------------------------------------------------------------------ stride-1 "same" convolutions: halo tile in LDS
// k_conv_gemm re-reads every input pixel once per tap from L2 (9x for 3x3) and, at small Cout, that fragment
// traffic — not the MFMA pipe — sets its speed.  For stride-1, pad=(K-1)/2 convolutions (the bulk of both U-Nets,
// forward AND dgrad) a workgroup instead stages a (TH+KH-1) x (TW+KW-1) halo tile of KC=32 channels in LDS once
// per chunk (coalesced 16-B rows, zero-filled outside the image) and serves all taps from it:
//   * 256 threads = 4 waves; the TH x TW = 128 output pixels are split 32 per wave (2 MFMA column tiles);
//   * the activation (B) fragments are ds_read_b128 from the tile at (ty+kh, tx+kw) — pitch 36 floats is
//     conflict-free for the 16 consecutive pixels of a lane group;
//   * the weight (A) fragments stream L2 -> registers one (tap, 16-channel group) ahead (all waves read the same
//     few KB, they stay in L1);
//   * chunks are double-buffered: the next chunk's global loads are in flight during the MFMAs of this one.
// flip = 1 evaluates the transposed stride-1 gather i = o + pad - k (the dgrad of the same convolution).
#define CT_KC 32
#define CT_P 36
// KS = kernel size (1 or 3; the 1-D variants have KH = 1): compile-time, so the halo index arithmetic of the staging
// (hp / HW per staged 16 B) is multiplications by constants instead of integer divisions.
// PT = 16-pixel tiles per wave (TH*TW = 64*PT pixels per workgroup), DB = two LDS buffers + register prefetch of the
// next item.  <PT=2, DB> is the small-image / 1x1 form; 3-tap kernels on images that have 256-pixel tiles run
// <PT=4, !DB>: twice the pixels per wave give each weight fragment 4 MFMAs instead of 2 and halve the halo overhead,
// and the bigger halo is single-buffered so that three workgroups still share a CU (their MFMAs cover the staging).
// 1x1 kernels stay at ~0.3 of the MFMA peak / 3 TB/s whatever the tiles per workgroup; two attempts at their memory
// pipeline measured no gain and were removed: two items in flight in registers, and the weight chunk staged through
// LDS with the activations (so that no vmcnt wait for a weight fragment also waits for the next item's loads) —
// the second costs the third resident workgroup (55 KB of LDS) and was 5-10 % slower.
// Also measured and removed: requesting the next item's halo two pairs before the end of the single-buffered 1-D
// NCO = 4 form (36 more registers, still two waves per SIMD) — 4-8 % slower (C3 106.6 -> 109 ms).
// B6 (opt-in experiment; sampler forward, wide 2-D form only): the MFMA work in bf16-SPLIT arithmetic — every fp32
// operand as three bf16 pieces (h, m, l: 24 mantissa bits), six v_mfma_f32_16x16x32_bf16 products (lh, hl, mm, mh, hm, hh;
// fp32 accumulate) per 32-channel chunk and tap instead of eight fp32 MFMAs: as accurate as the fp32 MFMA and 1.96x its
// LDS-fed register-tile rate (tools/probe_bf16x3.hip).  The staging splits each halo element once (after the folded
// GroupNorm + SiLU) into three bf16 planes of the pixel's LDS row (208 B per pixel: 3 x 64 B + 16 B pad, conflict-free
// 16-byte fragment reads), the weight image is pre-split ([3 planes][tap][CoutP][Ktot] bf16, msgm_b6_split_weights).
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
template <int TH, int TW, int NCO, int KS, int PT = 2, bool DB = true, bool B6 = false>
#ifndef CT_MINWG
#define CT_MINWG 1
#endif
#ifndef CT_WIDE_MINWG
#define CT_WIDE_MINWG 1
#endif
__global__ void __launch_bounds__(256, (PT == 4 && !DB && NCO == 4) ? CT_WIDE_MINWG : CT_MINWG) k_conv_tile(ConvArgs A, int flip, int tiles_x, int tiles_y, int tiles_per_wg, int n_tiles,
                                                             int n_cob, int n_tgrp) {
  extern __shared__ __attribute__((aligned(16))) float ct_lds[];
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, il = lane & 15, q = lane >> 4;
  const ConvGeom g = A.g;
  constexpr int KH = TH == 1 ? 1 : KS, KW = KS, taps = KH * KW;
  constexpr int HH = TH + KH - 1, HW = TW + KW - 1, halo = HH * HW;
  constexpr int PITCH = B6 ? 52 : CT_P;                   // floats per halo pixel in LDS
  static_assert(!B6 || (!DB && PT == 4 && KS == 3 && TH == 16), "the bf16-split form exists for the wide 2-D kernel only");
  float* cur = ct_lds;
  float* nxt = ct_lds + halo * PITCH;
  // XCD-aware 1-D grid: workgroup ids are dealt round-robin to the 8 XCDs (each with its own L2), so the n_cob
  // output-channel blocks that read the SAME input tiles are given ids 8 apart — same XCD, dispatched back to back —
  // and the re-reads of the input hit that L2 instead of HBM (matters for the 1x1 convolutions, which are HBM-bound).
  int cob, tgrp;
  if (n_cob > 0) {
    const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
    cob = loc % n_cob; tgrp = (loc / n_cob) * 8 + xcd;
  } else {                       // diagnostic (MSGM_CONV_NO_XCD): tile-fastest order, channel blocks far apart
    const int ntp = 8 * ((n_tgrp + 7) / 8);
    cob = blockIdx.x / ntp; tgrp = blockIdx.x - cob * ntp;
  }
  if (tgrp >= n_tgrp) return;
  const int co0 = cob * (NCO * 16);
  // A workgroup walks over tiles_per_wg consecutive spatial tiles; the pipeline unit is a (tile, channel chunk)
  // item, so the halo of the NEXT tile is in flight during the MFMAs of this one even when Cin fits one chunk.
  const int t_beg = tgrp * tiles_per_wg, t_end = min(t_beg + tiles_per_wg, n_tiles);
  if (t_beg >= t_end) return;
  static_assert(TH * TW == 64 * PT, "tile = 4 waves x PT MFMA column tiles");
  // this lane's PT output pixels inside a tile
  // pixel pt of this lane: p = 16 PT w + 16 pt + il.  For 16-wide tiles that is row PT w + pt (wave-uniform), column il;
  // for 1-row tiles row 0, column p.  Derived where used (no per-pt registers kept across the kernel).
  const int wu = __builtin_amdgcn_readfirstlane(w);
  auto pty_of = [&](int pt) __attribute__((always_inline)) { return TH == 1 ? 0 : PT * wu + pt; };
  auto ptx_of = [&](int pt) __attribute__((always_inline)) { return TH == 1 ? 16 * PT * wu + 16 * pt + il : il; };
  static_assert(TH == 1 || TW == 16, "2-D tiles are 16 pixels wide");
  f32x4 acc[NCO][PT];

  // chunk list: (source, channel offset), flattened
  int nch[CONV_MAX_SRC];
  int total_chunks = 0;
#pragma unroll
  for (int s = 0; s < CONV_MAX_SRC; ++s) { nch[s] = s < A.nsrc ? (A.C[s] + CT_KC - 1) / CT_KC : 0; total_chunks += nch[s]; }

  // ---- staging: halo pixel hp, 16-B column c4 (8 per pixel)
  constexpr int MAXST = (halo * (CT_KC / 4) + 255) / 256;
  constexpr int HALF = MAXST;          // (staging in two halves measured no register gain: the epilogue is the peak)
  f32x4 st[HALF];
  f32x4 st_ga, st_gb;                    // the staged item's input affine and which of its elements are real pixels
  unsigned st_valid = 0;
  long st_aoff = -1;
  constexpr int n_items = halo * (CT_KC / 4);
  const int padH = g.padH, padW = g.padW;
  const int up = g.ups ? 1 : 0;          // Upsample folded into the gather (model/unet.py:60-73)
  const int ctot_all = A.C[0] + (A.nsrc > 1 ? A.C[1] : 0);
  auto tile_origin = [&](int t, int& n, int& y0, int& x0) {
    const int tx_i = t % tiles_x; t /= tiles_x;
    const int ty_i = t % tiles_y;
    n = t / tiles_y; y0 = ty_i * TH; x0 = tx_i * TW;
  };
  auto stage_load = [&](f32x4* dst, int t, int s, int c0, int K0, int K1) __attribute__((always_inline)) {
    int n, y0, x0;
    tile_origin(t, n, y0, x0);
    const int C = A.C[s];
    const float* base = A.src[s] + (size_t)n * g.Hi * g.Wi * C;
    // folded GroupNorm(+SiLU): this thread's 4 channels are the same for every k (256 % 8 == 0), so a and b are
    // fetched once per staged chunk.  They are APPLIED in stage_store, after all the halo loads of the item are in
    // flight: applied here, inside the bounds check of each load, every load waited for the previous one's data (11
    // serialised round trips per item: +10 % on the 32-channel layers of the sampler, tools/bench_conv_epi.py).
    // (double-buffered form: only their offset is kept across the MFMAs of the previous item — 8 registers less there)
    const int cq = c0 + 4 * (tid & 7);
    st_aoff = (A.in_scale && cq < C) ? (long)n * ctot_all + (s ? A.C[0] : 0) + cq : -1;
    if (!DB) {
      st_ga = f32x4{1.f, 1.f, 1.f, 1.f}; st_gb = f32x4{0.f, 0.f, 0.f, 0.f};
      if (st_aoff >= 0) { st_ga = *reinterpret_cast<const f32x4*>(A.in_scale + st_aoff); st_gb = *reinterpret_cast<const f32x4*>(A.in_shift + st_aoff); }
    }
    unsigned valid = 0;
    // this thread's halo positions (row, column) are the same for every tile but are RE-derived per item (HW is a
    // compile-time constant: a multiply-high each): kept in registers they were MAXST live values at the kernel's
    // register peak (the epilogue) — the opaque copy of tid keeps the compiler from hoisting them back out of the loop
    int tv = tid;
    asm volatile("" : "+v"(tv));
#pragma unroll
    for (int k = K0; k < K1; ++k) {
      const int idx = tid + 256 * k;
      f32x4 v = {0, 0, 0, 0};
      if (idx < n_items) {
        const int c4 = idx & 7;
        const int hp = (tv + 256 * k) >> 3;
        const int hy = hp / HW, hx = hp - hy * HW;
        const int iy = y0 + hy - padH, ix = x0 + hx - padW;     // on the (2x nearest-upsampled, if ups) input grid
        const int c = c0 + 4 * c4;
#ifdef CT_EXP_NOSTAGE    // diagnostic: no halo loads / index arithmetic
        if (false) {
#else
        if (iy >= 0 && iy < (g.Hi << up) && ix >= 0 && ix < (g.Wi << up) && c < C) {
#endif
          v = *reinterpret_cast<const f32x4*>(base + ((size_t)(iy >> up) * g.Wi + (ix >> up)) * C + c);
          valid |= 1u << k;                                    // a real pixel: zero padding stays zero
        }
      }
      dst[k - K0] = v;
    }
    st_valid = valid;
  };
  // item after (t_, s_, c_); returns whether (t_, s_, c_) is the last chunk of its tile
  auto advance = [&](int t_, int s_, int c_, int& nt, int& ns_, int& nc_) {
    nt = t_; ns_ = s_; nc_ = c_ + 1;
    if (nc_ == (s_ ? nch[1] : nch[0])) { ns_ = s_ + 1; nc_ = 0; }   // selects, not a dynamically indexed array (scratch)
    const bool last = ns_ >= A.nsrc || (ns_ ? nch[1] : nch[0]) == 0;
    if (last) { nt = t_ + 1; ns_ = 0; nc_ = 0; }
    return last;
  };
  auto stage_store = [&](float* buf, int K0, int K1) __attribute__((always_inline)) {
    int tv = tid;
    asm volatile("" : "+v"(tv));                            // LDS addresses re-derived per item, not kept in registers
    if (A.in_scale) {
      if (DB) {
        st_ga = f32x4{1.f, 1.f, 1.f, 1.f}; st_gb = f32x4{0.f, 0.f, 0.f, 0.f};
        if (st_aoff >= 0) { st_ga = *reinterpret_cast<const f32x4*>(A.in_scale + st_aoff); st_gb = *reinterpret_cast<const f32x4*>(A.in_shift + st_aoff); }
      }
#pragma unroll
      for (int k = K0; k < K1; ++k) {
        if ((st_valid >> k) & 1u) {
          f32x4 v = st[k - K0] * st_ga + st_gb;
          if (A.in_act == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = v[r] * __builtin_amdgcn_rcpf(1.0f + __expf(-v[r]));
          }
          st[k - K0] = v;
        }
      }
    }
    if constexpr (B6) {
#pragma unroll
      for (int k = K0; k < K1; ++k) {
        const int idx = tv + 256 * k;
        if (idx < n_items) {
          bf16x4_t h4, m4, l4;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float x = st[k - K0][r];
            const __bf16 h = (__bf16)x;
            const float r1 = x - (float)h;
            const __bf16 m = (__bf16)r1;
            h4[r] = h; m4[r] = m; l4[r] = (__bf16)(r1 - (float)m);
          }
          float* pp = buf + (idx >> 3) * PITCH + 2 * (idx & 7);          // 4 channels = 8 bytes in each plane
          *reinterpret_cast<bf16x4_t*>(pp) = h4;
          *reinterpret_cast<bf16x4_t*>(pp + 16) = m4;
          *reinterpret_cast<bf16x4_t*>(pp + 32) = l4;
        }
      }
    } else {
#pragma unroll
    for (int k = K0; k < K1; ++k) {
      const int idx = tv + 256 * k;
      if (idx < n_items) *reinterpret_cast<f32x4*>(buf + (idx >> 3) * CT_P + 4 * (idx & 7)) = st[k - K0];
    }
    }
  };
  auto stage_all = [&](float* buf, int t, int s, int c0) __attribute__((always_inline)) {     // global -> registers -> LDS
    stage_load(st, t, s, c0, 0, MAXST); stage_store(buf, 0, MAXST);
  };

  const size_t a_co_stride = (size_t)16 * A.Ktot;
  f32x4 an[B6 ? 1 : NCO];                // the next (tap, group) pair's weight fragments
  bf16x8_t an6[B6 ? NCO : 1][3];         // B6: the next tap's fragments, three planes (the whole 32-channel chunk: K = 32)
  using WT = typename std::conditional<B6, __bf16, float>::type;       // weight pointers count ELEMENTS of the image
  const WT* Wimg = reinterpret_cast<const WT*>(A.Wp);
  const size_t plane_stride = (size_t)taps * A.CoutP * A.Ktot;
  auto fetch_w = [&](const WT* wp) {
    if constexpr (B6) {
#pragma unroll
      for (int c = 0; c < NCO; ++c)
#pragma unroll
        for (int p_ = 0; p_ < 3; ++p_) an6[c][p_] = *reinterpret_cast<const bf16x8_t*>(wp + c * a_co_stride + p_ * plane_stride);
    } else {
#pragma unroll
    for (int c = 0; c < NCO; ++c) an[c] = *reinterpret_cast<const f32x4*>(wp + c * a_co_stride);
    }
  };
  auto wptr = [&](int s_, int c_) { return Wimg + (size_t)(co0 + il) * A.Ktot + A.koff[s_] + c_ * CT_KC + (B6 ? 8 : 4) * q; };
  auto chunk_mask = [&](int s_, int c_) {
    unsigned tm = (1u << taps) - 1u;
    const int chunk_flat = (s_ ? nch[0] : 0) + c_;
    const unsigned mi = chunk_flat < 16 ? A.tapmask_in[chunk_flat] : 0u, mo = cob < 8 ? A.tapmask_out[cob] : 0u;
    if (mi) tm &= mi;
    if (mo) tm &= mo;
    return tm;
  };
  {
    const unsigned tm0 = chunk_mask(0, 0);
    if (tm0) fetch_w(wptr(0, 0) + (size_t)(__ffs(tm0) - 1) * A.CoutP * A.Ktot);
  }
  int tile = t_beg, cs = 0, cc = 0;      // current item: tile, source cs, chunk index cc within it
  stage_all(cur, tile, 0, 0);
  __syncthreads();
  for (;;) {
    // next item
    int ntile, ns, nc;
    const bool last_chunk = advance(tile, cs, cc, ntile, ns, nc);
    const bool more = ntile < t_end;
    if (DB && more) stage_load(st, ntile, ns, nc * CT_KC, 0, MAXST);
    if (cs == 0 && cc == 0) {
#pragma unroll
      for (int c = 0; c < NCO; ++c)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) acc[c][pt] = f32x4{0, 0, 0, 0};
    }
    // ---- MFMAs of this chunk: (tap, 16-channel group) pairs, weight fragments one pair ahead
    const int C = A.C[cs];
    const int c0 = cc * CT_KC;
    const int ngrp = B6 ? 1 : ((C - c0 >= CT_KC) ? 2 : ((C - c0 + 15) >> 4));       // B6: one K = 32 step per tap (host: C % 32 == 0)
    const WT* wbase = wptr(cs, cc);
    // taps whose weight block is structurally zero for this (input chunk, output block) are skipped (Stride2PairOp)
    const unsigned tmask = chunk_mask(cs, cc);
    auto next_tap = [&](int t) { const unsigned rem = tmask & ~((2u << t) - 1u); return rem ? __ffs(rem) - 1 : taps; };
    const int tap0 = tmask ? __ffs(tmask) - 1 : taps;
    const int npairs = __popc(tmask) * ngrp;
    // Weight fragments are fetched one (tap, group) pair ahead (one pair = 4*PT*NCO MFMAs); `an` already holds this
    // item's first pair — it was requested during the LAST pair of the previous item, so that no item starts by
    // waiting for an L2 round trip (for 1x1 kernels an item is only two pairs long).
    int ftap = tap0, fgrp = 1;             // next pair to fetch
    if (fgrp == ngrp) { fgrp = 0; ftap = next_tap(tap0); }
    auto fetch_next_item = [&]() {
      const unsigned ntm = chunk_mask(ns, nc);
      if (ntm) fetch_w(wptr(ns, nc) + (size_t)(__ffs(ntm) - 1) * A.CoutP * A.Ktot);
    };
    if (npairs == 0 && more) fetch_next_item();
    int tap = tap0, grp = 0;
    for (int pr = 0; pr < npairs; ++pr) {
      f32x4 a[B6 ? 1 : NCO];
      bf16x8_t a6[B6 ? NCO : 1][3];
      if constexpr (B6) {
#pragma unroll
        for (int c = 0; c < NCO; ++c)
#pragma unroll
          for (int p_ = 0; p_ < 3; ++p_) a6[c][p_] = an6[c][p_];
      } else {
#pragma unroll
      for (int c = 0; c < NCO; ++c) a[c] = an[c];
      }
      int ntap = tap, ngr = grp + 1;
      if (ngr == ngrp) { ngr = 0; ntap = next_tap(tap); }
      if (pr + 1 < npairs) {
        fetch_w(wbase + (size_t)ftap * A.CoutP * A.Ktot + 16 * fgrp);
        if (++fgrp == ngrp) { fgrp = 0; ftap = next_tap(ftap); }
      } else if (more) {
        fetch_next_item();
      }
      const int kh = tap / KW, kw = tap - kh * KW;
      const int oy = flip ? (KH - 1 - kh) : kh, ox = flip ? (KW - 1 - kw) : kw;
      if constexpr (B6) {
        bf16x8_t b6[PT][3];
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
#pragma unroll
          for (int p_ = 0; p_ < 3; ++p_)
            b6[pt][p_] = *reinterpret_cast<const bf16x8_t*>(cur + ((pty_of(pt) + oy) * HW + ptx_of(pt) + ox) * PITCH + 16 * p_ + 4 * q);
#pragma unroll
        for (int c = 0; c < NCO; ++c)
#pragma unroll
          for (int pt = 0; pt < PT; ++pt) {
            f32x4 d = acc[c][pt];                            // small terms first
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a6[c][2], b6[pt][0], d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a6[c][0], b6[pt][2], d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a6[c][1], b6[pt][1], d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a6[c][1], b6[pt][0], d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a6[c][0], b6[pt][1], d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a6[c][0], b6[pt][0], d, 0, 0, 0);
            acc[c][pt] = d;
          }
        tap = ntap; grp = ngr;
        continue;
      }
      f32x4 b[PT];
#ifndef CT_EXP_NOLDS    // diagnostic: -DCT_EXP_NOLDS feeds the MFMAs from registers (no activation reads)
#pragma unroll
      for (int pt = 0; pt < PT; ++pt)
        b[pt] = *reinterpret_cast<const f32x4*>(cur + ((pty_of(pt) + oy) * HW + ptx_of(pt) + ox) * CT_P + 16 * grp + 4 * q);
#else
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) b[pt] = a[pt % NCO];
      (void)oy; (void)ox;
#endif
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < NCO; ++c)
#pragma unroll
          for (int pt = 0; pt < PT; ++pt) acc[c][pt] = mfma16c(a[c][r], b[pt][r], acc[c][pt]);
      tap = ntap; grp = ngr;
    }

    // ---- epilogue of a finished tile (same contract as k_conv_gemm)
#ifdef CT_EXP_NOEPI      // diagnostic: one lane stores one value
    if (last_chunk && tid == 0 && tile == t_beg) A.out[0] = acc[0][0][0] + acc[NCO - 1][PT - 1][3];
    if (false) {
#else
    if (last_chunk) {
#endif
      int n, y0, x0;
      tile_origin(tile, n, y0, x0);
      const bool primal = n < A.n_bias;
      const bool vec = (A.Cout & 3) == 0;                 // 16-B accesses along the output channels
#pragma unroll
      for (int c = 0; c < NCO; ++c) {
        const int co = co0 + 16 * c + 4 * q;
        if (co >= A.Cout) continue;
        const bool full = vec && (co + 3 < A.Cout);
        // bias / per-sample bias of this lane's 4 channels: fetched once, shared by its two pixels
        f32x4 add = {0.f, 0.f, 0.f, 0.f};
        if (primal && A.bias) {
          if (full) add = *reinterpret_cast<const f32x4*>(A.bias + co);
          else
#pragma unroll
            for (int r = 0; r < 4; ++r) if (co + r < A.Cout) add[r] = A.bias[co + r];
        }
        if (A.samp_bias && n < A.n_samp) {
          const float* sbp = A.samp_bias + (size_t)n * A.Cout + co;
          if (full) add += *reinterpret_cast<const f32x4*>(sbp);
          else
#pragma unroll
            for (int r = 0; r < 4; ++r) if (co + r < A.Cout) add[r] += sbp[r];
        }
        f32x4 cs = {0.f, 0.f, 0.f, 0.f}, css = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
          if (!((y0 + pty_of(pt) < g.Ho) && (x0 + ptx_of(pt) < g.Wo))) continue;
          const size_t m = ((size_t)n * g.Ho + y0 + pty_of(pt)) * g.Wo + x0 + ptx_of(pt);
          f32x4 v = acc[c][pt] + add;
          float* op = A.out + m * A.Cout + co;
          if (full) {
            if (A.accumulate) v += *reinterpret_cast<const f32x4*>(op);
            if (A.residual) v += *reinterpret_cast<const f32x4*>(A.residual + (op - A.out));
            *reinterpret_cast<f32x4*>(op) = v;
            cs += v; css += v * v;
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (co + r < A.Cout) op[r] = (A.accumulate ? op[r] + v[r] : v[r]) + (A.residual ? A.residual[(op - A.out) + r] : 0.f);
          }
        }
        if (A.cstat) cstat_store(A, n, (tile - n * tiles_x * tiles_y) * 4 + w, co, cs, css, il);   // Cout % 4 == 0 (host)
      }
    }
    if (!more) break;
    if (!DB) {             // one LDS buffer: the next item is fetched after this one's MFMAs (other workgroups cover it)
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      stage_all(cur, ntile, ns, nc * CT_KC);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      tile = ntile; cs = ns; cc = nc;
      continue;
    }
    stage_store(nxt, 0, MAXST);
    // LDS hand-off only: __syncthreads() would also drain vmcnt, i.e. wait for this tile's output stores to land
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    float* t = cur; cur = nxt; nxt = t;
    tile = ntile; cs = ns; cc = nc;
  }
}

// ------------------------------------------------------------------ 1x1 convolutions: pixel-stationary streaming kernel
// A 1x1 (stride 1) convolution is a plain GEMM out[p][co] = sum_c x[p][c] W[co][c] with K = Cin of only 32..256: per
// 32-channel chunk the halo-tile kernel above has 2 (tap, group) pairs = 64 MFMAs per wave between two barriers, and it
// measured a FLAT ~45-60 TFLOP/s whatever the shape (1.2-3.7 TB/s of algorithmic bytes; tools/bench_1x1.py) — bound by
// its own barrier / staging cadence, not by HBM.  Here nothing is staged and nothing is shared: a wave owns 16*PT pixels,
// reads their activations ONCE from global memory straight into registers as MFMA B fragments (16 B per lane; the 4
// lanes of a pixel cover 64 contiguous bytes per 16-channel group) and keeps them for the whole kernel, then walks over
// the output-channel tiles with the weight fragments streamed L2 -> registers one (tile, group) pair ahead: PT*4 MFMAs
// per 16-byte weight load, no LDS, no barrier, every activation byte read once, every output byte written once.
// Same fused options as k_conv_tile: second source (concatenated K), per-(sample, channel) input affine (+SiLU), bias,
// per-sample bias, accumulate, residual.  Also the dgrad of the same convolution (with the Wd image).
// The loop over the output-channel tiles is STRAIGHT-LINE code: on gfx950 loads and stores retire through one in-order
// counter (vmcnt), so a wait for a load also waits for every store issued before it, and a conditional load (or any
// control flow around a memory operation) makes the compiler fall back to "wait for everything".  The first version of
// this kernel had both — per-tile conditional bias / weight loads — and its MFMA time and its store time simply added up
// (64 -> 192 channels at 1024 x 1024 pixels: 0.27 ms without the stores, 0.46 ms with them).  Here every per-tile load
// (weight ring, bias + per-sample bias, residual) is unconditional (clamped index or a 0/1 factor), is issued a tile
// ahead, BEFORE the previous tile's stores, and nothing in the loop waits on a store.
// What is left (diagnostic builds -DC1_EXP_NOMFMA / -DC1_EXP_COALESCED / -DC1_EXP_NOSTORE, 64 -> 192 channels): the memory
// streams alone take 0.35 ms with these 64-byte-per-pixel store segments (0.27 ms if every store wrote 1 KB contiguous), the
// MFMAs alone 0.27 ms, both together 0.43 ms — three waves per SIMD do not overlap the two completely.
// Host-side contract (conv_route): Cout % 16 == 0, P % (16 PT) == 0 (a wave's pixels belong to one sample), C0 % 16 == 0
// (and C1), at most one of accumulate / residual (EXTRA = 1: one more addend read per output element).
#define C1_LP 36       // LDS pitch of a pixel's 32 channels (floats): 16 consecutive pixels land on distinct banks for 16-B accesses
template <int PT, int KG, int EXTRA>     // PT pixel tiles of 16 per wave; KG = 16-channel groups of the (concatenated) input
__global__ void __launch_bounds__(256) k_conv1x1(ConvArgs A, int P /* pixels per sample */, long Mtot, int ct_per) {
  // full-line output stores through LDS pay with 64 pixels per wave (64-channel inputs: -7..-13 %), not with 32 or 16
  // (128+ channels: +0..5 %, measured) — those keep the direct stores
  constexpr bool LDS_OUT = PT == 4;
  __shared__ __attribute__((aligned(16))) float c1_lds[4][LDS_OUT ? 16 * PT * C1_LP : 4];     // per wave: [16 PT pixels][32 channels]
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, il = lane & 15, q = lane >> 4;
  const long m0 = ((long)blockIdx.x * 4 + w) * (16 * PT);
  if (m0 >= Mtot) return;                                  // no barriers in this kernel
  float* wl = c1_lds[w];
  const int n = __builtin_amdgcn_readfirstlane((int)(m0 / P));      // wave-uniform
  const int g0 = A.C[0] >> 4;                              // groups of the first source
  const int ctot = A.C[0] + (A.nsrc > 1 ? A.C[1] : 0);
  // ---- activations of this wave's pixels: B fragments, resident
  f32x4 b[PT][KG];
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    const long m = m0 + 16 * pt + il;
#pragma unroll
    for (int g = 0; g < KG; ++g) {
      const bool second = g >= g0;
      const int c = (second ? 16 * (g - g0) : 16 * g) + 4 * q;           // channel inside its source
      const int C = second ? A.C[1] : A.C[0];
      b[pt][g] = *reinterpret_cast<const f32x4*>((second ? A.src[1] : A.src[0]) + (size_t)m * C + c);
    }
  }
  if (A.in_scale) {             // folded GroupNorm(+SiLU): per-(sample, channel) affine — after ALL the loads are in flight
#pragma unroll
    for (int g = 0; g < KG; ++g) {
      const bool second = g >= g0;
      const size_t o = (size_t)n * ctot + (second ? A.C[0] + 16 * (g - g0) : 16 * g) + 4 * q;
      const f32x4 ga = *reinterpret_cast<const f32x4*>(A.in_scale + o), gb = *reinterpret_cast<const f32x4*>(A.in_shift + o);
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
        f32x4 v = b[pt][g] * ga + gb;
        if (A.in_act == 1) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = v[r] * __builtin_amdgcn_rcpf(1.0f + __expf(-v[r]));
        }
        b[pt][g] = v;
      }
    }
  }
  // ---- output-channel tiles.  Small launches (the 32-row per-GPU shard of C4) spread the tiles over blockIdx.y as well: a
  // wave that walked over all of them alone would leave most of the chip idle
  const int ct0 = blockIdx.y * ct_per, ntile = min(A.Cout >> 4, ct0 + ct_per);
  const float* wrow = A.Wp + (size_t)il * A.Ktot + 4 * q;   // row co = 16*ct + il, k = 16*g + 4*q (sources are 16-padded in K)
  const size_t tile_stride = (size_t)16 * A.Ktot;
  constexpr int D = (KG % 4 == 0) ? 4 : 2;                  // weight ring: D (tile, group) pairs ahead; D divides KG
  static_assert(KG % D == 0 && D <= KG, "ring slots must line up from one tile to the next");
  f32x4 an[D];
#pragma unroll
  for (int d = 0; d < D; ++d) an[d] = *reinterpret_cast<const f32x4*>(wrow + (size_t)ct0 * tile_stride + 16 * d);
  // bias (primal rows) + per-sample bias of this lane's 4 channels, one tile ahead; absent terms read a valid dummy address
  // (the weight image) and are multiplied by 0
  const bool has_b = A.bias && n < A.n_bias, has_s = A.samp_bias && n < A.n_samp;
  const float fb = has_b ? 1.f : 0.f, fs = has_s ? 1.f : 0.f;
  const float* bp = has_b ? A.bias + 4 * q : A.Wp;
  const float* sp = has_s ? A.samp_bias + (size_t)n * A.Cout + 4 * q : A.Wp;
  const int bstep = has_b ? 16 : 0, sstep = has_s ? 16 : 0;
  f32x4 add_n = *reinterpret_cast<const f32x4*>(bp + (size_t)ct0 * bstep) * fb + *reinterpret_cast<const f32x4*>(sp + (size_t)ct0 * sstep) * fs;
  const float* ex = EXTRA ? (A.residual ? A.residual : A.out) : nullptr;
  const size_t orow = (size_t)(m0 + il) * A.Cout + 4 * q;   // element offset of (pixel il of tile 0, channel 4q) in out
  const size_t ptstep = (size_t)16 * A.Cout;
  f32x4 ex_n[EXTRA ? PT : 1];
  if (EXTRA) {
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) ex_n[pt] = *reinterpret_cast<const f32x4*>(ex + orow + pt * ptstep + 16 * ct0);
  }
  for (int ct = ct0; ct < ntile; ++ct) {
    const int ctn = min(ct + 1, ntile - 1);                 // the last tile re-requests itself: no branch around a load
    f32x4 acc[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) acc[pt] = f32x4{0, 0, 0, 0};
    const f32x4 add = add_n;
    f32x4 exv[EXTRA ? PT : 1];
    if (EXTRA) {
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) exv[pt] = ex_n[pt];
    }
#pragma unroll
    for (int g = 0; g < KG; ++g) {
      const f32x4 a = an[g % D];
      // pair D ahead: (ct, g + D) or (ct + 1, g + D - KG); the K offset of group g is 16*g (koff[1] = 16*g0, second source)
      const int ng = (g + D < KG) ? g + D : g + D - KG, nct = (g + D < KG) ? ct : ctn;
      an[g % D] = *reinterpret_cast<const f32x4*>(wrow + (size_t)nct * tile_stride + 16 * ng);
      if (g == 0) {                                          // next tile's addends, requested before this tile's stores
        add_n = *reinterpret_cast<const f32x4*>(bp + (size_t)ctn * bstep) * fb + *reinterpret_cast<const f32x4*>(sp + (size_t)ctn * sstep) * fs;
        if (EXTRA) {
#pragma unroll
          for (int pt = 0; pt < PT; ++pt) ex_n[pt] = *reinterpret_cast<const f32x4*>(ex + orow + pt * ptstep + 16 * ctn);
        }
      }
      __builtin_amdgcn_sched_barrier(0);     // keep the requests HERE, D pairs ahead of their use (the scheduler sinks them)
#ifdef C1_EXP_NOMFMA       // diagnostic (WRONG results): the kernel's memory streams without its matrix work
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) acc[pt] += a * b[pt][g];
#else
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) acc[pt] = mfma16c(a[r], b[pt][g][r], acc[pt]);
#endif
    }
    // ---- epilogue of this tile: lane (pixel il of tile pt, channels 16ct + 4q .. +3).
    // Output stores go out as FULL 128-byte lines: two consecutive channel tiles (32 channels = 128 bytes per pixel) meet in
    // a wave-private LDS image [16 PT pixels][32 channels] and leave as rows — 8 lanes per pixel, 8 complete lines per store
    // instruction instead of 16 half lines (the kernel's memory streams alone: 0.35 -> 0.27 ms for 64 -> 192 channels).
    // A trailing unpaired tile is stored directly.
    f32x4 cs = {0.f, 0.f, 0.f, 0.f}, css = {0.f, 0.f, 0.f, 0.f};
    const int half = (ct - ct0) & 1;
    const bool paired = LDS_OUT && (half == 1 || ct + 1 < ntile);        // wave-uniform
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      f32x4 v = acc[pt] + add;
      if (EXTRA) v += exv[pt];
#if defined(C1_EXP_NOSTORE)    // diagnostic: what the kernel costs without its output stores
      if (v[0] == 12345.678f) *reinterpret_cast<f32x4*>(A.out + orow + pt * ptstep + 16 * ct) = v;
#elif defined(C1_EXP_COALESCED)    // diagnostic (WRONG results): the same bytes as fully coalesced 1-KB stores
      *reinterpret_cast<f32x4*>(A.out + ((size_t)(m0 / 16 + pt) * (A.Cout / 16) + ct) * 256 + lane * 4) = v;
#else
      if (paired) *reinterpret_cast<f32x4*>(wl + (16 * pt + il) * C1_LP + 16 * half + 4 * q) = v;
      else *reinterpret_cast<f32x4*>(A.out + orow + pt * ptstep + 16 * ct) = v;
#endif
      cs += v; css += v * v;
    }
#if !defined(C1_EXP_NOSTORE) && !defined(C1_EXP_COALESCED)
    if (LDS_OUT && half == 1) {                             // both halves are in LDS: rows out, 8 pixels per instruction
      const size_t obase = (size_t)m0 * A.Cout + 16 * (ct - 1) + 4 * (lane & 7);
#pragma unroll
      for (int j = 0; j < 2 * PT; ++j) {
        const int px = 8 * j + (lane >> 3);
        const f32x4 v = *reinterpret_cast<const f32x4*>(wl + px * C1_LP + 4 * (lane & 7));
        *reinterpret_cast<f32x4*>(A.out + obase + (size_t)px * A.Cout) = v;
        if (j & 1) __builtin_amdgcn_sched_barrier(0);      // two rows in flight, not all 2 PT (registers)
      }
    }
#endif
    if (A.cstat) cstat_store(A, n, (int)((m0 % P) / (16 * PT)), 16 * ct + 4 * q, cs, css, il);
  }
}

// ------------------------------------------------------------------ the U-Net's first and last 3x3 convolutions (vector ALU)
// model/unet.py:353-359 (image channels -> model_channels) and :442-446 (model_channels -> image channels): 1 or 3 channels
// on one side.  On the MFMA kernels those channels are padded to 16 (5x / 10x wasted matrix work) and the implicit GEMM
// re-reads its input once per tap from L2: 0.91 ms and 0.78 ms per 1024-row sampler step for 7 + 7 GFLOP of useful work
// (rocprofv3, C5).  Both are 864 multiply-adds per pixel — a job for the vector ALU with the weights in SCALAR registers
// (uniform addresses: s_load), one thread per output pixel:
//   k_conv3x3_cin_small : <= 4 input channels read straight from global memory / L1 (27 floats per pixel), CO accumulators;
//                         optional channel-statistics by-product (full-wave DPP reduction);
//   k_conv3x3_cout_small: <= 4 output channels; the 32-channel input halo tile is staged in LDS exactly as k_conv_tile does
//                         (same fused GroupNorm + SiLU input transform), each thread reads its 9 x 32 inputs as ds_read_b128.
// Forward only (mode 0); same bias / per-sample bias / accumulate / residual contract as the other forward kernels.
__device__ __forceinline__ float wave64_sum_to_last(float v) {        // lane 63 ends with the sum over the wave
  v = row16_sum(v);
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false));   // row_bcast:15
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, false));   // row_bcast:31
  return v;
}

template <int CO>
__global__ void __launch_bounds__(256) k_conv3x3_cin_small(ConvArgs A, long Mtot) {
  const ConvGeom g = A.g;
  const int P = g.Ho * g.Wo, Cin = A.C[0];
  const long m = (long)blockIdx.x * 256 + threadIdx.x;
  const bool live = m < Mtot;
  const long mm = live ? m : Mtot - 1;
  const int n = (int)(mm / P), r = (int)(mm - (long)n * P), y = r / g.Wo, x = r - y * g.Wo;
  const float* src = A.src[0] + (size_t)n * g.Hi * g.Wi * Cin;
  float acc[CO];
#pragma unroll
  for (int co = 0; co < CO; ++co) acc[co] = 0.f;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int iy = y + tap / 3 - 1, ix = x + tap % 3 - 1;
    const bool ok = iy >= 0 && iy < g.Hi && ix >= 0 && ix < g.Wi;
    const float* px = src + ((size_t)(ok ? iy : 0) * g.Wi + (ok ? ix : 0)) * Cin;
    // mode 1 = the transposed gather i = o + pad - k (dgrad of the same convolution with the Wd image): tap 8 - t's weights
    const float* wt = A.Wp + (size_t)(g.mode ? 8 - tap : tap) * A.CoutP * A.Ktot;   // W[tap][co][c] at wt[co * Ktot + c]: uniform
    for (int c = 0; c < Cin; ++c) {
      const float xv = ok ? px[c] : 0.f;
#pragma unroll
      for (int co = 0; co < CO; ++co) acc[co] = __builtin_fmaf(xv, wt[(size_t)co * A.Ktot + c], acc[co]);
    }
  }
  const bool primal = n < A.n_bias && A.bias;
  float* op = A.out + (size_t)mm * CO;
#pragma unroll
  for (int c4 = 0; c4 < CO / 4; ++c4) {
    f32x4 v = {acc[4 * c4], acc[4 * c4 + 1], acc[4 * c4 + 2], acc[4 * c4 + 3]};
    if (primal) v += *reinterpret_cast<const f32x4*>(A.bias + 4 * c4);
    if (A.samp_bias && n < A.n_samp) v += *reinterpret_cast<const f32x4*>(A.samp_bias + (size_t)n * CO + 4 * c4);
    if (live) {
      if (A.accumulate) v += *reinterpret_cast<const f32x4*>(op + 4 * c4);
      if (A.residual) v += *reinterpret_cast<const f32x4*>(A.residual + (size_t)mm * CO + 4 * c4);
      *reinterpret_cast<f32x4*>(op + 4 * c4) = v;
    } else {
      v = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[4 * c4 + k] = v[k];
  }
  if (A.cstat) {       // host: P % 64 == 0, so a wave's 64 pixels are one slot of one sample
    const int lane = threadIdx.x & 63;
    float* cp = A.cstat + (((size_t)n * A.cs_S + (r >> 6)) * 2) * CO;
#pragma unroll
    for (int co = 0; co < CO; ++co) {
      const float s = wave64_sum_to_last(acc[co]), ss = wave64_sum_to_last(acc[co] * acc[co]);
      if (lane == 63) { cp[co] = s; cp[CO + co] = ss; }
    }
  }
}

template <int CO>
__global__ void __launch_bounds__(256) k_conv3x3_cout_small(ConvArgs A, int tiles_x, int tiles_y) {
  extern __shared__ __attribute__((aligned(16))) float cs_lds[];      // [18 * 18][CT_P]
  const ConvGeom g = A.g;
  const int tid = threadIdx.x;
  constexpr int HW = 18, n_items = HW * HW * 8;
  int t = blockIdx.x;
  const int tx_i = t % tiles_x; t /= tiles_x;
  const int ty_i = t % tiles_y, n = t / tiles_y, y0 = ty_i * 16, x0 = tx_i * 16;
  const float* base = A.src[0] + (size_t)n * g.Hi * g.Wi * 32;
  // ---- halo tile of the 32 input channels, GroupNorm(+SiLU) applied while staging (zero padding stays zero)
  f32x4 ga = {1.f, 1.f, 1.f, 1.f}, gb = {0.f, 0.f, 0.f, 0.f};
  const int c4 = tid & 7;
  if (A.in_scale) {
    ga = *reinterpret_cast<const f32x4*>(A.in_scale + (size_t)n * 32 + 4 * c4);
    gb = *reinterpret_cast<const f32x4*>(A.in_shift + (size_t)n * 32 + 4 * c4);
  }
  constexpr int NST = (n_items + 255) / 256;
  f32x4 st[NST];
  unsigned valid = 0;
#pragma unroll
  for (int k = 0; k < NST; ++k) {                           // all the loads first, then the transform (see k_conv_tile)
    const int idx = tid + 256 * k;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (idx < n_items) {
      const int hp = idx >> 3, hy = hp / HW, hx = hp - hy * HW;
      const int iy = y0 + hy - 1, ix = x0 + hx - 1;
      if (iy >= 0 && iy < g.Hi && ix >= 0 && ix < g.Wi) {
        v = *reinterpret_cast<const f32x4*>(base + ((size_t)iy * g.Wi + ix) * 32 + 4 * c4);
        valid |= 1u << k;
      }
    }
    st[k] = v;
  }
#pragma unroll
  for (int k = 0; k < NST; ++k) {
    const int idx = tid + 256 * k;
    if (idx < n_items) {
      f32x4 v = st[k];
      if (A.in_scale && ((valid >> k) & 1u)) {
        v = v * ga + gb;
        if (A.in_act == 1) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = v[r] * __builtin_amdgcn_rcpf(1.0f + __expf(-v[r]));
        }
      }
      *reinterpret_cast<f32x4*>(cs_lds + (idx >> 3) * CT_P + 4 * c4) = v;
    }
  }
  __syncthreads();
  const int py = tid >> 4, px = tid & 15;
  float acc[CO];
#pragma unroll
  for (int co = 0; co < CO; ++co) acc[co] = 0.f;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const float* xp = cs_lds + ((py + tap / 3) * HW + px + tap % 3) * CT_P;
    const float* wt = A.Wp + (size_t)tap * A.CoutP * A.Ktot;           // W[tap][co][c] at wt[co * Ktot + c]: uniform
#pragma unroll
    for (int q4 = 0; q4 < 8; ++q4) {
      const f32x4 xv = *reinterpret_cast<const f32x4*>(xp + 4 * q4);
#pragma unroll
      for (int co = 0; co < CO; ++co) {
        const float* w = wt + (size_t)co * A.Ktot + 4 * q4;
        acc[co] = __builtin_fmaf(xv[0], w[0], acc[co]); acc[co] = __builtin_fmaf(xv[1], w[1], acc[co]);
        acc[co] = __builtin_fmaf(xv[2], w[2], acc[co]); acc[co] = __builtin_fmaf(xv[3], w[3], acc[co]);
      }
    }
    // pin the accumulators at the tap boundary: left alone the compiler runs each output channel's 288-FMA chain on its
    // own, which needs all 72 LDS reads (288 registers) live at once
#pragma unroll
    for (int co = 0; co < CO; ++co) asm volatile("" : "+v"(acc[co]));
  }
  const int oy = y0 + py, ox = x0 + px;
  if (oy < g.Ho && ox < g.Wo) {
    const size_t m = ((size_t)n * g.Ho + oy) * g.Wo + ox;
    float* op = A.out + m * CO;
#pragma unroll
    for (int co = 0; co < CO; ++co) {
      float v = acc[co];
      if (n < A.n_bias && A.bias) v += A.bias[co];
      if (A.samp_bias && n < A.n_samp) v += A.samp_bias[(size_t)n * CO + co];
      if (A.accumulate) v += op[co];
      if (A.residual) v += A.residual[m * CO + co];
      op[co] = v;
    }
  }
}

// The 4x4 input patch of a lane's Winograd tile (4 channels): tile (ty, tx) reads halo pixels (2ty + i, 2tx + j).  Columns 2, 3 of
// tile tx ARE columns 0, 1 of tile tx + 1, which is the next lane of the same 16-lane row (lanes il = 0..7 hold tiles tx = 0..7 of
// one tile row, il = 8..15 the next row): instead of reading them from LDS again the lane takes them from its neighbour's
// registers (DPP row_shl:1); only the last tile of a row (tx = 7) still reads its own.  16 -> 9 ds_read_b128 worth of LDS traffic
// per lane and group (the Winograd kernels with their weights in LDS are LDS-bandwidth-bound); the same values, bit for bit.
// Measured (AFF=1 tools/bench_wino.py 1024): +2-3 % on the LDS-weight shapes (135 -> 138, 151 -> 155, 157 -> 160 TFLOP/s as written).
__device__ __forceinline__ float dpp_from_next_lane(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x101 /* row_shl:1 */, 0xf, 0xf, true));
}
template <int HW_>
__device__ __forceinline__ void wino_patch_load(f32x4 (&d)[4][4], const float* __restrict__ p00, bool last_in_row) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    d[i][0] = *reinterpret_cast<const f32x4*>(p00 + (i * HW_ + 0) * CT_P);
    d[i][1] = *reinterpret_cast<const f32x4*>(p00 + (i * HW_ + 1) * CT_P);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) d[i][2 + j][r] = dpp_from_next_lane(d[i][j][r]);
  if (last_in_row) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      d[i][2] = *reinterpret_cast<const f32x4*>(p00 + (i * HW_ + 2) * CT_P);
      d[i][3] = *reinterpret_cast<const f32x4*>(p00 + (i * HW_ + 3) * CT_P);
    }
  }
}

// ------------------------------------------------------------------ Winograd F(2x2, 3x3) forward (sampler path)
// The reverse-SDE sampler spends 55 % of a step in stride-1 3x3 convolutions (C5, rocprofv3), a third of that in the
// 32-output-channel layers where the direct halo-tile kernel reaches only ~65 TFLOP/s (the halo staging is amortised over
// half as many MFMAs).  For those convolutions the minimal-filtering form Y = A^T [ (G g G^T) o (B^T d B) ] A computes a
// 2x2 output tile from a 4x4 input patch with 16 instead of 36 multiplications per (co, ci): 2.25x fewer MFMAs, all in
// fp32 (the transforms are additions and halvings; the result differs from the direct form by fp32 rounding only —
// measured in tests/test_conv_gpu.py).  No tangent-specific code (tangent rows are batch rows); r3: the training step's forward
// and dgrad (flipped, transposed kernel image) run on it as well, the weight gradients keep their own kernels.
//   * workgroup = 16x16 output pixels = 8x8 Winograd tiles x 32 output channels, the (16+2)^2 halo of a 32-channel chunk
//     staged in LDS exactly as k_conv_tile does (same fused GroupNorm(+SiLU) input transform, two sources, folded
//     2x upsample);
//   * wave w owns 16 tiles (lane&15) for ALL 16 transform positions: per 16-channel group a lane reads its 4x4 patch of
//     4 channels (16 ds_read_b128), transforms it in registers (B^T d B: 32 additions per channel) and the 16 results
//     are the B operands of 16 x NCO MFMAs per k-step — so the output transform A^T M A needs no exchange between
//     lanes: the 16 position-accumulators of a tile sit in the same lane;
//   * the transformed weights U[pos][co][ci] (packed like Wp with 16 "taps" by k_wino_pack) stream L2 -> registers one
//     position ahead.
//   * WL (weights through LDS, the default): the register form above keeps ONE position's weight fragments in flight — 8 MFMAs
//     (256 cycles) to cover an L2 round trip of 500-900, and at 256 registers (two workgroups per CU) a deeper ring spills:
//     MfmaUtil 32 %, 62 % of the wave cycles waiting (profiles/r03, PMC of the 1024-row EM step).  With WL the transformed
//     weights of HALF a channel group (8 positions x 32 channels x 16 inputs = 16 KB) travel L2 -> LDS by LDS-DMA
//     (global_load_lds_dwordx4: no registers), laid out as the A fragments themselves (one 1-KB piece per (position, co tile),
//     lane-linear, conflict-free ds_read_b128), double-buffered: the DMA of half-group s + 1 is issued right after the barrier
//     that opens half-group s and has that half-group's 64 MFMAs per wave (2 k cycles) to land.  LDS = halo 46.7 KB + 2 x 16 KB.
template <int NCO, bool WL = false>
__global__ void __launch_bounds__(256, 2) k_conv_wino(ConvArgs A, int tiles_x, int tiles_y, int n_tiles, int n_cob, int n_tgrp) {
  extern __shared__ __attribute__((aligned(16))) float cw_lds[];
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, il = lane & 15, q = lane >> 4;
  const ConvGeom g = A.g;
  constexpr int HW = 18, halo = 18 * 18;
  float* cur = cw_lds;
  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;            // XCD-aware order, as k_conv_tile
  const int cob = loc % n_cob, tile = (loc / n_cob) * 8 + xcd;
  (void)n_tgrp;
  if (tile >= n_tiles) return;
  const int co0 = cob * (NCO * 16);
  // this lane's Winograd tile inside the 16x16 output tile: 8x8 tiles, wave w owns tile rows 2w, 2w+1
  const int tt = 16 * w + il, ty = tt >> 3, tx = tt & 7;
  const int pbase = (2 * ty * HW + 2 * tx) * CT_P + 4 * q;            // LDS offset of patch element (0,0), channel 4q
  f32x4 acc[16][NCO];
#pragma unroll
  for (int p = 0; p < 16; ++p)
#pragma unroll
    for (int c = 0; c < NCO; ++c) acc[p][c] = f32x4{0, 0, 0, 0};

  int nch[CONV_MAX_SRC];
#pragma unroll
  for (int s = 0; s < CONV_MAX_SRC; ++s) nch[s] = s < A.nsrc ? (A.C[s] + CT_KC - 1) / CT_KC : 0;
  constexpr int MAXST = (halo * (CT_KC / 4) + 255) / 256;
  constexpr int n_items = halo * (CT_KC / 4);
  const int up = g.ups ? 1 : 0;
  const int ctot_all = A.C[0] + (A.nsrc > 1 ? A.C[1] : 0);
  int n, y0, x0;
  { int t = tile; const int tx_i = t % tiles_x; t /= tiles_x; const int ty_i = t % tiles_y; n = t / tiles_y; y0 = ty_i * 16; x0 = tx_i * 16; }
  auto stage = [&](int s_, int c0) __attribute__((always_inline)) {    // global -> (transform) -> LDS, one 32-channel chunk
    const int C = A.C[s_];
    const float* base = A.src[s_] + (size_t)n * g.Hi * g.Wi * C;
    f32x4 ga = {1.f, 1.f, 1.f, 1.f}, gb = {0.f, 0.f, 0.f, 0.f};
    const int cq = c0 + 4 * (tid & 7);
    if (A.in_scale && cq < C) {
      const size_t o = (size_t)n * ctot_all + (s_ ? A.C[0] : 0) + cq;
      ga = *reinterpret_cast<const f32x4*>(A.in_scale + o);
      gb = *reinterpret_cast<const f32x4*>(A.in_shift + o);
    }
    f32x4 st[MAXST];
    unsigned valid = 0;
#pragma unroll
    for (int k = 0; k < MAXST; ++k) {                       // all the loads first (see k_conv_tile's stage_load)
      const int idx = tid + 256 * k;
      f32x4 v = {0, 0, 0, 0};
      if (idx < n_items) {
        const int hp = idx >> 3, hy = hp / HW, hx = hp - hy * HW;        // division by the constant 18
        const int iy = y0 + hy - 1, ix = x0 + hx - 1;
        const int c = c0 + 4 * (idx & 7);
        if (iy >= 0 && iy < (g.Hi << up) && ix >= 0 && ix < (g.Wi << up) && c < C) {
          v = *reinterpret_cast<const f32x4*>(base + ((size_t)(iy >> up) * g.Wi + (ix >> up)) * C + c);
          valid |= 1u << k;
        }
      }
      st[k] = v;
    }
    if (A.in_scale) {
#pragma unroll
      for (int k = 0; k < MAXST; ++k) {
        if ((valid >> k) & 1u) {
          f32x4 v = st[k] * ga + gb;
          if (A.in_act == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = v[r] * __builtin_amdgcn_rcpf(1.0f + __expf(-v[r]));
          }
          st[k] = v;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < MAXST; ++k) {
      const int idx = tid + 256 * k;
      if (idx < n_items) *reinterpret_cast<f32x4*>(cur + (idx >> 3) * CT_P + 4 * (idx & 7)) = st[k];
    }
  };

  const size_t a_co_stride = (size_t)16 * A.Ktot, pos_stride = (size_t)A.CoutP * A.Ktot;
  if constexpr (WL) {
    static_assert(NCO == 2, "the LDS weight image is laid out for two output-channel tiles");
    float* wbuf = cw_lds + halo * CT_P;                       // [2 buffers][8 positions][2 co tiles][64 lanes][4]
    auto ngrp_of = [&](int s_, int c_) { const int rem = A.C[s_] - c_ * CT_KC; return rem >= CT_KC ? 2 : ((rem + 15) >> 4); };
    // the 16 pieces of one half-group: wave w copies pieces 4w .. 4w+3 (piece = 2 (position in the half) + co tile); lane
    // (il, q) brings U[pos][co0 + 16 c + il][k .. k+3], k = 4 q of the group — exactly its A fragment
    auto wfill = [&](int s_, int c_, int g_, int half, int b) __attribute__((always_inline)) {
      const float* src = A.Wp + (size_t)(co0 + il) * A.Ktot + A.koff[s_] + c_ * CT_KC + 16 * g_ + 4 * q + (size_t)(8 * half) * pos_stride;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int pw = 4 * w + j;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)(pw >> 1) * pos_stride + (pw & 1) * a_co_stride),
                                         (__attribute__((address_space(3))) void*)(wbuf + b * 4096 + pw * 256), 16, 0, 0);
      }
    };
    int cs = 0, cc = 0, grp = 0, step = 0;
    wfill(0, 0, 0, 0, 0);
    for (;;) {
      if (grp == 0) {                                         // a new 32-channel chunk: restage the halo
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");         // the previous chunk's readers are done
        stage(cs, cc * CT_KC);
      }
      // the group that follows this one (its first half-group is requested during this group's second)
      int ns = cs, nc = cc, ng = grp + 1;
      bool more = true;
      if (ng >= ngrp_of(cs, cc)) {
        ng = 0; ++nc;
        if (nc >= nch[cs]) { nc = 0; ++ns; more = ns < A.nsrc; }
      }
      f32x4 d[4][4];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        // this half-group's weights have landed (every wave waits for its own pieces, then all meet); the halo stores and
        // the previous half-group's reads of the other buffer are behind the same barrier
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (half == 0) wfill(cs, cc, grp, 1, (step + 1) & 1);
        else if (more) wfill(ns, nc, ng, 0, (step + 1) & 1);
        if (half == 0) {
          // ---- the lane's 4x4 patch (4 channels) and its transform V = B^T d B, in place
          wino_patch_load<HW>(d, cur + pbase + 16 * grp, tx == 7);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f32x4 r0 = d[0][j] - d[2][j], r1 = d[1][j] + d[2][j], r2 = d[2][j] - d[1][j], r3 = d[1][j] - d[3][j];
            d[0][j] = r0; d[1][j] = r1; d[2][j] = r2; d[3][j] = r3;
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const f32x4 c0_ = d[i][0] - d[i][2], c1_ = d[i][1] + d[i][2], c2_ = d[i][2] - d[i][1], c3_ = d[i][1] - d[i][3];
            d[i][0] = c0_; d[i][1] = c1_; d[i][2] = c2_; d[i][3] = c3_;
          }
        }
        const float* wb = wbuf + (step & 1) * 4096 + lane * 4;
        f32x4 a0 = *reinterpret_cast<const f32x4*>(wb), a1 = *reinterpret_cast<const f32x4*>(wb + 256);
#pragma unroll
        for (int pl = 0; pl < 8; ++pl) {
          const f32x4 x0 = a0, x1 = a1;
          if (pl < 7) {
            a0 = *reinterpret_cast<const f32x4*>(wb + (2 * pl + 2) * 256);
            a1 = *reinterpret_cast<const f32x4*>(wb + (2 * pl + 3) * 256);
          }
          const int pos = 8 * half + pl;
          const f32x4 b = d[pos >> 2][pos & 3];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            acc[pos][0] = mfma16c(x0[r], b[r], acc[pos][0]);
            acc[pos][1] = mfma16c(x1[r], b[r], acc[pos][1]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        ++step;
      }
      if (!more) break;
      cs = ns; cc = nc; grp = ng;
    }
  } else
  for (int cs = 0; cs < A.nsrc; ++cs)
    for (int cc = 0; cc < (cs ? nch[1] : nch[0]); ++cc) {
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");           // the previous chunk's readers are done
      stage(cs, cc * CT_KC);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      const int C = A.C[cs], c0 = cc * CT_KC;
      const int ngrp = (C - c0 >= CT_KC) ? 2 : ((C - c0 + 15) >> 4);
      const float* wbase = A.Wp + (size_t)(co0 + il) * A.Ktot + A.koff[cs] + c0 + 4 * q;
      f32x4 an[NCO];
#pragma unroll
      for (int c = 0; c < NCO; ++c) an[c] = *reinterpret_cast<const f32x4*>(wbase + c * a_co_stride);
      // ONE running weight pointer (kept opaque to the optimiser: with 16 x NCO precomputed 64-bit addresses per group the
      // kernel spilled at 256 registers), advanced by one position per step and to the next channel group after 16
      const float* wq = wbase;
      for (int grp = 0; grp < ngrp; ++grp) {
        // ---- the lane's 4x4 patch (4 channels) and its transform V = B^T d B, in place
        f32x4 d[4][4];
        wino_patch_load<HW>(d, cur + pbase + 16 * grp, tx == 7);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 r0 = d[0][j] - d[2][j], r1 = d[1][j] + d[2][j], r2 = d[2][j] - d[1][j], r3 = d[1][j] - d[3][j];
          d[0][j] = r0; d[1][j] = r1; d[2][j] = r2; d[3][j] = r3;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const f32x4 c0_ = d[i][0] - d[i][2], c1_ = d[i][1] + d[i][2], c2_ = d[i][2] - d[i][1], c3_ = d[i][1] - d[i][3];
          d[i][0] = c0_; d[i][1] = c1_; d[i][2] = c2_; d[i][3] = c3_;
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- 16 positions: M_p += U_p V_p
#pragma unroll
        for (int pos = 0; pos < 16; ++pos) {
          f32x4 a[NCO];
#pragma unroll
          for (int c = 0; c < NCO; ++c) a[c] = an[c];
          // weights one (group, position) pair ahead
          if (pos < 15) wq += pos_stride;
          else wq = wbase + 16 * (grp + 1);
          asm volatile("" : "+v"(wq));
          if (pos < 15 || grp + 1 < ngrp) {
#pragma unroll
            for (int c = 0; c < NCO; ++c) an[c] = *reinterpret_cast<const f32x4*>(wq + c * a_co_stride);
          }
          const f32x4 b = d[pos >> 2][pos & 3];
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < NCO; ++c) acc[pos][c] = mfma16c(a[c][r], b[r], acc[pos][c]);
          __builtin_amdgcn_sched_barrier(0);          // one position's weight fragments in flight, not all sixteen
        }
      }
    }

  // ---- output transform Y = A^T M A per (tile, co quad) and the epilogue of k_conv_tile (bias / per-sample bias /
  //      accumulate / residual), 2x2 pixels x 4 consecutive output channels per lane and co tile
  const bool primal = n < A.n_bias;
  const bool vec = (A.Cout & 3) == 0;
#pragma unroll
  for (int c = 0; c < NCO; ++c) {
    const int co = co0 + 16 * c + 4 * q;
    if (co >= A.Cout) continue;
    const bool full = vec && (co + 3 < A.Cout);
    f32x4 add = {0.f, 0.f, 0.f, 0.f};
    if (primal && A.bias) {
      if (full) add = *reinterpret_cast<const f32x4*>(A.bias + co);
      else
#pragma unroll
        for (int r = 0; r < 4; ++r) if (co + r < A.Cout) add[r] = A.bias[co + r];
    }
    if (A.samp_bias && n < A.n_samp) {
      const float* sbp = A.samp_bias + (size_t)n * A.Cout + co;
      if (full) add += *reinterpret_cast<const f32x4*>(sbp);
      else
#pragma unroll
        for (int r = 0; r < 4; ++r) if (co + r < A.Cout) add[r] += sbp[r];
    }
    f32x4 t0[4], t1[4];
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) {
      t0[nu] = acc[nu][c] + acc[4 + nu][c] + acc[8 + nu][c];
      t1[nu] = acc[4 + nu][c] - acc[8 + nu][c] - acc[12 + nu][c];
    }
    f32x4 Y[2][2];
    Y[0][0] = t0[0] + t0[1] + t0[2]; Y[0][1] = t0[1] - t0[2] - t0[3];
    Y[1][0] = t1[0] + t1[1] + t1[2]; Y[1][1] = t1[1] - t1[2] - t1[3];
    f32x4 cs = {0.f, 0.f, 0.f, 0.f}, css = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int oy = y0 + 2 * ty + dy, ox = x0 + 2 * tx + dx;
        if (oy >= g.Ho || ox >= g.Wo) continue;
        const size_t m = ((size_t)n * g.Ho + oy) * g.Wo + ox;
        f32x4 v = Y[dy][dx] + add;
        float* op = A.out + m * A.Cout + co;
        if (full) {
          if (A.accumulate) v += *reinterpret_cast<const f32x4*>(op);
          if (A.residual) v += *reinterpret_cast<const f32x4*>(A.residual + (op - A.out));
          *reinterpret_cast<f32x4*>(op) = v;
          cs += v; css += v * v;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (co + r < A.Cout) op[r] = (A.accumulate ? op[r] + v[r] : v[r]) + (A.residual ? A.residual[(op - A.out) + r] : 0.f);
        }
      }
    // the wave's 16 Winograd tiles = 64 pixels: one statistics slot, as the direct kernel's 16x16 tiles (Cout % 4 == 0, host)
    if (A.cstat) cstat_store(A, n, (tile - n * tiles_x * tiles_y) * 4 + w, co, cs, css, il);
  }
}

// ------------------------------------------------------------------ Winograd, 32 input channels: persistent workgroups
// The 32 -> 32 convolutions of the 64x64 level (one 32-channel chunk: K = 32) were the slowest Winograd shape: nothing to
// pipeline across chunks, the weight fragments one position ahead from L2 — MfmaUtil 26-31 % (profiles/r03/pmc_c4_step.json;
// 14 launches per training step, 11 % of the sampler's step).  Their whole transformed weight image is 16 positions x 32
// channels x 32 inputs = 64 KB: here it is copied into LDS ONCE per workgroup (LDS-DMA, fragment order as in the WL form) and the
// workgroup — one per CU, its waves alone on their SIMDs with the whole register file — walks over tiles: the NEXT tile's halo
// travels global -> registers under this tile's MFMAs (no weight loads in the loop, so the in-order vmcnt only ever waits
// for halo data it needs) and is written (after the folded GroupNorm + SiLU) into the second halo buffer; one barrier per tile.
// LDS: 64 KB weights + 2 x 46.7 KB halo = 155 KB.  Same staging options and epilogue as k_conv_wino.
__global__ void __launch_bounds__(256, 1) k_conv_wino_p32(ConvArgs A, int tiles_x, int tiles_y, int n_tiles) {
  extern __shared__ __attribute__((aligned(16))) float cw_lds[];
  constexpr int NCO = 2, HW = 18, halo = 18 * 18;
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, il = lane & 15, q = lane >> 4;
  const ConvGeom g = A.g;
  float* wbuf = cw_lds;                                     // [2 groups][16 positions][2 co tiles][64 lanes][4]
  float* hb0 = cw_lds + 2 * 16 * 2 * 256;
  float* hb1 = hb0 + halo * CT_P;
  const int tt = 16 * w + il, ty = tt >> 3, tx = tt & 7;
  const int pbase = (2 * ty * HW + 2 * tx) * CT_P + 4 * q;
  const size_t a_co_stride = (size_t)16 * A.Ktot, pos_stride = (size_t)A.CoutP * A.Ktot;
  int t = blockIdx.x;
  if (t >= n_tiles) return;
  // ---- the weights, once: 64 pieces of 1 KB, 16 per wave (piece = (group, position, co tile))
  {
    const float* src = A.Wp + (size_t)il * A.Ktot + 4 * q;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int pw = 16 * w + j, grp = pw >> 5, pos = (pw >> 1) & 15, c = pw & 1;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)pos * pos_stride + c * a_co_stride + 16 * grp),
                                       (__attribute__((address_space(3))) void*)(wbuf + pw * 256), 16, 0, 0);
    }
  }
  constexpr int MAXST = (halo * (CT_KC / 4) + 255) / 256;
  constexpr int n_items = halo * (CT_KC / 4);
  const int up = g.ups ? 1 : 0;
  const int C = A.C[0];
  f32x4 st[MAXST];
  f32x4 ga = {1.f, 1.f, 1.f, 1.f}, gb = {0.f, 0.f, 0.f, 0.f};
  unsigned valid = 0;
  auto origin = [&](int tile, int& n, int& y0, int& x0) {
    const int tx_i = tile % tiles_x; tile /= tiles_x;
    const int ty_i = tile % tiles_y; n = tile / tiles_y; y0 = ty_i * 16; x0 = tx_i * 16;
  };
  auto stage_load = [&](int tile) __attribute__((always_inline)) {
    int n, y0, x0;
    origin(tile, n, y0, x0);
    const float* base = A.src[0] + (size_t)n * g.Hi * g.Wi * C;
    const int cq = 4 * (tid & 7);
    ga = f32x4{1.f, 1.f, 1.f, 1.f}; gb = f32x4{0.f, 0.f, 0.f, 0.f};
    if (A.in_scale && cq < C) {
      const size_t o = (size_t)n * C + cq;
      ga = *reinterpret_cast<const f32x4*>(A.in_scale + o);
      gb = *reinterpret_cast<const f32x4*>(A.in_shift + o);
    }
    valid = 0;
#pragma unroll
    for (int k = 0; k < MAXST; ++k) {
      const int idx = tid + 256 * k;
      f32x4 v = {0, 0, 0, 0};
      if (idx < n_items) {
        const int hp = idx >> 3, hy = hp / HW, hx = hp - hy * HW;
        const int iy = y0 + hy - 1, ix = x0 + hx - 1;
        const int c = 4 * (idx & 7);
        if (iy >= 0 && iy < (g.Hi << up) && ix >= 0 && ix < (g.Wi << up) && c < C) {
          v = *reinterpret_cast<const f32x4*>(base + ((size_t)(iy >> up) * g.Wi + (ix >> up)) * C + c);
          valid |= 1u << k;
        }
      }
      st[k] = v;
    }
  };
  auto stage_store = [&](float* buf) __attribute__((always_inline)) {
    if (A.in_scale) {
#pragma unroll
      for (int k = 0; k < MAXST; ++k) {
        if ((valid >> k) & 1u) {
          f32x4 v = st[k] * ga + gb;
          if (A.in_act == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = v[r] * __builtin_amdgcn_rcpf(1.0f + __expf(-v[r]));
          }
          st[k] = v;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < MAXST; ++k) {
      const int idx = tid + 256 * k;
      if (idx < n_items) *reinterpret_cast<f32x4*>(buf + (idx >> 3) * CT_P + 4 * (idx & 7)) = st[k];
    }
  };
  stage_load(t);
  stage_store(hb0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");        // weights and the first halo are in LDS
  float* cur = hb0;
  float* nxt = hb1;
  const int ngrp = (C + 15) >> 4;
  const bool vec = (A.Cout & 3) == 0;
  for (; t < n_tiles; t += gridDim.x) {
    const int tn = t + gridDim.x;
    const bool more = tn < n_tiles;
    if (more) stage_load(tn);                               // the next tile's halo: in flight under this tile's MFMAs
    f32x4 acc[16][NCO];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
      for (int c = 0; c < NCO; ++c) acc[p][c] = f32x4{0, 0, 0, 0};
    for (int grp = 0; grp < ngrp; ++grp) {
      // (computing group g + 1's transformed patch under group g's MFMAs — a second patch register set, no scheduling fences —
      // measured SLOWER, 0.625 -> 0.666 ms at 1024 x 64 x 64: hipcc bunches the loads and additions in front of the MFMAs anyway)
      f32x4 d[4][4];
      // (plain reads here: this form is not LDS-bound, and the neighbour exchange of wino_patch_load measured 3 % slower)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) d[i][j] = *reinterpret_cast<const f32x4*>(cur + pbase + (i * HW + j) * CT_P + 16 * grp);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 r0 = d[0][j] - d[2][j], r1 = d[1][j] + d[2][j], r2 = d[2][j] - d[1][j], r3 = d[1][j] - d[3][j];
        d[0][j] = r0; d[1][j] = r1; d[2][j] = r2; d[3][j] = r3;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x4 c0_ = d[i][0] - d[i][2], c1_ = d[i][1] + d[i][2], c2_ = d[i][2] - d[i][1], c3_ = d[i][1] - d[i][3];
        d[i][0] = c0_; d[i][1] = c1_; d[i][2] = c2_; d[i][3] = c3_;
      }
      const float* wb = wbuf + grp * (16 * 2 * 256) + lane * 4;
      f32x4 a0 = *reinterpret_cast<const f32x4*>(wb), a1 = *reinterpret_cast<const f32x4*>(wb + 256);
#pragma unroll
      for (int pos = 0; pos < 16; ++pos) {
        const f32x4 x0_ = a0, x1_ = a1;
        if (pos < 15) {
          a0 = *reinterpret_cast<const f32x4*>(wb + (2 * pos + 2) * 256);
          a1 = *reinterpret_cast<const f32x4*>(wb + (2 * pos + 3) * 256);
        }
        const f32x4 b = d[pos >> 2][pos & 3];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          acc[pos][0] = mfma16c(x0_[r], b[r], acc[pos][0]);
          acc[pos][1] = mfma16c(x1_[r], b[r], acc[pos][1]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ---- output transform and epilogue (as k_conv_wino)
    int n, y0, x0;
    origin(t, n, y0, x0);
    const bool primal = n < A.n_bias;
#pragma unroll
    for (int c = 0; c < NCO; ++c) {
      const int co = 16 * c + 4 * q;
      if (co >= A.Cout) continue;
      const bool full = vec && (co + 3 < A.Cout);
      f32x4 add = {0.f, 0.f, 0.f, 0.f};
      if (primal && A.bias) {
        if (full) add = *reinterpret_cast<const f32x4*>(A.bias + co);
        else
#pragma unroll
          for (int r = 0; r < 4; ++r) if (co + r < A.Cout) add[r] = A.bias[co + r];
      }
      if (A.samp_bias && n < A.n_samp) {
        const float* sbp = A.samp_bias + (size_t)n * A.Cout + co;
        if (full) add += *reinterpret_cast<const f32x4*>(sbp);
        else
#pragma unroll
          for (int r = 0; r < 4; ++r) if (co + r < A.Cout) add[r] += sbp[r];
      }
      f32x4 t0[4], t1[4];
#pragma unroll
      for (int nu = 0; nu < 4; ++nu) {
        t0[nu] = acc[nu][c] + acc[4 + nu][c] + acc[8 + nu][c];
        t1[nu] = acc[4 + nu][c] - acc[8 + nu][c] - acc[12 + nu][c];
      }
      f32x4 Y[2][2];
      Y[0][0] = t0[0] + t0[1] + t0[2]; Y[0][1] = t0[1] - t0[2] - t0[3];
      Y[1][0] = t1[0] + t1[1] + t1[2]; Y[1][1] = t1[1] - t1[2] - t1[3];
      f32x4 cs = {0.f, 0.f, 0.f, 0.f}, css = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          const int oy = y0 + 2 * ty + dy, ox = x0 + 2 * tx + dx;
          if (oy >= g.Ho || ox >= g.Wo) continue;
          const size_t m = ((size_t)n * g.Ho + oy) * g.Wo + ox;
          f32x4 v = Y[dy][dx] + add;
          float* op = A.out + m * A.Cout + co;
          if (full) {
            if (A.accumulate) v += *reinterpret_cast<const f32x4*>(op);
            if (A.residual) v += *reinterpret_cast<const f32x4*>(A.residual + (op - A.out));
            *reinterpret_cast<f32x4*>(op) = v;
            cs += v; css += v * v;
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (co + r < A.Cout) op[r] = (A.accumulate ? op[r] + v[r] : v[r]) + (A.residual ? A.residual[(op - A.out) + r] : 0.f);
          }
        }
      if (A.cstat) cstat_store(A, n, (t - n * tiles_x * tiles_y) * 4 + w, co, cs, css, il);
    }
    if (more) stage_store(nxt);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");               // the other buffer is complete; this one is free
    float* sw = cur; cur = nxt; nxt = sw;
  }
}

// U[pos = 4 xi + nu][r][kp_off + c] = (G g G^T)[xi][nu] of the 3x3 kernel g = W[r][col_off + c][:, :]; same job table
// as k_pack_batched (taps = 9 on input: element (r, c, t) at W + r*sr + (col_off + c)*sc + t*st), 16 positions out
__global__ void __launch_bounds__(256) k_wino_pack_batched(const msgm_pack_job_t* __restrict__ jobs) {
  const msgm_pack_job_t J = jobs[blockIdx.y];
  const int64_t tot = (int64_t)J.rows * J.ncols;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < tot; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % J.ncols), r = (int)(e / J.ncols);
    const float* w = J.W + r * J.sr + (J.col_off + c) * J.sc;
    float gk[3][3], h[4][3];
#pragma unroll
    for (int t = 0; t < 9; ++t) gk[t / 3][t % 3] = w[t * J.st];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      h[0][j] = gk[0][j];
      h[1][j] = 0.5f * (gk[0][j] + gk[1][j] + gk[2][j]);
      h[2][j] = 0.5f * (gk[0][j] - gk[1][j] + gk[2][j]);
      h[3][j] = gk[2][j];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float u0 = h[i][0], u1 = 0.5f * (h[i][0] + h[i][1] + h[i][2]), u2 = 0.5f * (h[i][0] - h[i][1] + h[i][2]), u3 = h[i][2];
      float* p = J.Wp + ((int64_t)(4 * i) * J.rowsP + r) * J.Ktot + J.kp_off + c;
      const int64_t ps = (int64_t)J.rowsP * J.Ktot;
      p[0] = u0; p[ps] = u1; p[2 * ps] = u2; p[3 * ps] = u3;
    }
  }
}

// ------------------------------------------------------------------ wgrad
struct WgradArgs {
  ConvGeom g;
  const float* gy;            // [N][Ho][Wo][Cout]
  const float* src;           // one source [N][Hi][Wi][C]
  int C, koff;
  float* dWp;                 // [taps][CoutP][Ktot], accumulated with float atomics
  int Cout, CoutP, Ktot;
  int chunk;                  // output positions per workgroup
  float* dbias;               // optional: dbias[co] += sum over primal rows (n < n_bias) and pixels of gy (tile kernel)
  int n_bias;
  // structurally-zero weight blocks (tile kernel): bit t = tap t present, 0 = all; per 32-channel block of c / of co
  unsigned short tm_c[16], tm_o[16];
  // deterministic mode (msgm_conv_wgrad_det): instead of float atomics into dWp / dbias every workgroup column
  // blockIdx.x STORES its partial block into its own slab [taps][CoutP][C] (+ [CoutP] bias partials) and
  // k_wgrad_slab_reduce adds the slabs in slot order.  slab == nullptr: atomics.
  float* slab; long slab_stride;
};

// One workgroup = one (position chunk, tap, co block of 16*MT, c block of 16*KT).
// Reduction over positions: A lane (co, q) <- gy[m+q][co], B lane (c, q) <- in[src(m+q)][c].
template <int MT, int KT>
__global__ void __launch_bounds__(256) k_conv_wgrad(WgradArgs A) {
  __shared__ float red[4][MT * KT * 256];
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, il = lane & 15, q = lane >> 4;
  const ConvGeom g = A.g;
  const int HoWo = g.Ho * g.Wo;
  const int Mtot = g.N * HoWo;
  const int cblocks = (A.C + 16 * KT - 1) / (16 * KT);
  const int coblk = blockIdx.y / cblocks, cblk = blockIdx.y - coblk * cblocks;
  const int co0 = coblk * 16 * MT, c0 = cblk * 16 * KT;
  const int tap = blockIdx.z;
  const int kh = tap / g.KW, kw = tap - kh * g.KW;
  const int Hup = g.ups ? 2 * g.Hi : g.Hi, Wup = g.ups ? 2 * g.Wi : g.Wi;
  const int mbeg = blockIdx.x * A.chunk;
  const int mend = min(mbeg + A.chunk, Mtot);

  f32x4 acc[MT][KT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) acc[mt][kt] = f32x4{0, 0, 0, 0};

  // the four waves interleave groups of 4 positions; (n, oh, ow) of this lane's position advance by 16 per
  // iteration with carries instead of two integer divisions per step
  int m = mbeg + 4 * w + q;
  int n = m / HoWo, oh, ow;
  { const int r = m - n * HoWo; oh = r / g.Wo; ow = r - oh * g.Wo; }
  const float* gyp[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) gyp[mt] = A.gy + (size_t)m * A.Cout + co0 + 16 * mt + il;
  bool vco[MT], vc[KT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) vco[mt] = co0 + 16 * mt + il < A.Cout;
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) vc[kt] = c0 + 16 * kt + il < A.C;
  const size_t gy_step = (size_t)16 * A.Cout;
  for (int mg = mbeg + 4 * w; mg < mend; mg += 16) {     // wave-uniform trip count
    const bool pin = m < mend;
    float a[MT], b[KT];
    int ih = 0, iw = 0;
    const bool ok = pin && src_coord(g, oh, kh, Hup, g.strideH, g.padH, ih) && src_coord(g, ow, kw, Wup, g.strideW, g.padW, iw);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a[mt] = (pin && vco[mt]) ? *gyp[mt] : 0.f;
    const float* sp = A.src + ((size_t)(n * g.Hi + ih) * g.Wi + iw) * A.C + c0 + il;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) b[kt] = (ok && vc[kt]) ? sp[16 * kt] : 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) acc[mt][kt] = mfma16c(a[mt], b[kt], acc[mt][kt]);
    m += 16; ow += 16;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) gyp[mt] += gy_step;
    while (ow >= g.Wo) { ow -= g.Wo; ++oh; }
    while (oh >= g.Ho) { oh -= g.Ho; ++n; }
  }
  // cross-wave sum, then one atomic per element per workgroup
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[w][((mt * KT + kt) * 4 + r) * 64 + lane] = acc[mt][kt][r];
  __syncthreads();
  if (w == 0) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int idx = ((mt * KT + kt) * 4 + r) * 64 + lane;
          const float s = (red[0][idx] + red[1][idx]) + (red[2][idx] + red[3][idx]);
          const int co = co0 + 16 * mt + 4 * q + r, c = c0 + 16 * kt + il;
          if (co < A.Cout && c < A.C) {
            if (A.slab) A.slab[(size_t)blockIdx.x * A.slab_stride + (size_t)(tap * A.CoutP + co) * A.C + c] = s;
            else atomicAdd(A.dWp + ((size_t)(tap * A.CoutP + co) * A.Ktot + A.koff + c), s);
          }
        }
  }
}

// ------------------------------------------------------------------ wgrad of stride-1 "same" convolutions: LDS tiles
// k_conv_wgrad re-reads gy and the input once per (tap, channel block) with dword loads.  Here a workgroup owns a
// (32 co) x (32 c) block of dW for ALL taps (TAPS x 2 x 2 MFMA tiles = 36 accumulators for 3x3) and walks over
// spatial tiles of TH x TW = 128 output pixels: per tile it stages gy [pixel][co] and the input halo tile
// [halo pixel][c] ONCE (coalesced 16-B global loads, stored transposed as [channel][pixel] so that the reduction
// index — the pixel — is contiguous for the MFMA fragments), then every tap reads its shifted window from LDS.
// Each wave reduces its own 32 pixels; the four partial sums meet in LDS at the end and are added to dWp with one
// float atomic per element per workgroup.
#define WT_GP 132
// gy^T rows are [channel][pixel ^ WT_SWZ(channel)]: the transposed store of a staged float4 (lane = 4 channels of one pixel, 8
// lanes per pixel) put 32 lanes on 8 banks (pitch 132 = 4 mod 32: channel quads c4 and c4 + 2 collide, 4-way) — measured
// SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.52 in the 3x3 wgrad (profiles/r03/pmc_c4_step.json).  XOR-ing the pixel with
// 4 * (channel quad & 7) spreads the eight channel quads over all 32 banks and keeps groups of 4 consecutive pixels (the
// 16-byte MFMA fragments) intact.  Readers apply the same XOR (an involution).
#define WT_SWZ(ch) ((((ch) >> 2) & 7) << 2)
#ifndef WT_DBUF
#define WT_DBUF(TAPS) ((TAPS) == 9)
#endif
// RAG: channel counts that are not multiples of 4 (the U-Net's input conv has 1 or 3 input channels, its output conv 1 or 3
// output channels): the staging loads go element by element with a channel mask instead of 16 bytes at a time.  Those two
// layers ran on k_conv_wgrad at 1.2 ms each per C4 step (rocprofv3), 1.8 % of it.
template <int TH, int TW, int TAPS, bool RAG = false>
__global__ void __launch_bounds__(256, 2) k_wgrad_tile(WgradArgs A, int tiles_x, int tiles_y, int tiles_per_wg, int n_tiles) {
  extern __shared__ __attribute__((aligned(16))) float wt_lds[];
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, il = lane & 15, q = lane >> 4;
  const ConvGeom g = A.g;
  constexpr int KW = (TH == 1) ? TAPS : (TAPS == 9 ? 3 : 1), KH = TAPS / KW;    // compile-time: constant index arithmetic
  constexpr int HH = TH + KH - 1, HW = TW + KW - 1, halo = HH * HW;
  // pitch of an input channel row: = 5 (mod 8) keeps the (channel il, pixel 4q) scalar reads at <= 3 lanes per bank
  // (brute-forced; 3 (mod 8) is equivalent) and makes the 3x3 buffer 80.1 KB, so TWO such workgroups share a CU
  constexpr int IP = ((halo + 2) & ~7) + 5;
  constexpr int BUF = 32 * WT_GP + 32 * IP;               // one staging buffer: gy^T [32][WT_GP] | input^T [32][IP]
  auto gyT = [&](int b) { return wt_lds + b * BUF; };
  auto inT = [&](int b) { return wt_lds + b * BUF + 32 * WT_GP; };
  const int cblocks = (A.C + 31) / 32;
  const int coblk = blockIdx.y / cblocks, cblk = blockIdx.y - coblk * cblocks;
  const int co0 = coblk * 32, c0 = cblk * 32;
  const int t_beg = blockIdx.x * tiles_per_wg, t_end = min(t_beg + tiles_per_wg, n_tiles);
  if (t_beg >= t_end) return;

  f32x4 acc[TAPS][2][2];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int m = 0; m < 2; ++m) { acc[t][m][0] = f32x4{0, 0, 0, 0}; acc[t][m][1] = f32x4{0, 0, 0, 0}; }

  constexpr int MAXIN = ((TH == 1 ? 1 : TH + 2) * (TW + 2) * 8 + 255) / 256;   // 1-D convolutions have no vertical halo
  f32x4 sg[4], si[MAXIN];
  const int up = g.ups ? 1 : 0;          // input nearest-upsampled 2x on the fly (folded Upsample)
  const int n_in = halo * 8;
  auto stage_load = [&](int tile) {
    int bx = tile;
    const int tx_i = bx % tiles_x; bx /= tiles_x;
    const int ty_i = bx % tiles_y;
    const int n = bx / tiles_y;
    const int y0 = ty_i * TH, x0 = tx_i * TW;
#pragma unroll
    for (int k = 0; k < 4; ++k) {                           // gy: 128 pixels x 8 column quads
      const int idx = tid + 256 * k;
      const int p = idx >> 3, c4 = idx & 7;
      const int py = p / TW, px = p - py * TW;
      const int oy = y0 + py, ox = x0 + px, co = co0 + 4 * c4;
      f32x4 v = {0, 0, 0, 0};
      if (oy < g.Ho && ox < g.Wo && co < A.Cout) {
        const float* gp = A.gy + (((size_t)n * g.Ho + oy) * g.Wo + ox) * A.Cout + co;
        if (!RAG) v = *reinterpret_cast<const f32x4*>(gp);
        else {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (co + r < A.Cout) v[r] = gp[r];
        }
      }
      sg[k] = v;
    }
#pragma unroll
    for (int k = 0; k < MAXIN; ++k) {
      const int idx = tid + 256 * k;
      f32x4 v = {0, 0, 0, 0};
      if (idx < n_in) {
        const int hp = idx >> 3, c4 = idx & 7;
        const int hy = hp / HW, hx = hp - hy * HW;
        const int iy = y0 + hy - g.padH, ix = x0 + hx - g.padW, c = c0 + 4 * c4;   // on the 2x grid if ups
        if (iy >= 0 && iy < (g.Hi << up) && ix >= 0 && ix < (g.Wi << up) && c < A.C) {
          const float* sp = A.src + (((size_t)n * g.Hi + (iy >> up)) * g.Wi + (ix >> up)) * A.C + c;
          if (!RAG) v = *reinterpret_cast<const f32x4*>(sp);
          else {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (c + r < A.C) v[r] = sp[r];
          }
        }
      }
      si[k] = v;
    }
  };
  auto stage_store = [&](int b) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int idx = tid + 256 * k;
      const int p = idx >> 3, c4 = idx & 7;
#pragma unroll
      for (int r = 0; r < 4; ++r) gyT(b)[(4 * c4 + r) * WT_GP + (p ^ WT_SWZ(4 * c4))] = sg[k][r];
    }
#pragma unroll
    for (int k = 0; k < MAXIN; ++k) {
      const int idx = tid + 256 * k;
      if (idx < n_in) {
        const int hp = idx >> 3, c4 = idx & 7;
#pragma unroll
        for (int r = 0; r < 4; ++r) inT(b)[(4 * c4 + r) * IP + hp] = si[k][r];
      }
    }
  };

  unsigned tmask = (1u << TAPS) - 1u;
  if (cblk < 16 && A.tm_c[cblk]) tmask &= A.tm_c[cblk];
  if (coblk < 16 && A.tm_o[coblk]) tmask &= A.tm_o[coblk];
  const bool do_bias = A.dbias != nullptr && cblk == 0;
  const int tiles_per_sample = tiles_x * tiles_y;
  float bsum = 0.f;                                        // thread (co = tid>>3, 16-pixel part = tid&7)
  // LDS buffering: the 3x3 variant is register-bound to two workgroups per CU and keeps two LDS buffers (one barrier
  // per tile); the others have registers for three and keep ONE buffer (34 KB) so that three fit — the next tile
  // still travels global -> registers under this tile's MFMAs, only its LDS store waits for an extra barrier.
  constexpr bool DBUF = WT_DBUF(TAPS);
  stage_load(t_beg);
  stage_store(0);
  __syncthreads();
  int cur = 0;
  for (int tile = t_beg; tile < t_end; ++tile) {
    const bool more = tile + 1 < t_end;
    if (more) stage_load(tile + 1);
    const float* gT = gyT(cur);
    const float* iT = inT(cur);
    if (do_bias && tile / tiles_per_sample < A.n_bias) {   // bias gradient as a by-product of the staged gy tile
      const float* gr = gT + (tid >> 3) * WT_GP + 16 * (tid & 7);
      f32x4 t4 = *reinterpret_cast<const f32x4*>(gr);
      t4 += *reinterpret_cast<const f32x4*>(gr + 4);
      t4 += *reinterpret_cast<const f32x4*>(gr + 8);
      t4 += *reinterpret_cast<const f32x4*>(gr + 12);
      bsum += (t4[0] + t4[1]) + (t4[2] + t4[3]);
    }
    // (Round 3 tried a branch-free tap loop — no per-tap `continue`, the next tap's LDS fragments requested ahead: hipcc then
    // keeps more fragments live, spills 19 registers at the 256-register budget of two workgroups per CU, and the C4 step
    // got SLOWER, 125.7 -> 133.6 ms.  The per-tap branch stays; the second resident wave covers the LDS latency.)
#pragma unroll
    for (int pg2 = 0; pg2 < 2; ++pg2) {
      const int pg = 2 * w + pg2;                           // this wave's 16-pixel group
      const int p0 = 16 * pg + 4 * q;                       // pixels p0..p0+3 <-> MFMA steps r = 0..3
      const int ty = p0 / TW, tx = p0 - ty * TW;
      f32x4 a[2];
      a[0] = *reinterpret_cast<const f32x4*>(gT + il * WT_GP + (p0 ^ WT_SWZ(il)));
      a[1] = *reinterpret_cast<const f32x4*>(gT + (16 + il) * WT_GP + (p0 ^ WT_SWZ(16 + il)));
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        if (!((tmask >> t) & 1u)) continue;                 // this (tap, block) of the weight is structurally zero
        const int kh = t / KW, kw = t - kh * KW;
        const int hb = (ty + kh) * HW + tx + kw;            // halo index of pixel p0 shifted by the tap
        float b[2][4];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) b[kt][r] = iT[(16 * kt + il) * IP + hb + r];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            acc[t][m][0] = mfma16c(a[m][r], b[0][r], acc[t][m][0]);
            acc[t][m][1] = mfma16c(a[m][r], b[1][r], acc[t][m][1]);
          }
      }
    }
    if (DBUF) {
      if (more) stage_store(cur ^ 1);
      __syncthreads();
      cur ^= 1;
    } else {
      __syncthreads();                                       // every wave is done reading the buffer
      if (more) stage_store(0);
      __syncthreads();
    }
  }
  if (do_bias) {
    bsum += __shfl_xor(bsum, 1, 64);
    bsum += __shfl_xor(bsum, 2, 64);
    bsum += __shfl_xor(bsum, 4, 64);
    const int co = co0 + (tid >> 3);
    if ((tid & 7) == 0 && co < A.Cout) {
      if (A.slab) A.slab[(size_t)blockIdx.x * A.slab_stride + (size_t)TAPS * A.CoutP * A.C + co] = bsum;
      else atomicAdd(A.dbias + co, bsum);
    }
  }
  // cross-wave reduction through LDS (reuse the staging area): up to four taps per round — every wave parks its partial
  // tiles of the round's taps, one barrier, then wave w adds the four partials of tap (round base + w) in wave order and
  // stores that tap's 32 x 32 block.  (One tap per round with wave 0 doing every store was 18 barriers and 144 scattered
  // stores by a single wave per workgroup: 10-15 % of a workgroup's time at 8 tiles per workgroup.)
  float* red = wt_lds;
  constexpr int TR = TAPS == 9 ? 4 : (TAPS == 3 ? 2 : 1);   // taps per round: TR * 4 waves * 1024 floats fit the staging area
  constexpr int STG = (WT_DBUF(TAPS) ? 2 : 1) * BUF;
  static_assert(TR * 4096 <= (STG > 4096 ? STG : 4096), "reduction scratch exceeds the staging area");
  __syncthreads();
#pragma unroll
  for (int t0 = 0; t0 < TAPS; t0 += TR) {
#pragma unroll
    for (int tt = 0; tt < TR; ++tt) {
      const int t = t0 + tt;
      if (t < TAPS) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(tt * 4 + w) * 1024 + ((m * 2 + kt) * 4 + r) * 64 + lane] = acc[t][m][kt][r];
      }
    }
    __syncthreads();
    {
      const int tt = w, t = t0 + tt;                       // wave-uniform
      // a skipped (structurally zero) block still stores its zeros into a slab (the slab reduction reads every slot)
      if (tt < TR && t < TAPS && (((tmask >> t) & 1u) || A.slab)) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int idx = tt * 4096 + ((m * 2 + kt) * 4 + r) * 64 + lane;
              const float sum = (red[idx] + red[1024 + idx]) + (red[2048 + idx] + red[3072 + idx]);
              const int co = co0 + 16 * m + 4 * q + r, c = c0 + 16 * kt + il;
              if (co < A.Cout && c < A.C) {
                if (A.slab) A.slab[(size_t)blockIdx.x * A.slab_stride + (size_t)(t * A.CoutP + co) * A.C + c] = sum;
                else atomicAdd(A.dWp + ((size_t)(t * A.CoutP + co) * A.Ktot + A.koff + c), sum);
              }
            }
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------ wgrad of 3x3 "same" convolutions, OUTPUT-stationary waves
// k_wgrad_tile splits a tile's 128 pixels over the four waves: every wave carries the full 32 x 32 x 9 block (144
// accumulator registers -> 251 VGPRs, two workgroups per CU, nothing left for the compiler to keep LDS fragments in flight)
// and the four partial blocks meet through LDS at the end.  Here the waves split the OUTPUT instead: wave (m, kt) owns the
// 16 co x 16 c tile (m, kt) of all nine taps — 36 accumulator registers — and walks all 128 pixels of every tile itself:
//   * no cross-wave reduction, no epilogue barriers: a wave stores its own tiles;
//   * ~130 VGPRs: three workgroups (12 waves) per CU instead of two (8);
//   * a tap row's fragments overlap: pixels 4q..4q+3 shifted by kw = 0..2 are 6 consecutive halo values, read once per
//     (pixel group, kh) — 9 LDS read pairs per 36 MFMAs.
// Same staging (coalesced 16-byte global loads one tile ahead, transposed [channel][pixel] LDS images), ONE LDS buffer
// (40 KB), same slab / bias outputs as k_wgrad_tile<8, 16, 9>.  Aligned channel counts, no tap masks, no folded upsample
// restrictions beyond the old kernel's.
__global__ void __launch_bounds__(256, 3) k_wgrad_tile9(WgradArgs A, int tiles_x, int tiles_y, int tiles_per_wg, int n_tiles) {
  extern __shared__ __attribute__((aligned(16))) float wt_lds[];
  constexpr int TH = 8, TW = 16, KW = 3, KH = 3, TAPS = 9;
  constexpr int HH = TH + KH - 1, HW = TW + KW - 1, halo = HH * HW;
  constexpr int IP = ((halo + 2) & ~7) + 5;
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, il = lane & 15, q = lane >> 4;
  const int m = w >> 1, kt = w & 1;                         // this wave's 16 x 16 output tile of every tap
  const ConvGeom g = A.g;
  float* gT = wt_lds;                                       // gy^T    [32][WT_GP]
  float* iT = wt_lds + 32 * WT_GP;                          // input^T [32][IP]
  const int cblocks = (A.C + 31) / 32;
  const int coblk = blockIdx.y / cblocks, cblk = blockIdx.y - coblk * cblocks;
  const int co0 = coblk * 32, c0 = cblk * 32;
  const int t_beg = blockIdx.x * tiles_per_wg, t_end = min(t_beg + tiles_per_wg, n_tiles);
  if (t_beg >= t_end) return;
  f32x4 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t) acc[t] = f32x4{0, 0, 0, 0};
  constexpr int MAXIN = (halo * 8 + 255) / 256;
  f32x4 sg[4], si[MAXIN];
  const int up = g.ups ? 1 : 0;
  constexpr int n_in = halo * 8;
  auto stage_load = [&](int tile) {
    int bx = tile;
    const int tx_i = bx % tiles_x; bx /= tiles_x;
    const int ty_i = bx % tiles_y;
    const int n = bx / tiles_y;
    const int y0 = ty_i * TH, x0 = tx_i * TW;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int idx = tid + 256 * k;
      const int p = idx >> 3, c4 = idx & 7;
      const int py = p / TW, px = p - py * TW;
      const int oy = y0 + py, ox = x0 + px, co = co0 + 4 * c4;
      f32x4 v = {0, 0, 0, 0};
      if (oy < g.Ho && ox < g.Wo && co < A.Cout)
        v = *reinterpret_cast<const f32x4*>(A.gy + (((size_t)n * g.Ho + oy) * g.Wo + ox) * A.Cout + co);
      sg[k] = v;
    }
#pragma unroll
    for (int k = 0; k < MAXIN; ++k) {
      const int idx = tid + 256 * k;
      f32x4 v = {0, 0, 0, 0};
      if (idx < n_in) {
        const int hp = idx >> 3, c4 = idx & 7;
        const int hy = hp / HW, hx = hp - hy * HW;
        const int iy = y0 + hy - g.padH, ix = x0 + hx - g.padW, c = c0 + 4 * c4;
        if (iy >= 0 && iy < (g.Hi << up) && ix >= 0 && ix < (g.Wi << up) && c < A.C)
          v = *reinterpret_cast<const f32x4*>(A.src + (((size_t)n * g.Hi + (iy >> up)) * g.Wi + (ix >> up)) * A.C + c);
      }
      si[k] = v;
    }
  };
  auto stage_store = [&]() {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int idx = tid + 256 * k;
      const int p = idx >> 3, c4 = idx & 7;
#pragma unroll
      for (int r = 0; r < 4; ++r) gT[(4 * c4 + r) * WT_GP + (p ^ WT_SWZ(4 * c4))] = sg[k][r];
    }
#pragma unroll
    for (int k = 0; k < MAXIN; ++k) {
      const int idx = tid + 256 * k;
      if (idx < n_in) {
        const int hp = idx >> 3, c4 = idx & 7;
#pragma unroll
        for (int r = 0; r < 4; ++r) iT[(4 * c4 + r) * IP + hp] = si[k][r];
      }
    }
  };
  const bool do_bias = A.dbias != nullptr && cblk == 0;
  const int tiles_per_sample = tiles_x * tiles_y;
  float bsum = 0.f;                                        // thread (co = tid>>3, 16-pixel part = tid&7)
  stage_load(t_beg);
  stage_store();
  __syncthreads();
  const float* ga = gT + (16 * m + il) * WT_GP;            // + ((16 pg + 4 q) ^ swizzle of this lane's channel)
  const int gsw = WT_SWZ(16 * m + il);
  const float* ib = iT + (16 * kt + il) * IP + 4 * q;      // + (pg + kh) * HW   (TW = 16: pixel group pg is tile row pg)
  for (int tile = t_beg; tile < t_end; ++tile) {
    const bool more = tile + 1 < t_end;
    if (more) stage_load(tile + 1);
    if (do_bias && tile / tiles_per_sample < A.n_bias) {
      const float* gr = gT + (tid >> 3) * WT_GP + 16 * (tid & 7);
      f32x4 t4 = *reinterpret_cast<const f32x4*>(gr);
      t4 += *reinterpret_cast<const f32x4*>(gr + 4);
      t4 += *reinterpret_cast<const f32x4*>(gr + 8);
      t4 += *reinterpret_cast<const f32x4*>(gr + 12);
      bsum += (t4[0] + t4[1]) + (t4[2] + t4[3]);
    }
#pragma unroll
    for (int pg = 0; pg < 8; ++pg) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(ga + ((16 * pg + 4 * q) ^ gsw));
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const float* vp = ib + (pg + kh) * HW;
        float v[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) v[j] = vp[j];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) acc[kh * 3 + kw] = mfma16c(a[r], v[kw + r], acc[kh * 3 + kw]);
      }
    }
    __syncthreads();                                        // every wave is done reading the buffer
    if (more) { stage_store(); __syncthreads(); }
  }
  if (do_bias) {
    bsum += __shfl_xor(bsum, 1, 64);
    bsum += __shfl_xor(bsum, 2, 64);
    bsum += __shfl_xor(bsum, 4, 64);
    const int co = co0 + (tid >> 3);
    if ((tid & 7) == 0 && co < A.Cout) {
      if (A.slab) A.slab[(size_t)blockIdx.x * A.slab_stride + (size_t)TAPS * A.CoutP * A.C + co] = bsum;
      else atomicAdd(A.dbias + co, bsum);
    }
  }
  const int c = c0 + 16 * kt + il;
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = co0 + 16 * m + 4 * q + r;
      if (co < A.Cout && c < A.C) {
        if (A.slab) A.slab[(size_t)blockIdx.x * A.slab_stride + (size_t)(t * A.CoutP + co) * A.C + c] = acc[t][r];
        else atomicAdd(A.dWp + ((size_t)(t * A.CoutP + co) * A.Ktot + A.koff + c), acc[t][r]);
      }
    }
}

// ------------------------------------------------------------------ wgrad of 1x1 convolutions: pixel-streaming kernel
// dW[co][c] = sum_p gy[p][co] x[p][c] with the pixel as the reduction index: 2 Cout C / (4 (Cout + C)) = 8..48 FLOPs per
// byte, i.e. HBM-bound below ~128 x 128 channels and MFMA-bound above.  k_wgrad_tile<., ., 1> splits dW into 32 x 32
// blocks over blockIdx.y: every block re-stages its 128-pixel tile of gy and x (transposed, scalar LDS stores) for 8
// MFMAs per wave between two barriers, and gy / x are fetched C / 32 and Cout / 32 times — it measured 51 TFLOP/s and
// 2.4 TB/s whatever the shape (profiles/r03/bench_default_c4.json, "1x1 / strided conv wgrad").
// Here ONE workgroup owns the whole C and up to 192 output channels, and streams over pixels: per PX-pixel tile it copies
// the rows [gy | x] into LDS as they lie in memory (16-byte loads and stores, no transposition: the MFMA fragments are
// dword reads, pixel = 16 g + 4 q + r for lane (il, q) and step r, conflict-free with a row pitch = 4 (mod 8) floats), the
// four waves split the OUTPUT block (wave tile MT x NT: MT + NT LDS reads per MT NT MFMAs), two LDS buffers, one barrier
// per tile, the next tile's global loads in flight under this tile's MFMAs.  Every gy / x byte is read once per output-
// channel block (one block for Cout <= 192); the workgroup's partial dW goes to its slab (slot = blockIdx.x, reduced in
// slot order like every other wgrad) or to dWp with float atomics, the bias gradient (column sums of the primal pixels'
// gy rows) comes from the staged tile.
// Pixels per staged tile for a row of W floats: the largest multiple of 16 with PX W <= 8192 floats (2 x 33 KB of LDS, two
// workgroups per CU) whose float4 count is a multiple of 256, so that every thread stages exactly PX W / 1024 items.
__host__ __device__ constexpr int w1_px(int W) {
  int px = (8192 / W) / 16 * 16;
  while (px > 16 && (px * W) % 1024) px -= 16;
  return px > 128 ? 128 : px;
}
template <int MT, int NT, int WM>      // wave grid WM x (4 / WM) over (co, c); the workgroup covers 16 MT WM x 16 NT (4 / WM)
__global__ void __launch_bounds__(256, 2) k_wgrad1x1(WgradArgs A, long Mtot, long Mbias, int tiles_per_wg, long n_tiles) {
  extern __shared__ __attribute__((aligned(16))) float w1_lds[];
  constexpr int WN = 4 / WM, COB = 16 * MT * WM, CB = 16 * NT * WN, W = COB + CB, PITCH = W + 4;
  constexpr int G4 = COB / 4, R4 = W / 4;                       // float4 per row: gy part, both parts
  constexpr int PX = w1_px(W), NST = PX * R4 / 256;             // staged items (float4) per thread and tile
  static_assert(PX * R4 % 256 == 0 && NST >= 1 && NST <= 8, "tile does not divide over the threads");
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, il = lane & 15, q = lane >> 4;
  const int wm = w / WN, wn = w - wm * WN;
  const int co_blk = blockIdx.y * COB;
  const long t_beg = (long)blockIdx.x * tiles_per_wg, t_end = min(t_beg + (long)tiles_per_wg, n_tiles);
  if (t_beg >= t_end) return;
  constexpr int buf_floats = PX * PITCH;
  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0, 0, 0, 0};
  // staging: item k of thread tid = float4 (tid + 256 k) of the tile = (row, j).  Everything about an item except the tile
  // is fixed: its global pointer (advanced by one tile per iteration: two adds per item — the first version recomputed
  // row / column / predicate / 64-bit address per item and tile, ~250 vector instructions per tile in front of 3-4 k MFMA
  // cycles, with a branch around every load) and its LDS offset
  const float* gp[NST];
  int lo[NST], rowk[NST];
  long gstep[NST];
#pragma unroll
  for (int k = 0; k < NST; ++k) {
    const int idx = tid + 256 * k, row = idx / R4, j = idx - row * R4;       // divisions by a compile-time constant
    const bool isg = j < G4;
    gp[k] = isg ? A.gy + (size_t)(t_beg * PX + row) * A.Cout + co_blk + 4 * j : A.src + (size_t)(t_beg * PX + row) * A.C + 4 * (j - G4);
    gstep[k] = (long)PX * (isg ? A.Cout : A.C);
    lo[k] = row * PITCH + 4 * j;
    rowk[k] = row;
  }
  f32x4 stA[NST];
  // all tiles but a ragged last one are whole (Mtot is a multiple of PX for every U-Net layer): no predicate, no branch
  const long n_full = Mtot / PX;
  auto stage_load = [&](f32x4* st, long tile) __attribute__((always_inline)) {
    if (tile < n_full) {
#pragma unroll
      for (int k = 0; k < NST; ++k) st[k] = *reinterpret_cast<const f32x4*>(gp[k]);
    } else {
#pragma unroll
      for (int k = 0; k < NST; ++k) {
        f32x4 v = {0, 0, 0, 0};
        if (tile < n_tiles && tile * PX + rowk[k] < Mtot) v = *reinterpret_cast<const f32x4*>(gp[k]);
        st[k] = v;
      }
    }
#pragma unroll
    for (int k = 0; k < NST; ++k) gp[k] += gstep[k];
  };
  auto stage_store = [&](const f32x4* st, float* buf) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < NST; ++k) *reinterpret_cast<f32x4*>(buf + lo[k]) = st[k];
  };
  // bias gradient = column sums of gy over the primal pixels (p < Mbias, a multiple of 16): the wn == 0 waves add up the A
  // fragments they read anyway (one fma per fragment), the four q lanes of a channel meet at the end
  const bool do_bias = A.dbias != nullptr && wn == 0;
  float ab[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) ab[m] = 0.f;
  const int a_off = 4 * q * PITCH + 16 * wm * MT + il, b_off = 4 * q * PITCH + COB + 16 * wn * NT + il;
  auto compute = [&](const float* buf, long tile) __attribute__((always_inline)) {
#pragma unroll 2
    for (int g = 0; g < PX / 16; ++g) {
      const float* ap = buf + a_off + 16 * g * PITCH;
      const float* bp = buf + b_off + 16 * g * PITCH;
      const float pf = (do_bias && tile * PX + 16 * g < Mbias) ? 1.f : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float a[MT], b[NT];
#pragma unroll
        for (int m = 0; m < MT; ++m) a[m] = ap[r * PITCH + 16 * m];
#pragma unroll
        for (int n = 0; n < NT; ++n) b[n] = bp[r * PITCH + 16 * n];
#pragma unroll
        for (int m = 0; m < MT; ++m) ab[m] = fmaf(pf, a[m], ab[m]);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[m][n] = mfma16c(a[m], b[n], acc[m][n]);
      }
    }
  };
  float* buf0 = w1_lds;
  float* buf1 = w1_lds + buf_floats;
  auto tl = [&](long t) { return t < t_end ? t : n_tiles; };    // past the workgroup's range: the ragged form, nothing loaded
  stage_load(stA, t_beg);
  stage_store(stA, buf0);
  __syncthreads();
  // (two tiles in flight — a second register set — and half-size tiles with four workgroups per CU were both measured:
  // neither is faster, tools/bench_wgrad1x1.py; what paid was taking the address arithmetic out of the tile loop)
  for (long tile = t_beg; tile < t_end; ++tile) {
    stage_load(stA, tl(tile + 1));
    compute(((tile - t_beg) & 1) ? buf1 : buf0, tile);
    stage_store(stA, ((tile - t_beg) & 1) ? buf0 : buf1);
    __syncthreads();
  }
  if (do_bias) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      float t = ab[m];
      t += __shfl_xor(t, 16, 64);
      t += __shfl_xor(t, 32, 64);
      const int co = co_blk + 16 * (wm * MT + m) + il;
      if (q == 0) {
        if (A.slab) A.slab[(size_t)blockIdx.x * A.slab_stride + (size_t)A.CoutP * A.C + co] = t;
        else atomicAdd(A.dbias + co, t);
      }
    }
  }
  if (A.slab) {
    float* sl = A.slab + (size_t)blockIdx.x * A.slab_stride;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          sl[(size_t)(co_blk + 16 * (wm * MT + m) + 4 * q + r) * A.C + 16 * (wn * NT + n) + il] = acc[m][n][r];
  } else {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          atomicAdd(A.dWp + ((size_t)(co_blk + 16 * (wm * MT + m) + 4 * q + r) * A.Ktot + A.koff + 16 * (wn * NT + n) + il), acc[m][n][r]);
  }
}

// ------------------------------------------------------------------ weight (un)packing
// Wp[t][r][kp_off + c] = W[r*sr + (col_off + c)*sc + t*st]   (r < rows, c < ncols); zero elsewhere is
// provided by a memset of Wp before packing.
__global__ void k_pack_w(const float* __restrict__ W, float* __restrict__ Wp, int rows, int ncols, int col_off, int taps,
                         int64_t sr, int64_t sc, int64_t st, int rowsP, int Ktot, int kp_off) {
  const int64_t tot = (int64_t)taps * rows * ncols;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < tot; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % ncols);
    const int r = (int)((e / ncols) % rows);
    const int t = (int)(e / ((int64_t)ncols * rows));
    Wp[((int64_t)t * rowsP + r) * Ktot + kp_off + c] = W[r * sr + (col_off + c) * sc + t * st];
  }
}
// dW[r*sr + (col_off+c)*sc + t*st] (+)= dWp[t][r][kp_off + c]
__global__ void k_unpack_w(float* __restrict__ dW, const float* __restrict__ dWp, int rows, int ncols, int col_off,
                           int taps, int64_t sr, int64_t sc, int64_t st, int rowsP, int Ktot, int kp_off, int accumulate) {
  const int64_t tot = (int64_t)taps * rows * ncols;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < tot; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % ncols);
    const int r = (int)((e / ncols) % rows);
    const int t = (int)(e / ((int64_t)ncols * rows));
    const float v = dWp[((int64_t)t * rowsP + r) * Ktot + kp_off + c];
    float* d = dW + r * sr + (col_off + c) * sc + t * st;
    *d = accumulate ? *d + v : v;
  }
}

// ------------------------------------------------------------------ pointwise dual kernels
// GELU (exact erf form, nn.GELU default: NNUnet1D.py:18,20) on a (primal | tangent) stacked
// batch: hP = g(zP), hT = g'(zP) zT.  half = elements of one half.
__device__ __forceinline__ void gelu012(float z, float& g0, float& g1, float& g2) {
  const float Phi = 0.5f * (1.0f + erff(z * 0.70710678118654752f));
  const float phi = 0.3989422804014327f * __expf(-0.5f * z * z);
  g0 = z * Phi;
  g1 = Phi + z * phi;
  g2 = phi * (2.0f - z * z);
}
__device__ __forceinline__ void silu012(float z, float& s0, float& s1, float& s2) {
  const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-z));
  const float om = 1.0f - sg;
  s0 = z * sg;
  s1 = sg * (1.0f + z * om);
  s2 = sg * om * (2.0f + z * (1.0f - 2.0f * sg));
}

template <int ACT>   // 0 GELU, 1 SiLU
__global__ void k_act_dual_fwd(const float* __restrict__ z, float* __restrict__ h, int64_t half, int dual) {
  const int64_t nq = half >> 2;   // half is a multiple of 4 (channels-last rows of >= 4... checked on host)
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nq; i += (int64_t)gridDim.x * blockDim.x) {
    const f32x4 zp = *reinterpret_cast<const f32x4*>(z + 4 * i);
    f32x4 zt = {0, 0, 0, 0};
    if (dual) zt = *reinterpret_cast<const f32x4*>(z + half + 4 * i);
    f32x4 hp, ht;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float a0, a1, a2;
      if (ACT == 0) gelu012(zp[k], a0, a1, a2); else silu012(zp[k], a0, a1, a2);
      hp[k] = a0; ht[k] = a1 * zt[k];
    }
    *reinterpret_cast<f32x4*>(h + 4 * i) = hp;
    if (dual) *reinterpret_cast<f32x4*>(h + half + 4 * i) = ht;
  }
}
// cotangents (gP,gT) of (hP,hT) -> cotangents of (zP,zT), written over g
template <int ACT>
__global__ void k_act_dual_bwd(const float* __restrict__ z, float* __restrict__ g, int64_t half) {
  const int64_t nq = half >> 2;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nq; i += (int64_t)gridDim.x * blockDim.x) {
    const f32x4 zp = *reinterpret_cast<const f32x4*>(z + 4 * i), zt = *reinterpret_cast<const f32x4*>(z + half + 4 * i);
    f32x4 gp = *reinterpret_cast<const f32x4*>(g + 4 * i), gt = *reinterpret_cast<const f32x4*>(g + half + 4 * i);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float a0, a1, a2;
      if (ACT == 0) gelu012(zp[k], a0, a1, a2); else silu012(zp[k], a0, a1, a2);
      const float np = gp[k] * a1 + gt[k] * (a2 * zt[k]);
      gt[k] = gt[k] * a1;
      gp[k] = np;
    }
    *reinterpret_cast<f32x4*>(g + 4 * i) = gp;
    *reinterpret_cast<f32x4*>(g + half + 4 * i) = gt;
  }
}

// S[n][c] (+)= sum over positions of x[n][pos][c]  (bias / embedding gradients).  Grid (n, position
// chunk): threads along channels, LDS across pixel lanes, one float atomic per (n, c) per block into a
// zeroed S (a per-sample workgroup would leave the chip idle at small batch).
// 32 elements x 8 slot slices per workgroup: every slice adds its slots in order, the slices are added in order.
// out[map(e)] (+)= sum_slot part[slot * stride + e] — the deterministic replacement of float atomics (no run-to-run
// difference in the weight / bias gradients).  map: wgrad image element e = (tap*CoutP + co)*C + c ->
// dWp[(tap*CoutP + co)*Ktot + koff + c] when Ktot > 0, identity otherwise.
// A second, identity-mapped element range [n_elem, n_elem + n_elem2) of the same slots (the bias gradient that the tiled
// wgrad leaves behind its weight image) is reduced into out2 by the trailing workgroups of the SAME launch.
__device__ __forceinline__ void slot_reduce_block(long blk, const float* __restrict__ part, int nslots, long stride, long n_elem,
                                                  float* __restrict__ out, int C, int Ktot, int koff, int accumulate,
                                                  int rowsP, int rows, long n_elem2, float* __restrict__ out2) {
  __shared__ float red[8][32];
  const int el = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const long nb1 = (n_elem + 31) / 32;
  if (blk >= nb1) {                                    // workgroup-uniform
    const long e2 = (blk - nb1) * 32 + el;
    const bool ok2 = e2 < n_elem2;
    float t2 = 0.f;
    if (ok2)
      for (int s = sl; s < nslots; s += 8) t2 += part[(size_t)s * stride + n_elem + e2];
    red[sl][el] = t2;
    __syncthreads();
    if (sl == 0 && ok2) {
      float r = red[0][el];
#pragma unroll
      for (int k = 1; k < 8; ++k) r += red[k][el];
      out2[e2] = accumulate ? out2[e2] + r : r;
    }
    return;
  }
  const long e = blk * 32 + el;
  // padding rows (co >= Cout of a [taps][CoutP][C] image) are never written by the producers
  const bool ok = e < n_elem && (rowsP == 0 || (int)((e / C) % rowsP) < rows);
  float t = 0.f;
  if (ok)
    for (int s = sl; s < nslots; s += 8) t += part[(size_t)s * stride + e];
  red[sl][el] = t;
  __syncthreads();
  if (sl == 0 && ok) {
    float r = red[0][el];
#pragma unroll
    for (int k = 1; k < 8; ++k) r += red[k][el];
    const long o = Ktot > 0 ? (e / C) * Ktot + koff + (e % C) : e;
    out[o] = accumulate ? out[o] + r : r;
  }
}
__global__ void __launch_bounds__(256) k_slot_reduce(const float* __restrict__ part, int nslots, long stride, long n_elem,
                                                      float* __restrict__ out, int C, int Ktot, int koff, int accumulate,
                                                      int rowsP, int rows, long n_elem2, float* __restrict__ out2) {
  slot_reduce_block((long)blockIdx.x, part, nslots, stride, n_elem, out, C, Ktot, koff, accumulate, rowsP, rows, n_elem2, out2);
}
// Many slot reductions in ONE launch (the weight / bias gradients of a whole backward pass: ~140 launches of a few
// microseconds each otherwise): the jobs sit in a device table with the first workgroup of each; a workgroup finds its
// job by bisection.  Same arithmetic, same order as k_slot_reduce.
__global__ void __launch_bounds__(256) k_slot_reduce_batched(const msgm_reduce_job_t* __restrict__ jobs, int n_jobs) {
  int lo = 0, hi = n_jobs - 1;
  const long blk = blockIdx.x;
  while (lo < hi) {                                    // last job whose block_begin <= blk
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].block_begin <= blk) lo = mid; else hi = mid - 1;
  }
  const msgm_reduce_job_t J = jobs[lo];
  slot_reduce_block(blk - J.block_begin, J.part, J.nslots, (long)J.stride, (long)J.n_elem, J.out, J.C, J.Ktot, J.koff, J.accumulate,
                    J.rowsP, J.rows, (long)J.n_elem2, J.out2);
}

__global__ void __launch_bounds__(256) k_colsum(const float* __restrict__ x, float* __restrict__ S, int P, int C, int chunk, int acc) {
  __shared__ float red[256];
  const int n = blockIdx.x;
  const float* xn = x + (size_t)n * P * C;
  const int p0 = blockIdx.y * chunk, p1 = min(p0 + chunk, P);
  for (int cb = 0; cb < C; cb += 256) {
    const int lanes_c = min(C - cb, 256);
    const int rows = 256 / lanes_c;                  // position lanes per channel
    const int c = threadIdx.x % lanes_c, pr = threadIdx.x / lanes_c;
    float s = 0.f;
    if (pr < rows)
      for (int p = p0 + pr; p < p1; p += rows) s += xn[(size_t)p * C + cb + c];
    red[threadIdx.x] = (pr < rows) ? s : 0.f;
    __syncthreads();
    if (pr == 0) {
      float t = 0.f;
      for (int r = 0; r < rows; ++r) t += red[r * lanes_c + c];
      if (acc == 2) S[((size_t)blockIdx.y * gridDim.x + n) * C + cb + c] = t;     // slot mode: S = [chunk][n][C] partials
      else if (gridDim.y == 1 && !acc) S[(size_t)n * C + cb + c] = t;
      else atomicAdd(S + (size_t)n * C + cb + c, t);
    }
    __syncthreads();
  }
}

// x[n][pos][c] += sgn * E[n][c] for one position per sample (border terms of the broadcast-embedding
// channels, NNUnet1D.py:156: zero padding makes taps 0 / 2 miss at l = 0 / L-1).
__global__ void k_add_row(float* __restrict__ x, const float* __restrict__ E, int N, int P, int C, int pos, float sgn) {
  const int64_t tot = (int64_t)N * C;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < tot; e += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(e / C), c = (int)(e - (int64_t)n * C);
    x[((size_t)n * P + pos) * C + c] += sgn * E[e];
  }
}

__global__ void k_gather_row(const float* __restrict__ x, float* __restrict__ out, int N, int P, int C, int pos) {
  const int64_t tot = (int64_t)N * C;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < tot; e += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(e / C), c = (int)(e - (int64_t)n * C);
    out[e] = x[((size_t)n * P + pos) * C + c];
  }
}

// ============================================================ C ABI
static inline hipStream_t S(msgm_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static int check_geom(const msgm_conv_geom_t* g) {
  if (!g || g->N <= 0 || g->Hi <= 0 || g->Wi <= 0 || g->Ho <= 0 || g->Wo <= 0 || g->KH <= 0 || g->KW <= 0 ||
      g->strideH <= 0 || g->padH < 0 || g->strideW <= 0 || g->padW < 0 || (g->mode != 0 && g->mode != 1) ||
      (g->ups != 0 && g->ups != 1))
    return MSGM_E_BADARG;
  if ((int64_t)g->N * g->Ho * g->Wo >= (1ll << 31) || (int64_t)g->N * g->Hi * g->Wi >= (1ll << 31)) return MSGM_E_UNSUPPORTED;
  return MSGM_OK;
}
static ConvGeom to_geom(const msgm_conv_geom_t* g) {
  return ConvGeom{g->N, g->Hi, g->Wi, g->Ho, g->Wo, g->KH, g->KW, g->strideH, g->padH, g->strideW, g->padW, g->mode, g->ups};
}

extern "C" {

// is this (geometry, channels) served by the halo-tile kernel?  (the only one that can transform its input)
static bool conv_tile_eligible(const msgm_conv_geom_t* geom, int32_t C0, const float* src1, int32_t C1, int32_t CoutP) {
  const bool fast = (C0 % 16 == 0) && (!src1 || C1 % 16 == 0);
  const int ups_sh = geom->ups ? 1 : 0;
  const bool same = geom->strideH == 1 && geom->strideW == 1 && (!geom->ups || geom->mode == 0) &&
                    (geom->Hi << ups_sh) == geom->Ho && (geom->Wi << ups_sh) == geom->Wo &&
                    (geom->KH & 1) && (geom->KW & 1) && geom->padH == (geom->KH - 1) / 2 && geom->padW == (geom->KW - 1) / 2 &&
                    geom->KH <= 3 && geom->KW <= 3;
  return same && fast && CoutP % 32 == 0 && (int64_t)geom->Ho * geom->Wo >= 64 &&
         (geom->Ho > 1 ? geom->KH == geom->KW : geom->KH == 1) &&          // square kernels in 2-D, KH = 1 in 1-D
         !getenv("MSGM_NO_CONV_TILE");
}

int msgm_conv_input_transform_supported(const msgm_conv_geom_t* geom, int32_t C0, int32_t C1, int32_t CoutP) {
  if (check_geom(geom)) return 0;
  static const float dummy = 0.f;
  return conv_tile_eligible(geom, C0, C1 > 0 ? &dummy : nullptr, C1, CoutP) ? 1 : 0;
}

static bool conv_wino_eligible(const msgm_conv_geom_t* geom, int32_t C0, int32_t C1, int32_t CoutP) {
  const int ups_sh = geom->ups ? 1 : 0;
  return geom->mode == 0 && geom->KH == 3 && geom->KW == 3 && geom->strideH == 1 && geom->strideW == 1 && geom->padH == 1 &&
         geom->padW == 1 && (geom->Hi << ups_sh) == geom->Ho && (geom->Wi << ups_sh) == geom->Wo && geom->Ho % 16 == 0 &&
         geom->Wo % 16 == 0 && C0 % 16 == 0 && C1 % 16 == 0 && CoutP % 32 == 0;
}

int msgm_conv_wino_supported(const msgm_conv_geom_t* geom, int32_t C0, int32_t C1, int32_t CoutP) {
  if (check_geom(geom)) return 0;
  return conv_wino_eligible(geom, C0, C1, CoutP) ? 1 : 0;
}

int msgm_wino_pack_weights_batched(const msgm_pack_job_t* jobs, int32_t n_jobs, msgm_stream_t stream) {
  if (!jobs || n_jobs <= 0) return MSGM_E_BADARG;
  hipLaunchKernelGGL(k_wino_pack_batched, dim3(8, (unsigned)n_jobs), dim3(256), 0, S(stream), jobs);
  return msgm_check_launch();
}

int msgm_conv_forward_wino(const msgm_conv_geom_t* geom, const float* src0, int32_t C0, const float* src1, int32_t C1,
                           const float* WpW, int32_t Cout, int32_t CoutP, int32_t Ktot, const float* bias,
                           const float* samp_bias, int32_t n_bias, int32_t n_samp, float* out, int32_t accumulate,
                           const msgm_conv_fuse_t* fuse, msgm_stream_t stream) {
  int rc = check_geom(geom);
  if (rc) return rc;
  if (fuse && ((fuse->in_scale == nullptr) != (fuse->in_shift == nullptr) || (fuse->in_act != 0 && fuse->in_act != 1)))
    return MSGM_E_BADARG;
  if (!src0 || !WpW || !out || C0 <= 0 || Cout <= 0 || (src1 && C1 <= 0)) return MSGM_E_BADARG;
  if (!conv_wino_eligible(geom, C0, src1 ? C1 : 0, CoutP)) return MSGM_E_UNSUPPORTED;
  const int k0 = ((C0 + 15) / 16) * 16, k1 = src1 ? ((C1 + 15) / 16) * 16 : 0;
  if (Ktot != k0 + k1 || CoutP < Cout) return MSGM_E_BADARG;
  ConvArgs A{};
  A.g = to_geom(geom);
  A.src[0] = src0; A.C[0] = C0; A.koff[0] = 0;
  A.src[1] = src1; A.C[1] = src1 ? C1 : 0; A.koff[1] = k0;
  A.nsrc = src1 ? 2 : 1;
  A.Wp = WpW; A.Cout = Cout; A.CoutP = CoutP; A.Ktot = Ktot;
  A.bias = bias; A.samp_bias = samp_bias; A.n_bias = n_bias; A.n_samp = n_samp; A.out = out; A.accumulate = accumulate;
  if (fuse && fuse->chanstats && (Cout & 3)) return MSGM_E_UNSUPPORTED;
  if (fuse) { A.residual = fuse->residual; A.in_scale = fuse->in_scale; A.in_shift = fuse->in_shift; A.in_act = fuse->in_act; }
  const int tiles_x = geom->Wo / 16, tiles_y = geom->Ho / 16;
  if (fuse && fuse->chanstats) { A.cstat = fuse->chanstats; A.cs_S = tiles_x * tiles_y * 4; }   // [N][Ho/16 * Wo/16 * 4][2][Cout]
  const int n_tiles = tiles_x * tiles_y * geom->N, gy = CoutP / 32;
  const size_t lds = (size_t)18 * 18 * CT_P * sizeof(float);
  dim3 grid((unsigned)(8 * gy * ((n_tiles + 7) / 8)));
  // one 32-channel chunk (Ktot = 32: the 64x64 32 -> 32 layers) has nothing to pipeline and pays the extra barriers: 96 vs
  // 104 TFLOP/s as written; from 64 input channels on the LDS weights win, +2 .. +18 % (AFF=1 tools/bench_wino.py)
  static const bool reg_weights = getenv("MSGM_WINO_REGW") != nullptr;    // A/B: the register weight ring everywhere
  static const bool no_p32 = getenv("MSGM_WINO_NO_P32") != nullptr;       // A/B: the 32-channel layers without the persistent form
  static const int p32_min = getenv("MSGM_WINO_P32_MIN") ? atoi(getenv("MSGM_WINO_P32_MIN")) : 1024;      // >= 4 tiles per workgroup (B = 32: 18.6 -> 18.5 ms)
  if (!reg_weights && !no_p32 && !src1 && Ktot == 32 && CoutP == 32 && n_tiles >= p32_min) {
    // one persistent workgroup per CU (>= 8 tiles each): weights resident in LDS, halo double-buffered
    static const int once32 = [] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_wino_p32), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      return 0;
    }();
    (void)once32;
    static const int n_cu = [] { int d = 0, n = 256; (void)hipGetDevice(&d); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d); return n > 0 ? n : 256; }();
    const size_t lds32 = ((size_t)2 * 16 * 2 * 256 + 2 * 18 * 18 * CT_P) * sizeof(float);
    hipLaunchKernelGGL(k_conv_wino_p32, dim3((unsigned)n_cu), dim3(256), lds32, S(stream), A, tiles_x, tiles_y, n_tiles);
    return msgm_check_launch();
  }
  if (reg_weights || Ktot < 64) {
    hipLaunchKernelGGL((k_conv_wino<2, false>), grid, dim3(256), lds, S(stream), A, tiles_x, tiles_y, n_tiles, gy, n_tiles);
    return msgm_check_launch();
  }
  static const int once = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_wino<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return 0;
  }();
  (void)once;
  hipLaunchKernelGGL((k_conv_wino<2, true>), grid, dim3(256), lds + 2 * 4096 * sizeof(float), S(stream), A, tiles_x, tiles_y, n_tiles, gy, n_tiles);
  return msgm_check_launch();
}

// ------------------------------------------------------------------ bf16-split 3x3 forward (opt-in experiment, sampler path)
// Wb[p][e] = piece p (h, m, l) of the packed fp32 image Wp[e] ([tap][CoutP][Ktot]): x = h + m + l, each piece a bf16
__global__ void __launch_bounds__(256) k_b6_split(const float* __restrict__ Wp, __bf16* __restrict__ Wb, long n) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    const float x = Wp[e];
    const __bf16 h = (__bf16)x;
    const float r1 = x - (float)h;
    const __bf16 m = (__bf16)r1;
    Wb[e] = h; Wb[n + e] = m; Wb[2 * n + e] = (__bf16)(r1 - (float)m);
  }
}

static bool conv_b6_eligible(const msgm_conv_geom_t* geom, int32_t C0, int32_t C1, int32_t CoutP) {
  const int ups_sh = geom->ups ? 1 : 0;
  return geom->mode == 0 && geom->KH == 3 && geom->KW == 3 && geom->strideH == 1 && geom->strideW == 1 && geom->padH == 1 &&
         geom->padW == 1 && (geom->Hi << ups_sh) == geom->Ho && (geom->Wi << ups_sh) == geom->Wo && geom->Ho % 16 == 0 && geom->Wo % 16 == 0 &&
         C0 % 32 == 0 && C1 % 32 == 0 && CoutP % 32 == 0;
}

int msgm_conv_b6_supported(const msgm_conv_geom_t* geom, int32_t C0, int32_t C1, int32_t CoutP) {
  if (check_geom(geom)) return 0;
  return conv_b6_eligible(geom, C0, C1, CoutP) ? 1 : 0;
}

int msgm_b6_split_weights(const float* Wp, void* Wb, int64_t n_elem, msgm_stream_t stream) {
  if (!Wp || !Wb || n_elem <= 0) return MSGM_E_BADARG;
  hipLaunchKernelGGL(k_b6_split, dim3((unsigned)grid_for(n_elem, 256, 2048)), dim3(256), 0, S(stream), Wp, static_cast<__bf16*>(Wb), (long)n_elem);
  return msgm_check_launch();
}

int msgm_conv_forward_b6(const msgm_conv_geom_t* geom, const float* src0, int32_t C0, const float* src1, int32_t C1,
                         const void* Wb, int32_t Cout, int32_t CoutP, int32_t Ktot, const float* bias,
                         const float* samp_bias, int32_t n_bias, int32_t n_samp, float* out, int32_t accumulate,
                         const msgm_conv_fuse_t* fuse, msgm_stream_t stream) {
  int rc = check_geom(geom);
  if (rc) return rc;
  if (fuse && ((fuse->in_scale == nullptr) != (fuse->in_shift == nullptr) || (fuse->in_act != 0 && fuse->in_act != 1)))
    return MSGM_E_BADARG;
  if (!src0 || !Wb || !out || C0 <= 0 || Cout <= 0 || (src1 && C1 <= 0)) return MSGM_E_BADARG;
  if (!conv_b6_eligible(geom, C0, src1 ? C1 : 0, CoutP)) return MSGM_E_UNSUPPORTED;
  if (Ktot != C0 + (src1 ? C1 : 0) || CoutP < Cout) return MSGM_E_BADARG;
  if (fuse)
    for (int i = 0; i < 16; ++i)
      if (fuse->tapmask_in[i] || (i < 8 && fuse->tapmask_out[i])) return MSGM_E_UNSUPPORTED;
  ConvArgs A{};
  A.g = to_geom(geom);
  A.src[0] = src0; A.C[0] = C0; A.koff[0] = 0;
  A.src[1] = src1; A.C[1] = src1 ? C1 : 0; A.koff[1] = C0;
  A.nsrc = src1 ? 2 : 1;
  A.Wp = static_cast<const float*>(Wb); A.Cout = Cout; A.CoutP = CoutP; A.Ktot = Ktot;
  A.bias = bias; A.samp_bias = samp_bias; A.n_bias = n_bias; A.n_samp = n_samp; A.out = out; A.accumulate = accumulate;
  if (fuse) { A.residual = fuse->residual; A.in_scale = fuse->in_scale; A.in_shift = fuse->in_shift; A.in_act = fuse->in_act; }
  const int nco = (CoutP % 64 == 0) ? 4 : 2;
  const int tiles_x = (geom->Wo + 15) / 16, tiles_y = (geom->Ho + 15) / 16;
  if (fuse && fuse->chanstats) {                            // the direct wide kernel's layout: one slot per (16x16 tile, wave)
    if (Cout & 3) return MSGM_E_UNSUPPORTED;
    A.cstat = fuse->chanstats; A.cs_S = tiles_x * tiles_y * 4;
  }
  const int n_tiles = tiles_x * tiles_y * geom->N, gy = CoutP / (16 * nco);
  int per = (int)(((int64_t)n_tiles * gy) / (512 * 8));     // two resident workgroups per CU (67 KB of LDS each)
  if (per > 4) per = 4;
  if (per < 1) per = 1;
  const int n_tgrp = (n_tiles + per - 1) / per;
  dim3 grid((unsigned)(8 * gy * ((n_tgrp + 7) / 8)));
  const size_t lds = (size_t)18 * 18 * 52 * sizeof(float);
  static const int once = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_tile<16, 16, 4, 3, 4, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_tile<16, 16, 2, 3, 4, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return 0;
  }();
  (void)once;
  if (nco == 4) hipLaunchKernelGGL((k_conv_tile<16, 16, 4, 3, 4, false, true>), grid, dim3(256), lds, S(stream), A, 0, tiles_x, tiles_y, per, n_tiles, gy, n_tgrp);
  else hipLaunchKernelGGL((k_conv_tile<16, 16, 2, 3, 4, false, true>), grid, dim3(256), lds, S(stream), A, 0, tiles_x, tiles_y, per, n_tiles, gy, n_tgrp);
  return msgm_check_launch();
}

// Which kernel serves a forward convolution, and with what tiling — ONE decision shared by the launcher and by
// msgm_conv_chanstats_slots (the channel-statistics by-product is laid out per (tile, wave) of that tiling).
struct ConvRoute {
  int kind;                 // 0 implicit GEMM from L2, 1 pixel-stationary 1x1, 2 halo tile, 3 / 4 vector-ALU first / last conv
  int pt, kg;               // 1x1: 16-pixel tiles per wave, 16-channel input groups
  bool wide, two_d; int nco, TH, TW, tiles_x, tiles_y;      // halo tile
};
// first / last convolution of the U-Net: 3x3 "same", one source, <= 4 channels on one side and 32 on the other
static bool conv_small_shape(const msgm_conv_geom_t* geom, bool has1, bool masks) {
  static const bool off = getenv("MSGM_NO_CONV_SMALL") != nullptr;         // diagnostic A/B
  return !off && geom->KH == 3 && geom->KW == 3 && geom->strideH == 1 && geom->strideW == 1 &&
         geom->padH == 1 && geom->padW == 1 && !geom->ups && geom->Hi == geom->Ho && geom->Wi == geom->Wo && geom->Ho > 1 &&
         !has1 && !masks;
}
static ConvRoute conv_route(const msgm_conv_geom_t* geom, int32_t C0, bool has1, int32_t C1, int32_t Cout, int32_t CoutP,
                            bool masks, bool both_extra = false /* accumulate AND residual */, bool any_size = false) {
  ConvRoute r{};
  const int64_t Mtot = (int64_t)geom->N * geom->Ho * geom->Wo;
  if (conv_small_shape(geom, has1, masks)) {
    if (C0 <= 4 && Cout == 32 && CoutP == 32) { r.kind = 3; return r; }      // forward, or (mode 1) the dgrad of the output conv
    if (C0 == 32 && Cout <= 4 && geom->mode == 0) { r.kind = 4; return r; }
  }
  const bool fast = (C0 % 16 == 0) && (!has1 || C1 % 16 == 0);
  const int Ktot = ((C0 + 15) / 16) * 16 + (has1 ? ((C1 + 15) / 16) * 16 : 0);
  static const bool no1 = getenv("MSGM_NO_CONV1X1") != nullptr;            // diagnostic A/B
  const int kg = Ktot / 16;
  if (!no1 && geom->KH == 1 && geom->KW == 1 && geom->strideH == 1 && geom->strideW == 1 && geom->padH == 0 && geom->padW == 0 &&
      !geom->ups && geom->Hi == geom->Ho && geom->Wi == geom->Wo && fast && !masks && (Mtot >= 4096 || any_size) &&
      (kg == 2 || kg == 4 || kg == 6 || kg == 8 || kg == 12 || kg == 16) && Cout % 16 == 0 && !both_extra) {
    // resident activations: PT * KG float4 per lane (<= 64 registers).  Fewer pixels per wave (more waves per SIMD) measured
    // equal at 64 input channels and 1.4x slower at 128 (tools/bench_1x1.py)
    // r3: 192 input channels with 32 pixels per wave (96 resident registers) 268 -> 207 us (192 -> 64 at 524 k pixels) and
    // 111 -> 78 us (192 -> 128 at 131 k); 128 channels with 64 pixels per wave were SLOWER (60 -> 74 us), and a 384-channel
    // instance lost to the halo-tile kernel (223 vs 203 us) — tools/bench_1x1.py
    const int pt = kg <= 4 ? 4 : (kg <= 8 ? 2 : (kg == 12 && (geom->Ho * geom->Wo) % 32 == 0 ? 2 : 1));
    if ((geom->Ho * geom->Wo) % (16 * pt) == 0) { r.kind = 1; r.kg = kg; r.pt = pt; return r; }
  }
  static const float dummy = 0.f;
  if (conv_tile_eligible(geom, C0, has1 ? &dummy : nullptr, C1, CoutP)) {
    r.kind = 2;
    r.two_d = geom->Ho > 1;
    r.nco = (CoutP % 64 == 0) ? 4 : 2;
    // 3-tap / 3x3 kernels: 256-pixel tiles (4 MFMA column tiles per wave, one LDS buffer) when the image has them —
    // measured +10..27 % in 2-D and +3..8 % in 1-D over the 128-pixel double-buffered form (tools/exp_wide.py);
    // 1x1 kernels (no tap reuse, HBM-bound) stay on the 128-pixel form, which measured equal or better
    const bool no_wide = getenv("MSGM_NO_CONV_WIDE") != nullptr;            // diagnostic A/B
    bool wide = geom->KW == 3 && !no_wide && (r.two_d ? (geom->Ho >= 16 && geom->Wo >= 16) : geom->Wo >= 256);
    if (wide) {
      // small launches (the 32-row per-GPU shard of C4): below two 256-pixel workgroups per CU the chip is not
      // filled — 128-pixel tiles double the workgroup count (B = 32 step 26.3 -> 25.5 ms; no change at B >= 128)
      const int64_t wtiles = (int64_t)geom->N * ((geom->Wo + (r.two_d ? 15 : 255)) / (r.two_d ? 16 : 256)) *
                             (r.two_d ? (geom->Ho + 15) / 16 : 1) * (CoutP / (16 * r.nco));
      if (wtiles < 512) wide = false;
    }
    r.wide = wide;
    r.TH = r.two_d ? (wide ? 16 : 8) : 1; r.TW = r.two_d ? 16 : (wide ? 256 : 128);
    r.tiles_x = (geom->Wo + r.TW - 1) / r.TW; r.tiles_y = (geom->Ho + r.TH - 1) / r.TH;
    // still fewer than two workgroups per CU (the 16x16 layers of the 32-row shard: 128 tiles x 2 channel blocks): 32
    // instead of 64 output channels per workgroup doubles the count; the input tile is then staged twice, from L2
    static const bool no_split = getenv("MSGM_NO_CONV_COSPLIT") != nullptr;  // diagnostic A/B
    // (not with tap masks: tapmask_out is indexed by the 64-channel block when CoutP % 64 == 0)
    if (!wide && r.nco == 4 && !no_split && !masks && (int64_t)geom->N * r.tiles_x * r.tiles_y * (CoutP / 64) < 512) r.nco = 2;
    return r;
  }
  r.pt = (CoutP >= 64 && CoutP % 64 == 0) ? 2 : 4;         // NT of the k_conv_gemm<MT, NT> instantiation launched below
  return r;
}
static int conv_route_slots(const ConvRoute& r, const msgm_conv_geom_t* geom, int32_t Cout) {
  if (Cout % 4) return 0;
  if (r.kind == 1) { const int P = geom->Ho * geom->Wo; return P % (16 * r.pt) == 0 ? P / (16 * r.pt) : 0; }
  if (r.kind == 2) return r.tiles_x * r.tiles_y * 4;
  if (r.kind == 3) { const int P = geom->Ho * geom->Wo; return P % 64 == 0 ? P / 64 : 0; }
  if (r.kind == 4) return 0;
  const int P = geom->Ho * geom->Wo, px = 16 * r.pt;      // implicit GEMM: a wave owns 16 NT consecutive pixels
  return P % px == 0 ? P / px : 0;
}

int32_t msgm_conv_chanstats_slots(const msgm_conv_geom_t* geom, int32_t C0, int32_t C1, int32_t Cout, int32_t CoutP) {
  if (check_geom(geom) || C0 <= 0 || C1 < 0 || Cout <= 0 || CoutP < Cout || CoutP % 16) return 0;
  return conv_route_slots(conv_route(geom, C0, C1 > 0, C1, Cout, CoutP, false), geom, Cout);
}

int msgm_conv_small_cout_supported(const msgm_conv_geom_t* geom, int32_t C0, int32_t C1, int32_t Cout) {
  if (check_geom(geom) || C1 != 0) return 0;
  return conv_route(geom, C0, false, 0, Cout, 16, false).kind == 4 ? 1 : 0;
}

int msgm_conv_forward(const msgm_conv_geom_t* geom, const float* src0, int32_t C0, const float* src1, int32_t C1,
                      const float* Wp, int32_t Cout, int32_t CoutP, int32_t Ktot, const float* bias,
                      const float* samp_bias, int32_t n_bias, int32_t n_samp, float* out, int32_t accumulate,
                      msgm_stream_t stream) {
  return msgm_conv_forward_fused(geom, src0, C0, src1, C1, Wp, Cout, CoutP, Ktot, bias, samp_bias, n_bias, n_samp, out,
                                 accumulate, nullptr, stream);
}

int msgm_conv_forward_fused(const msgm_conv_geom_t* geom, const float* src0, int32_t C0, const float* src1, int32_t C1,
                            const float* Wp, int32_t Cout, int32_t CoutP, int32_t Ktot, const float* bias,
                            const float* samp_bias, int32_t n_bias, int32_t n_samp, float* out, int32_t accumulate,
                            const msgm_conv_fuse_t* fuse, msgm_stream_t stream) {
  int rc = check_geom(geom);
  if (rc) return rc;
  if (fuse && ((fuse->in_scale == nullptr) != (fuse->in_shift == nullptr) || (fuse->in_act != 0 && fuse->in_act != 1)))
    return MSGM_E_BADARG;
  if (!src0 || !Wp || !out || C0 <= 0 || Cout <= 0 || (src1 && C1 <= 0)) return MSGM_E_BADARG;
  const int k0 = ((C0 + 15) / 16) * 16, k1 = src1 ? ((C1 + 15) / 16) * 16 : 0;
  if (Ktot != k0 + k1 || CoutP % 16 || CoutP < Cout) return MSGM_E_BADARG;
  bool masks = false;
  if (fuse) {
    for (int i = 0; i < 16; ++i) masks = masks || fuse->tapmask_in[i];
    for (int i = 0; i < 8; ++i) masks = masks || fuse->tapmask_out[i];
  }
  const bool both_extra = accumulate && fuse && fuse->residual;
  // A Linear layer (H = W = 1: the embedding / time MLPs, model/unet.py:334-340,128-134) whose rows all take the same bias
  // path is the 1x1 convolution of ONE sample with N pixels: that shape takes the pixel-stationary kernel (a few
  // microseconds at N = 32..1024 rows) instead of the implicit GEMM's chain of L2 round trips (18-37 us, 49 launches per
  // training step at the 32-row shard).
  msgm_conv_geom_t lin = *geom;
  const bool as_pixels = geom->Hi == 1 && geom->Wi == 1 && geom->Ho == 1 && geom->Wo == 1 && geom->KH == 1 && geom->KW == 1 &&
                         geom->N >= 32 && !samp_bias && (!bias || n_bias >= geom->N) && !(fuse && (fuse->in_scale || fuse->chanstats)) &&
                         !getenv("MSGM_NO_LINEAR_AS_PIXELS");
  if (as_pixels) {
    lin.Wi = lin.Wo = geom->N; lin.N = 1; lin.mode = 0;
    const ConvRoute rl = conv_route(&lin, C0, src1 != nullptr, C1, Cout, CoutP, masks, both_extra, true);
    if (rl.kind == 1) { geom = &lin; n_bias = bias ? 1 : 0; n_samp = 0; }
  }
  const ConvRoute rt = conv_route(geom, C0, src1 != nullptr, C1, Cout, CoutP, masks, both_extra, geom == &lin);
  if (fuse && fuse->in_scale && rt.kind != 4 && !conv_tile_eligible(geom, C0, src1, C1, CoutP)) return MSGM_E_UNSUPPORTED;
  ConvArgs A{};
  A.g = to_geom(geom);
  A.src[0] = src0; A.C[0] = C0; A.koff[0] = 0;
  A.src[1] = src1; A.C[1] = src1 ? C1 : 0; A.koff[1] = k0;
  A.nsrc = src1 ? 2 : 1;
  A.Wp = Wp; A.Cout = Cout; A.CoutP = CoutP; A.Ktot = Ktot;
  A.bias = bias; A.samp_bias = samp_bias; A.n_bias = n_bias; A.n_samp = n_samp; A.out = out; A.accumulate = accumulate;
  if (fuse) {
    A.residual = fuse->residual; A.in_scale = fuse->in_scale; A.in_shift = fuse->in_shift; A.in_act = fuse->in_act;
    for (int i = 0; i < 16; ++i) A.tapmask_in[i] = fuse->tapmask_in[i];
    for (int i = 0; i < 8; ++i) A.tapmask_out[i] = fuse->tapmask_out[i];
  }
  const int64_t Mtot = (int64_t)geom->N * geom->Ho * geom->Wo;
  const bool fast = (C0 % 16 == 0) && (!src1 || C1 % 16 == 0);
  if (fuse && fuse->chanstats) {
    A.cs_S = masks ? 0 : conv_route_slots(rt, geom, Cout);
    // the caller sized the buffer from msgm_conv_chanstats_slots(), which sees neither masks nor accumulate + residual
    if (A.cs_S == 0 || A.cs_S != msgm_conv_chanstats_slots(geom, C0, src1 ? C1 : 0, Cout, CoutP)) return MSGM_E_UNSUPPORTED;
    A.cstat = fuse->chanstats;
  }
  // the U-Net's first / last 3x3 convolution: vector ALU, weights in scalar registers
  if (rt.kind == 3) {
    hipLaunchKernelGGL((k_conv3x3_cin_small<32>), dim3((unsigned)((Mtot + 255) / 256)), dim3(256), 0, S(stream), A, (long)Mtot);
    return msgm_check_launch();
  }
  if (rt.kind == 4) {
    const int tiles_x = (geom->Wo + 15) / 16, tiles_y = (geom->Ho + 15) / 16;
    const dim3 grid((unsigned)(tiles_x * tiles_y * geom->N));
    const size_t lds = (size_t)18 * 18 * CT_P * sizeof(float);
    switch (Cout) {
      case 1: hipLaunchKernelGGL((k_conv3x3_cout_small<1>), grid, dim3(256), lds, S(stream), A, tiles_x, tiles_y); break;
      case 2: hipLaunchKernelGGL((k_conv3x3_cout_small<2>), grid, dim3(256), lds, S(stream), A, tiles_x, tiles_y); break;
      case 3: hipLaunchKernelGGL((k_conv3x3_cout_small<3>), grid, dim3(256), lds, S(stream), A, tiles_x, tiles_y); break;
      default: hipLaunchKernelGGL((k_conv3x3_cout_small<4>), grid, dim3(256), lds, S(stream), A, tiles_x, tiles_y); break;
    }
    return msgm_check_launch();
  }
  // 1x1 stride-1 convolution (forward or dgrad): pixel-stationary streaming kernel, no LDS
  if (rt.kind == 1) {
    const int P = geom->Ho * geom->Wo;
    const int ntile = (Cout + 15) / 16;
    const long wgs = (Mtot + 64 * rt.pt - 1) / (64 * rt.pt);
    int split = wgs >= 768 ? 1 : (int)((768 + wgs - 1) / wgs);          // aim at >= 3 workgroups per CU
    if (split > ntile) split = ntile;
    const int ct_per = (ntile + split - 1) / split, gy = (ntile + ct_per - 1) / ct_per;
    const bool extra = accumulate || (fuse && fuse->residual);
#define C1_LAUNCH(PT_, KG_)                                                                                          \
  do {                                                                                                               \
    if (extra) hipLaunchKernelGGL((k_conv1x1<PT_, KG_, 1>), dim3((unsigned)wgs, (unsigned)gy), dim3(256), 0, S(stream), A, P, (long)Mtot, ct_per); \
    else hipLaunchKernelGGL((k_conv1x1<PT_, KG_, 0>), dim3((unsigned)wgs, (unsigned)gy), dim3(256), 0, S(stream), A, P, (long)Mtot, ct_per); \
  } while (0)
    switch (rt.kg) {
      case 2: C1_LAUNCH(4, 2); break;
      case 4: C1_LAUNCH(4, 4); break;
      case 6: C1_LAUNCH(2, 6); break;
      case 8: C1_LAUNCH(2, 8); break;
      case 12: if (rt.pt == 2) C1_LAUNCH(2, 12); else C1_LAUNCH(1, 12); break;
      default: C1_LAUNCH(1, 16); break;
    }
#undef C1_LAUNCH
    return msgm_check_launch();
  }
  // stride-1 "same" convolution (or its dgrad) on a big enough image: halo-tile kernel
  if (rt.kind == 2) {
    const bool two_d = rt.two_d, wide = rt.wide;
    const int nco = rt.nco, TH = rt.TH, TW = rt.TW, tiles_x = rt.tiles_x, tiles_y = rt.tiles_y;
    const int halo = (TH + geom->KH - 1) * (TW + geom->KW - 1);
    const size_t lds = (size_t)(wide ? 1 : 2) * halo * CT_P * sizeof(float);
    const int n_tiles = tiles_x * tiles_y * geom->N, gy = CoutP / (16 * nco);
    // several consecutive tiles per workgroup (the next tile's halo loads overlap this tile's MFMAs) — but only
    // while >= 8 rounds of resident workgroups (3 per CU) remain: below that the tail of the last round costs more
    // than the hidden prologues gain (measured, tools/bench_conv.py)
    const int max_per = getenv("MSGM_CONV_TILES") ? atoi(getenv("MSGM_CONV_TILES")) : 4;
    const int rounds = getenv("MSGM_CONV_ROUNDS") ? atoi(getenv("MSGM_CONV_ROUNDS")) : 8;
    int per = (int)(((int64_t)n_tiles * gy) / (768 * rounds));
    if (per > max_per) per = max_per;
    if (per < 1) per = 1;
    const int n_tgrp = (n_tiles + per - 1) / per;
    dim3 grid((unsigned)(8 * gy * ((n_tgrp + 7) / 8)));
    const int flip = geom->mode;
#define CT_LAUNCH(TH_, TW_, NCO_, KS_, PT_, DB_) \
  hipLaunchKernelGGL((k_conv_tile<TH_, TW_, NCO_, KS_, PT_, DB_>), grid, dim3(256), lds, S(stream), A, flip, tiles_x, tiles_y, per, n_tiles, getenv("MSGM_CONV_NO_XCD") ? -gy : gy, n_tgrp)
    const bool k3 = geom->KW == 3;
    if (two_d) {
      if (wide) { if (nco == 4) CT_LAUNCH(16, 16, 4, 3, 4, false); else CT_LAUNCH(16, 16, 2, 3, 4, false); }
      else if (nco == 4) { if (k3) CT_LAUNCH(8, 16, 4, 3, 2, true); else CT_LAUNCH(8, 16, 4, 1, 2, true); }
      else { if (k3) CT_LAUNCH(8, 16, 2, 3, 2, true); else CT_LAUNCH(8, 16, 2, 1, 2, true); }
    } else {
      if (wide) { if (nco == 4) CT_LAUNCH(1, 256, 4, 3, 4, false); else CT_LAUNCH(1, 256, 2, 3, 4, false); }
      else if (nco == 4) { if (k3) CT_LAUNCH(1, 128, 4, 3, 2, true); else CT_LAUNCH(1, 128, 4, 1, 2, true); }
      else { if (k3) CT_LAUNCH(1, 128, 2, 3, 2, true); else CT_LAUNCH(1, 128, 2, 1, 2, true); }
    }
#undef CT_LAUNCH
    return msgm_check_launch();
  }
  if (CoutP >= 64 && CoutP % 64 == 0) {
    dim3 grid((unsigned)((Mtot + 4 * 32 - 1) / (4 * 32)), (unsigned)(CoutP / 64));
    if (fast) hipLaunchKernelGGL((k_conv_gemm<4, 2, true>), grid, dim3(256), 0, S(stream), A);
    else hipLaunchKernelGGL((k_conv_gemm<4, 2, false>), grid, dim3(256), 0, S(stream), A);
  } else if (CoutP % 32 == 0) {
    dim3 grid((unsigned)((Mtot + 4 * 64 - 1) / (4 * 64)), (unsigned)(CoutP / 32));
    if (fast) hipLaunchKernelGGL((k_conv_gemm<2, 4, true>), grid, dim3(256), 0, S(stream), A);
    else hipLaunchKernelGGL((k_conv_gemm<2, 4, false>), grid, dim3(256), 0, S(stream), A);
  } else {
    dim3 grid((unsigned)((Mtot + 4 * 64 - 1) / (4 * 64)), (unsigned)(CoutP / 16));
    if (fast) hipLaunchKernelGGL((k_conv_gemm<1, 4, true>), grid, dim3(256), 0, S(stream), A);
    else hipLaunchKernelGGL((k_conv_gemm<1, 4, false>), grid, dim3(256), 0, S(stream), A);
  }
  return msgm_check_launch();
}

// launch geometry shared by the launcher and the workspace query
struct WgradPlan { bool tile; int wgs, per, tiles_x, tiles_y, n_tiles, yblocks; int64_t nchunks, chunk; int64_t bias_slots, bias_chunk;
                   bool one; int mt, nt, wm, px; int64_t n_tiles1; };
// The pixel-streaming 1x1 wgrad (k_wgrad1x1): instance (MT, NT, WM) for (Cout, C), or mt = 0
static void wgrad1x1_shape(int C, int Cout, int* mt, int* nt, int* wm) {
  *mt = 0; *nt = 0; *wm = 4;
  if (C != 32 && C != 64 && C != 128 && C != 256) return;
  if (Cout == 32 && C <= 128) { *mt = 1; *nt = C / 32; *wm = 2; return; }
  if (Cout % 64) return;
  *nt = C / 16;
  if (C == 256) { *mt = 1; return; }
  *mt = Cout % 192 == 0 ? 3 : (Cout % 128 == 0 ? 2 : 1);
}
static WgradPlan wgrad_plan(const msgm_conv_geom_t* geom, int C, int Cout, int n_bias) {
  WgradPlan p{};
  const int64_t Mtot = (int64_t)geom->N * geom->Ho * geom->Wo;
  const int taps = geom->KH * geom->KW;
  const int ups_sh = geom->ups ? 1 : 0;
  const bool same = geom->mode == 0 && geom->strideH == 1 && geom->strideW == 1 && (geom->Hi << ups_sh) == geom->Ho &&
                    (geom->Wi << ups_sh) == geom->Wo && (geom->KH & 1) && (geom->KW & 1) && geom->padH == (geom->KH - 1) / 2 &&
                    geom->padW == (geom->KW - 1) / 2 && geom->KH <= 3 && geom->KW <= 3;
  if (taps == 1 && same && !geom->ups && geom->padH == 0 && geom->padW == 0 && ((int64_t)n_bias * geom->Ho * geom->Wo) % 16 == 0 &&
      !getenv("MSGM_NO_WGRAD1X1")) {
    wgrad1x1_shape(C, Cout, &p.mt, &p.nt, &p.wm);
    if (p.mt) {
      p.one = true;
      const int COB = 16 * p.mt * p.wm, Wd = COB + C;
      p.px = w1_px(Wd);                                     // = k_wgrad1x1's PX
      p.yblocks = Cout / COB;
      p.n_tiles1 = (Mtot + p.px - 1) / p.px;
      int wgs = 512 / p.yblocks;                            // two resident workgroups per CU (2 x 33 KB of LDS each)
      int64_t per = (p.n_tiles1 + wgs - 1) / wgs;
      int per_min = COB * C / 4096;                         // a workgroup's slab (COB x C) should stay well below what it reads
      per_min = per_min < 2 ? 2 : (per_min > 8 ? 8 : per_min);
      if (per < per_min) per = p.n_tiles1 < per_min ? p.n_tiles1 : per_min;
      p.per = (int)per;
      p.wgs = (int)((p.n_tiles1 + per - 1) / per);
      return p;
    }
  }
  const bool aligned = C % 4 == 0 && Cout % 4 == 0;      // otherwise only the 2-D 3x3 form has the element-wise staging (RAG)
  p.tile = same && (aligned || (taps == 9 && geom->Ho > 1 && !getenv("MSGM_NO_WGRAD_RAG"))) && (int64_t)geom->Ho * geom->Wo >= 64 &&
           (taps == 1 || taps == 3 || (taps == 9 && geom->Ho > 1)) && (geom->Ho > 1 || geom->KH == 1) && !getenv("MSGM_NO_WGRAD_TILE");
  if (p.tile) {
    const bool two_d = geom->Ho > 1;
    const int TH = two_d ? 8 : 1, TW = two_d ? 16 : 128;
    p.tiles_x = (geom->Wo + TW - 1) / TW; p.tiles_y = (geom->Ho + TH - 1) / TH;
    p.n_tiles = p.tiles_x * p.tiles_y * geom->N;
    p.yblocks = ((Cout + 31) / 32) * ((C + 31) / 32);
    static const int wg_target = getenv("MSGM_WGRAD_WGS") ? (atoi(getenv("MSGM_WGRAD_WGS")) > 0 ? atoi(getenv("MSGM_WGRAD_WGS")) : 1) : 768;    // = 3 resident workgroups per CU (k_wgrad_tile9 and the 1- / 3-tap forms); r3 measured 768 / 1024 / 1536 / 3072: 124.0 / 126.4 / 124.7 / 124.9 ms per C4 step (r2, two workgroups per CU: 1024 was best)
    int wgs = wg_target / p.yblocks;                       // ~4 workgroups per CU overall
    if (wgs < 1) wgs = 1;
    int per = (p.n_tiles + wgs - 1) / wgs;
    static const int min_per = getenv("MSGM_WGRAD_PER") ? (atoi(getenv("MSGM_WGRAD_PER")) > 0 ? atoi(getenv("MSGM_WGRAD_PER")) : 1) : 8;   // >= 8 tiles per workgroup amortise the cross-wave sum + atomics
    if (per < min_per) per = p.n_tiles < min_per ? p.n_tiles : min_per;
    if (per < 1) per = 1;
    p.per = per;
    p.wgs = (p.n_tiles + per - 1) / per;
    return p;
  }
  const int coblocks = (Cout + 31) / 32, cblocks = (C + 63) / 64;
  // aim at ~2048 workgroups overall, at least 256 positions each
  int64_t nchunks = 2048 / (int64_t)(coblocks * cblocks * taps);
  if (nchunks < 1) nchunks = 1;
  int64_t chunk = (Mtot + nchunks - 1) / nchunks;
  if (chunk < 256) chunk = 256;
  chunk = ((chunk + 15) / 16) * 16;
  p.nchunks = (Mtot + chunk - 1) / chunk;
  p.chunk = chunk;
  p.yblocks = coblocks * cblocks;
  if (n_bias > 0) {
    const int64_t Pb = (int64_t)n_bias * geom->Ho * geom->Wo;
    int64_t cch = (Pb + 1023) / 1024;
    if (cch < 64) cch = 64;
    if (cch > Pb) cch = Pb;
    p.bias_chunk = cch;
    p.bias_slots = (Pb + cch - 1) / cch;
  }
  return p;
}

static int wgrad_impl(const msgm_conv_geom_t* geom, const float* gy, const float* src, int32_t C, int32_t koff,
                      float* dWp, int32_t Cout, int32_t CoutP, int32_t Ktot, float* dbias, int32_t n_bias,
                      const uint16_t* tapmask_c32, const uint16_t* tapmask_co32, float* ws, size_t ws_bytes, bool det,
                      msgm_stream_t stream, msgm_reduce_job_t* jobs_out = nullptr, int32_t* n_jobs_out = nullptr);

size_t msgm_conv_wgrad_workspace(const msgm_conv_geom_t* geom, int32_t C, int32_t Cout, int32_t CoutP, int32_t n_bias) {
  if (check_geom(geom) || C <= 0 || Cout <= 0 || CoutP < Cout) return 0;
  const WgradPlan p = wgrad_plan(geom, C, Cout, n_bias);
  const size_t stride = (size_t)geom->KH * geom->KW * CoutP * C + CoutP;
  const size_t slots = (p.tile || p.one) ? (size_t)p.wgs : (size_t)p.nchunks;
  return (slots * stride + (size_t)p.bias_slots * Cout) * sizeof(float);
}

int msgm_conv_wgrad(const msgm_conv_geom_t* geom, const float* gy, const float* src, int32_t C, int32_t koff,
                    float* dWp, int32_t Cout, int32_t CoutP, int32_t Ktot, float* dbias, int32_t n_bias,
                    const uint16_t* tapmask_c32, const uint16_t* tapmask_co32, msgm_stream_t stream) {
  return wgrad_impl(geom, gy, src, C, koff, dWp, Cout, CoutP, Ktot, dbias, n_bias, tapmask_c32, tapmask_co32, nullptr, 0, false,
                    stream);
}

int msgm_conv_wgrad_det(const msgm_conv_geom_t* geom, const float* gy, const float* src, int32_t C, int32_t koff,
                        float* dWp, int32_t Cout, int32_t CoutP, int32_t Ktot, float* dbias, int32_t n_bias,
                        const uint16_t* tapmask_c32, const uint16_t* tapmask_co32, void* workspace, size_t workspace_bytes,
                        msgm_stream_t stream) {
  if (!workspace) return MSGM_E_BADARG;
  return wgrad_impl(geom, gy, src, C, koff, dWp, Cout, CoutP, Ktot, dbias, n_bias, tapmask_c32, tapmask_co32,
                    static_cast<float*>(workspace), workspace_bytes, true, stream);
}

int msgm_conv_wgrad_slabs(const msgm_conv_geom_t* geom, const float* gy, const float* src, int32_t C, int32_t koff,
                          float* dWp, int32_t Cout, int32_t CoutP, int32_t Ktot, float* dbias, int32_t n_bias,
                          const uint16_t* tapmask_c32, const uint16_t* tapmask_co32, void* workspace, size_t workspace_bytes,
                          msgm_reduce_job_t* jobs_out, int32_t* n_jobs_out, msgm_stream_t stream) {
  if (!workspace || !jobs_out || !n_jobs_out) return MSGM_E_BADARG;
  *n_jobs_out = 0;
  return wgrad_impl(geom, gy, src, C, koff, dWp, Cout, CoutP, Ktot, dbias, n_bias, tapmask_c32, tapmask_co32,
                    static_cast<float*>(workspace), workspace_bytes, true, stream, jobs_out, n_jobs_out);
}

int msgm_slot_reduce_batched(const msgm_reduce_job_t* jobs_dev, int32_t n_jobs, int64_t total_blocks, msgm_stream_t stream) {
  if (!jobs_dev || n_jobs <= 0 || total_blocks <= 0 || total_blocks > 0x7fffffffLL) return MSGM_E_BADARG;
  hipLaunchKernelGGL(k_slot_reduce_batched, dim3((unsigned)total_blocks), dim3(256), 0, S(stream), jobs_dev, n_jobs);
  return msgm_check_launch();
}

static int wgrad_impl(const msgm_conv_geom_t* geom, const float* gy, const float* src, int32_t C, int32_t koff,
                      float* dWp, int32_t Cout, int32_t CoutP, int32_t Ktot, float* dbias, int32_t n_bias,
                      const uint16_t* tapmask_c32, const uint16_t* tapmask_co32, float* ws, size_t ws_bytes, bool det,
                      msgm_stream_t stream, msgm_reduce_job_t* jobs_out, int32_t* n_jobs_out) {
  int rc = check_geom(geom);
  if (rc) return rc;
  if (!gy || !src || !dWp || C <= 0 || Cout <= 0 || koff < 0 || koff + C > Ktot || (dbias && n_bias <= 0)) return MSGM_E_BADARG;
  if (det && ws_bytes < msgm_conv_wgrad_workspace(geom, C, Cout, CoutP, dbias ? n_bias : 0)) return MSGM_E_WORKSPACE;
  WgradArgs A{to_geom(geom), gy, src, C, koff, dWp, Cout, CoutP, Ktot, 0, dbias, n_bias, {0}, {0}, nullptr, 0};
  const WgradPlan pl = wgrad_plan(geom, C, Cout, dbias ? n_bias : 0);
  const long img = (long)geom->KH * geom->KW * CoutP * C;
  if (det) { A.slab = ws; A.slab_stride = img + CoutP; }
  for (int i = 0; i < 16; ++i) {
    A.tm_c[i] = (tapmask_c32 && i < (C + 31) / 32) ? tapmask_c32[i] : 0;
    A.tm_o[i] = (tapmask_co32 && i < (Cout + 31) / 32) ? tapmask_co32[i] : 0;
  }
  const int taps = geom->KH * geom->KW;
  auto reduce_slabs = [&](int nslots, bool with_bias) {     // deterministic mode: slabs -> dWp (+ dbias), slot order
    if (jobs_out) {                                          // deferred: the caller batches the reductions of a whole pass
      msgm_reduce_job_t& J = jobs_out[(*n_jobs_out)++];
      J = msgm_reduce_job_t{ws, dWp, with_bias ? dbias : nullptr, (int64_t)A.slab_stride, (int64_t)img, with_bias ? (int64_t)Cout : 0, 0,
                            nslots, C, Ktot, koff, CoutP, Cout, 1, 0};
      return;
    }
    const long nb = (img + 31) / 32 + (with_bias ? (Cout + 31) / 32 : 0);
    hipLaunchKernelGGL(k_slot_reduce, dim3((unsigned)nb), dim3(256), 0, S(stream), (const float*)ws, nslots, A.slab_stride, img,
                       dWp, C, Ktot, koff, 1, CoutP, Cout, with_bias ? (long)Cout : 0L, dbias);
  };
  if (pl.one) {
    if (tapmask_c32 || tapmask_co32) return MSGM_E_UNSUPPORTED;      // a 1x1 kernel has no taps to mask
    const long Mtot = (long)geom->N * geom->Ho * geom->Wo, Mbias = dbias ? (long)n_bias * geom->Ho * geom->Wo : 0;
    const int COB = 16 * pl.mt * pl.wm;
    const size_t lds = (size_t)2 * pl.px * (COB + C + 4) * sizeof(float);
    dim3 grid((unsigned)pl.wgs, (unsigned)pl.yblocks);
#define W1_LAUNCH(MT_, NT_, WM_)                                                                                      \
  do {                                                                                                               \
    static const int once = [] {                                                                                     \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad1x1<MT_, NT_, WM_>),                           \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                             \
      return 0;                                                                                                      \
    }();                                                                                                             \
    (void)once;                                                                                                      \
    hipLaunchKernelGGL((k_wgrad1x1<MT_, NT_, WM_>), grid, dim3(256), lds, S(stream), A, Mtot, Mbias, pl.per, (long)pl.n_tiles1); \
  } while (0)
    const int key = pl.wm * 10000 + pl.mt * 100 + pl.nt;
    switch (key) {
      case 20101: W1_LAUNCH(1, 1, 2); break;
      case 20102: W1_LAUNCH(1, 2, 2); break;
      case 20104: W1_LAUNCH(1, 4, 2); break;
      case 40102: W1_LAUNCH(1, 2, 4); break;
      case 40202: W1_LAUNCH(2, 2, 4); break;
      case 40302: W1_LAUNCH(3, 2, 4); break;
      case 40104: W1_LAUNCH(1, 4, 4); break;
      case 40204: W1_LAUNCH(2, 4, 4); break;
      case 40304: W1_LAUNCH(3, 4, 4); break;
      case 40108: W1_LAUNCH(1, 8, 4); break;
      case 40208: W1_LAUNCH(2, 8, 4); break;
      case 40308: W1_LAUNCH(3, 8, 4); break;
      case 40116: W1_LAUNCH(1, 16, 4); break;
      default: return MSGM_E_UNSUPPORTED;
    }
#undef W1_LAUNCH
    if (det) reduce_slabs(pl.wgs, dbias != nullptr);
    return msgm_check_launch();
  }
  if (pl.tile) {
    const bool two_d = geom->Ho > 1;
    const int TH = two_d ? 8 : 1, TW = two_d ? 16 : 128;
    const int tiles_x = pl.tiles_x, tiles_y = pl.tiles_y, n_tiles = pl.n_tiles, yblocks = pl.yblocks, per = pl.per, wgs = pl.wgs;
    const int halo = (TH + geom->KH - 1) * (TW + geom->KW - 1);
    const int IP = ((halo + 2) & ~7) + 5;
    size_t lds = (size_t)(WT_DBUF(taps) ? 2 : 1) * (32 * WT_GP + 32 * IP) * sizeof(float);
    if (lds < 4096 * sizeof(float)) lds = 4096 * sizeof(float);
    dim3 grid((unsigned)wgs, (unsigned)yblocks);
    // more than 64 KB of dynamic LDS has to be opted into per kernel (160 KB per CU on gfx950)
#define WT_LAUNCH(TH_, TW_, TP_) WT_LAUNCH2(TH_, TW_, TP_, false)
#define WT_LAUNCH2(TH_, TW_, TP_, RAG_)                                                                              \
  do {                                                                                                               \
    static const int once = [] {                                                                                     \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad_tile<TH_, TW_, TP_, RAG_>),                   \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                             \
      return 0;                                                                                                      \
    }();                                                                                                             \
    (void)once;                                                                                                      \
    hipLaunchKernelGGL((k_wgrad_tile<TH_, TW_, TP_, RAG_>), grid, dim3(256), lds, S(stream), A, tiles_x, tiles_y, per, n_tiles); \
  } while (0)
    const bool rag = (C & 3) || (Cout & 3);
    if (two_d) {
      if (taps == 9) {
        static const bool old9 = getenv("MSGM_WGRAD_OLD9") != nullptr;      // A/B: the pixel-split kernel
        if (rag) WT_LAUNCH2(8, 16, 9, true);
        else if (old9 || tapmask_c32 || tapmask_co32) WT_LAUNCH(8, 16, 9);
        else {
          const size_t lds9 = (size_t)(32 * WT_GP + 32 * IP) * sizeof(float);
          hipLaunchKernelGGL(k_wgrad_tile9, grid, dim3(256), lds9, S(stream), A, tiles_x, tiles_y, per, n_tiles);
        }
      }
      else if (taps == 3) WT_LAUNCH(8, 16, 3); else WT_LAUNCH(8, 16, 1);
    }
    else { if (taps == 3) WT_LAUNCH(1, 128, 3); else WT_LAUNCH(1, 128, 1); }
#undef WT_LAUNCH
#undef WT_LAUNCH2
    if (det) reduce_slabs(wgs, dbias != nullptr);
    return msgm_check_launch();
  }
  const int64_t nchunks = pl.nchunks;
  A.chunk = (int)pl.chunk;
  dim3 grid((unsigned)nchunks, (unsigned)pl.yblocks, (unsigned)taps);
  hipLaunchKernelGGL((k_conv_wgrad<2, 4>), grid, dim3(256), 0, S(stream), A);
  if (det) reduce_slabs((int)nchunks, false);
  if (dbias) {                                             // no by-product in this kernel: a separate accumulating column sum
    const int64_t Pb = (int64_t)n_bias * geom->Ho * geom->Wo;
    if (det) {
      float* part = ws + (size_t)nchunks * A.slab_stride;   // [bias_slots][Cout] partials, then slot-ordered sum
      hipLaunchKernelGGL(k_colsum, dim3(1, (unsigned)pl.bias_slots), dim3(256), 0, S(stream), gy, part, (int)Pb, Cout, (int)pl.bias_chunk, 2);
      if (jobs_out) {
        msgm_reduce_job_t& J = jobs_out[(*n_jobs_out)++];
        J = msgm_reduce_job_t{part, dbias, nullptr, (int64_t)Cout, (int64_t)Cout, 0, 0, (int32_t)pl.bias_slots, 1, 0, 0, 0, 0, 1, 0};
      } else
      hipLaunchKernelGGL(k_slot_reduce, dim3((unsigned)((Cout + 31) / 32)), dim3(256), 0, S(stream), (const float*)part,
                         (int)pl.bias_slots, (long)Cout, (long)Cout, dbias, 1, 0, 0, 1, 0, 0, 0L, (float*)nullptr);
    } else {
      hipLaunchKernelGGL(k_colsum, dim3(1, (unsigned)pl.bias_slots), dim3(256), 0, S(stream), gy, dbias, (int)Pb, Cout, (int)pl.bias_chunk, 1);
    }
  }
  return msgm_check_launch();
}

int msgm_pack_weight(const float* W, float* Wp, int32_t rows, int32_t ncols, int32_t col_off, int32_t taps, int64_t sr,
                     int64_t sc, int64_t st, int32_t rowsP, int32_t Ktot, int32_t kp_off, msgm_stream_t stream) {
  if (!W || !Wp || rows <= 0 || ncols <= 0 || taps <= 0 || rowsP < rows || kp_off + ncols > Ktot) return MSGM_E_BADARG;
  hipLaunchKernelGGL(k_pack_w, dim3(grid_for((int64_t)rows * ncols * taps, 256)), dim3(256), 0, S(stream), W, Wp, rows,
                     ncols, col_off, taps, sr, sc, st, rowsP, Ktot, kp_off);
  return msgm_check_launch();
}

// All (un)pack jobs of a network in ONE launch: blockIdx.y = job, blockIdx.x strides over its elements.  The job
// table is static (parameter buckets and packed images do not move), so the host uploads it once.
__global__ void __launch_bounds__(256) k_pack_batched(const msgm_pack_job_t* __restrict__ jobs, int unpack) {
  const msgm_pack_job_t J = jobs[blockIdx.y];
  const int tot = J.taps * J.rows * J.ncols, per_tap = J.ncols * J.rows;        // 32-bit divisions: a packed image has far fewer than 2^31 elements
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += gridDim.x * blockDim.x) {
    const int t = e / per_tap, rc = e - t * per_tap;
    const int r = rc / J.ncols, c = rc - r * J.ncols;
    float* w = J.W + r * J.sr + (J.col_off + c) * J.sc + t * J.st;
    float* p = J.Wp + ((int64_t)t * J.rowsP + r) * J.Ktot + J.kp_off + c;
    if (!unpack) *p = *w;
    else if (J.reserved) atomicAdd(w, *p);               // several images fold into one parameter (paired-stride bias)
    else *w = *p;
  }
}

int msgm_pack_weights_batched(const msgm_pack_job_t* jobs, int32_t n_jobs, int32_t unpack, msgm_stream_t stream) {
  if (!jobs || n_jobs <= 0) return MSGM_E_BADARG;
  hipLaunchKernelGGL(k_pack_batched, dim3(32, (unsigned)n_jobs), dim3(256), 0, S(stream), jobs, unpack);
  return msgm_check_launch();
}

int msgm_unpack_weight(float* dW, const float* dWp, int32_t rows, int32_t ncols, int32_t col_off, int32_t taps, int64_t sr,
                       int64_t sc, int64_t st, int32_t rowsP, int32_t Ktot, int32_t kp_off, int32_t accumulate,
                       msgm_stream_t stream) {
  if (!dW || !dWp || rows <= 0 || ncols <= 0 || taps <= 0 || rowsP < rows || kp_off + ncols > Ktot) return MSGM_E_BADARG;
  hipLaunchKernelGGL(k_unpack_w, dim3(grid_for((int64_t)rows * ncols * taps, 256)), dim3(256), 0, S(stream), dW, dWp, rows,
                     ncols, col_off, taps, sr, sc, st, rowsP, Ktot, kp_off, accumulate);
  return msgm_check_launch();
}

int msgm_act_dual_forward(int32_t act, const float* z, float* h, int64_t half, int32_t dual, msgm_stream_t stream) {
  if (!z || !h || half <= 0 || (half & 3) || (act != 0 && act != 1)) return MSGM_E_BADARG;
  const int grid = grid_for(half / 4, 256);
  if (act == 0) hipLaunchKernelGGL(k_act_dual_fwd<0>, dim3(grid), dim3(256), 0, S(stream), z, h, half, dual);
  else hipLaunchKernelGGL(k_act_dual_fwd<1>, dim3(grid), dim3(256), 0, S(stream), z, h, half, dual);
  return msgm_check_launch();
}

int msgm_act_dual_backward(int32_t act, const float* z, float* g, int64_t half, msgm_stream_t stream) {
  if (!z || !g || half <= 0 || (half & 3) || (act != 0 && act != 1)) return MSGM_E_BADARG;
  const int grid = grid_for(half / 4, 256);
  if (act == 0) hipLaunchKernelGGL(k_act_dual_bwd<0>, dim3(grid), dim3(256), 0, S(stream), z, g, half);
  else hipLaunchKernelGGL(k_act_dual_bwd<1>, dim3(grid), dim3(256), 0, S(stream), z, g, half);
  return msgm_check_launch();
}

int msgm_colsum(const float* x, float* Sout, int32_t N, int32_t P, int32_t C, msgm_stream_t stream) {
  if (!x || !Sout || N <= 0 || P <= 0 || C <= 0) return MSGM_E_BADARG;
  int nch = (1024 + N - 1) / N;
  int chunk = (P + nch - 1) / nch;
  if (chunk < 64) chunk = 64;
  if (chunk > P) chunk = P;
  nch = (P + chunk - 1) / chunk;
  if (nch > 1 && msgm_zero_async(Sout, (size_t)N * C * sizeof(float), S(stream)) != MSGM_OK) return MSGM_E_LAUNCH;
  hipLaunchKernelGGL(k_colsum, dim3(N, nch), dim3(256), 0, S(stream), x, Sout, P, C, chunk, 0);
  return msgm_check_launch();
}

size_t msgm_colsum_workspace(int32_t N, int32_t P, int32_t C) {
  if (N <= 0 || P <= 0 || C <= 0) return 0;
  int nch = (1024 + N - 1) / N;
  int chunk = (P + nch - 1) / nch;
  if (chunk < 64) chunk = 64;
  if (chunk > P) chunk = P;
  nch = (P + chunk - 1) / chunk;
  return nch > 1 ? (size_t)nch * N * C * sizeof(float) : 0;
}

// the same sums without float atomics: per-chunk partials [chunk][N][C] in the workspace, added in chunk order
int msgm_colsum_det(const float* x, float* Sout, int32_t N, int32_t P, int32_t C, void* workspace, size_t workspace_bytes,
                    msgm_stream_t stream) {
  if (!x || !Sout || N <= 0 || P <= 0 || C <= 0) return MSGM_E_BADARG;
  int nch = (1024 + N - 1) / N;
  int chunk = (P + nch - 1) / nch;
  if (chunk < 64) chunk = 64;
  if (chunk > P) chunk = P;
  nch = (P + chunk - 1) / chunk;
  if (nch == 1) {
    hipLaunchKernelGGL(k_colsum, dim3(N, 1), dim3(256), 0, S(stream), x, Sout, P, C, chunk, 0);
    return msgm_check_launch();
  }
  if (!workspace || workspace_bytes < msgm_colsum_workspace(N, P, C)) return MSGM_E_WORKSPACE;
  float* part = static_cast<float*>(workspace);
  hipLaunchKernelGGL(k_colsum, dim3(N, nch), dim3(256), 0, S(stream), x, part, P, C, chunk, 2);
  const long n_elem = (long)N * C;
  hipLaunchKernelGGL(k_slot_reduce, dim3((unsigned)((n_elem + 31) / 32)), dim3(256), 0, S(stream), (const float*)part, nch, n_elem,
                     n_elem, Sout, 1, 0, 0, 0, 0, 0, 0L, (float*)nullptr);
  return msgm_check_launch();
}

int msgm_gather_row(const float* x, float* out, int32_t N, int32_t P, int32_t C, int32_t pos, msgm_stream_t stream) {
  if (!x || !out || N <= 0 || P <= 0 || C <= 0 || pos < 0 || pos >= P) return MSGM_E_BADARG;
  hipLaunchKernelGGL(k_gather_row, dim3(grid_for((int64_t)N * C, 256)), dim3(256), 0, S(stream), x, out, N, P, C, pos);
  return msgm_check_launch();
}

int msgm_add_row(float* x, const float* E, int32_t N, int32_t P, int32_t C, int32_t pos, float sgn, msgm_stream_t stream) {
  if (!x || !E || N <= 0 || P <= 0 || C <= 0 || pos < 0 || pos >= P) return MSGM_E_BADARG;
  hipLaunchKernelGGL(k_add_row, dim3(grid_for((int64_t)N * C, 256)), dim3(256), 0, S(stream), x, E, N, P, C, pos, sgn);
  return msgm_check_launch();
}

}  // extern "C"
