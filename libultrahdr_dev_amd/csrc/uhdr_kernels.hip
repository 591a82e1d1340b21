// uhdr_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels for the Ultra HDR gain-map path.
//
//   k_generate<TF,ALIGNED>   UltraHdr::generateGainMap hot loop   (ref lib/src/ultrahdr.cpp:308-338)
//   k_apply_s4<FMT,MASK>     UltraHdr::applyGainMap hot loop, scale 4, FAST arithmetic (:427-496)
//   k_apply_px<FMT,EXACT>    same loop, any integer scale / any alignment / EXACT arithmetic
//   k_apply_s4_est, k_apply_px_est, k_apply_resolve   EXACT arithmetic as an f32 estimate + the exact path on the pixels in doubt
//   k_apply_lut_s4, k_apply_lut   the loops with the reference's LUT accessors (opt-in LUT mode)
//   k_tonemap_*              UltraHdr::toneMap                    (:517-558)
//   k_convert_yuv<ALIGNED>   JpegR::convertYuv + transformYuv420  (lib/src/jpegr.cpp:1199-1203,
//                                                                  lib/src/gainmapmath.cpp:483-520)
//   k_effect, k_effect_rot   crop / mirror / rotate / resize      (lib/src/editorhelper.cpp:26-360)
//   k_eval_transfer          scalar transfer functions over arrays (diagnostics for the exhaustive tests)
//
// All of it is pointwise byte/float work: no MFMA.  Measured: generate runs within 10 % of what its own
// read pattern reaches on the same box, apply 3-8 % above a build of itself without arithmetic (its writes cap at 5.5 TB/s);
// both got there by taking work off the VALU -- an f32 pre-filter with the exact path deferred to a resolve kernel in generate,
// both transfer functions as line-segment tables in LDS in apply.  Design rules followed here:
//   * one wave64 reads whole contiguous row segments (16 B/lane for P010, 8 B/lane for 8-bit luma),
//     stores are 16 B/lane; no LDS round trip is needed because every input byte is consumed by
//     exactly one lane;
//   * every per-call variant (gamut matrices, transfer function, boosts) is a kernel argument held
//     in SGPRs or a template parameter -- never a per-lane function pointer;
//   * a batch of equally sized images is ONE launch (grid.y = image), descriptors in the kernarg
//     segment, so 64 frames x 2 kernels fill all 256 CUs without per-image launch gaps;
//   * compiled with -ffp-contract=off; float op order is the reference's.
#include "uhdr_kernels.h"

#include <hip/hip_runtime.h>

#include <cmath>

#include "uhdr_device_math.h"
#include "uhdr_wave_scan.h"

namespace uhdr {

// Shepard IDW weights for map scale 4: [table][oy*16 + ox*4 + k]; tables: 0 std, 1 no-right,
// 2 no-bottom, 3 corner (gainmapmath.h:184-228).  Filled by the host at uhdr_hip_init().
__constant__ float c_idw4[4 * 64];

// the standard table once more, arranged for two horizontally adjacent pixels per instruction:
// [oy][pair][tap][pixel of the pair] (an aligned 8-byte scalar load is one packed operand)
__constant__ float c_idw4p[64];

hipError_t upload_idw4(const float* tables) {
  float p[64];
  for (int oy = 0; oy < 4; ++oy)
    for (int pr = 0; pr < 2; ++pr)
      for (int k = 0; k < 4; ++k)
        for (int h = 0; h < 2; ++h) p[((oy * 2 + pr) * 4 + k) * 2 + h] = tables[oy * 16 + (2 * pr + h) * 4 + k];
  hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(c_idw4p), p, sizeof(p));
  if (e != hipSuccess) return e;
  return hipMemcpyToSymbol(HIP_SYMBOL(c_idw4), tables, sizeof(float) * 4 * 64);
}

constexpr float k255 = 1 / 255.0f;  // gainmapmath.cpp:579
constexpr float k876 = 1 / 876.0f;  // gainmapmath.cpp:598
constexpr float k896 = 1 / 896.0f;  // gainmapmath.cpp:599

// BT.601 YUV->RGB used by apply for every image (ultrahdr.cpp:431, gainmapmath.cpp:184-202)
constexpr float kP3YR = 0.299f, kP3YG = 0.587f, kP3YB = 0.114f;
constexpr float kP3Cb = 1.772f, kP3Cr = 1.402f;
constexpr float kP3GCb = kP3YB * kP3Cb / kP3YG;
constexpr float kP3GCr = kP3YR * kP3Cr / kP3YG;

// two floats per lane: v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 process both in one 4-cycle issue slot
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 splat(float v) { return (f2){v, v}; }
// one half of a pair in both lanes of a packed operand (VOP3P op_sel / op_sel_hi: no instruction)
__device__ __forceinline__ f2 bc_lo(f2 v) { return __builtin_shufflevector(v, v, 0, 0); }
__device__ __forceinline__ f2 bc_hi(f2 v) { return __builtin_shufflevector(v, v, 1, 1); }

// packed add / mul with the VOP3P clamp modifier: both lanes saturate to [0, 1] for free (hipcc only
// folds clamps into scalar ops, so these two are spelled out)
__device__ __forceinline__ f2 pk_add_sat(f2 a, f2 b) {
  f2 r;
  asm("v_pk_add_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ f2 pk_mul_sat(f2 a, f2 b) {
  f2 r;
  asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ f2 pk_fma_sat(f2 a, f2 b, f2 c) {
  f2 r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// a * b + (c.lo, c.lo) or (c.hi, c.hi), clamped: VOP3P's op_sel broadcasts one half of the third operand for free
template <int HI>
__device__ __forceinline__ f2 pk_fma_sat_bc(f2 a, f2 b, f2 c) {
  f2 r;
  if (HI) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,1] clamp" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,1,0] clamp" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}


// =================================================================================================
// LUT mode: the reference's table accessors (gainmapmath.cpp:162-171,269-277,292-302,316-324,344-354) and
// GainLUT (gainmapmath.h:151-182).  Everything here is float/integer arithmetic plus table reads, so the
// LUT pipelines are bit-exact by construction once the tables are (they are built by the exact functions).
// =================================================================================================
// `uint32_t value = static_cast<uint32_t>(e * (N - 1) + 0.5); value = CLIP3(value, 0, N - 1);`
// float product, double sum, truncation.  Outside [0, 2^32) the C++ conversion is undefined; the oracle pins
// what the reference's x86-64 build does (64-bit cvttsd2si, low word kept) and this restates it.
__device__ __attribute__((noinline)) uint32_t lut_index_wild(float t) {
  const double pos = (double)t + 0.5;
  uint32_t v = 0u;  // NaN, inf and |pos| >= 2^63 convert to 0x8000000000000000: low word 0
  if (__builtin_fabs(pos) < 9223372036854775808.0) v = (uint32_t)(long long)pos;
  return v;
}
// floor(t + 0.5) in ONE instruction: v_cvt_rpi_i32_f32 rounds to the nearest integer, ties toward +infinity, with no intermediate
// rounding -- for every float in [0, 2^31) and for -0 the value of (uint32_t)((double)t + 0.5), the reference's expression
// (scripts/ab/cvt_rpi_check.hip compares all 1.3e9 of them; tests/test_gpu_lut.py through eval case 40).  The f64 form costs
// v_cvt_f64_f32 + v_add_f64 + v_cvt_u32_f64.
__device__ __forceinline__ uint32_t round_half_up(float t) {
  int r;
  asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(t));
  return (uint32_t)r;
}
__device__ __forceinline__ uint32_t lut_index(float e, uint32_t n) {
  const float t = e * (float)(n - 1u);
  uint32_t v;
  if (__builtin_expect(__float_as_uint(t) < 0x4F000000u, 1)) v = round_half_up(t);  // +0 <= t < 2^31
  else v = lut_index_wild(t);
  return min(v, n - 1u);
}
// e in [0, 1] (what clampPixelFloat returns: the sRGB lookups of apply): no clip can bite and the conversion is in range
__device__ __forceinline__ uint32_t lut_index_unit(float e, uint32_t n) {
  return round_half_up(e * (float)(n - 1u));
}
// GainLUT::mGainTable[idx] (gainmapmath.h:153-168); boost_factor == 1.0f reproduces the one-argument constructor
__device__ __forceinline__ float gain_lut_entry(uint32_t idx, double log2_min, double log2_max, float boost_factor) {
  const float value = (float)idx / (float)(kGainLutN - 1u);
  const float log_boost = (float)(log2_min * (double)(1.0f - value) + log2_max * (double)value);
  return (float)exp2((double)(log_boost * boost_factor));
}

__global__ void __launch_bounds__(256) k_build_luts(float* lut) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= kLutTotal) return;
  float y;
  if (i < kLutHlgInv) y = srgb_inv_oetf_exact((float)(i - kLutSrgbInv) / (float)(kLutSrgbInvN - 1u));
  else if (i < kLutPqInv) y = hlg_inv_oetf_exact((float)(i - kLutHlgInv) / (float)(kLutHlgInvN - 1u));
  else if (i < kLutHlg) y = pq_inv_oetf_exact((float)(i - kLutPqInv) / (float)(kLutPqInvN - 1u));
  else if (i < kLutPq) y = hlg_oetf_exact((float)(i - kLutHlg) / (float)(kLutHlgN - 1u));
  else y = pq_oetf_exact((float)(i - kLutPq) / (float)(kLutPqN - 1u));
  lut[i] = y;
}
hipError_t launch_build_luts(float* lut, hipStream_t s) {
  hipLaunchKernelGGL(k_build_luts, dim3((kLutTotal + 255u) / 256u), dim3(256), 0, s, lut);
  return hipGetLastError();
}
__global__ void __launch_bounds__(256) k_build_lut_codes(float* lut) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;   // 2 x 65536 entries
  const uint32_t t = i >> 16, k = i & 0xFFFFu;
  uint16_t* codes = reinterpret_cast<uint16_t*>(lut + (t ? kCodePq : kCodeHlg));
  codes[k] = (uint16_t)(0x3ffu & (uint32_t)(lut[(t ? kLutPq : kLutHlg) + k] * 1023.0f));
}
hipError_t launch_build_lut_codes(float* lut, hipStream_t s) {
  hipLaunchKernelGGL(k_build_lut_codes, dim3(2u * 65536u / 256u), dim3(256), 0, s, lut);
  return hipGetLastError();
}
__global__ void __launch_bounds__(256) k_build_gain_lut(float* table, double log2_min, double log2_max, float boost_factor) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < kGainLutN) table[i] = gain_lut_entry(i, log2_min, log2_max, boost_factor);
}
hipError_t launch_build_gain_lut(float* table, double log2_min, double log2_max, float boost_factor, hipStream_t s) {
  hipLaunchKernelGGL(k_build_gain_lut, dim3(kGainLutN / 256u), dim3(256), 0, s, table, log2_min, log2_max, boost_factor);
  return hipGetLastError();
}

// =================================================================================================
// generate
// =================================================================================================

template <int TF>
__device__ __forceinline__ float hdr_inv_oetf(float e) {
  if (TF == 1) return hlg_inv_oetf_guarded(e);
  if (TF == 2) return pq_inv_oetf_guarded(e);
  return e;  // ULTRAHDR_TF_LINEAR: identityConversion (ultrahdr.cpp:223-228)
}

// Two horizontally adjacent map pixels (index 0|1 of every array) from their 4x4 blocks, processed as one
// packed float2 wherever the arithmetic is float: both pixels run the identical IEEE operation sequence,
// so packing changes nothing but the issue count.  hy[k][r][0|1]: P010 luma cols (0,1)|(2,3) packed lo/hi
// 16 bits; huv[k][r][0|1]: (U,V) of chroma col 0|1 of chroma row r; y8[k][r]: 4 luma bytes; u8/v8[k][r]:
// 2 chroma bytes in bits 0-15.  Accumulation order is samplePixels' (gainmapmath.cpp:605-615): dy outer,
// dx inner, one running float sum per channel.
// float of the low / high 16-bit word of w in ONE instruction (SDWA operand select); hipcc emits an extract
// (v_and_b32_sdwa / v_lshrrev_b32) followed by v_cvt_f32_u32 for the plain C expression
__device__ __forceinline__ float cvt_word0(uint32_t w) {
  float r;
  asm("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(r) : "v"(w));
  return r;
}
__device__ __forceinline__ float cvt_word1(uint32_t w) {
  float r;
  asm("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(r) : "v"(w));
  return r;
}


// LUT: the inverse OETFs are the reference's table accessors (ultrahdr.cpp:230,238,319 with USE_*_LUT = 1), read
// from the LDS copies s_srgb (1024 entries) / s_hdr (4096 entries of the HLG or PQ table)
// FILTER: before the exact (f64) evaluation, the same chain is run with the f32 special-function unit.  Its gain is
// within kFilterRelErr of the exact one (bounds measured exhaustively per function), so the
// truncated code value can only differ when it lies within flt_delta of an integer; only waves holding such a
// pixel (or one whose clamp decision is in doubt) pay for the exact path.  Bytes are identical either way.
// Returns a mask: bit k (k = 0, 1) set <=> gain[k] is the exact (reference) unclamped gain; a clear bit means gain[k] is the
// filter's estimate, within GenConsts::flt_gain_rel of it -- unless bit 2 + k is set: that pixel is in doubt (FILTER only), its
// byte and gain are provisional.
// PART: the function in two halves, for the pairs the streaming kernel hands to k_generate_resolve WITH their sampled values:
// kGenFront stops in front of the transfer functions and returns their inputs in *mid (both pixels' r, g, b, hr, hg, hb -- what the
// filter and the exact path both start from), kGenBack takes them from *mid instead of sampling; kGenWhole with mid != nullptr runs
// as ever and leaves a copy in *mid.
constexpr int kGenWhole = 0, kGenFront = 1, kGenBack = 2;
struct GenMid { f2 r, g, b, hr, hg, hb; };
template <int TF, bool LUT, bool FILTER, bool DEFER = false, int PART = kGenWhole>
__device__ __forceinline__ uint32_t gen_pair(const GenConsts& c, const uint32_t (&hy)[2][4][2],
                                             const uint32_t (&huv)[2][2][2], const uint32_t (&y8)[2][4],
                                             const uint32_t (&u8)[2][2], const uint32_t (&v8)[2][2],
                                             uint8_t (&out)[2], float (&gain)[2], const float* s_srgb, const float* s_hdr,
                                             GenMid* mid = nullptr) {
  f2 r, g, b, hr, hg, hb;
  if (PART == kGenBack) {
    r = mid->r; g = mid->g; b = mid->b; hr = mid->hr; hg = mid->hg; hb = mid->hb;
  } else {
  f2 sy = splat(0.0f), su = splat(0.0f), sv = splat(0.0f);
  f2 hsy = splat(0.0f), hsu = splat(0.0f), hsv = splat(0.0f);
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    f2 uf[2], vf[2], huf[2], hvf[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      // float(u - 128) * (1/255.0f)   (gainmapmath.cpp:579-580): ONE rounding of an exact product -- which is what
      // fma(float(u), k, -128 k) computes as well: 128 k is exact (a power of two times k), so the fma's unrounded value is
      // (u - 128) k exactly.  One packed fma in place of a packed subtract and a packed multiply, the same float.
      const f2 ub = (f2){(float)((u8[0][r] >> (8 * k)) & 0xffu), (float)((u8[1][r] >> (8 * k)) & 0xffu)};
      const f2 vb = (f2){(float)((v8[0][r] >> (8 * k)) & 0xffu), (float)((v8[1][r] >> (8 * k)) & 0xffu)};
      uf[k] = pk_fma(ub, splat(k255), splat(-128.0f * k255));
      vf[k] = pk_fma(vb, splat(k255), splat(-128.0f * k255));
      // float((x >> 6) - 64) * (1/896) - 0.5   (gainmapmath.cpp:593-600), evaluated as
      // fma(float(x & 0xFFC0), (1/896)/64, -64 (1/896)) - 0.5: the masked 16-bit word is 64 * (x >> 6), so one mask per
      // two samples and one SDWA word->float conversion per sample replace shift + mask + subtract + convert; both
      // constants are exact scalings of 1/896 by powers of two, so the fma rounds the reference's exact product
      // ((x >> 6) - 64) * (1/896) -- once, like the reference's multiply.  (Round 4: the fma form; a subtract and a
      // multiply before, 32 packed instructions more per pixel pair over the three places it applies.)
      const uint32_t m0 = huv[0][r][k] & 0xFFC0FFC0u, m1 = huv[1][r][k] & 0xFFC0FFC0u;
      const f2 hu = (f2){cvt_word0(m0), cvt_word0(m1)};
      const f2 hv = (f2){cvt_word1(m0), cvt_word1(m1)};
      huf[k] = pk_fma(hu, splat(k896 * 0.015625f), splat(-64.0f * k896)) - splat(0.5f);
      hvf[k] = pk_fma(hv, splat(k896 * 0.015625f), splat(-64.0f * k896)) - splat(0.5f);
    }
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const int dy = 2 * r + d;
#pragma unroll
      for (int dx = 0; dx < 4; ++dx) {
        const f2 yb = (f2){(float)((y8[0][dy] >> (8 * dx)) & 0xffu), (float)((y8[1][dy] >> (8 * dx)) & 0xffu)};
        sy += yb * splat(k255);
        su += uf[dx >> 1];
        sv += vf[dx >> 1];
        const uint32_t w0 = hy[0][dy][dx >> 1] & 0xFFC0FFC0u, w1 = hy[1][dy][dx >> 1] & 0xFFC0FFC0u;
        const f2 hb = (dx & 1) ? (f2){cvt_word1(w0), cvt_word1(w1)} : (f2){cvt_word0(w0), cvt_word0(w1)};
        hsy += pk_fma(hb, splat(k876 * 0.015625f), splat(-64.0f * k876));   // ((x >> 6) - 64) * (1/876), one rounding (see above)
        hsu += huf[dx >> 1];
        hsv += hvf[dx >> 1];
      }
    }
  }
  // e / float(scale*scale): division by 16 == multiplication by 1/16 exactly
  sy *= splat(0.0625f); su *= splat(0.0625f); sv *= splat(0.0625f);
  hsy *= splat(0.0625f); hsu *= splat(0.0625f); hsv *= splat(0.0625f);

  // SDR: YUV->RGB (gainmapmath.cpp:142-146 shape), sRGB EOTF, luminance * 203 (ultrahdr.cpp:316-324)
  // clampPixelFloat (gainmapmath.cpp:115-118) rides on the last add of each expression: the sum is the same IEEE
  // operation, and min(max(x,0),1) differs from the reference's compare chain only in the sign of a zero,
  // which no later step can observe
  r = pk_add_sat(sy, splat(c.sdr_cr) * sv);
  g = pk_add_sat(sy - splat(c.sdr_gcb) * su, -(splat(c.sdr_gcr) * sv));
  b = pk_add_sat(sy, splat(c.sdr_cb) * su);
  // HDR: YUV->RGB (ultrahdr.cpp:326-327)
  hr = pk_add_sat(hsy, splat(c.hdr_cr) * hsv);
  hg = pk_add_sat(hsy - splat(c.hdr_gcb) * hsu, -(splat(c.hdr_gcr) * hsv));
  hb = pk_add_sat(hsy, splat(c.hdr_cb) * hsu);
  }
  if (PART != kGenBack && mid != nullptr) { mid->r = r; mid->g = g; mid->b = b; mid->hr = hr; mid->hg = hg; mid->hb = hb; }
  if (PART == kGenFront) return 0u;

  if (FILTER && !LUT) {
    const f2 fr = (f2){srgb_inv_oetf_fast(r.x), srgb_inv_oetf_fast(r.y)};
    const f2 fg = (f2){srgb_inv_oetf_fast(g.x), srgb_inv_oetf_fast(g.y)};
    const f2 fb = (f2){srgb_inv_oetf_fast(b.x), srgb_inv_oetf_fast(b.y)};
    const f2 fs = (splat(c.lum_r) * fr + splat(c.lum_g) * fg + splat(c.lum_b) * fb) * splat(203.0f);
    f2 qr = hr, qg = hg, qb = hb;
    if (TF == 1) {
      qr = (f2){hlg_inv_oetf_fast(hr.x), hlg_inv_oetf_fast(hr.y)};
      qg = (f2){hlg_inv_oetf_fast(hg.x), hlg_inv_oetf_fast(hg.y)};
      qb = (f2){hlg_inv_oetf_fast(hb.x), hlg_inv_oetf_fast(hb.y)};
    } else if (TF == 2) {
      qr = (f2){pq_inv_oetf_fast(hr.x), pq_inv_oetf_fast(hr.y)};
      qg = (f2){pq_inv_oetf_fast(hg.x), pq_inv_oetf_fast(hg.y)};
      qb = (f2){pq_inv_oetf_fast(hb.x), pq_inv_oetf_fast(hb.y)};
    }
    if (!c.gm_identity) {
      const f2 t0 = splat(c.gm[0]) * qr + splat(c.gm[1]) * qg + splat(c.gm[2]) * qb;
      const f2 t1 = splat(c.gm[3]) * qr + splat(c.gm[4]) * qg + splat(c.gm[5]) * qb;
      const f2 t2 = splat(c.gm[6]) * qr + splat(c.gm[7]) * qg + splat(c.gm[8]) * qb;
      qr = t0; qg = t1; qb = t2;
    }
    const f2 fh = (splat(c.lum_r) * qr + splat(c.lum_g) * qg + splat(c.lum_b) * qb) * splat(c.hdr_white_nits);
    uint32_t fbyte[2], fexact = 0u, doubt = 0u;
    float fgain[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float ys = k ? fs.y : fs.x, yh = k ? fh.y : fh.x;
      const float gf = (ys > 0.0f) ? yh * __builtin_amdgcn_rcpf(ys) : 1.0f;
      const float gc = __builtin_amdgcn_fmed3f(gf, c.min_boost, c.max_boost);
      const float v = (__builtin_amdgcn_logf(gc) - c.log2_min) * c.flt_scale;
      const float fl = __builtin_floorf(v), fr2 = v - fl;
      const bool lo = gf < c.flt_lo, hi = gf > c.flt_hi;
      const bool mid = (fr2 >= c.flt_delta) && (fr2 <= 1.0f - c.flt_delta);
      // luminances below 1e-10 leave the relative-error regime of the fast functions (denormal intermediates)
      const bool sane = !(ys > 0.0f && ys < 1e-10f) && !(yh != 0.0f && __builtin_fabsf(yh) < 1e-10f);
      if (!((lo || hi || mid) && sane)) doubt |= 1u << k;
      fbyte[k] = lo ? c.enc_byte_min : hi ? c.enc_byte_max : (uint32_t)fl;
      fgain[k] = gf;
      // the fast and the exact sRGB EOTF are zero for the same inputs only, so "SDR luminance is zero" (gain := 1,
      // gainmapmath.cpp:531) is decided identically on both paths
      if (!(ys > 0.0f)) fexact |= 1u << k;
    }
    // DEFER (large launches): the kernel never runs the exact path: a pixel in doubt keeps its estimate's byte for now and is
    // handed (bits 2, 3 of the result) to k_generate_resolve, which replaces the byte and accounts for the pixel's exact gain.
    // Otherwise (a launch too small to be worth a second kernel's latency) a wave holding such a pixel runs the exact path below.
    if (DEFER || __builtin_amdgcn_ballot_w64(doubt != 0u) == 0ull) {
      out[0] = (uint8_t)fbyte[0]; out[1] = (uint8_t)fbyte[1];
      gain[0] = fgain[0]; gain[1] = fgain[1];
      return fexact | (doubt << 2);
    }
  }
  // independent f64 evaluations advanced in lock step: 6 = 3 channels x 2 pixels.  Measured on MI355X
  // (scripts/ab): 6 -> 0.388 ms per 32-frame launch, 3 -> 0.432, 2 -> 0.490 although the narrower forms
  // need fewer VGPRs (98 / 84 / 78): exposed f64 FMA latency costs more than the lost occupancy.
  constexpr int kLockstep = 6;
  if (LUT) {
    r = (f2){s_srgb[lut_index_unit(r.x, kLutSrgbInvN)], s_srgb[lut_index_unit(r.y, kLutSrgbInvN)]};
    g = (f2){s_srgb[lut_index_unit(g.x, kLutSrgbInvN)], s_srgb[lut_index_unit(g.y, kLutSrgbInvN)]};
    b = (f2){s_srgb[lut_index_unit(b.x, kLutSrgbInvN)], s_srgb[lut_index_unit(b.y, kLutSrgbInvN)]};
  } else {
    float ch[6] = {r.x, r.y, g.x, g.y, b.x, b.y};
#pragma unroll
    for (int i = 0; i < 6; i += kLockstep) {
      float part[kLockstep];
#pragma unroll
      for (int j = 0; j < kLockstep; ++j) part[j] = ch[i + j];
      srgb_inv_oetf_guarded_n<kLockstep>(part);
#pragma unroll
      for (int j = 0; j < kLockstep; ++j) ch[i + j] = part[j];
    }
    r = (f2){ch[0], ch[1]}; g = (f2){ch[2], ch[3]}; b = (f2){ch[4], ch[5]};
  }
  const f2 sdr_nits = (splat(c.lum_r) * r + splat(c.lum_g) * g + splat(c.lum_b) * b) * splat(203.0f);

  // HDR: inverse OETF, gamut conversion, luminance * white (ultrahdr.cpp:328-330)
  if (TF != 0 && LUT) {  // both HDR tables have 4096 entries (gainmapmath.h:342-343,368-369)
    hr = (f2){s_hdr[lut_index_unit(hr.x, kLutHlgInvN)], s_hdr[lut_index_unit(hr.y, kLutHlgInvN)]};
    hg = (f2){s_hdr[lut_index_unit(hg.x, kLutHlgInvN)], s_hdr[lut_index_unit(hg.y, kLutHlgInvN)]};
    hb = (f2){s_hdr[lut_index_unit(hb.x, kLutHlgInvN)], s_hdr[lut_index_unit(hb.y, kLutHlgInvN)]};
  } else if (TF != 0) {  // ULTRAHDR_TF_LINEAR: identityConversion (ultrahdr.cpp:223-228)
    float ch[6] = {hr.x, hr.y, hg.x, hg.y, hb.x, hb.y};
#pragma unroll
    for (int i = 0; i < 6; i += kLockstep) {
      float part[kLockstep];
#pragma unroll
      for (int j = 0; j < kLockstep; ++j) part[j] = ch[i + j];
      if (TF == 1) hlg_inv_oetf_guarded_n<kLockstep>(part);
      else pq_inv_oetf_guarded_n<kLockstep>(part);
#pragma unroll
      for (int j = 0; j < kLockstep; ++j) ch[i + j] = part[j];
    }
    hr = (f2){ch[0], ch[1]}; hg = (f2){ch[2], ch[3]}; hb = (f2){ch[4], ch[5]};
  }
  if (!c.gm_identity) {
    const f2 t0 = splat(c.gm[0]) * hr + splat(c.gm[1]) * hg + splat(c.gm[2]) * hb;
    const f2 t1 = splat(c.gm[3]) * hr + splat(c.gm[4]) * hg + splat(c.gm[5]) * hb;
    const f2 t2 = splat(c.gm[6]) * hr + splat(c.gm[7]) * hg + splat(c.gm[8]) * hb;
    hr = t0; hg = t1; hb = t2;
  }
  const f2 hdr_nits = (splat(c.lum_r) * hr + splat(c.lum_g) * hg + splat(c.lum_b) * hb) * splat(c.hdr_white_nits);

  gain[0] = raw_gain(sdr_nits.x, hdr_nits.x);
  gain[1] = raw_gain(sdr_nits.y, hdr_nits.y);
#pragma unroll
  for (int k = 0; k < 2; ++k)
    out[k] = encode_gain_guarded(gain[k], c.min_boost, c.max_boost, c.log2_min, c.log2_max, c.enc_scale, c.enc_byte_min,
                                 c.enc_byte_max);
  return 3u;
}

// Frames are read once and outputs written once: the aligned fast paths use non-temporal loads and stores
// so that 2 GB of apply output do not sit dirty in L2 / Infinity Cache when the next kernel starts reading.
// Same-box A/B (scripts/ab, 64 x 4K, ms per launch): generate 0.484 -> 0.444 with either, 0.394 with both;
// apply 0.98 -> 0.996.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint4 ld_stream(const uint4* p) {
  const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint2 ld_stream(const uint2* p) {
  const u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(p));
  return make_uint2(v.x, v.y);
}
__device__ __forceinline__ uint32_t ld_stream(const uint32_t* p) {
  return __builtin_nontemporal_load(p);
}
__device__ __forceinline__ void st_stream(uint4* p, uint4 v) {
  __builtin_nontemporal_store((u32x4){v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4*>(p));
}
__device__ __forceinline__ void st_stream(uint2* p, uint2 v) {
  __builtin_nontemporal_store((u32x2){v.x, v.y}, reinterpret_cast<u32x2*>(p));
}
__device__ __forceinline__ uint32_t ld8(const uint8_t* p) { return *p; }
__device__ __forceinline__ uint32_t ld16(const uint16_t* p) { return *p; }

// launch geometry of k_generate, chosen by same-box A/B (scripts/ab, ms per 32-frame 4K launch):
//   block 64/128/256/512/1024, 1 span: 0.64 / 0.50 / 0.39 / 0.355 / 0.43;   block 256, 4 spans: 0.350
constexpr int kGenBlock = 256, kGenTiles = 4;

// inputs of pair `idx` (two horizontally adjacent map pixels = an 8x4 pixel block of both images) -> registers
template <bool ALIGNED>
__device__ __forceinline__ void load_pair(const GenConsts& c, const GenImage& im, const uint8_t* im_v, uint32_t my, uint32_t pr,
                                          bool two, uint32_t (&hy)[2][4][2], uint32_t (&huv)[2][2][2], uint32_t (&y8)[2][4],
                                          uint32_t (&u8)[2][2], uint32_t (&v8)[2][2]) {
  const uint32_t mx = pr * 2u;
  if (ALIGNED) {
    // 32-bit element offsets (every plane is < 4 GiB on this path): one 64-bit add per address
    const uint32_t hoff = 4u * my * im.hy_stride + 8u * pr, yoff = 4u * my * im.y_stride + 8u * pr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint4 q = ld_stream(reinterpret_cast<const uint4*>(im.hy + (hoff + r * im.hy_stride)));
      hy[0][r][0] = q.x; hy[0][r][1] = q.y; hy[1][r][0] = q.z; hy[1][r][1] = q.w;
      const uint2 p = ld_stream(reinterpret_cast<const uint2*>(im.y + (yoff + r * im.y_stride)));
      y8[0][r] = p.x; y8[1][r] = p.y;
    }
    const uint32_t huvoff = 2u * my * im.huv_stride + 8u * pr, coff = 2u * my * im.c_stride + 4u * pr;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const uint4 q = ld_stream(reinterpret_cast<const uint4*>(im.huv + (huvoff + r * im.huv_stride)));
      huv[0][r][0] = q.x; huv[0][r][1] = q.y; huv[1][r][0] = q.z; huv[1][r][1] = q.w;
      const uint32_t uu = ld_stream(reinterpret_cast<const uint32_t*>(im.u + (coff + r * im.c_stride)));
      const uint32_t vv = ld_stream(reinterpret_cast<const uint32_t*>(im_v + (coff + r * im.c_stride)));
      u8[0][r] = uu & 0xffffu; u8[1][r] = uu >> 16;
      v8[0][r] = vv & 0xffffu; v8[1][r] = vv >> 16;
    }
  } else {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const bool on = (k == 0) || two;
      const uint32_t x0 = 4u * (mx + k);  // first image column of this map pixel
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint16_t* hrow = im.hy + (size_t)(4u * my + r) * im.hy_stride + x0;
        const uint8_t* yrow = im.y + (size_t)(4u * my + r) * im.y_stride + x0;
        hy[k][r][0] = on ? (ld16(hrow) | (ld16(hrow + 1) << 16)) : 0u;
        hy[k][r][1] = on ? (ld16(hrow + 2) | (ld16(hrow + 3) << 16)) : 0u;
        y8[k][r] = on ? (ld8(yrow) | (ld8(yrow + 1) << 8) | (ld8(yrow + 2) << 16) | (ld8(yrow + 3) << 24)) : 0u;
      }
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const uint16_t* crow = im.huv + (size_t)(2u * my + r) * im.huv_stride + x0;  // (x & ~1)
        const uint8_t* urow = im.u + (size_t)(2u * my + r) * im.c_stride + (x0 >> 1);
        const uint8_t* vrow = im_v + (size_t)(2u * my + r) * im.c_stride + (x0 >> 1);
        huv[k][r][0] = on ? (ld16(crow) | (ld16(crow + 1) << 16)) : 0u;
        huv[k][r][1] = on ? (ld16(crow + 2) | (ld16(crow + 3) << 16)) : 0u;
        u8[k][r] = on ? (ld8(urow) | (ld8(urow + 1) << 8)) : 0u;
        v8[k][r] = on ? (ld8(vrow) | (ld8(vrow + 1) << 8)) : 0u;
      }
    }
  }
}

// wave64 reduction: every lane ends up with the wave's min / max.  Six DPP steps -- the lane exchanges ride on the VALU operand
// path: quad_perm (xor 1, xor 2), row_half_mirror, row_mirror put a row's extreme into its 16 lanes, row_bcast:15 / row_bcast:31
// carry it across the four rows into lane 63 -- and one v_readlane.  (Round 4.  __shfl_xor compiles to ds_bpermute_b32: six
// dependent trips through the LDS crossbar per value, 36 of them in k_generate's tail, ~0.6 us during which a wave that has
// finished its pixels still holds its slot.)  All 64 lanes are active wherever this is called.
template <int CTRL, int ROWS>
__device__ __forceinline__ float dpp_move(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROWS, 0xF, false));
}
__device__ __forceinline__ void wave_minmax(float& vmin, float& vmax) {
  vmin = fminf(vmin, dpp_move<0xB1, 0xF>(vmin));  vmax = fmaxf(vmax, dpp_move<0xB1, 0xF>(vmax));    // quad_perm:[1,0,3,2]
  vmin = fminf(vmin, dpp_move<0x4E, 0xF>(vmin));  vmax = fmaxf(vmax, dpp_move<0x4E, 0xF>(vmax));    // quad_perm:[2,3,0,1]
  vmin = fminf(vmin, dpp_move<0x141, 0xF>(vmin)); vmax = fmaxf(vmax, dpp_move<0x141, 0xF>(vmax));   // row_half_mirror
  vmin = fminf(vmin, dpp_move<0x140, 0xF>(vmin)); vmax = fmaxf(vmax, dpp_move<0x140, 0xF>(vmax));   // row_mirror
  vmin = fminf(vmin, dpp_move<0x142, 0xA>(vmin)); vmax = fmaxf(vmax, dpp_move<0x142, 0xA>(vmax));   // row_bcast:15 -> rows 1, 3
  vmin = fminf(vmin, dpp_move<0x143, 0xC>(vmin)); vmax = fmaxf(vmax, dpp_move<0x143, 0xC>(vmax));   // row_bcast:31 -> rows 2, 3
  vmin = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vmin), 63));
  vmax = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vmax), 63));
}
// one lane per wave publishes; keys[0] holds ~key(min), keys[1] holds key(max); both only grow, both start at 0
__device__ __forceinline__ void publish_minmax(float gmin, float gmax, uint32_t* keys) {
  if ((threadIdx.x & 63u) == 0u && gmin <= gmax) {  // wave saw at least one pixel
    const uint32_t kmin = ~float_to_key(gmin), kmax = float_to_key(gmax);
    // a stale (smaller) value read here only costs an unnecessary atomic, never a lost update
    if (__hip_atomic_load(&keys[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < kmin) atomicMax(&keys[0], kmin);
    if (__hip_atomic_load(&keys[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < kmax) atomicMax(&keys[1], kmax);
  }
}

// exact bytes and unclamped gains of pair (my, pr): what k_generate_resolve runs on the pixels the filtered kernel left to it
template <int TF, bool ALIGNED, bool LUT>
__device__ __forceinline__ void exact_pair(const GenConsts& c, const GenImage& im, const uint8_t* im_v, uint32_t my,
                                           uint32_t pr, bool two, const float* s_srgb, const float* s_hdr, uint8_t (&o)[2], float (&gn)[2]) {
  uint32_t hy[2][4][2], huv[2][2][2], y8[2][4], u8[2][2], v8[2][2];
  load_pair<ALIGNED>(c, im, im_v, my, pr, two, hy, huv, y8, u8, v8);
  gen_pair<TF, LUT, false>(c, hy, huv, y8, u8, v8, o, gn, s_srgb, s_hdr);
}

// (the relative distance within which a filter estimate may sit from the exact gain is GenConsts::flt_gain_rel: generate_consts' kRel, doubled)

// Thread = 2 horizontally adjacent map pixels = an 8x4 pixel block of both images.
// A wave64 therefore consumes 1 KiB contiguous per P010 row (dwordx4/lane), 512 B per 8-bit luma
// row (dwordx2/lane) and 256 B per chroma row (dword/lane).
//
// Statistics (content min / max of the unclamped gain) stay EXACT under the filter: every thread keeps the gains of its
// pixels with an "exact" bit; when a wave has finished its tiles, only pixels whose estimate could be the wave's exact
// extreme -- and could still beat the extreme already published for the image -- are re-evaluated on the exact path
// (typically none).  Waves never wait for each other: a block-wide reduction here measured +25 % on the whole kernel,
// because one wave in seven takes the exact path in some tile and its three siblings would idle at the barrier.
template <int TF, bool ALIGNED, bool LUT, bool FILTER, int TILES, bool DEFER, int BLOCK = kGenBlock>
__global__ void __launch_bounds__(BLOCK, 1) k_generate(const GenConsts c, const GenBatch b) {
  // LUT mode: block-private copies of the two tables (4 KiB + 16 KiB), so every lookup is an LDS gather
  __shared__ float s_srgb[LUT ? kLutSrgbInvN : 1];
  __shared__ float s_hdr[(LUT && TF != 0) ? kLutHlgInvN : 1];
  __shared__ float s_kept[(FILTER && DEFER) ? TILES * 2 * BLOCK : 1];  // FILTER: every thread's gains, for the candidate pass
  if (LUT) {
    for (uint32_t i = threadIdx.x; i < kLutSrgbInvN / 4u; i += BLOCK)
      reinterpret_cast<float4*>(s_srgb)[i] = reinterpret_cast<const float4*>(c.lut + kLutSrgbInv)[i];
    if (TF != 0) {
      const float4* src = reinterpret_cast<const float4*>(c.lut + (TF == 1 ? kLutHlgInv : kLutPqInv));
      for (uint32_t i = threadIdx.x; i < kLutHlgInvN / 4u; i += BLOCK) reinterpret_cast<float4*>(s_hdr)[i] = src[i];
    }
    __syncthreads();
  }
  // consecutive workgroups belong to different images (grid.x = image): the images of a launch progress together, so
  // the extremes a finished wave publishes prune the statistics candidates of the image's later waves
  const uint32_t img_i = blockIdx.x, blk = blockIdx.y;
  const GenImage& im = b.img[img_i];
  const uint8_t* im_v = im.u + (size_t)im.c_stride * (c.height / 2u);
  const uint32_t pairs_per_row = (c.map_w + 1u) >> 1;
  const uint32_t total = pairs_per_row * c.map_h;
  const bool stats = c.stat_keys != nullptr;
  float emin = __builtin_inff(), emax = -__builtin_inff();   // over gains known exactly
  float amin = __builtin_inff(), amax = -__builtin_inff();   // over all gains (estimates included)
  uint32_t kept_exact = 0u, kept_valid = 0u, kept_doubt = 0u;   // 2 bits per tile
  uint32_t n_saved = 0u;   // slot mode: pairs of this wave that have left with their sampled values (wave-uniform count)
  uint32_t* wslot = (FILTER && DEFER && c.stat_slots != 0u)
                        ? c.stat_ws + (size_t)kStatWords * blockIdx.x + kStatHdr + (blockIdx.y * (uint32_t)(BLOCK / 64) + (threadIdx.x >> 6)) * kStatSlotWords
                        : nullptr;

  // each block walks TILES consecutive spans of BLOCK pairs: fewer, longer-lived waves
  // (wave launch + descriptor fetch is a measurable share of a ~10 us wave)
#pragma unroll 1
  for (uint32_t t = 0; t < (uint32_t)TILES; ++t) {
    const uint32_t idx = (blk * (uint32_t)TILES + t) * (uint32_t)BLOCK + threadIdx.x;
    if (idx >= total) break;
    const uint32_t my = idx / pairs_per_row;
    const uint32_t pr = idx - my * pairs_per_row;
    const uint32_t mx = pr * 2u;
    const bool two = ALIGNED || (mx + 1u < c.map_w);

    uint32_t hy[2][4][2], huv[2][2][2], y8[2][4], u8[2][2], v8[2][2];
    load_pair<ALIGNED>(c, im, im_v, my, pr, two, hy, huv, y8, u8, v8);

    uint8_t o[2];
    float gn[2];
    // a missing second pixel is computed on zeros and dropped
    GenMid mid;
    const uint32_t ex = gen_pair<TF, LUT, FILTER, DEFER>(c, hy, huv, y8, u8, v8, o, gn, s_srgb, s_hdr, (FILTER && DEFER) ? &mid : nullptr);
    const uint8_t o0 = o[0], o1 = o[1];
    if (FILTER && DEFER) {
      // pixels in doubt: their bytes are provisional -- the pair goes onto the image's list once the wave has finished its tiles,
      // flags = which of its two pixels k_generate_resolve has to redo -- and their gains take no part in the statistics here
      uint32_t dm = (ex >> 2) & (two ? 3u : 1u);
      if (c.stat_slots != 0u) {
        // slot mode: the pair leaves WITH its sampled values, now, while they are still in registers (one tile in five holds such a
        // pair; the wave's first kStatSlotSaved go this way, any beyond that onto the plain list below)
        const unsigned long long dmask = __builtin_amdgcn_ballot_w64(dm != 0u);
        if (dmask != 0ull) {
          const uint32_t pos = n_saved + (uint32_t)__builtin_popcountll(dmask & ((1ull << (threadIdx.x & 63u)) - 1ull));
          n_saved += (uint32_t)__builtin_popcountll(dmask);
          if (dm != 0u && pos < kStatSlotSaved) {
            float4* e = reinterpret_cast<float4*>(wslot + 16u + pos * 16u);
            e[0] = make_float4(__uint_as_float((idx << 3) | dm), mid.r.x, mid.r.y, mid.g.x);
            e[1] = make_float4(mid.g.y, mid.b.x, mid.b.y, mid.hr.x);
            e[2] = make_float4(mid.hr.y, mid.hg.x, mid.hg.y, mid.hb.x);
            e[3] = make_float4(mid.hb.y, 0.0f, 0.0f, 0.0f);
            dm = 0u;   // (not onto the plain list; its gains stay out of the statistics here all the same: `valid` below)
          }
        }
      }
      kept_doubt |= dm << (2u * t);
      if (stats) {
        s_kept[(2u * t) * BLOCK + threadIdx.x] = gn[0];
        s_kept[(2u * t + 1u) * BLOCK + threadIdx.x] = gn[1];
        const uint32_t valid = (two ? 3u : 1u) & ~((ex >> 2) & 3u);
        kept_exact |= (ex & 3u) << (2u * t);
        kept_valid |= valid << (2u * t);
        if (valid & 1u) { amin = fminf(amin, gn[0]); amax = fmaxf(amax, gn[0]); }
        if (valid & 2u) { amin = fminf(amin, gn[1]); amax = fmaxf(amax, gn[1]); }
        if (ex & valid & 1u) { emin = fminf(emin, gn[0]); emax = fmaxf(emax, gn[0]); }
        if (ex & valid & 2u) { emin = fminf(emin, gn[1]); emax = fmaxf(emax, gn[1]); }
      }
    } else {
      emin = fminf(emin, gn[0]); emax = fmaxf(emax, gn[0]);
      if (two) { emin = fminf(emin, gn[1]); emax = fmaxf(emax, gn[1]); }
    }
    uint8_t* mp = im.map + (size_t)my * c.map_w + mx;
    if (ALIGNED) {
      *reinterpret_cast<uint16_t*>(mp) = (uint16_t)(o0 | ((uint32_t)o1 << 8));
    } else {
      mp[0] = o0;
      if (two) mp[1] = o1;
    }
  }
  if (FILTER && !DEFER) return;   // (small launches without statistics: launch_generate_t)
  if (!FILTER && !stats) return;
  uint32_t* keys = stats ? c.stat_keys + (size_t)c.stat_stride * img_i : nullptr;

  if (FILTER) {
    uint32_t* ws = c.stat_ws + (size_t)kStatWords * img_i;
    uint32_t cand = 0u;   // one bit per tile: a statistics candidate
    if (stats) {
      // Which estimates could be the image's exact minimum / maximum?  An estimate a stands for an exact gain within e|a| of it, so
      // the exact minimum of the image lies at or below (1 + e) x any estimate and at or below any exact gain seen -- by this wave
      // or, through the image's words in device memory, by any wave before it.  A pixel whose a (1 - e) lies above that bound
      // cannot be the minimum; the others (the pixel that IS the minimum always among them) go onto the image's list, which
      // k_generate_resolve evaluates on the exact path after this kernel.
      wave_minmax(amin, amax);
      float wmin = emin, wmax = emax;          // what this wave already knows exactly (gain := 1 of pixels without SDR luminance)
      wave_minmax(wmin, wmax);
      const float e = c.flt_gain_rel;
      // (the words the estimates' extremes are published in: one pair per image, or -- launches of few images, kStatEst -- per list)
      uint32_t* est = c.stat_spread != 0u ? ws + kStatEst + 2u * (blk % kStatLists) : ws;
      uint32_t e0 = __hip_atomic_load(&est[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      uint32_t e1 = __hip_atomic_load(&est[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint32_t k0 = __hip_atomic_load(&keys[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint32_t k1 = __hip_atomic_load(&keys[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (amin <= amax) {   // publish this wave's estimates (one lane; the words only grow)
        const uint32_t m0 = ~float_to_key(amin), m1 = float_to_key(amax);
        if ((threadIdx.x & 63u) == 0u) {
          if (e0 < m0) atomicMax(&est[0], m0);
          if (e1 < m1) atomicMax(&est[1], m1);
        }
        e0 = e0 < m0 ? m0 : e0; e1 = e1 < m1 ? m1 : e1;
      }
      float min_hi = fminf(k1 != 0u ? key_to_float(~k0) : __builtin_inff(), wmin);   // exact gains: the bound as it stands
      float max_lo = fmaxf(k1 != 0u ? key_to_float(k1) : -__builtin_inff(), wmax);
      if (e1 != 0u) {
        const float pmin = key_to_float(~e0), pmax = key_to_float(e1);
        min_hi = fminf(min_hi, pmin + e * __builtin_fabsf(pmin));
        max_lo = fmaxf(max_lo, pmax - e * __builtin_fabsf(pmax));
      }
#pragma unroll
      for (uint32_t t = 0; t < (uint32_t)TILES; ++t)
#pragma unroll
        for (uint32_t k = 0; k < 2u; ++k) {
          const uint32_t bit = 1u << (2u * t + k);
          if ((kept_valid & bit) && !(kept_exact & bit)) {
            const float a = s_kept[(2u * t + k) * BLOCK + threadIdx.x], slack = e * __builtin_fabsf(a);
            if ((a - slack <= min_hi) || (a + slack >= max_lo)) cand |= 1u << t;
          }
        }
    }
    // One append per wave for everything it leaves to k_generate_resolve: a returning atomic per tile would park the wave for a
    // trip to the memory side in one tile out of five.
    unsigned long long mask[TILES];
    uint32_t total = 0u;
#pragma unroll
    for (uint32_t t = 0; t < (uint32_t)TILES; ++t) {
      mask[t] = __builtin_amdgcn_ballot_w64((((kept_doubt >> (2u * t)) & 3u) | ((cand >> t) & 1u)) != 0u);
      total += (uint32_t)__builtin_popcountll(mask[t]);
    }
    if (total != 0u || c.stat_slots != 0u) {
      const uint32_t lane = threadIdx.x & 63u, list = blk % kStatLists;
      uint32_t base = 0u, cap = kStatCap;
      uint32_t* lw = ws + kStatHdr + list * kStatCap;
      if (c.stat_slots != 0u) {
        // the wave's own slots: plain stores, nothing to wait for (every wave of the launch writes its count, also a zero)
        lw = wslot;
        const bool over = total > kStatSlotPlain;
        const uint32_t ns = n_saved < kStatSlotSaved ? n_saved : kStatSlotSaved;
        if (lane == 0u) {
          ws[kStatSlotCnt + blk * (uint32_t)(BLOCK / 64) + (threadIdx.x >> 6)] = over ? 0u : (total | (ns << 8));
          if (over) __hip_atomic_store(&ws[6], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sweep this image
        }
        ++lw;
        cap = over ? 0u : kStatSlotPlain;
      } else {
        if (lane == 0u) base = atomicAdd(&ws[8u + list], total);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
      }
#pragma unroll
      for (uint32_t t = 0; t < (uint32_t)TILES; ++t) {
        const uint32_t flags = ((kept_doubt >> (2u * t)) & 3u) | (((cand >> t) & 1u) << 2);
        const uint32_t pos = base + (uint32_t)__builtin_popcountll(mask[t] & ((1ull << lane) - 1ull));
        if (flags != 0u && pos < cap) lw[pos] = (((blk * (uint32_t)TILES + t) * (uint32_t)BLOCK + threadIdx.x) << 3) | flags;
        base += (uint32_t)__builtin_popcountll(mask[t]);
      }
    }
    if (!stats) return;
  }
  wave_minmax(emin, emax);
  publish_minmax(emin, emax, keys);
}

__global__ void k_stats_finalize(uint32_t* keys, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t k0 = keys[2 * i], k1 = keys[2 * i + 1];
  float mn = __builtin_inff(), mx = -__builtin_inff();  // image without map pixels
  if (k1 != 0u) { mn = key_to_float(~k0); mx = key_to_float(k1); }
  keys[2 * i] = __float_as_uint(mn);
  keys[2 * i + 1] = __float_as_uint(mx);
}

// What the filtered kernel left for an image, on the exact path: list entries are (pair index << 3 | flags) -- flags 1 / 2: that
// pixel of the pair was in doubt, its byte is replaced and its exact gain joins the statistics; flag 4: a statistics candidate,
// both gains join.  An image whose list overflowed (flat content at a code boundary: every pixel in doubt) is swept whole.
// grid = (image, kResolveSlices), each block a slice of the work; the slice that finishes last writes the image's (min, max) and
// clears the header words for the next launch.
constexpr uint32_t kResolveSlices = 16;
static_assert(kStatLists == 64u, "k_generate_resolve reads the list counts with one wave");
template <int TF, bool ALIGNED>
__global__ void __launch_bounds__(256) k_generate_resolve(const GenConsts c, const GenBatch b) {
  const uint32_t img_i = blockIdx.x;
  const GenImage& im = b.img[img_i];
  const uint8_t* im_v = im.u + (size_t)im.c_stride * (c.height / 2u);
  uint32_t* ws = c.stat_ws + (size_t)kStatWords * img_i;
  const uint32_t pairs_per_row = (c.map_w + 1u) >> 1, total = pairs_per_row * c.map_h;
  const bool stats = c.stat_out != nullptr;
  float emin = __builtin_inff(), emax = -__builtin_inff();
  // one loop for both forms (the exact path is large): thread g of the image takes pair g (sweep) or entry g of the image's lists
  // laid end to end (their exclusive prefix sums in LDS), so that the few entries fill whole waves
  __shared__ uint32_t s_first[kStatSlotWaves + 1u];
  __shared__ uint8_t s_ns[kStatSlotWaves];   // slot mode: the saved entries among a wave's entries (they come first)
  __shared__ uint32_t s_sweep;
  const bool slots = c.stat_slots != 0u;
  if (slots) {
    // slot mode: the counts of the image's waves (word 0 of every wave's slots), four per thread, prefix sums over the block
    static_assert(kStatSlotWaves == 4u * 256u, "four waves' counts per thread");
    __shared__ uint32_t s_part[4];
    uint32_t n4[4];
    const uint4 cw4 = *reinterpret_cast<const uint4*>(ws + kStatSlotCnt + 4u * threadIdx.x);   // plain entries | saved entries << 8, four waves'
    const uint32_t cws[4] = {cw4.x, cw4.y, cw4.z, cw4.w};
#pragma unroll
    for (uint32_t j = 0; j < 4u; ++j) {
      const uint32_t w = 4u * threadIdx.x + j;
      const uint32_t cw = w < c.stat_slots ? cws[j] : 0u;
      s_ns[w] = (uint8_t)(cw >> 8);
      n4[j] = (cw & 0xFFu) + (cw >> 8);
    }
    const uint32_t over = __hip_atomic_load(&ws[6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t mine = n4[0] + n4[1] + n4[2] + n4[3];
    const uint32_t before = block_exclusive_sum<256>(mine, s_part);
    s_first[4u * threadIdx.x] = before;
    s_first[4u * threadIdx.x + 1u] = before + n4[0];
    s_first[4u * threadIdx.x + 2u] = before + n4[0] + n4[1];
    s_first[4u * threadIdx.x + 3u] = before + n4[0] + n4[1] + n4[2];
    if (threadIdx.x == 255u) { s_first[kStatSlotWaves] = before + mine; s_sweep = over; }
  } else if (threadIdx.x < 64u) {   // the counts in one trip to memory (kStatLists == 64), prefix sums by wave shuffles
    const uint32_t cnt = ws[8u + threadIdx.x];
    const bool over = __builtin_amdgcn_ballot_w64(cnt > kStatCap) != 0ull;
    uint32_t inc = min(cnt, kStatCap);
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t v = (uint32_t)__shfl_up((int)inc, off, 64);
      if ((int)threadIdx.x >= off) inc += v;
    }
    s_first[threadIdx.x + 1u] = inc;
    if (threadIdx.x == 0) { s_first[0] = 0u; s_sweep = over ? 1u : 0u; }
  }
  __syncthreads();
  const bool sweep = s_sweep != 0u;
  const uint32_t nl = slots ? kStatSlotWaves : kStatLists;
  const uint32_t n = sweep ? total : s_first[nl];
#pragma unroll 1
  for (uint32_t g = blockIdx.y * 256u + threadIdx.x; g < n; g += kResolveSlices * 256u) {
    uint32_t entry = (g << 3) | 7u;
    GenMid mid;
    bool have = false;   // the pair came with its sampled values
    if (!sweep) {
      uint32_t l = 0u;
      for (uint32_t s2 = nl / 2u; s2 != 0u; s2 >>= 1) if (s_first[l + s2] <= g) l += s2;   // the list (the wave's slots) that holds entry g
      const uint32_t k = g - s_first[l];
      if (!slots) {
        entry = ws[kStatHdr + l * kStatCap + k];
      } else if (k >= (uint32_t)s_ns[l]) {
        entry = ws[kStatHdr + l * kStatSlotWords + 1u + (k - (uint32_t)s_ns[l])];
      } else {
        const uint4* e = reinterpret_cast<const uint4*>(ws + kStatHdr + l * kStatSlotWords + 16u + k * 16u);
        const uint4 e0 = e[0], e1 = e[1], e2 = e[2], e3 = e[3];
        entry = e0.x;
        mid.r = (f2){__uint_as_float(e0.y), __uint_as_float(e0.z)};
        mid.g = (f2){__uint_as_float(e0.w), __uint_as_float(e1.x)};
        mid.b = (f2){__uint_as_float(e1.y), __uint_as_float(e1.z)};
        mid.hr = (f2){__uint_as_float(e1.w), __uint_as_float(e2.x)};
        mid.hg = (f2){__uint_as_float(e2.y), __uint_as_float(e2.z)};
        mid.hb = (f2){__uint_as_float(e2.w), __uint_as_float(e3.x)};
        have = true;
      }
    }
    const uint32_t idx = entry >> 3;
    const uint32_t my = idx / pairs_per_row, pr = idx - my * pairs_per_row;
    const bool two = ALIGNED || (pr * 2u + 1u < c.map_w);
    uint8_t o[2];
    float gn[2];
    uint32_t hy[2][4][2], huv[2][2][2], y8[2][4], u8[2][2], v8[2][2];
    if (!have) {   // sample the pair (again): fourteen scattered lines
      load_pair<ALIGNED>(c, im, im_v, my, pr, two, hy, huv, y8, u8, v8);
      gen_pair<TF, false, false, false, kGenFront>(c, hy, huv, y8, u8, v8, o, gn, nullptr, nullptr, &mid);
    }
    gen_pair<TF, false, false, false, kGenBack>(c, hy, huv, y8, u8, v8, o, gn, nullptr, nullptr, &mid);
    uint8_t* mp = im.map + (size_t)my * c.map_w + 2u * pr;
    if (entry & 1u) mp[0] = o[0];
    if ((entry & 2u) && two) mp[1] = o[1];
    if ((entry & 5u) != 0u) { emin = fminf(emin, gn[0]); emax = fmaxf(emax, gn[0]); }
    if ((entry & 6u) != 0u && two) { emin = fminf(emin, gn[1]); emax = fmaxf(emax, gn[1]); }
  }
  if (stats) {
    wave_minmax(emin, emax);
    publish_minmax(emin, emax, ws + 4);
  }
  // (no fence: a release at agent scope would write this XCD's whole L2 back -- the maps k_generate has just stored.  The keys
  // are only ever touched by agent-scope atomics, which are performed at the memory side; waiting for this wave's to be
  // acknowledged before the slice counts itself finished is the order that is needed.)
  __shared__ uint32_t s_last;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) s_last = atomicAdd(&ws[3], 1u);
  __syncthreads();
  if (s_last == kResolveSlices - 1u && threadIdx.x == 0) {
    if (stats) {
      const uint32_t k0 = __hip_atomic_load(&ws[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint32_t k1 = __hip_atomic_load(&ws[5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      float mn = __builtin_inff(), mx = -__builtin_inff();  // image without map pixels
      if (k1 != 0u) { mn = key_to_float(~k0); mx = key_to_float(k1); }
      c.stat_out[2u * img_i] = mn;
      c.stat_out[2u * img_i + 1u] = mx;
    }
  }
  __syncthreads();   // (words 4 and 5 have been read)
  static_assert(kStatSlotCnt <= 256u, "one header word per thread");
  if (s_last == kResolveSlices - 1u && threadIdx.x < kStatSlotCnt)   // the header, cleared for the next launch (the waves' count words behind it are rewritten by every launch)
    __hip_atomic_store(&ws[threadIdx.x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int TF>
static hipError_t launch_stats_resolve_tf(const GenConsts& c, const GenBatch& b, int n, bool aligned, hipStream_t s) {
  if (aligned) hipLaunchKernelGGL((k_generate_resolve<TF, true>), dim3((unsigned)n, kResolveSlices), dim3(256), 0, s, c, b);
  else hipLaunchKernelGGL((k_generate_resolve<TF, false>), dim3((unsigned)n, kResolveSlices), dim3(256), 0, s, c, b);
  return hipGetLastError();
}
hipError_t launch_stats_resolve(const GenConsts& c, const GenBatch& b, int n, int hdr_tf, bool aligned, hipStream_t s) {
  if (n <= 0 || c.stat_ws == nullptr) return hipSuccess;
  switch (hdr_tf) {
    case 0: return launch_stats_resolve_tf<0>(c, b, n, aligned, s);
    case 1: return launch_stats_resolve_tf<1>(c, b, n, aligned, s);
    case 2: return launch_stats_resolve_tf<2>(c, b, n, aligned, s);
    default: return hipErrorInvalidValue;
  }
}

bool generate_is_small(const GenConsts& c, int n) {
  const uint32_t total = ((c.map_w + 1u) >> 1) * c.map_h;
  const uint32_t span = (uint32_t)kGenBlock * (uint32_t)kGenTiles;
  return (uint64_t)((total + span - 1u) / span) * (uint64_t)n < 1024u;
}

// waves per image of the filtered + deferred launch launch_generate_t makes of n such images (one span per block when the launch is
// small, kGenTiles otherwise); 0 when they do not fit the per-wave slots of the statistics workspace
uint32_t generate_slot_waves(const GenConsts& c, int n) {
  const uint32_t total = ((c.map_w + 1u) >> 1) * c.map_h;
  const uint32_t per = (uint32_t)kGenBlock * (generate_is_small(c, n) ? 1u : (uint32_t)kGenTiles);
  const uint64_t waves = (uint64_t)((total + per - 1u) / per) * (uint64_t)(kGenBlock / 64);
  return waves <= kStatSlotWaves ? (uint32_t)waves : 0u;
}

// a launch WITH statistics: is the filtered kernel + k_generate_resolve faster than the exact kernel?  (from 128 spans of 1024 pairs)
bool generate_resolve_pays(const GenConsts& c, int n) {
  const uint32_t total = ((c.map_w + 1u) >> 1) * c.map_h;
  const uint32_t span = (uint32_t)kGenBlock * (uint32_t)kGenTiles;
  return (uint64_t)((total + span - 1u) / span) * (uint64_t)n >= 128u;
}

template <int TF, bool ALIGNED, bool LUT, bool FILTER>
static hipError_t launch_generate_t(const GenConsts& c, const GenBatch& b, int n, hipStream_t s) {
  const uint32_t total = ((c.map_w + 1u) >> 1) * c.map_h;
  if (total == 0 || n == 0) return hipSuccess;
  // Spans per block: kGenTiles (fewer, longer-lived waves) where the launch still fills the chip, one otherwise -- a single
  // 4K image is 1013 spans: 254 blocks of 4 would put one wave on each SIMD and leave the memory system nothing to overlap.
  // Filtered kernel: a large launch leaves its pixels in doubt to k_generate_resolve (c.stat_ws set by the caller); a small one
  // (c.stat_ws null) runs the exact path in place, for the waves that hold such a pixel.
  const bool small = generate_is_small(c, n);
  const unsigned b4 = (total + (uint32_t)kGenBlock * (uint32_t)kGenTiles - 1u) / ((uint32_t)kGenBlock * (uint32_t)kGenTiles);
  const unsigned b1 = (total + kGenBlock - 1u) / kGenBlock;
  const dim3 g4 = dim3((unsigned)n, b4, 1);
  const dim3 g1 = dim3((unsigned)n, b1, 1);
  if (FILTER) {
    if (c.stat_ws != nullptr) {
      if (small) hipLaunchKernelGGL((k_generate<TF, ALIGNED, LUT, FILTER, 1, true>), g1, dim3(kGenBlock, 1, 1), 0, s, c, b);
      else hipLaunchKernelGGL((k_generate<TF, ALIGNED, LUT, FILTER, kGenTiles, true>), g4, dim3(kGenBlock, 1, 1), 0, s, c, b);
    } else {
      if (c.stat_keys != nullptr) return hipErrorInvalidValue;   // statistics of a filtered launch need the workspace
      // a launch whose waves are all resident at once, from one 4K image up (1013 blocks of 256 threads): 128-thread blocks give
      // the dispatcher twice the workgroups to spread -- same-box A/B: one 4K frame 11.9 -> 11.0 us; one 1080p frame (254 blocks)
      // is better off as it is (6.9 against 7.9 us)
      const unsigned h1 = (total + 127u) / 128u;
      if ((uint64_t)h1 * (uint64_t)n >= 2000u) {
        const dim3 gh = dim3((unsigned)n, h1, 1);
        hipLaunchKernelGGL((k_generate<TF, ALIGNED, LUT, FILTER, 1, false, 128>), gh, dim3(128, 1, 1), 0, s, c, b);
      } else {
        hipLaunchKernelGGL((k_generate<TF, ALIGNED, LUT, FILTER, 1, false>), g1, dim3(kGenBlock, 1, 1), 0, s, c, b);
      }
    }
  } else if (small) {
    hipLaunchKernelGGL((k_generate<TF, ALIGNED, LUT, FILTER, 1, false>), g1, dim3(kGenBlock, 1, 1), 0, s, c, b);
  } else {
    hipLaunchKernelGGL((k_generate<TF, ALIGNED, LUT, FILTER, kGenTiles, false>), g4, dim3(kGenBlock, 1, 1), 0, s, c, b);
  }
  return hipGetLastError();
}

template <int TF>
static hipError_t launch_generate_tf(const GenConsts& c, const GenBatch& b, int n, bool aligned, bool lut, bool filter,
                                     hipStream_t s) {
  if (lut) return aligned ? launch_generate_t<TF, true, true, false>(c, b, n, s) : launch_generate_t<TF, false, true, false>(c, b, n, s);
  if (filter && aligned) return launch_generate_t<TF, true, false, true>(c, b, n, s);
  return aligned ? launch_generate_t<TF, true, false, false>(c, b, n, s) : launch_generate_t<TF, false, false, false>(c, b, n, s);
}
hipError_t launch_generate(const GenConsts& c, const GenBatch& b, int n, int hdr_tf, bool aligned, bool lut,
                           bool filter, hipStream_t s) {
  if (lut && c.lut == nullptr) return hipErrorInvalidValue;
  switch (hdr_tf) {
    case 0: return launch_generate_tf<0>(c, b, n, aligned, lut, filter, s);
    case 1: return launch_generate_tf<1>(c, b, n, aligned, lut, filter, s);
    case 2: return launch_generate_tf<2>(c, b, n, aligned, lut, filter, s);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_stats_init(uint32_t* keys, int n, hipStream_t s) {
  return hipMemsetAsync(keys, 0, sizeof(uint32_t) * 2 * (size_t)n, s);
}
hipError_t launch_stats_finalize(uint32_t* keys, int n, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_stats_finalize, dim3((n + 63) / 64), dim3(64), 0, s, keys, n);
  return hipGetLastError();
}

// =================================================================================================
// apply
// =================================================================================================

// FMT: 1 HDR_LINEAR (F16), 2 HDR_PQ, 3 HDR_HLG, 4 HDR_LINEAR_RGB_10BIT (ultrahdr.h:56-64)

// Everything of ultrahdr.cpp:431-451 after the loads, for one pixel: BT.601 YUV->RGB was folded
// into (yf, crv, gcbu, gcrv, cbu) by the caller, `gain` is sampleMap's result.
template <bool EXACT>
__device__ __forceinline__ F3 recover_hdr(const AppConsts& c, float yf, float crv, float gcbu,
                                          float gcrv, float cbu, float gain) {
  float r = clamp01(yf + crv);
  float g = clamp01(yf - gcbu - gcrv);
  float b = clamp01(yf + cbu);
  // applyGain(e, gain, metadata, displayBoost)   gainmapmath.cpp:550-555
  const float log_boost = (float)(c.log2_min_d * (double)(1.0f - gain) + c.log2_max_d * (double)gain);
  F3 o;
  if (EXACT) {
    // the reference's double-precision pow / exp2, as lean f64 + a rounding test (uhdr_device_math.h): the float each call
    // returns is the one the libm call rounds to (tests/test_gpu_transfer_exhaustive.py), three channels in lock step
    float ch[3] = {r, g, b};
    srgb_inv_oetf_guarded_n<3>(ch);
    r = ch[0]; g = ch[1]; b = ch[2];
    const float factor = exp2_to_float_guarded(log_boost * c.display_boost / c.max_boost);
    o.x = (r * factor) / c.display_boost;  // ultrahdr.cpp:451
    o.y = (g * factor) / c.display_boost;
    o.z = (b * factor) / c.display_boost;
  } else {
    r = srgb_inv_oetf_fast(r); g = srgb_inv_oetf_fast(g); b = srgb_inv_oetf_fast(b);
    const float factor = __builtin_amdgcn_exp2f((log_boost * c.display_boost) * c.inv_max_boost);
    o.x = (r * factor) * c.inv_display_boost;
    o.y = (g * factor) * c.inv_display_boost;
    o.z = (b * factor) * c.inv_display_boost;
  }
  return o;
}

template <int FMT, bool EXACT>
__device__ __forceinline__ F3 hdr_oetf(F3 e) {
  if (FMT == 3) {
    if (EXACT) { float ch[3] = {e.x, e.y, e.z}; hlg_oetf_guarded_n<3>(ch); e.x = ch[0]; e.y = ch[1]; e.z = ch[2]; }
    else { e.x = hlg_oetf_fast(e.x); e.y = hlg_oetf_fast(e.y); e.z = hlg_oetf_fast(e.z); }
  } else if (FMT == 2) {
    if (EXACT) { float ch[3] = {e.x, e.y, e.z}; pq_oetf_guarded_n<3>(ch); e.x = ch[0]; e.y = ch[1]; e.z = ch[2]; }
    else { e.x = pq_oetf_est(e.x); e.y = pq_oetf_est(e.y); e.z = pq_oetf_est(e.z); }   // (within 3e-4 of a code; pq_oetf_fast: 0.3)
  }
  return e;
}

__device__ __forceinline__ float map_to_float(uint32_t v) { return (float)v / 255.0f; }  // gainmapmath.cpp:632
// the same value in 3 issue slots instead of the ~10 of an IEEE division: exact for all 256 map bytes
// (tests/test_gpu_transfer_exhaustive.py::test_map_byte_to_float_is_exact)
__device__ __forceinline__ float map_to_float_fast(uint32_t v) { return div_const((float)v, 255.0f, 1.0f / 255.0f); }

// ---- FAST path, scale factor 4 ---------------------------------------------------------------
// Thread = one gain-map cell = a 4x4 pixel block.  The four map taps and the four 2x2 chroma samples
// are loaded once and shared by the 16 pixels; each output row of the block leaves as one 16 B
// (1010102) / 2x16 B (F16) / 3x8 B (planar 10 bit) store per lane, so a wave64 writes 1 KiB contiguous
// per row.  The kernel is bound by VALU issue, not by HBM (13 special-function ops per pixel, see
// DESIGN.md section 6), so the arithmetic is organised to minimise issue slots:
//   * two horizontally adjacent pixels (they share one chroma sample) are processed as one packed
//     float2 (v_pk_mul/add/fma_f32: two pixels per 4-cycle issue slot);
//   * gain -> log-boost -> display scaling collapse into one exponent:
//       factor/display_boost = 2^E,  E = sum_i e_i*(w_i*A) + B,
//       A = (log2 max - log2 min) * display_boost / max,  B = log2 min * display_boost / max - log2 display_boost
//     with the products w_i*A pre-multiplied on the host (AppFast::wA, SGPR operands);
//   * the *1023 of the 10-bit pack is folded into the OETF constants;
//   * clamps are written as min(max()) so that they fold into the producing add's clamp modifier.
__device__ __forceinline__ f2 log2_2(f2 v) { return (f2){__builtin_amdgcn_logf(v.x), __builtin_amdgcn_logf(v.y)}; }
__device__ __forceinline__ f2 exp2_2(f2 v) { return (f2){__builtin_amdgcn_exp2f(v.x), __builtin_amdgcn_exp2f(v.y)}; }
__device__ __forceinline__ f2 sqrt_2(f2 v) { return (f2){__builtin_amdgcn_sqrtf(v.x), __builtin_amdgcn_sqrtf(v.y)}; }
__device__ __forceinline__ f2 rcp_2(f2 v) { return (f2){__builtin_amdgcn_rcpf(v.x), __builtin_amdgcn_rcpf(v.y)}; }
__device__ __forceinline__ f2 sel_le(f2 x, float thr, f2 a, f2 b) {  // x <= thr ? a : b
  return (f2){x.x <= thr ? a.x : b.x, x.y <= thr ? a.y : b.y};
}

// The piecewise transfer functions are evaluated WITHOUT per-lane selects (v_cmp + v_cndmask cost 2 issue
// slots per value and cannot be packed).  For a function  f(e) = A(e) for e <= j,  B(e) for e > j :
//     f(e) = A(min(e, j)) - A(j) + B(max(e, j)),
// with  min(e, j) = j * sat(e / j)  and  max(e, j) = j + sat(e - j)  (0 <= e <= ~1), both packed ops with the
// clamp modifier.  The identity is exact for e > j; for e <= j it adds B(j) - A(j), the reference function's own
// jump at the junction (2.3e-9 for sRGB, ~1e-7 for HLG): far inside the 1-LSB tolerance.

// sRGB EOTF of two values (gainmapmath.cpp:149-155): x^2.4 = x^2 * 2^(0.4 log2 x)
__device__ __forceinline__ f2 srgb_eotf2(f2 e) {
  constexpr float kThr = 0.04045f;
  const f2 a = pk_mul_sat(e, splat(1.0f / kThr));                                      // min(e, thr) / thr
  const f2 lin = pk_fma(a, splat(kThr / 12.92f), splat(-(kThr / 12.92f)));             // lin(min) - lin(thr)
  const f2 d = pk_add_sat(e, splat(-kThr));                                            // max(e, thr) - thr
  const f2 x = pk_fma(d, splat(1.0f / 1.055f), splat((kThr + 0.055f) / 1.055f));
  const f2 t = exp2_2(log2_2(x) * splat(0.4f));
  return pk_fma(x * x, t, lin);
}

// ---- transfer functions as line segments in LDS ---------------------------------------------------
// Both transfer functions of a channel -- the sRGB EOTF in front of the gain, the HLG / PQ OETF behind it -- cost 13 (PQ: 22)
// special-function evaluations per pixel on the VALU; they are read from line-segment tables in LDS instead (the LDS pipe is
// otherwise idle in this kernel).  What bounds a table in LDS is not its size but bank conflicts: 32 lanes reading 32 random
// 8-byte entries take 3-4 cycles instead of 1, and six such gathers per pixel made the LDS the bottleneck (PMC: 2/3 of its
// cycles were conflicts).  So the path is arranged for ONE conflicting gather per value (uhdr_kernels.h has the arithmetic):
//   stage 1: T(c) = EOTF(c)^g.  Cell = bits 14..3 of the half-precision bit pattern of c: v_cvt_pkrtz_f16_f32 converts two
//     values in one issue slot, one v_and each (the upper half through SDWA's WORD_1 select) gives the byte offset of the
//     8-byte (c0, c1) entry as it stands; one ds_read_b64, one fma.  Shared table, 15 KiB, conflicts as they come.
//   u = T * 2^(g E) is formed as s = 2 + 2u by one packed fma; s lies in [2, 4], so byte 2 of its bit pattern IS the number of
//     its 1/128-wide cell: v_perm_b32 puts that byte next to the lane's slot number -- the address of a private copy of the
//     entry (cell * 256 + (lane & 31) * 8).  32 lanes, 32 different bank pairs: no conflict whatever the pixels hold.
//   stage 2: the fma on that entry yields -(2 + code 2^-22) rounded toward zero (the kernel runs in that rounding mode), whose
//     bits are 0xC0000000 | code: the pixel is two v_lshl_or_b32 of the three results.
// Errors: stage 1 <= 2e-6 relative; stage 2 <= 0.05 codes next to the HLG junction, <= 0.01 elsewhere and for PQ
// (tests/test_gpu_apply_tables.py measures both for every float).
// byte offset of the cell of one value (diagnostics; the kernel converts two at a time)
__device__ __forceinline__ uint32_t tab_offset(float x) {
  typedef __fp16 h2 __attribute__((ext_vector_type(2)));
  const h2 q = __builtin_amdgcn_cvt_pkrtz(x, 0.0f);
  return __builtin_bit_cast(uint32_t, q) & 0x7FF8u;
}
// stage-1 tables for g < 1: the byte offset of a value's cell is its own bits 27..19 (the low five exponent bits -- the kernel's
// inputs are 0 or lie in [2^-31, 1], exponents 96..127: build_stage1 in uhdr_capi.hip -- and the top four of the mantissa), in
// place: one SDWA v_and on the upper half.  4 KiB per table; offset 0 is the entry of the input 0.
__device__ __forceinline__ uint32_t pow_offset(float x) {
  uint32_t off;
  asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(off) : "v"(0x0FF8u), "v"(x));
  return off;
}
__device__ __forceinline__ void pow_pair(f2 x, const char* lut, float2& a, float2& b) {
  a = *reinterpret_cast<const float2*>(lut + pow_offset(x.x));
  b = *reinterpret_cast<const float2*>(lut + pow_offset(x.y));
}
template <uint32_t BASE>
__device__ __forceinline__ void tab_pair(f2 x, const char* lut, float2& a, float2& b) {
  typedef __fp16 h2 __attribute__((ext_vector_type(2)));
  const h2 q = __builtin_amdgcn_cvt_pkrtz(x.x, x.y);
  const uint32_t d = __builtin_bit_cast(uint32_t, q);
  uint32_t hi_off;
  asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(hi_off) : "v"(0x7FF8u), "v"(d));
  a = *reinterpret_cast<const float2*>(lut + BASE + (d & 0x7FF8u));
  b = *reinterpret_cast<const float2*>(lut + BASE + hi_off);
}
// c0 + c1 * x for both values.  Two plain fmas on purpose (asm: the vectoriser would re-pack them behind three v_movs that shuffle
// the operands into pairs)
template <uint32_t BASE>
__device__ __forceinline__ f2 tab_eval2(f2 x, const char* lut) {
  float2 a, b;
  tab_pair<BASE>(x, lut, a, b);
  f2 o;
  asm("v_fma_f32 %0, %1, %2, %3" : "=v"(o.x) : "v"(a.y), "v"(x.x), "v"(a.x));
  asm("v_fma_f32 %0, %1, %2, %3" : "=v"(o.y) : "v"(b.y), "v"(x.y), "v"(b.x));
  return o;
}
// stage 2 on s = 2 + 2u (both values): the lane's private copy of the cell's entry sits at byte2(s) * 256 + slot8
__device__ __forceinline__ uint32_t s2_address(float s, uint32_t slot8) {
  return __builtin_amdgcn_perm(__float_as_uint(s), slot8, 0x0C0C0600u);   // {0, 0, s.byte2, slot8.byte0}
}
__device__ __forceinline__ f2 code_eval2(f2 s, uint32_t slot8, const char* lut) {   // lut: the stage-2 table
  const float2 a = *reinterpret_cast<const float2*>(lut + s2_address(s.x, slot8));
  const float2 b = *reinterpret_cast<const float2*>(lut + s2_address(s.y, slot8));
  f2 o;
  asm("v_fma_f32 %0, %1, %2, %3" : "=v"(o.x) : "v"(a.y), "v"(s.x), "v"(a.x));
  asm("v_fma_f32 %0, %1, %2, %3" : "=v"(o.y) : "v"(b.y), "v"(s.y), "v"(b.x));
  return o;
}

// LDS layout of the kernel for an output format: the replicated stage-2 table (HLG / PQ output of a call whose values stay within
// [0, 1]), then the stage-1 table
template <int FMT, bool MASK> struct ApplyTab {
  static constexpr bool kOetf = !MASK && (FMT == 2 || FMT == 3);
  static constexpr uint32_t kS1Bytes = kOetf ? kTabS1PowBytes : kTabS1Bytes;   // at LDS offset 0: its cell offsets are addresses
  static constexpr uint32_t kS2Base = kS1Bytes;
  static constexpr uint32_t kS2Bytes = kOetf ? kTabS2LdsBytes : 0u;
  static constexpr uint32_t kBytes = kS1Bytes + kS2Bytes;
  static constexpr uint32_t kS1Float = !kOetf ? kTabS1Lin : FMT == 3 ? kTabS1Hlg : kTabS1Pq;
  static constexpr uint32_t kS2Float = FMT == 3 ? kTabS2Hlg : kTabS2Pq;
};

// OETF of two linear values, scaled to 10-bit code units where the format is 10 bit: the special-function forms, used where the
// values may exceed 1.0 (max_display_boost < maxContentBoost: the tables end at 1.0, and the HLG junction is selected per lane
// because max(e, j) - j cannot be had from a [0,1] clamp) and for the two linear formats.
template <int FMT>
__device__ __forceinline__ f2 oetf2_scaled(f2 e) {
  if (FMT == 3) {  // HLG (gainmapmath.cpp:259-265): sqrt(3e) | a ln(12e-b)+c, times 1023
    const f2 lo = sqrt_2(e * splat(3.0f * 1023.0f * 1023.0f));
    const f2 hi = pk_fma(log2_2(pk_fma(e, splat(12.0f), splat(-UHDR_HLG_B))),
                         splat(UHDR_HLG_A * 0.693147180559945f * 1023.0f), splat(UHDR_HLG_C * 1023.0f));
    return sel_le(e, 1.0f / 12.0f, lo, hi);
  } else if (FMT == 2) {  // PQ (gainmapmath.cpp:309-312): ((c1 + c2 e^m1) / (1 + c3 e^m1))^m2, times 1023
    // e == 0: log2 -> -inf, e^m1 -> 0, result c1^m2 * 1023 = 7e-4, which truncates to the reference's 0
    const f2 p = exp2_2(log2_2(e) * splat(UHDR_PQ_M1));
    const f2 q = pk_fma(p, splat(UHDR_PQ_C2), splat(UHDR_PQ_C1)) * rcp_2(pk_fma(p, splat(UHDR_PQ_C3), splat(1.0f)));
    return exp2_2(pk_fma(log2_2(q), splat(UHDR_PQ_M2), splat(9.99859042974533f)));     // + log2(1023)
  } else if (FMT == 4) {
    return e * splat(1023.0f);
  }
  return e;
}

struct PairOut { f2 r, g, b; };

// inputs already *1023.  The reference's `& 0x3ff` (gainmapmath.cpp:723-725) can only bite when a channel
// reaches 1024, i.e. when max_display_boost < maxContentBoost lets values exceed 1.0; MASK is a per-call
// (wave-uniform) property, so the common case packs with two v_lshl_or_b32 and one v_or_b32.
template <bool MASK>
__device__ __forceinline__ uint32_t pack10_scaled(float r, float g, float b) {
  uint32_t ri = (uint32_t)r, gi = (uint32_t)g, bi = (uint32_t)b;
  if (MASK) { ri &= 0x3ffu; gi &= 0x3ffu; bi &= 0x3ffu; }
  uint32_t t;  // hipcc splits (g << 10) | r into a shift and an or; v_lshl_or_b32 does it in one slot
  asm("v_lshl_or_b32 %0, %1, 10, %2" : "=v"(t) : "v"(gi), "v"(ri));
  asm("v_lshl_or_b32 %0, %1, 20, %2" : "=v"(t) : "v"(bi), "v"(t));
  return t | 0xC0000000u;
}
// the same from three OETF-table results: red's bits are 0xC0000000 | code (alpha included); the shifts push the upper bits of
// green and blue out of the word
__device__ __forceinline__ uint32_t pack10_bits(float r, float g, float b) {
  uint32_t t;
  asm("v_lshl_or_b32 %0, %1, 10, %2" : "=v"(t) : "v"(__float_as_uint(g)), "v"(__float_as_uint(r)));
  asm("v_lshl_or_b32 %0, %1, 20, %2" : "=v"(t) : "v"(__float_as_uint(b)), "v"(t));
  return t;
}

// one float of byte k of a word: v_cvt_f32_ubyte<k>
template <int K>
__device__ __forceinline__ float cvt_byte(uint32_t w) { return (float)((w >> (8 * K)) & 0xffu); }

// One map cell (4x4 pixels) for HLG / PQ output through both tables, as a software pipeline over its 8 pixel pairs (two
// horizontally adjacent pixels share one chroma sample and run as one packed float2).  A pair needs two LDS round
// trips (stage 1, stage 2), and the compiler, left alone, waits for each pair of reads right after issuing it (a SIMD's four waves
// then spend 60 % of their time parked at s_waitcnt).  Here every iteration runs three stages of three different pairs,
//   F(k):   E, factor, r g b of pair k, their cells; issues the 6 stage-1 reads of pair k
//   M(k-1): the 6 fmas on the stage-1 entries of pair k-1, s = 2 + 2 T factor, the 6 slot addresses; issues its 6 stage-2 reads
//   B(k-2): the 6 fmas on the stage-2 entries of pair k-2, packs its two pixels (and stores the row when it is complete)
// so that every read has a whole iteration (~50 VALU instructions) between issue and use.  sched_barrier keeps the stages apart.
struct PairF { f2 c[3]; f2 factor; float2 e[6]; };
struct PairM { f2 s[3]; float2 e[6]; };
// the chroma terms of the BT.601 YUV->RGB of gainmapmath.cpp:184-202 for one row of two chroma samples, scaled like the luma
struct ChromaRow { f2 crv, ngs, cbu; };
__device__ __forceinline__ ChromaRow chroma_row(uint32_t uu, uint32_t vv) {
  constexpr float kCrS = kP3Cr * k255, kCbS = kP3Cb * k255, kGCbS = kP3GCb * k255, kGCrS = kP3GCr * k255;
  const f2 uf = (f2){cvt_byte<0>(uu), cvt_byte<1>(uu)};
  const f2 vf = (f2){cvt_byte<0>(vv), cvt_byte<1>(vv)};
  ChromaRow o;
  o.crv = pk_fma(vf, splat(kCrS), splat(-128.0f * kCrS));
  o.cbu = pk_fma(uf, splat(kCbS), splat(-128.0f * kCbS));
  o.ngs = pk_fma(uf, splat(-kGCbS), pk_fma(vf, splat(-kGCrS), splat(128.0f * kGCbS + 128.0f * kGCrS)));
  return o;
}
template <int FMT>
__device__ __forceinline__ void apply_cell_piped(const AppConsts& c, void* dst, uint32_t cx, uint32_t cy,
                                                 const uint32_t (&yrow)[4], const uint32_t (&uu)[2], const uint32_t (&vv)[2],
                                                 float m1, float m2, float m3, float m4, uint32_t slot8, const char* lut) {
  typedef ApplyTab<FMT, false> T;
  // only the chroma row in use is kept as floats (the second row's six values are formed when rows 2 and 3 start): the cell runs
  // at the register budget of four waves per SIMD
  ChromaRow cr = chroma_row(uu[0], vv[0]);
  const float a255 = c.fast.A255;
  // The four per-cell scalars of the exponent travel as two register PAIRS whose halves v_pk_fma_f32 broadcasts through op_sel: a
  // scalar splat by the compiler occupies a pair as well, with an undefined upper register -- which the allocator is free to
  // place on a register an outstanding load of the next cell will write, and the s_waitcnt in front of the fma then waits for
  // that load (seen in the ISA: one full trip to HBM per cell).
  f2 d23 = (f2){m2 - m1, m3 - m1}, d4b = (f2){m4 - m1, __builtin_fmaf(m1, a255, c.fast.B)};
  asm("" : "+v"(d23), "+v"(d4b));   // (opaque: or the pairs are taken apart again)
  PairF pf[2];
  PairM pm[2];
  uint32_t px[4];
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    if (k < 8) {  // ---- F(k)
      const int oy = k >> 1, pr = k & 1;
      // sampleMap's standard weights (cells of the last column / row: apply_cell_edge) times A / 255: SGPR pairs
      const f2 w1 = (f2){c.fast.wD[oy][pr][0][0], c.fast.wD[oy][pr][0][1]}, w2 = (f2){c.fast.wD[oy][pr][1][0], c.fast.wD[oy][pr][1][1]};
      const f2 w3 = (f2){c.fast.wD[oy][pr][2][0], c.fast.wD[oy][pr][2][1]};
      if (k == 4) cr = chroma_row(uu[1], vv[1]);
      PairF& f = pf[k & 1];
      const f2 E = pk_fma(bc_lo(d4b), w3, pk_fma(bc_hi(d23), w2, pk_fma(bc_lo(d23), w1, bc_hi(d4b))));
      const f2 yraw = pr ? (f2){cvt_byte<2>(yrow[oy]), cvt_byte<3>(yrow[oy])} : (f2){cvt_byte<0>(yrow[oy]), cvt_byte<1>(yrow[oy])};
      if (pr) {
        f.c[0] = pk_fma_sat_bc<1>(yraw, splat(k255), cr.crv); f.c[1] = pk_fma_sat_bc<1>(yraw, splat(k255), cr.ngs);
        f.c[2] = pk_fma_sat_bc<1>(yraw, splat(k255), cr.cbu);
      } else {
        f.c[0] = pk_fma_sat_bc<0>(yraw, splat(k255), cr.crv); f.c[1] = pk_fma_sat_bc<0>(yraw, splat(k255), cr.ngs);
        f.c[2] = pk_fma_sat_bc<0>(yraw, splat(k255), cr.cbu);
      }
      f.factor = exp2_2(E);
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) pow_pair(f.c[ch], lut, f.e[2 * ch], f.e[2 * ch + 1]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (k >= 1 && k <= 8) {  // ---- M(k-1)
      PairF& f = pf[(k - 1) & 1];
      PairM& m = pm[(k - 1) & 1];
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        f2 t;
        asm("v_fma_f32 %0, %1, %2, %3" : "=v"(t.x) : "v"(f.e[2 * ch].y), "v"(f.c[ch].x), "v"(f.e[2 * ch].x));
        asm("v_fma_f32 %0, %1, %2, %3" : "=v"(t.y) : "v"(f.e[2 * ch + 1].y), "v"(f.c[ch].y), "v"(f.e[2 * ch + 1].x));
        m.s[ch] = pk_fma(t, f.factor, splat(2.0f));
      }
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        m.e[2 * ch] = *reinterpret_cast<const float2*>(lut + T::kS2Base + s2_address(m.s[ch].x, slot8));
        m.e[2 * ch + 1] = *reinterpret_cast<const float2*>(lut + T::kS2Base + s2_address(m.s[ch].y, slot8));
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (k >= 2) {  // ---- B(k-2)
      const int q = k - 2, oy = q >> 1, pr = q & 1;
      PairM& m = pm[q & 1];
      float o[6];
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        asm("v_fma_f32 %0, %1, %2, %3" : "=v"(o[2 * ch]) : "v"(m.e[2 * ch].y), "v"(m.s[ch].x), "v"(m.e[2 * ch].x));
        asm("v_fma_f32 %0, %1, %2, %3" : "=v"(o[2 * ch + 1]) : "v"(m.e[2 * ch + 1].y), "v"(m.s[ch].y), "v"(m.e[2 * ch + 1].x));
      }
      px[2 * pr] = pack10_bits(o[0], o[2], o[4]);
      px[2 * pr + 1] = pack10_bits(o[1], o[3], o[5]);
      if (pr) {
        const uint32_t off = ((4u * cy + oy) * c.width + 4u * cx) * 4u;  // (a 32-bit byte offset: app_fast_s4 keeps the image below 4 GiB)
        st_stream(reinterpret_cast<uint4*>(static_cast<char*>(dst) + off), make_uint4(px[0], px[1], px[2], px[3]));
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// The cell for the outputs without a stage-2 table (the two linear formats; HLG / PQ of a call whose values may exceed 1.0) in the
// same manner, two stages: F(k) as above on the g = 1 table, B(k-1) = the 6 fmas on its entries, the factor, the special-function
// OETF where there is one, the pack.
constexpr uint32_t kXchSecond = 72;    // in uint4: the second halves start 1152 bytes into the wave's exchange area ...
constexpr uint32_t kXchPerWave = 136;  // ... of 2176 bytes
template <int FMT, bool MASK>
__device__ __forceinline__ void apply_cell_piped1(const AppConsts& c, const AppImage& im, uint32_t cx, uint32_t cy,
                                                  const uint32_t (&yrow)[4], const uint32_t (&uu)[2], const uint32_t (&vv)[2],
                                                  float m1, float m2, float m3, float m4, const char* lut, uint4* xch) {
  ChromaRow cr = chroma_row(uu[0], vv[0]);
  const float a255 = c.fast.A255;
  // The four per-cell scalars of the exponent travel as two register PAIRS whose halves v_pk_fma_f32 broadcasts through op_sel: a
  // scalar splat by the compiler occupies a pair as well, with an undefined upper register -- which the allocator is free to
  // place on a register an outstanding load of the next cell will write, and the s_waitcnt in front of the fma then waits for
  // that load (seen in the ISA: one full trip to HBM per cell).
  f2 d23 = (f2){m2 - m1, m3 - m1}, d4b = (f2){m4 - m1, __builtin_fmaf(m1, a255, c.fast.B)};
  asm("" : "+v"(d23), "+v"(d4b));   // (opaque: or the pairs are taken apart again)
  PairF pf[2];
  PairOut po[2];
  const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    if (k < 8) {  // ---- F(k)
      const int oy = k >> 1, pr = k & 1;
      // sampleMap's standard weights (cells of the last column / row: apply_cell_edge) times A / 255: SGPR pairs
      const f2 w1 = (f2){c.fast.wD[oy][pr][0][0], c.fast.wD[oy][pr][0][1]}, w2 = (f2){c.fast.wD[oy][pr][1][0], c.fast.wD[oy][pr][1][1]};
      const f2 w3 = (f2){c.fast.wD[oy][pr][2][0], c.fast.wD[oy][pr][2][1]};
      if (k == 4) cr = chroma_row(uu[1], vv[1]);
      PairF& f = pf[k & 1];
      const f2 E = pk_fma(bc_lo(d4b), w3, pk_fma(bc_hi(d23), w2, pk_fma(bc_lo(d23), w1, bc_hi(d4b))));
      const f2 yraw = pr ? (f2){cvt_byte<2>(yrow[oy]), cvt_byte<3>(yrow[oy])} : (f2){cvt_byte<0>(yrow[oy]), cvt_byte<1>(yrow[oy])};
      if (pr) {
        f.c[0] = pk_fma_sat_bc<1>(yraw, splat(k255), cr.crv); f.c[1] = pk_fma_sat_bc<1>(yraw, splat(k255), cr.ngs);
        f.c[2] = pk_fma_sat_bc<1>(yraw, splat(k255), cr.cbu);
      } else {
        f.c[0] = pk_fma_sat_bc<0>(yraw, splat(k255), cr.crv); f.c[1] = pk_fma_sat_bc<0>(yraw, splat(k255), cr.ngs);
        f.c[2] = pk_fma_sat_bc<0>(yraw, splat(k255), cr.cbu);
      }
      f.factor = exp2_2(E);
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) tab_pair<0>(f.c[ch], lut, f.e[2 * ch], f.e[2 * ch + 1]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (k >= 1) {  // ---- B(k-1)
      const int q = k - 1, oy = q >> 1, pr = q & 1;
      PairF& f = pf[q & 1];
      f2 lin[3];
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        f2 t;
        asm("v_fma_f32 %0, %1, %2, %3" : "=v"(t.x) : "v"(f.e[2 * ch].y), "v"(f.c[ch].x), "v"(f.e[2 * ch].x));
        asm("v_fma_f32 %0, %1, %2, %3" : "=v"(t.y) : "v"(f.e[2 * ch + 1].y), "v"(f.c[ch].y), "v"(f.e[2 * ch + 1].x));
        lin[ch] = oetf2_scaled<FMT>(t * f.factor);
      }
      po[pr].r = lin[0]; po[pr].g = lin[1]; po[pr].b = lin[2];
      if (pr) {
        const uint32_t pix0 = (4u * cy + oy) * c.width + 4u * cx;  // < 2^27 pixels per image
        if (FMT == 2 || FMT == 3) {
          uint4 o;
          o.x = pack10_scaled<MASK>(po[0].r.x, po[0].g.x, po[0].b.x); o.y = pack10_scaled<MASK>(po[0].r.y, po[0].g.y, po[0].b.y);
          o.z = pack10_scaled<MASK>(po[1].r.x, po[1].g.x, po[1].b.x); o.w = pack10_scaled<MASK>(po[1].r.y, po[1].g.y, po[1].b.y);
          st_stream(reinterpret_cast<uint4*>(static_cast<uint32_t*>(im.dst) + pix0), o);
        } else if (FMT == 1) {
          const uint2 a = pack_f16_hw(po[0].r.x, po[0].g.x, po[0].b.x), bb = pack_f16_hw(po[0].r.y, po[0].g.y, po[0].b.y);
          const uint2 cc = pack_f16_hw(po[1].r.x, po[1].g.x, po[1].b.x), d = pack_f16_hw(po[1].r.y, po[1].g.y, po[1].b.y);
          if (xch != nullptr) {
            // A lane's row is 32 bytes: stored as it stands, every store instruction would fill half of each line it touches
            // (measured: 2.0 instead of 5.6 TB/s, scripts/ab/store_pattern).  The wave's 64 rows pass through LDS so that each of
            // the two instructions writes 1 KiB contiguous: lane L stores half (L & 1) of the row of lane L / 2 (then of lane
            // 32 + L / 2).  First halves at 16 L, second halves at 1152 + 16 L: no bank conflict either way.
            xch[lane] = make_uint4(a.x, a.y, bb.x, bb.y);
            xch[kXchSecond + lane] = make_uint4(cc.x, cc.y, d.x, d.y);
            __builtin_amdgcn_wave_barrier();
            const uint32_t from = (lane & 1u) * kXchSecond + (lane >> 1);
            const uint4 s0 = xch[from], s1 = xch[from + 32u];
            __builtin_amdgcn_wave_barrier();
            uint4* o = reinterpret_cast<uint4*>(static_cast<uint2*>(im.dst) + (uint32_t)__builtin_amdgcn_readfirstlane((int)pix0));
            st_stream(o + lane, s0);
            st_stream(o + 64u + lane, s1);
          } else {
            uint4* o = reinterpret_cast<uint4*>(static_cast<uint2*>(im.dst) + pix0);
            o[0] = make_uint4(a.x, a.y, bb.x, bb.y);
            o[1] = make_uint4(cc.x, cc.y, d.x, d.y);
          }
        } else {  // FMT == 4: planar R,G,B uint16 (ultrahdr.cpp:460-468)
          const size_t plane = (size_t)c.width * c.height;
          uint16_t* base16 = static_cast<uint16_t*>(im.dst);
          const f2 ch[3][2] = {{po[0].r, po[1].r}, {po[0].g, po[1].g}, {po[0].b, po[1].b}};
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            const uint32_t q0 = 0x3ffu & (uint32_t)ch[p][0].x, q1 = 0x3ffu & (uint32_t)ch[p][0].y;
            const uint32_t q2 = 0x3ffu & (uint32_t)ch[p][1].x, q3 = 0x3ffu & (uint32_t)ch[p][1].y;
            st_stream(reinterpret_cast<uint2*>(base16 + p * plane + pix0), make_uint2(q0 | (q1 << 16), q2 | (q3 << 16)));
          }
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// the bytes of one map cell's 4x4 pixels: four luma words, two rows of two chroma samples per plane, the four sampleMap taps
// (gainmapmath.cpp:690-703: the reference indexes the map with map->width).  The taps stay bytes: the / 255 of gainmapmath.cpp:632
// is folded into the weights.  32-bit offsets (an image plane is < 4 GiB): one 64-bit add per address.
struct ApplyCellIn { uint32_t yrow[4], uu[2], vv[2], mb[4]; };
__device__ __forceinline__ void apply_load_cell(const AppConsts& c, const AppImage& im, uint32_t cx, uint32_t cy, ApplyCellIn& o) {
  const uint32_t yoff = 4u * cy * im.y_stride + 4u * cx;
#pragma unroll
  for (int r = 0; r < 4; ++r) o.yrow[r] = ld_stream(reinterpret_cast<const uint32_t*>(im.y + (yoff + r * im.y_stride)));
  const uint32_t coff = 2u * cy * im.c_stride + 2u * cx;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    o.uu[r] = *reinterpret_cast<const uint16_t*>(im.u + (coff + r * im.c_stride));
    o.vv[r] = *reinterpret_cast<const uint16_t*>(im.v + (coff + r * im.c_stride));
  }
  const uint32_t xu = min(cx + 1u, c.map_w - 1u), yu = min(cy + 1u, c.map_h - 1u);
  const uint32_t m0 = cy * c.map_w, m1 = yu * c.map_w;
  o.mb[0] = im.map[m0 + cx]; o.mb[1] = im.map[m1 + cx]; o.mb[2] = im.map[m0 + xu]; o.mb[3] = im.map[m1 + xu];
}

// A cell with per-lane weights (the last column / row of the map: the NR / NB / C tables of gainmapmath.h:184-228): the arithmetic
// of the walk's cells, row by row in a rolled loop, a row's two pixel pairs side by side.  Some 1500 cells of a 4K frame take this
// path, in the waves of the walk that touch the last column / row.  It is short and needs few registers -- a second copy of the
// pipelined cell with per-lane weights, inlined next to the first, made the allocator spill in the loop they share -- and nothing
// in it is pipelined.
template <int FMT, bool MASK>
__device__ __forceinline__ void apply_cell_edge(const AppConsts& c, const AppImage& im, uint32_t cx, uint32_t cy,
                                                const uint32_t (&yrow)[4], const uint32_t (&uu)[2], const uint32_t (&vv)[2],
                                                float m1, float m2, float m3, float m4, const float* wt /* the lane's table, in LDS */,
                                                uint32_t slot8, const char* lut) {
  typedef ApplyTab<FMT, MASK> T;
  const float a255 = c.fast.A255;
  const float base = __builtin_fmaf(m1, a255, c.fast.B), d2 = m2 - m1, d3 = m3 - m1, d4 = m4 - m1;
#pragma unroll 1
  for (uint32_t oy = 0; oy < 4u; ++oy) {   // (uniform: the selects below are scalar)
    const uint32_t yw = oy == 0u ? yrow[0] : oy == 1u ? yrow[1] : oy == 2u ? yrow[2] : yrow[3];
    const ChromaRow cr = chroma_row(oy < 2u ? uu[0] : uu[1], oy < 2u ? vv[0] : vv[1]);
    const uint32_t pix0 = (4u * cy + oy) * c.width + 4u * cx;  // < 2^27 pixels per image
    uint32_t px[4];
    PairOut po[2];
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      const f2 yraw = pr ? (f2){cvt_byte<2>(yw), cvt_byte<3>(yw)} : (f2){cvt_byte<0>(yw), cvt_byte<1>(yw)};
      const float* p0 = wt + oy * 16u + pr * 8;
      const f2 w1 = (f2){p0[1], p0[5]}, w2 = (f2){p0[2], p0[6]}, w3 = (f2){p0[3], p0[7]};   // (times A / 255 already: the kernel's prologue)
      const f2 E = pk_fma(splat(d4), w3, pk_fma(splat(d3), w2, pk_fma(splat(d2), w1, splat(base))));
      const f2 factor = exp2_2(E);
      f2 ch[3];
      if (pr) {
        ch[0] = pk_fma_sat_bc<1>(yraw, splat(k255), cr.crv); ch[1] = pk_fma_sat_bc<1>(yraw, splat(k255), cr.ngs); ch[2] = pk_fma_sat_bc<1>(yraw, splat(k255), cr.cbu);
      } else {
        ch[0] = pk_fma_sat_bc<0>(yraw, splat(k255), cr.crv); ch[1] = pk_fma_sat_bc<0>(yraw, splat(k255), cr.ngs); ch[2] = pk_fma_sat_bc<0>(yraw, splat(k255), cr.cbu);
      }
      if (T::kOetf) {
        float o[6];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          float2 a, b;
          pow_pair(ch[q], lut, a, b);
          f2 t;
          asm("v_fma_f32 %0, %1, %2, %3" : "=v"(t.x) : "v"(a.y), "v"(ch[q].x), "v"(a.x));
          asm("v_fma_f32 %0, %1, %2, %3" : "=v"(t.y) : "v"(b.y), "v"(ch[q].y), "v"(b.x));
          const f2 s = pk_fma(t, factor, splat(2.0f));
          const float2 ea = *reinterpret_cast<const float2*>(lut + T::kS2Base + s2_address(s.x, slot8));
          const float2 eb = *reinterpret_cast<const float2*>(lut + T::kS2Base + s2_address(s.y, slot8));
          asm("v_fma_f32 %0, %1, %2, %3" : "=v"(o[2 * q]) : "v"(ea.y), "v"(s.x), "v"(ea.x));
          asm("v_fma_f32 %0, %1, %2, %3" : "=v"(o[2 * q + 1]) : "v"(eb.y), "v"(s.y), "v"(eb.x));
        }
        px[2 * pr] = pack10_bits(o[0], o[2], o[4]);
        px[2 * pr + 1] = pack10_bits(o[1], o[3], o[5]);
      } else {
        f2 lin[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          float2 a, b;
          tab_pair<0>(ch[q], lut, a, b);
          f2 t;
          asm("v_fma_f32 %0, %1, %2, %3" : "=v"(t.x) : "v"(a.y), "v"(ch[q].x), "v"(a.x));
          asm("v_fma_f32 %0, %1, %2, %3" : "=v"(t.y) : "v"(b.y), "v"(ch[q].y), "v"(b.x));
          lin[q] = oetf2_scaled<FMT>(t * factor);
        }
        po[pr].r = lin[0]; po[pr].g = lin[1]; po[pr].b = lin[2];
      }
    }
    if (T::kOetf) {
      st_stream(reinterpret_cast<uint4*>(static_cast<uint32_t*>(im.dst) + pix0), make_uint4(px[0], px[1], px[2], px[3]));
    } else if (FMT == 2 || FMT == 3) {
      uint4 o;
      o.x = pack10_scaled<MASK>(po[0].r.x, po[0].g.x, po[0].b.x); o.y = pack10_scaled<MASK>(po[0].r.y, po[0].g.y, po[0].b.y);
      o.z = pack10_scaled<MASK>(po[1].r.x, po[1].g.x, po[1].b.x); o.w = pack10_scaled<MASK>(po[1].r.y, po[1].g.y, po[1].b.y);
      st_stream(reinterpret_cast<uint4*>(static_cast<uint32_t*>(im.dst) + pix0), o);
    } else if (FMT == 1) {
      const uint2 a = pack_f16_hw(po[0].r.x, po[0].g.x, po[0].b.x), bb = pack_f16_hw(po[0].r.y, po[0].g.y, po[0].b.y);
      const uint2 cc = pack_f16_hw(po[1].r.x, po[1].g.x, po[1].b.x), d = pack_f16_hw(po[1].r.y, po[1].g.y, po[1].b.y);
      uint4* o = reinterpret_cast<uint4*>(static_cast<uint2*>(im.dst) + pix0);
      o[0] = make_uint4(a.x, a.y, bb.x, bb.y);
      o[1] = make_uint4(cc.x, cc.y, d.x, d.y);
    } else {  // FMT == 4: planar R,G,B uint16 (ultrahdr.cpp:460-468)
      const size_t plane = (size_t)c.width * c.height;
      uint16_t* base16 = static_cast<uint16_t*>(im.dst);
      const f2 pl[3][2] = {{po[0].r, po[1].r}, {po[0].g, po[1].g}, {po[0].b, po[1].b}};
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const uint32_t q0 = 0x3ffu & (uint32_t)pl[p][0].x, q1 = 0x3ffu & (uint32_t)pl[p][0].y;
        const uint32_t q2 = 0x3ffu & (uint32_t)pl[p][1].x, q3 = 0x3ffu & (uint32_t)pl[p][1].y;
        st_stream(reinterpret_cast<uint2*>(base16 + p * plane + pix0), make_uint2(q0 | (q1 << 16), q2 | (q3 << 16)));
      }
    }
  }
}

// The same bytes as the walk of k_apply_s4 keeps them across a cell (the next cell's are in flight while the current one is
// computed, so every dword counts against the 128 registers of four waves per SIMD): the two taps of a map row arrive as ONE
// 16-bit load, mrow[j] = the bytes (mx, mx + 1) of map row cy (j = 0) / cy1 (j = 1).  The loads are unaligned half the time; global
// memory takes that.  Needs map_w >= 2 (launch_apply_t).
struct ApplyCellPk { uint32_t yrow[4], uu[2], vv[2], mrow[2]; };
struct __attribute__((packed)) U16Any { uint16_t v; };
// (mx < map_w - 1: the column of the byte pairs; cy1: the row of the lower taps)
__device__ __forceinline__ void apply_load_cell_pk(const AppConsts& c, const AppImage& im, uint32_t cx, uint32_t cy, uint32_t mx, uint32_t cy1, ApplyCellPk& o) {
  const uint32_t yoff = 4u * cy * im.y_stride + 4u * cx;
#pragma unroll
  for (int r = 0; r < 4; ++r) o.yrow[r] = ld_stream(reinterpret_cast<const uint32_t*>(im.y + (yoff + r * im.y_stride)));
  const uint32_t coff = 2u * cy * im.c_stride + 2u * cx;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    o.uu[r] = *reinterpret_cast<const uint16_t*>(im.u + (coff + r * im.c_stride));
    o.vv[r] = *reinterpret_cast<const uint16_t*>(im.v + (coff + r * im.c_stride));
  }
  o.mrow[0] = reinterpret_cast<const U16Any*>(im.map + (cy * c.map_w + mx))->v;
  o.mrow[1] = reinterpret_cast<const U16Any*>(im.map + (cy1 * c.map_w + mx))->v;
}

// Each block copies its tables into LDS once (4 KiB + 32 KiB for the replicated stage-2 table of HLG / PQ output, 15 KiB for the
// other outputs, 1 KiB of sampleMap weights: two blocks of 512 threads per CU) and then walks c.cells_per_thread map cells per
// thread.  Consecutive blocks belong to different images, as in generate (grid.x = image): the images of a launch progress
// together and the blocks in flight spread over the whole batch's memory (same-box A/B, round 2: 0.608 -> 0.576 ms per 64 frames).
//
// The walk is ONE straight line of code per cell: no memory instruction sits behind a branch (the request for the next cell is
// issued even when there is none -- it repeats the current cell's --, no lane is masked, and where a wave chooses between the two
// forms of the cell both issue the same memory instructions).  That is what lets s_waitcnt count: behind a branch the compiler
// has to assume an instruction was not issued, and a wait for the inputs of the next cell then also waits for every store issued
// after them -- round 2's loop drained its stores at every cell, reloaded six spilled registers, and waited for loads of the NEXT
// cell through registers the allocator had placed under undefined halves of operand pairs (apply_cell_piped).  Same box, 64 x 4K:
// 0.579 -> 0.536 ms.
//
// The cells of the last column and row have per-lane weights (the NR / NB / C tables of gainmapmath.h:184-228): a wave that
// touches one (one in fifteen on a 4K frame) runs apply_cell_edge for its 64 cells instead of the pipelined form.  Measured and
// dropped: handing those cells to blocks of their own behind a walk on SGPR weights only (a lane on the last column recomputing
// its left neighbour): 2 % slower on 64 x 4K (0.604 against 0.594 ms, same box), and in a one-round launch the handful of
// latency-bound edge blocks is what the launch waits for (one 4K frame: 13 -> 17-23 us).
constexpr uint32_t kApplyBlock = 512;
constexpr uint32_t kApplyMaxCellsPerThread = 32;   // (64 x 4K, same box: 32 cells per thread 0.594 ms, 64: 0.598, 128: 0.607; 48 / 80 / 96: 0.63-0.65)

// One cell of the walk: requests the inputs of the thread's next cell into `nxt` (a block's waves start together, and without this
// they would also all wait for HBM together and all compute together), computes the cell whose inputs `cur` holds, and steps
// (cx, cy) on.  The kernel calls it with its two register sets swapped from cell to cell, so nothing is copied between them.
// Returns whether there is a next cell; when there is none the request repeats the current cell's (issued all the same: see above).
template <int FMT, bool MASK>
__device__ __forceinline__ bool apply_walk_cell(const AppConsts& c, const AppImage& im, uint32_t& cx, uint32_t& cy, uint32_t& left,
                                                const ApplyCellPk& cur, ApplyCellPk& nxt, uint32_t slot8, const char* lut,
                                                const float* s_idw, uint4* s_xch) {
  typedef ApplyTab<FMT, MASK> T;
  // the block's next stretch of cells lies kApplyBlock = step_y * map_w + step_x cells further on
  uint32_t ncx = cx + c.step_x, ncy = cy + c.step_y;
  if (ncx >= c.map_w) { ncx -= c.map_w; ++ncy; }
  --left;
  const bool more = left != 0u && ncy < c.map_h;
  if (!more) { ncx = cx; ncy = cy; }
  // In the last column sampleMap's right tap IS the left one (gainmapmath.cpp:690-703, xu == xl): the two bytes are loaded one
  // column to the left, the right tap is byte 1 either way, the left tap byte 0 -- or byte 1 in the last column.
  apply_load_cell_pk(c, im, ncx, ncy, ncx - (ncx + 1u == c.map_w ? 1u : 0u), min(ncy + 1u, c.map_h - 1u), nxt);
  const bool edge_x = cx + 1u == c.map_w, edge_y = cy + 1u == c.map_h;
  const float e3 = cvt_byte<1>(cur.mrow[0]), e4 = cvt_byte<1>(cur.mrow[1]);
  const float e1 = edge_x ? e3 : cvt_byte<0>(cur.mrow[0]), e2 = edge_x ? e4 : cvt_byte<0>(cur.mrow[1]);
  // all waves but those touching the last column / row take the pipelined form on SGPR weights
  const bool interior = __builtin_amdgcn_ballot_w64(edge_x || edge_y) == 0ull;
  const float* wt = s_idw + (edge_x ? (edge_y ? 192 : 64) : (edge_y ? 128 : 0));   // tables: 0 std, 1 no-right, 2 no-bottom, 3 corner
  if (T::kOetf) {
    if (interior) apply_cell_piped<FMT>(c, im.dst, cx, cy, cur.yrow, cur.uu, cur.vv, e1, e2, e3, e4, slot8, lut);
    else apply_cell_edge<FMT, MASK>(c, im, cx, cy, cur.yrow, cur.uu, cur.vv, e1, e2, e3, e4, wt, slot8, lut);
  } else {
    // F16: a full wave on one row of cells stores through the exchange area (apply_cell_piped1)
    uint4* xch = nullptr;
    if (FMT == 1 && __builtin_amdgcn_ballot_w64(true) == ~0ull &&
        __builtin_amdgcn_ballot_w64(cy != (uint32_t)__builtin_amdgcn_readfirstlane((int)cy)) == 0ull)
      xch = s_xch + (threadIdx.x >> 6) * kXchPerWave;
    if (interior) apply_cell_piped1<FMT, MASK>(c, im, cx, cy, cur.yrow, cur.uu, cur.vv, e1, e2, e3, e4, lut, xch);
    else apply_cell_edge<FMT, MASK>(c, im, cx, cy, cur.yrow, cur.uu, cur.vv, e1, e2, e3, e4, wt, slot8, lut);
  }
  cx = ncx; cy = ncy;
  return more;
}

template <int FMT, bool MASK>
__global__ void __launch_bounds__(kApplyBlock, 4) k_apply_s4(const AppConsts c, const AppBatch b) {
  typedef ApplyTab<FMT, MASK> T;
  __shared__ uint4 s_tab[T::kBytes / 16u];
  __shared__ uint4 s_xch[FMT == 1 ? (kApplyBlock / 64) * kXchPerWave : 1];   // F16: the waves' exchange areas
  __shared__ float s_idw[4 * 64];   // sampleMap's four weight tables times A / 255: per-lane weights are an LDS read, not a trip to L2 per pixel pair
  const uint32_t img_i = blockIdx.x, span = blockIdx.y;
  const AppImage& im = b.img[img_i];
  const uint32_t idx = span * c.cells_per_thread * kApplyBlock + threadIdx.x;
  uint32_t cy = idx / c.map_w;
  uint32_t cx = idx - cy * c.map_w;
  const bool any = cy < c.map_h;
  // the first cell's pixels are requested before the tables: both trips to memory overlap
  ApplyCellPk ca, cb;
  if (any) apply_load_cell_pk(c, im, cx, cy, cx - (cx + 1u == c.map_w ? 1u : 0u), min(cy + 1u, c.map_h - 1u), ca);
  {
    // all loads first, then all stores: one trip through L2's latency per block instead of one per piece
    constexpr uint32_t kN1 = T::kS1Bytes / 16u, kPer1 = (kN1 + kApplyBlock - 1u) / kApplyBlock;
    constexpr uint32_t kN2 = T::kOetf ? kTabS2Cells * 32u : 0u, kPer2 = (kN2 + kApplyBlock - 1u) / kApplyBlock;
    const uint4* src1 = reinterpret_cast<const uint4*>(c.tab + T::kS1Float);
    const uint2* src2 = reinterpret_cast<const uint2*>(c.tab + T::kS2Float);
    uint4 t1[kPer1];
    uint2 t2[kPer2 ? kPer2 : 1u];
    const float w = c_idw4[threadIdx.x & 255u] * c.fast.A255;   // (before the rounding mode changes: launch_apply_t's product)
#pragma unroll
    for (uint32_t k = 0; k < kPer1; ++k) { const uint32_t i = k * kApplyBlock + threadIdx.x; t1[k] = src1[i < kN1 ? i : kN1 - 1u]; }
#pragma unroll
    for (uint32_t k = 0; k < kPer2; ++k) { const uint32_t i = k * kApplyBlock + threadIdx.x; t2[k] = src2[(i < kN2 ? i : 0u) >> 5]; }
    if (threadIdx.x < 256u) s_idw[threadIdx.x] = w;
#pragma unroll
    for (uint32_t k = 0; k < kPer1; ++k) { const uint32_t i = k * kApplyBlock + threadIdx.x; if (i < kN1) s_tab[i] = t1[k]; }
#pragma unroll
    for (uint32_t k = 0; k < kPer2; ++k) { const uint32_t i = k * kApplyBlock + threadIdx.x; if (i < kN2) reinterpret_cast<uint2*>(s_tab)[T::kS2Base / 8u + i] = t2[k]; }
  }
  __syncthreads();
  const char* lut = reinterpret_cast<const char*>(s_tab);
  const uint32_t slot8 = (threadIdx.x & 31u) << 3;
  if (!any) return;
  if (T::kOetf) {
    // Round toward zero from here on: stage 2 truncates the code inside its fma.  The integer division above is expanded into
    // float operations that assume round-to-nearest, hence the (fake) dependence of its results on this statement; the cells that
    // follow are reached by addition.
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3" : "+v"(cx), "+v"(cy));
  }
  uint32_t left = c.cells_per_thread;
#pragma unroll 1
  for (;;) {
    if (!apply_walk_cell<FMT, MASK>(c, im, cx, cy, left, ca, cb, slot8, lut, s_idw, s_xch)) return;
    if (!apply_walk_cell<FMT, MASK>(c, im, cx, cy, left, cb, ca, slot8, lut, s_idw, s_xch)) return;
  }
}

// General path: one thread per pixel; any integer scale, any pointer/stride alignment, FAST or
// EXACT arithmetic.  Mirrors ultrahdr.cpp:427-496 + gainmapmath.cpp:686-720 literally.
struct PxIn { float yf, u, v, gain; };
// the loads of ultrahdr.cpp:431-438 and sampleMap (gainmapmath.cpp:686-720) for pixel idx
__device__ __forceinline__ PxIn px_inputs(const AppConsts& c, const AppImage& im, uint32_t x, uint32_t y) {
  PxIn in;
  in.yf = (float)im.y[(size_t)y * im.y_stride + x] * k255;
  const size_t ci = (size_t)(y >> 1) * im.c_stride + (x >> 1);
  in.u = (float)((int)im.u[ci] - 128) * k255;
  in.v = (float)((int)im.v[ci] - 128) * k255;

  const uint32_t s = c.scale;
  uint32_t xl, yl, ox, oy;
  if ((s & (s - 1u)) == 0u) {   // (a launch-uniform branch: scale factors are powers of two in practice)
    const uint32_t sh = (uint32_t)__builtin_ctz(s);
    xl = x >> sh; yl = y >> sh; ox = x & (s - 1u); oy = y & (s - 1u);
  } else {
    xl = x / s; yl = y / s; ox = x - xl * s; oy = y - yl * s;
  }
  uint32_t xu = xl + 1u, yu = yl + 1u;
  xl = min(xl, c.map_w - 1u); xu = min(xu, c.map_w - 1u);
  yl = min(yl, c.map_h - 1u); yu = min(yu, c.map_h - 1u);
  // (byte / 255.0f through the constant division that is proven equal for all 256 bytes, tests/test_gpu_transfer_exhaustive.py)
  const float e1 = map_to_float_fast(im.map[(size_t)yl * c.map_w + xl]);
  const float e2 = map_to_float_fast(im.map[(size_t)yu * c.map_w + xl]);
  const float e3 = map_to_float_fast(im.map[(size_t)yl * c.map_w + xu]);
  const float e4 = map_to_float_fast(im.map[(size_t)yu * c.map_w + xu]);
  int tbl = 0;
  if (xl == xu && yl == yu) tbl = 3;
  else if (xl == xu) tbl = 1;
  else if (yl == yu) tbl = 2;
  const float* w = c.idw + (size_t)tbl * s * s * 4u + (size_t)oy * s * 4u + ox * 4u;
  in.gain = e1 * w[0] + e2 * w[1] + e3 * w[2] + e4 * w[3];
  return in;
}
// the per-pixel kernels run on a (column block, row, image) grid: no division by the width; rows beyond the grid's 65535 are
// reached by striding
__host__ __device__ inline dim3 px_grid(uint32_t width, uint32_t height, int n) {
  return dim3((width + 255u) / 256u, height < 65535u ? height : 65535u, (unsigned)n);
}
template <int FMT>
__device__ __forceinline__ void px_store(const AppImage& im, size_t idx, size_t total, F3 e) {
  if (FMT == 2 || FMT == 3) {
    static_cast<uint32_t*>(im.dst)[idx] = pack_1010102(e.x, e.y, e.z);
  } else if (FMT == 1) {
    static_cast<uint2*>(im.dst)[idx] = pack_f16(e.x, e.y, e.z);
  } else {
    uint16_t* base = static_cast<uint16_t*>(im.dst);
    base[idx] = (uint16_t)(0x3ffu & (uint32_t)(e.x * 1023.0f));
    base[total + idx] = (uint16_t)(0x3ffu & (uint32_t)(e.y * 1023.0f));
    base[2 * total + idx] = (uint16_t)(0x3ffu & (uint32_t)(e.z * 1023.0f));
  }
}

template <int FMT, bool EXACT>
__global__ void __launch_bounds__(256) k_apply_px(const AppConsts c, const AppBatch b) {
  const AppImage& im = b.img[blockIdx.z];
  const size_t total = (size_t)c.width * c.height;
  const uint32_t x = blockIdx.x * 256u + threadIdx.x;
  if (x >= c.width) return;
  for (uint32_t y = blockIdx.y; y < c.height; y += gridDim.y) {
    const PxIn in = px_inputs(c, im, x, y);
    const F3 lin = recover_hdr<EXACT>(c, in.yf, kP3Cr * in.v, kP3GCb * in.u, kP3GCr * in.v, kP3Cb * in.u, in.gain);
    px_store<FMT>(im, (size_t)y * c.width + x, total, hdr_oetf<FMT, EXACT>(lin));
  }
}

// ---- EXACT mode behind an f32 pre-filter (HLG, F16 and planar 10-bit outputs) ----------------------------------------------------
// The bit-exact pixel costs ~40 f64 operations per channel; what it decides in the end is an integer code (or a half-precision
// pattern).  k_apply_px_est evaluates the pixel on the f32 units -- the front end up to the exponent's argument exactly as the
// reference rounds it, then v_exp_f32 / v_log_f32 forms whose distance from the exact functions is measured for every float
// (tests/test_gpu_exact_filter.py) -- and writes that result.  A channel is IN DOUBT when the interval the exact value must lie in
// straddles a code boundary; pixels with such a channel (~1 %) go to per-image lists and k_apply_resolve recomputes exactly them
// with the exact path and overwrites.  Error budget, in the reference's own quantities:
//   lin = (srgbInvOetf(c) * factor) / displayBoost: relative error of the estimate <= kEstRel (srgb_inv_oetf_fast 6e-7, v_exp_f32
//     2e-7, the two roundings are the reference's own), so the 10-bit planar code floor(lin * 1023) and the half-precision pattern are
//     settled when both ends of lin (1 +- kEstRel) give the same;
//   hlgOetf: x h'(x) <= 1/4, so the input's error moves the code value by <= 1023 / 4 * kEstRel = 4e-4; hlg_oetf_fast itself is
//     within 3e-7 (3e-4 codes) of the exact function up to 1 and within 3e-7 relative above: doubt when
//     floor(v - d) != floor(v + d), d = kEstOetfAbs + v * kEstOetfRel;
//   pqOetf: v S(x) <= 112 codes per unit of relative input error (2.2e-4), pq_oetf_est is within 2.5e-4 codes: the same d.
constexpr float kEstRel = 2.0e-6f;
constexpr float kEstOetfAbs = 8.0e-4f, kEstOetfRel = 6.0e-7f;
constexpr uint32_t kExSlices = 256;   // blocks of k_apply_resolve per image

__device__ __forceinline__ F3 recover_hdr_est(const AppConsts& c, float yf, float crv, float gcbu, float gcrv, float cbu, float gain) {
  const float r = clamp01(yf + crv), g = clamp01(yf - gcbu - gcrv), b = clamp01(yf + cbu);
  const float log_boost = (float)(c.log2_min_d * (double)(1.0f - gain) + c.log2_max_d * (double)gain);
  const float factor = __builtin_amdgcn_exp2f(log_boost * c.display_boost / c.max_boost);   // the argument as the reference rounds it
  F3 o;
  o.x = (srgb_inv_oetf_fast(r) * factor) / c.display_boost;
  o.y = (srgb_inv_oetf_fast(g) * factor) / c.display_boost;
  o.z = (srgb_inv_oetf_fast(b) * factor) / c.display_boost;
  return o;
}
// one channel: lin -> the value whose integer part is the code (the half-precision source for F16), and the doubt test (below)
template <int FMT>
__device__ __forceinline__ void est_channel(float lin, float& out, float& doubt_min) {
  if (FMT != 1) {
    const float v = (FMT == 3 ? hlg_oetf_fast(lin) : FMT == 2 ? pq_oetf_est(lin) : lin) * 1023.0f;
    const float d = FMT == 4 ? v * kEstRel : __builtin_fmaf(v, kEstOetfRel, kEstOetfAbs);
    const float f = __builtin_amdgcn_fractf(fmaxf(v, 0.5f));
    doubt_min = fminf(doubt_min, fminf(f, 1.0f - f) - d);   // < 0: in doubt
    out = v;
  } else {
    const _Float16 lo = (_Float16)(lin * (1.0f - kEstRel)), hi = (_Float16)(lin * (1.0f + kEstRel));
    // (beyond the largest half the reference's floatToHalf does not saturate to infinity like the conversion above: it leaves
    // 0x7C00 | mantissa bits, 0x7FFF from 2^17 on -- gainmapmath.cpp:745-780; reachable with minContentBoost > maxContentBoost)
    if (lo != hi || (lin != 0.0f && lin < 0x1p-13f) || !(lin * (1.0f + kEstRel) < 65504.0f)) doubt_min = -1.0f;
    out = lin;
  }
}

template <int FMT>
__global__ void __launch_bounds__(256) k_apply_px_est(const AppConsts c, const AppBatch b) {
  const AppImage& im = b.img[blockIdx.z];
  const size_t total = (size_t)c.width * c.height;
  const uint32_t x = blockIdx.x * 256u + threadIdx.x;
  uint32_t* hdr = c.ex_ws + (size_t)blockIdx.z * kExHdrWords;
  __shared__ uint32_t s_cnt[4], s_base;
  for (uint32_t y = blockIdx.y; y < c.height; y += gridDim.y) {   // (block-uniform trip count: the barriers below are safe)
    const size_t idx = (size_t)y * c.width + x;
    bool doubt = false;
    if (x < c.width) {
      const PxIn in = px_inputs(c, im, x, y);
      const F3 lin = recover_hdr_est(c, in.yf, kP3Cr * in.v, kP3GCb * in.u, kP3GCr * in.v, kP3Cb * in.u, in.gain);
      float o[3], dm = 1.0f;
      est_channel<FMT>(lin.x, o[0], dm);
      est_channel<FMT>(lin.y, o[1], dm);
      est_channel<FMT>(lin.z, o[2], dm);
      doubt = dm < 0.0f;
      if (FMT == 2 || FMT == 3) {
        static_cast<uint32_t*>(im.dst)[idx] = (0x3ffu & (uint32_t)o[0]) | ((0x3ffu & (uint32_t)o[1]) << 10) | ((0x3ffu & (uint32_t)o[2]) << 20) | (0x3u << 30);
      } else if (FMT == 1) {
        static_cast<uint2*>(im.dst)[idx] = pack_f16_hw(o[0], o[1], o[2]);
      } else {
        uint16_t* base = static_cast<uint16_t*>(im.dst);
        base[idx] = (uint16_t)(0x3ffu & (uint32_t)o[0]);
        base[total + idx] = (uint16_t)(0x3ffu & (uint32_t)o[1]);
        base[2 * total + idx] = (uint16_t)(0x3ffu & (uint32_t)o[2]);
      }
    }
    // one append per block and row: the list is chosen by the block so that neighbouring blocks do not share a counter
    const uint64_t mask = __ballot(doubt);
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    __syncthreads();   // (the previous row's s_cnt / s_base have been read)
    if (lane == 0) s_cnt[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    const uint32_t list = (y * gridDim.x + blockIdx.x) % kExLists;
    if (threadIdx.x == 0) {
      const uint32_t t = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
      s_base = t ? atomicAdd(hdr + 8u + list, t) : 0u;
    }
    __syncthreads();
    if (doubt) {
      uint32_t pos = s_base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
      for (uint32_t w = 0; w < wave; ++w) pos += s_cnt[w];
      if (pos < c.ex_cap)
        c.ex_ws[(size_t)kMaxChunk * kExHdrWords + ((size_t)blockIdx.z * kExLists + list) * c.ex_cap + pos] = (uint32_t)idx;
    }
  }
}

// The same estimate for the common geometry (scale 4, aligned planes: app_fast_s4): thread = one map cell = 4x4 pixels, as in
// k_apply_s4 -- the taps, the chroma terms and the addresses are formed once per cell, rows leave as 16-byte stores, and every
// operation of the front end is the reference's own (same order, no contraction) up to the exponent's argument.  Doubt, per channel,
// with v the value whose integer part is the code: the distance of v from the nearest integer against d (v < 1 can only be code 0
// from below, hence the max).  F16: the hardware conversion rounds ties to even where the reference rounds them up; a value the
// interval of which holds a tie is in doubt anyway, and so is the half-precision subnormal range.
template <int FMT, bool INTERIOR>
__device__ __forceinline__ uint32_t est_cell(const AppConsts& c, const AppImage& im, uint32_t cx, uint32_t cy, const ApplyCellIn& in, int tbl) {
  const float e1 = map_to_float_fast(in.mb[0]), e2 = map_to_float_fast(in.mb[1]);   // == byte / 255.0f for every byte
  const float e3 = map_to_float_fast(in.mb[2]), e4 = map_to_float_fast(in.mb[3]);
  float crv[2][2], gcbu[2][2], gcrv[2][2], cbu[2][2];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float u = (float)((int)((in.uu[r] >> (8 * k)) & 0xffu) - 128) * k255;
      const float v = (float)((int)((in.vv[r] >> (8 * k)) & 0xffu) - 128) * k255;
      crv[r][k] = kP3Cr * v; gcbu[r][k] = kP3GCb * u; gcrv[r][k] = kP3GCr * v; cbu[r][k] = kP3Cb * u;
    }
  const float* wt = c_idw4 + (INTERIOR ? 0 : tbl * 64);
  const bool min_is_one = c.log2_min_d == 0.0;   // log2(min) (1 - g) is then +-0 and leaves the sum as it is
  uint32_t doubt = 0u;
#pragma unroll
  for (int oy = 0; oy < 4; ++oy) {
    float o[4][3];
#pragma unroll
    for (int ox = 0; ox < 4; ++ox) {
      const float* w = wt + oy * 16 + ox * 4;
      const float gain = e1 * w[0] + e2 * w[1] + e3 * w[2] + e4 * w[3];
      const float log_boost = min_is_one ? (float)(c.log2_max_d * (double)gain)
                                         : (float)(c.log2_min_d * (double)(1.0f - gain) + c.log2_max_d * (double)gain);
      const float factor = __builtin_amdgcn_exp2f(log_boost * c.display_boost / c.max_boost) * c.inv_display_boost;
      const float yf = (float)((in.yrow[oy] >> (8 * ox)) & 0xffu) * k255;
      const int r2 = oy >> 1, k2 = ox >> 1;
      // clamp01 as one v_med3_f32 (the same value for every finite input but -0, which ends as code 0 either way)
      const float r = __builtin_amdgcn_fmed3f(yf + crv[r2][k2], 0.0f, 1.0f), g = __builtin_amdgcn_fmed3f(yf - gcbu[r2][k2] - gcrv[r2][k2], 0.0f, 1.0f);
      const float b = __builtin_amdgcn_fmed3f(yf + cbu[r2][k2], 0.0f, 1.0f);
      float dm = 1.0f;
      est_channel<FMT>(srgb_inv_oetf_fast(r) * factor, o[ox][0], dm);
      est_channel<FMT>(srgb_inv_oetf_fast(g) * factor, o[ox][1], dm);
      est_channel<FMT>(srgb_inv_oetf_fast(b) * factor, o[ox][2], dm);
      if (dm < 0.0f) doubt |= 1u << (oy * 4 + ox);
    }
    const uint32_t pix0 = (4u * cy + oy) * c.width + 4u * cx;
    if (FMT == 2 || FMT == 3) {
      uint4 q;
      uint32_t* qq = &q.x;
#pragma unroll
      for (int ox = 0; ox < 4; ++ox)
        qq[ox] = (0x3ffu & (uint32_t)o[ox][0]) | ((0x3ffu & (uint32_t)o[ox][1]) << 10) | ((0x3ffu & (uint32_t)o[ox][2]) << 20) | (0x3u << 30);
      st_stream(reinterpret_cast<uint4*>(static_cast<uint32_t*>(im.dst) + pix0), q);
    } else if (FMT == 1) {
      const uint2 a = pack_f16_hw(o[0][0], o[0][1], o[0][2]), bb = pack_f16_hw(o[1][0], o[1][1], o[1][2]);
      const uint2 cc = pack_f16_hw(o[2][0], o[2][1], o[2][2]), d = pack_f16_hw(o[3][0], o[3][1], o[3][2]);
      // (plain stores: a lane's 32 bytes leave as two instructions, each filling half of every line it touches -- L2 merges them,
      // a non-temporal store would send the halves on their own)
      uint4* dst = reinterpret_cast<uint4*>(static_cast<uint2*>(im.dst) + pix0);
      dst[0] = make_uint4(a.x, a.y, bb.x, bb.y);
      dst[1] = make_uint4(cc.x, cc.y, d.x, d.y);
    } else {
      const size_t plane = (size_t)c.width * c.height;
      uint16_t* base16 = static_cast<uint16_t*>(im.dst);
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const uint32_t q0 = 0x3ffu & (uint32_t)o[0][p], q1 = 0x3ffu & (uint32_t)o[1][p];
        const uint32_t q2 = 0x3ffu & (uint32_t)o[2][p], q3 = 0x3ffu & (uint32_t)o[3][p];
        *reinterpret_cast<uint2*>(base16 + p * plane + pix0) = make_uint2(q0 | (q1 << 16), q2 | (q3 << 16));
      }
    }
  }
  return doubt;
}

// ---- the same estimate, two horizontally adjacent pixels per instruction (interior waves of k_apply_s4_est) --------------------------
// The scalar cell above spends ~115 VALU instructions per pixel; v_pk_mul / v_pk_add / v_pk_fma_f32 do two floats per issue slot.
// Every elementary function below performs, per element, the operation sequence of its scalar twin (srgb_inv_oetf_fast,
// hlg_oetf_fast, pq_oetf_est: the error bounds tests/test_gpu_exact_filter.py measures for those hold unchanged); what differs:
//   * the exponent's argument.  The reference rounds log2(max) * gain (a double product) to float, multiplies by the display boost
//     and divides by the content boost.  Here, for |argument| <= 6 and minContentBoost == 1: the double product as a two-float
//     product (exact but for a 2^-48 relative slip that moves the rounded result by one ulp once in 2^24 pixels; one ulp of an
//     argument <= 6 moves the factor by <= 5e-7 relative: part of the budget the test checks) and the division as
//     q + fma(-q, a, x) / a (div_const); other launches keep the scalar cell;
//   * the doubt test: fract(max(v, 1/2) + d) < 2 d instead of min(f, 1 - f) < d -- the same interval but for one rounding at the
//     magnitude of the code value, which kEstPkAbs carries.
constexpr float kEstPkAbs = 6.2e-5f;   // one ulp at 1023
__device__ __forceinline__ f2 srgb_inv_oetf_fast2(f2 e) {
  const f2 lin = e * splat(1.0f / 12.92f);
  const f2 x = (e + splat(0.055f)) * splat(1.0f / 1.055f);
  const f2 p = (x * x) * exp2_2(splat(0.4f) * log2_2(x));
  return (f2){e.x <= 0.04045f ? lin.x : p.x, e.y <= 0.04045f ? lin.y : p.y};
}
__device__ __forceinline__ f2 hlg_oetf_fast2(f2 e) {
  const f2 lo = sqrt_2(splat(3.0f) * e);
  const f2 hi = splat(UHDR_HLG_A * 0.693147180559945f) * log2_2(splat(12.0f) * e - splat(UHDR_HLG_B)) + splat(UHDR_HLG_C);
  return (f2){e.x <= 1.0f / 12.0f ? lo.x : hi.x, e.y <= 1.0f / 12.0f ? lo.y : hi.y};
}
__device__ __forceinline__ f2 pq_oetf_est2(f2 e) {
  const f2 p = exp2_2(splat(UHDR_PQ_M1) * log2_2(e));
  const f2 w = (splat(1.0f - UHDR_PQ_C1) * (splat(1.0f) - p)) * rcp_2(pk_fma(splat(UHDR_PQ_C3), p, splat(1.0f)));
  const f2 s = w * rcp_2(splat(2.0f) - w);
  const f2 s2 = s * s;
  f2 h = pk_fma(s2, splat(1.0f / 7.0f), splat(1.0f / 5.0f));
  h = pk_fma(s2, h, splat(1.0f / 3.0f));
  h = pk_fma(s2 * s, h, s);
  const f2 r = exp2_2(h * splat(-2.0f * 1.4426950408889634f * UHDR_PQ_M2));
  return (f2){e.x <= 0.0f ? 0.0f : r.x, e.y <= 0.0f ? 0.0f : r.y};
}
// one channel of a pixel pair: lin -> code values; dx / dy: that pixel has a channel in doubt (conditions, so that the three
// channels of a pixel combine on the scalar unit)
template <int FMT>
__device__ __forceinline__ void est_channel2(f2 lin, f2& out, bool& dx, bool& dy) {
  if (FMT != 1) {
    const f2 v = (FMT == 3 ? hlg_oetf_fast2(lin) : FMT == 2 ? pq_oetf_est2(lin) : lin) * splat(1023.0f);
    const f2 d = FMT == 4 ? pk_fma(v, splat(kEstRel), splat(kEstPkAbs)) : pk_fma(v, splat(kEstOetfRel), splat(kEstOetfAbs + kEstPkAbs));
    const f2 w = (f2){fmaxf(v.x, 0.5f), fmaxf(v.y, 0.5f)} + d;
    const f2 d2 = d + d;
    out = v;
    dx = dx || __builtin_amdgcn_fractf(w.x) < d2.x;
    dy = dy || __builtin_amdgcn_fractf(w.y) < d2.y;
  } else {
    float o0, o1, m0 = 1.0f, m1 = 1.0f;
    est_channel<1>(lin.x, o0, m0);
    est_channel<1>(lin.y, o1, m1);
    out = (f2){o0, o1};
    dx = dx || m0 < 0.0f;
    dy = dy || m1 < 0.0f;
  }
}
template <int FMT>
__device__ __forceinline__ uint32_t est_cell_pk(const AppConsts& c, const AppImage& im, uint32_t cx, uint32_t cy, const ApplyCellIn& in) {
  const f2 e1 = splat(map_to_float_fast(in.mb[0])), e2 = splat(map_to_float_fast(in.mb[1]));
  const f2 e3 = splat(map_to_float_fast(in.mb[2])), e4 = splat(map_to_float_fast(in.mb[3]));
  float crv[2][2], gcbu[2][2], gcrv[2][2], cbu[2][2];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float u = (float)((int)((in.uu[r] >> (8 * k)) & 0xffu) - 128) * k255;
      const float v = (float)((int)((in.vv[r] >> (8 * k)) & 0xffu) - 128) * k255;
      crv[r][k] = kP3Cr * v; gcbu[r][k] = kP3GCb * u; gcrv[r][k] = kP3GCr * v; cbu[r][k] = kP3Cb * u;
    }
  // log2(max) as two floats (the caller has checked |log2 max| display / max <= 6 and min == 1)
  const float lh = (float)c.log2_max_d, ll = (float)(c.log2_max_d - (double)lh);
  uint32_t doubt = 0u;
#pragma unroll
  for (int oy = 0; oy < 4; ++oy) {
    f2 o[2][3];
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      const float* w = c_idw4p + (oy * 2 + pr) * 8;
      f2 gain = e1 * (f2){w[0], w[1]};
      gain = gain + e2 * (f2){w[2], w[3]};
      gain = gain + e3 * (f2){w[4], w[5]};
      gain = gain + e4 * (f2){w[6], w[7]};
      // (float)(log2(max) * (double)gain), then * display / max as the reference rounds them
      const f2 p = gain * splat(lh);
      const f2 log_boost = p + pk_fma(gain, splat(ll), pk_fma(gain, splat(lh), -p));
      const f2 x = log_boost * splat(c.display_boost);
      const f2 q = x * splat(c.inv_max_boost);
      const f2 arg = pk_fma(pk_fma(-q, splat(c.max_boost), x), splat(c.inv_max_boost), q);
      const f2 factor = exp2_2(arg) * splat(c.inv_display_boost);
      const f2 yf = (pr ? (f2){cvt_byte<2>(in.yrow[oy]), cvt_byte<3>(in.yrow[oy])} : (f2){cvt_byte<0>(in.yrow[oy]), cvt_byte<1>(in.yrow[oy])}) * splat(k255);
      const int r2 = oy >> 1;
      const f2 r = pk_add_sat(yf, splat(crv[r2][pr])), g = pk_add_sat(yf - splat(gcbu[r2][pr]), splat(-gcrv[r2][pr]));
      const f2 b = pk_add_sat(yf, splat(cbu[r2][pr]));
      bool dx = false, dy = false;
      est_channel2<FMT>(srgb_inv_oetf_fast2(r) * factor, o[pr][0], dx, dy);
      est_channel2<FMT>(srgb_inv_oetf_fast2(g) * factor, o[pr][1], dx, dy);
      est_channel2<FMT>(srgb_inv_oetf_fast2(b) * factor, o[pr][2], dx, dy);
      doubt |= (dx ? 1u << (oy * 4 + 2 * pr) : 0u) | (dy ? 2u << (oy * 4 + 2 * pr) : 0u);
    }
    const uint32_t pix0 = (4u * cy + oy) * c.width + 4u * cx;
    if (FMT == 2 || FMT == 3) {
      uint4 qv;
      uint32_t* qq = &qv.x;
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        qq[2 * pr] = (0x3ffu & (uint32_t)o[pr][0].x) | ((0x3ffu & (uint32_t)o[pr][1].x) << 10) | ((0x3ffu & (uint32_t)o[pr][2].x) << 20) | (0x3u << 30);
        qq[2 * pr + 1] = (0x3ffu & (uint32_t)o[pr][0].y) | ((0x3ffu & (uint32_t)o[pr][1].y) << 10) | ((0x3ffu & (uint32_t)o[pr][2].y) << 20) | (0x3u << 30);
      }
      st_stream(reinterpret_cast<uint4*>(static_cast<uint32_t*>(im.dst) + pix0), qv);
    } else if (FMT == 1) {
      const uint2 a = pack_f16_hw(o[0][0].x, o[0][1].x, o[0][2].x), bb = pack_f16_hw(o[0][0].y, o[0][1].y, o[0][2].y);
      const uint2 cc = pack_f16_hw(o[1][0].x, o[1][1].x, o[1][2].x), d = pack_f16_hw(o[1][0].y, o[1][1].y, o[1][2].y);
      uint4* dst = reinterpret_cast<uint4*>(static_cast<uint2*>(im.dst) + pix0);
      dst[0] = make_uint4(a.x, a.y, bb.x, bb.y);
      dst[1] = make_uint4(cc.x, cc.y, d.x, d.y);
    } else {
      const size_t plane = (size_t)c.width * c.height;
      uint16_t* base16 = static_cast<uint16_t*>(im.dst);
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const uint32_t q0 = 0x3ffu & (uint32_t)o[0][p].x, q1 = 0x3ffu & (uint32_t)o[0][p].y;
        const uint32_t q2 = 0x3ffu & (uint32_t)o[1][p].x, q3 = 0x3ffu & (uint32_t)o[1][p].y;
        *reinterpret_cast<uint2*>(base16 + p * plane + pix0) = make_uint2(q0 | (q1 << 16), q2 | (q3 << 16));
      }
    }
  }
  return doubt;
}

template <int FMT>
__global__ void __launch_bounds__(256) k_apply_s4_est(const AppConsts c, const AppBatch b) {
  const AppImage& im = b.img[blockIdx.y];
  const uint32_t cells = c.map_w * c.map_h;
  const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
  uint32_t doubt = 0u, cx = 0u, cy = 0u;
  if (idx < cells) {
    cy = idx / c.map_w;
    cx = idx - cy * c.map_w;
    ApplyCellIn in;
    apply_load_cell(c, im, cx, cy, in);
    const bool edge_x = cx + 1u == c.map_w, edge_y = cy + 1u == c.map_h;
    const int tbl = edge_x ? (edge_y ? 3 : 1) : (edge_y ? 2 : 0);
    // (launch-uniform: the packed cell's exponent is budgeted for arguments up to 6 and takes minContentBoost == 1)
    const bool pk = c.log2_min_d == 0.0 && __builtin_fabs(c.log2_max_d) * (double)(c.display_boost * c.inv_max_boost) <= 6.0;
    if (__builtin_amdgcn_ballot_w64(tbl != 0) != 0ull) doubt = est_cell<FMT, false>(c, im, cx, cy, in, tbl);
    else if (pk) doubt = est_cell_pk<FMT>(c, im, cx, cy, in);
    else doubt = est_cell<FMT, true>(c, im, cx, cy, in, 0);
  }
  // one append per block
  __shared__ uint32_t s_part[4], s_base;
  const uint32_t mine = (uint32_t)__popc(doubt);
  const uint32_t before = block_exclusive_sum<256>(mine, s_part);
  const uint32_t list = blockIdx.x % kExLists;
  uint32_t* hdr = c.ex_ws + (size_t)blockIdx.y * kExHdrWords;
  if (threadIdx.x == 255) s_base = (before + mine) ? atomicAdd(hdr + 8u + list, before + mine) : 0u;
  __syncthreads();
  uint32_t pos = s_base + before;
  uint32_t* entries = c.ex_ws + (size_t)kMaxChunk * kExHdrWords + ((size_t)blockIdx.y * kExLists + list) * c.ex_cap;
  while (doubt) {
    const uint32_t k = (uint32_t)__builtin_ctz(doubt);
    doubt &= doubt - 1u;
    if (pos < c.ex_cap) entries[pos] = (4u * cy + (k >> 2)) * c.width + 4u * cx + (k & 3u);
    ++pos;
  }
}

// grid (images, kExSlices).  Every block forms the prefix sums of the image's list lengths, takes its share of the entries and
// recomputes them with the exact path; a list that overflowed turns the image into a sweep of all pixels.  The block that
// finishes last clears the header for the next launch.
template <int FMT>
__global__ void __launch_bounds__(256) k_apply_resolve(const AppConsts c, const AppBatch b) {
  const AppImage& im = b.img[blockIdx.x];
  uint32_t* hdr = c.ex_ws + (size_t)blockIdx.x * kExHdrWords;
  const uint32_t* entries = c.ex_ws + (size_t)kMaxChunk * kExHdrWords + (size_t)blockIdx.x * kExLists * c.ex_cap;
  const size_t total = (size_t)c.width * c.height;
  static_assert(kExLists == 1024, "four lists per thread");
  __shared__ uint32_t s_first[kExLists + 1];
  __shared__ uint32_t s_over;
  if (threadIdx.x == 0) s_over = 0u;
  __syncthreads();
  const uint4 n4 = *reinterpret_cast<const uint4*>(hdr + 8u + 4u * threadIdx.x);
  if (max(max(n4.x, n4.y), max(n4.z, n4.w)) > c.ex_cap) s_over = 1u;
  __shared__ uint32_t s_part[4];
  const uint32_t mine = n4.x + n4.y + n4.z + n4.w;
  const uint32_t before = block_exclusive_sum<256>(mine, s_part);
  s_first[4u * threadIdx.x] = before;
  s_first[4u * threadIdx.x + 1u] = before + n4.x;
  s_first[4u * threadIdx.x + 2u] = before + n4.x + n4.y;
  s_first[4u * threadIdx.x + 3u] = before + n4.x + n4.y + n4.z;
  if (threadIdx.x == 255) s_first[kExLists] = before + mine;
  __syncthreads();
  const bool sweep = s_over != 0u;
  const size_t count = sweep ? total : (size_t)s_first[kExLists];
  for (size_t g = (size_t)blockIdx.y * 256u + threadIdx.x; g < count; g += (size_t)kExSlices * 256u) {
    size_t idx = g;
    if (!sweep) {
      uint32_t lo = 0, hi = kExLists;   // the list with s_first[list] <= g < s_first[list + 1]
      while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (s_first[mid] <= (uint32_t)g) lo = mid; else hi = mid; }
      idx = entries[(size_t)lo * c.ex_cap + ((uint32_t)g - s_first[lo])];
    }
    const uint32_t py = (uint32_t)(idx / c.width), px = (uint32_t)(idx - (size_t)py * c.width);
    const PxIn in = px_inputs(c, im, px, py);
    const F3 lin = recover_hdr<true>(c, in.yf, kP3Cr * in.v, kP3GCb * in.u, kP3GCr * in.v, kP3Cb * in.u, in.gain);
    px_store<FMT>(im, idx, total, hdr_oetf<FMT, true>(lin));
  }
  // the header is cleared by the slice that finishes last (every slice has read the counts by then)
  __shared__ uint32_t s_last;
  __syncthreads();
  if (threadIdx.x == 0) s_last = atomicAdd(hdr + 3u, 1u) == kExSlices - 1u;
  __syncthreads();
  if (s_last) {
    reinterpret_cast<uint4*>(hdr + 8u)[threadIdx.x] = make_uint4(0, 0, 0, 0);
    if (threadIdx.x == 0) hdr[3] = 0u;
  }
}

// LUT mode (ultrahdr.cpp:433,446,470,481 with USE_*_LUT = 1): srgbInvOetfLUT, applyGainLUT, hlg/pqOetfLUT.
// One thread per pixel, any integer scale.  Each block first builds its private copies of the sRGB table and of
// GainLUT(metadata, display_boost) in LDS (the latter costs 4 double exp2 per thread, hence kLutPixelsPerThread
// pixels per thread); the 65536-entry OETF tables (256 KiB each) are gathered from L2.
constexpr uint32_t kLutPixelsPerThread = 16;
template <int FMT>
__global__ void __launch_bounds__(256) k_apply_lut(const AppConsts c, const AppBatch b) {
  __shared__ float s_srgb[kLutSrgbInvN];
  __shared__ float s_gain[kGainLutN];
  for (uint32_t i = threadIdx.x; i < kLutSrgbInvN; i += 256u) {
    s_srgb[i] = c.lut[kLutSrgbInv + i];
    s_gain[i] = gain_lut_entry(i, c.log2_min_d, c.log2_max_d, c.lut_boost_factor);
  }
  __syncthreads();
  const float* oetf = c.lut + (FMT == 3 ? kLutHlg : kLutPq);
  const AppImage& im = b.img[blockIdx.y];
  const size_t total = (size_t)c.width * c.height;
  const uint32_t s = c.scale;
  for (size_t idx = (size_t)blockIdx.x * 256u + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256u) {
    const uint32_t y = (uint32_t)(idx / c.width);
    const uint32_t x = (uint32_t)(idx - (size_t)y * c.width);
    const float yf = (float)im.y[(size_t)y * im.y_stride + x] * k255;
    const size_t ci = (size_t)(y >> 1) * im.c_stride + (x >> 1);
    const float u = (float)((int)im.u[ci] - 128) * k255;
    const float v = (float)((int)im.v[ci] - 128) * k255;
    // sampleMap (gainmapmath.cpp:686-720), as in k_apply_px
    uint32_t xl = x / s, yl = y / s;
    uint32_t xu = xl + 1u, yu = yl + 1u;
    xl = min(xl, c.map_w - 1u); xu = min(xu, c.map_w - 1u);
    yl = min(yl, c.map_h - 1u); yu = min(yu, c.map_h - 1u);
    const float e1 = map_to_float(im.map[(size_t)yl * c.map_w + xl]);
    const float e2 = map_to_float(im.map[(size_t)yu * c.map_w + xl]);
    const float e3 = map_to_float(im.map[(size_t)yl * c.map_w + xu]);
    const float e4 = map_to_float(im.map[(size_t)yu * c.map_w + xu]);
    const uint32_t ox = x % s, oy = y % s;
    int tbl = 0;
    if (xl == xu && yl == yu) tbl = 3;
    else if (xl == xu) tbl = 1;
    else if (yl == yu) tbl = 2;
    const float* w = c.idw + (size_t)tbl * s * s * 4u + (size_t)oy * s * 4u + ox * 4u;
    const float gain = e1 * w[0] + e2 * w[1] + e3 * w[2] + e4 * w[3];

    const float r = s_srgb[lut_index_unit(clamp01(yf + kP3Cr * v), kLutSrgbInvN)];
    const float g = s_srgb[lut_index_unit(clamp01(yf - kP3GCb * u - kP3GCr * v), kLutSrgbInvN)];
    const float bl = s_srgb[lut_index_unit(clamp01(yf + kP3Cb * u), kLutSrgbInvN)];
    const float factor = s_gain[lut_index(gain, kGainLutN)];       // GainLUT::getGainFactor, gainmapmath.h:173-178
    F3 e;
    e.x = (r * factor) / c.display_boost;                          // applyGainLUT, then ultrahdr.cpp:451
    e.y = (g * factor) / c.display_boost;
    e.z = (bl * factor) / c.display_boost;
    if (FMT == 2 || FMT == 3) {
      e.x = oetf[lut_index(e.x, kLutHlgN)]; e.y = oetf[lut_index(e.y, kLutHlgN)]; e.z = oetf[lut_index(e.z, kLutHlgN)];
      static_cast<uint32_t*>(im.dst)[idx] = pack_1010102(e.x, e.y, e.z);
    } else if (FMT == 1) {
      static_cast<uint2*>(im.dst)[idx] = pack_f16(e.x, e.y, e.z);
    } else {
      uint16_t* base = static_cast<uint16_t*>(im.dst);
      base[idx] = (uint16_t)(0x3ffu & (uint32_t)(e.x * 1023.0f));
      base[total + idx] = (uint16_t)(0x3ffu & (uint32_t)(e.y * 1023.0f));
      base[2 * total + idx] = (uint16_t)(0x3ffu & (uint32_t)(e.z * 1023.0f));
    }
  }
}

// LUT mode for the common geometry (scale 4, aligned planes: app_fast_s4).  Thread = one map cell; 1024-thread blocks, one per CU,
// that live for the whole launch (block b takes the 1024-cell chunks b, b + G, ... of the batch): what makes them expensive to
// start is what makes them fast -- the OETF table sits in LDS as its 65536 10-bit codes (128 KiB; uhdr_kernels.h), next to the
// sRGB table and GainLUT(metadata, display_boost), so no lookup leaves the CU.  The arithmetic is k_apply_lut's, operation for
// operation (the taps' / 255 through the constant division that is proven equal, the weights from the scale-4 table).
template <int FMT, bool INTERIOR>
__device__ __forceinline__ void lut_cell(const AppConsts& c, const AppImage& im, uint32_t cx, uint32_t cy, const ApplyCellIn& in, int tbl,
                                         const float* s_srgb, const float* s_gain, const uint16_t* s_code) {
  const float e1 = map_to_float_fast(in.mb[0]), e2 = map_to_float_fast(in.mb[1]);
  const float e3 = map_to_float_fast(in.mb[2]), e4 = map_to_float_fast(in.mb[3]);
  float crv[2][2], gcbu[2][2], gcrv[2][2], cbu[2][2];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float u = (float)((int)((in.uu[r] >> (8 * k)) & 0xffu) - 128) * k255;
      const float v = (float)((int)((in.vv[r] >> (8 * k)) & 0xffu) - 128) * k255;
      crv[r][k] = kP3Cr * v; gcbu[r][k] = kP3GCb * u; gcrv[r][k] = kP3GCr * v; cbu[r][k] = kP3Cb * u;
    }
  const float* wt = c_idw4 + (INTERIOR ? 0 : tbl * 64);
#pragma unroll
  for (int oy = 0; oy < 4; ++oy) {
    float o[4][3];
#pragma unroll
    for (int ox = 0; ox < 4; ++ox) {
      const float* w = wt + oy * 16 + ox * 4;
      const float gain = e1 * w[0] + e2 * w[1] + e3 * w[2] + e4 * w[3];
      const float factor = s_gain[lut_index(gain, kGainLutN)];
      const float yf = (float)((in.yrow[oy] >> (8 * ox)) & 0xffu) * k255;
      const int r2 = oy >> 1, k2 = ox >> 1;
      const float r = s_srgb[lut_index_unit(clamp01(yf + crv[r2][k2]), kLutSrgbInvN)];
      const float g = s_srgb[lut_index_unit(clamp01(yf - gcbu[r2][k2] - gcrv[r2][k2]), kLutSrgbInvN)];
      const float bl = s_srgb[lut_index_unit(clamp01(yf + cbu[r2][k2]), kLutSrgbInvN)];
      o[ox][0] = (r * factor) / c.display_boost;
      o[ox][1] = (g * factor) / c.display_boost;
      o[ox][2] = (bl * factor) / c.display_boost;
    }
    const uint32_t pix0 = (4u * cy + oy) * c.width + 4u * cx;
    if (FMT == 2 || FMT == 3) {
      uint4 q;
      uint32_t* qq = &q.x;
#pragma unroll
      for (int ox = 0; ox < 4; ++ox)
        qq[ox] = (uint32_t)s_code[lut_index(o[ox][0], kLutHlgN)] | ((uint32_t)s_code[lut_index(o[ox][1], kLutHlgN)] << 10) |
                 ((uint32_t)s_code[lut_index(o[ox][2], kLutHlgN)] << 20) | (0x3u << 30);
      st_stream(reinterpret_cast<uint4*>(static_cast<uint32_t*>(im.dst) + pix0), q);
    } else if (FMT == 1) {
      const uint2 a = pack_f16(o[0][0], o[0][1], o[0][2]), bb = pack_f16(o[1][0], o[1][1], o[1][2]);
      const uint2 cc = pack_f16(o[2][0], o[2][1], o[2][2]), d = pack_f16(o[3][0], o[3][1], o[3][2]);
      uint4* dst = reinterpret_cast<uint4*>(static_cast<uint2*>(im.dst) + pix0);
      dst[0] = make_uint4(a.x, a.y, bb.x, bb.y);
      dst[1] = make_uint4(cc.x, cc.y, d.x, d.y);
    } else {
      const size_t plane = (size_t)c.width * c.height;
      uint16_t* base16 = static_cast<uint16_t*>(im.dst);
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const uint32_t q0 = 0x3ffu & (uint32_t)(o[0][p] * 1023.0f), q1 = 0x3ffu & (uint32_t)(o[1][p] * 1023.0f);
        const uint32_t q2 = 0x3ffu & (uint32_t)(o[2][p] * 1023.0f), q3 = 0x3ffu & (uint32_t)(o[3][p] * 1023.0f);
        *reinterpret_cast<uint2*>(base16 + p * plane + pix0) = make_uint2(q0 | (q1 << 16), q2 | (q3 << 16));
      }
    }
  }
}

// The same cell for interior waves of calls whose division cannot need operand scaling (AppConsts::lut_plain_div), two horizontally
// adjacent pixels per instruction: every float operation is lut_cell's, in its order, on both pixels at once (v_pk_mul / add / fma;
// clampPixelFloat through the packed clamp modifier: the same value but for the sign of a zero, which indexes the same table
// entry).  The division x / display_boost is the IEEE expansion hipcc emits for it --
//     y0 = rcp(b), e0 = fma(-b, y0, 1), y1 = fma(e0, y0, y0);  q0 = x y1, r0 = fma(-b, q0, x), q1 = fma(r0, y1, q0),
//     r1 = fma(-b, q1, x), q = fma(r1, y1, q1)
// -- without v_div_scale_f32 / v_div_fixup_f32 around it: for operands in the ranges the host has checked those change nothing
// (no rescaling, no special case but x = 0, which the chain maps to +0 by itself), so the quotient is the correctly rounded one,
// bit for bit what the scalar cell computes.  y1 is formed once per thread; a quotient costs five packed instructions per two
// values where the expansion spends eleven per value, a quarter-rate v_rcp_f32 among them.
template <int FMT>
__device__ __forceinline__ void lut_cell_pk(const AppConsts& c, const AppImage& im, uint32_t cx, uint32_t cy, const ApplyCellIn& in,
                                            const float* s_srgb, const float* s_gain, const uint16_t* s_code) {
  const f2 e1 = splat(map_to_float_fast(in.mb[0])), e2 = splat(map_to_float_fast(in.mb[1]));
  const f2 e3 = splat(map_to_float_fast(in.mb[2])), e4 = splat(map_to_float_fast(in.mb[3]));
  float crv[2][2], gcbu[2][2], gcrv[2][2], cbu[2][2];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float u = (float)((int)((in.uu[r] >> (8 * k)) & 0xffu) - 128) * k255;
      const float v = (float)((int)((in.vv[r] >> (8 * k)) & 0xffu) - 128) * k255;
      crv[r][k] = kP3Cr * v; gcbu[r][k] = kP3GCb * u; gcrv[r][k] = kP3GCr * v; cbu[r][k] = kP3Cb * u;
    }
  const float db = c.display_boost, y0 = __builtin_amdgcn_rcpf(db), e0 = __builtin_fmaf(-db, y0, 1.0f), y1 = __builtin_fmaf(e0, y0, y0);
  const f2 nb = splat(-db), yy = splat(y1);
#pragma unroll
  for (int oy = 0; oy < 4; ++oy) {
    f2 o[2][3];
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      const float* w = c_idw4p + (oy * 2 + pr) * 8;
      f2 gain = e1 * (f2){w[0], w[1]};
      gain = gain + e2 * (f2){w[2], w[3]};
      gain = gain + e3 * (f2){w[4], w[5]};
      gain = gain + e4 * (f2){w[6], w[7]};
      // (0 <= gain <= 1 + a few ulps -- weights that sum to 1, taps in [0, 1] -- so the product stays below N - 1/2: CLIP3 cannot bite)
      const f2 tg = gain * splat((float)(kGainLutN - 1u));
      const f2 factor = (f2){s_gain[round_half_up(tg.x)], s_gain[round_half_up(tg.y)]};
      const f2 yf = (pr ? (f2){cvt_byte<2>(in.yrow[oy]), cvt_byte<3>(in.yrow[oy])} : (f2){cvt_byte<0>(in.yrow[oy]), cvt_byte<1>(in.yrow[oy])}) * splat(k255);
      const int r2 = oy >> 1;
      const f2 rr = pk_add_sat(yf, splat(crv[r2][pr])), gg = pk_add_sat(yf - splat(gcbu[r2][pr]), splat(-gcrv[r2][pr]));
      const f2 bb = pk_add_sat(yf, splat(cbu[r2][pr]));
      const f2 tr = rr * splat((float)(kLutSrgbInvN - 1u)), tgr = gg * splat((float)(kLutSrgbInvN - 1u)), tb = bb * splat((float)(kLutSrgbInvN - 1u));
      const f2 ch[3] = {(f2){s_srgb[round_half_up(tr.x)], s_srgb[round_half_up(tr.y)]},
                        (f2){s_srgb[round_half_up(tgr.x)], s_srgb[round_half_up(tgr.y)]},
                        (f2){s_srgb[round_half_up(tb.x)], s_srgb[round_half_up(tb.y)]}};
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const f2 x = ch[q] * factor;                           // applyGainLUT
        const f2 q0 = x * yy, r0 = pk_fma(nb, q0, x), q1 = pk_fma(r0, yy, q0), r1 = pk_fma(nb, q1, x);
        // / display_boost (ultrahdr.cpp:451).  HLG / PQ: the quotient only indexes the OETF table, where everything from 1.0 up
        // lands on the last entry (CLIP3): saturating it to 1.0 -- the instruction's clamp bit -- stands in for the clip of the index
        if (FMT == 2 || FMT == 3) o[pr][q] = pk_fma_sat(r1, yy, q1) * splat((float)(kLutHlgN - 1u));
        else o[pr][q] = pk_fma(r1, yy, q1);
      }
    }
    const uint32_t pix0 = (4u * cy + oy) * c.width + 4u * cx;
    if (FMT == 2 || FMT == 3) {
      uint4 qv;
      uint32_t* qq = &qv.x;
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        qq[2 * pr] = (uint32_t)s_code[round_half_up(o[pr][0].x)] | ((uint32_t)s_code[round_half_up(o[pr][1].x)] << 10) |
                     ((uint32_t)s_code[round_half_up(o[pr][2].x)] << 20) | (0x3u << 30);
        qq[2 * pr + 1] = (uint32_t)s_code[round_half_up(o[pr][0].y)] | ((uint32_t)s_code[round_half_up(o[pr][1].y)] << 10) |
                         ((uint32_t)s_code[round_half_up(o[pr][2].y)] << 20) | (0x3u << 30);
      }
      st_stream(reinterpret_cast<uint4*>(static_cast<uint32_t*>(im.dst) + pix0), qv);
    } else if (FMT == 1) {
      const uint2 a = pack_f16(o[0][0].x, o[0][1].x, o[0][2].x), b2 = pack_f16(o[0][0].y, o[0][1].y, o[0][2].y);
      const uint2 cc = pack_f16(o[1][0].x, o[1][1].x, o[1][2].x), d = pack_f16(o[1][0].y, o[1][1].y, o[1][2].y);
      uint4* dst = reinterpret_cast<uint4*>(static_cast<uint2*>(im.dst) + pix0);
      dst[0] = make_uint4(a.x, a.y, b2.x, b2.y);
      dst[1] = make_uint4(cc.x, cc.y, d.x, d.y);
    } else {
      const size_t plane = (size_t)c.width * c.height;
      uint16_t* base16 = static_cast<uint16_t*>(im.dst);
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const uint32_t q0 = 0x3ffu & (uint32_t)(o[0][p].x * 1023.0f), q1 = 0x3ffu & (uint32_t)(o[0][p].y * 1023.0f);
        const uint32_t q2 = 0x3ffu & (uint32_t)(o[1][p].x * 1023.0f), q3 = 0x3ffu & (uint32_t)(o[1][p].y * 1023.0f);
        *reinterpret_cast<uint2*>(base16 + p * plane + pix0) = make_uint2(q0 | (q1 << 16), q2 | (q3 << 16));
      }
    }
  }
}

constexpr uint32_t kLutS4Block = 1024;
template <int FMT>
__global__ void __launch_bounds__(1024) k_apply_lut_s4(const AppConsts c, const AppBatch b, uint32_t n) {
  constexpr bool kOetf = FMT == 2 || FMT == 3;
  extern __shared__ uint4 s_dyn[];                       // [codes 128 KiB (HLG / PQ output only)] [sRGB 4 KiB] [gain 4 KiB]
  uint16_t* s_code = reinterpret_cast<uint16_t*>(s_dyn);
  float* s_srgb = reinterpret_cast<float*>(reinterpret_cast<char*>(s_dyn) + (kOetf ? kLutHlgN * 2u : 0u));
  float* s_gain = s_srgb + kLutSrgbInvN;
  static_assert(kLutSrgbInvN == kLutS4Block && kGainLutN == kLutS4Block, "one table entry per thread");
  s_srgb[threadIdx.x] = c.lut[kLutSrgbInv + threadIdx.x];
  s_gain[threadIdx.x] = gain_lut_entry(threadIdx.x, c.log2_min_d, c.log2_max_d, c.lut_boost_factor);
  if (kOetf) {
    const uint4* src = reinterpret_cast<const uint4*>(c.lut + (FMT == 3 ? kCodeHlg : kCodePq));
    uint4 t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = src[k * kLutS4Block + threadIdx.x];
#pragma unroll
    for (int k = 0; k < 8; ++k) s_dyn[k * kLutS4Block + threadIdx.x] = t[k];
  }
  __syncthreads();
  const uint32_t cells = c.map_w * c.map_h, per_img = (cells + kLutS4Block - 1u) / kLutS4Block, total = per_img * n;
  for (uint32_t id = blockIdx.x; id < total; id += gridDim.x) {
    const uint32_t img = id / per_img, idx = (id - img * per_img) * kLutS4Block + threadIdx.x;
    if (idx >= cells) continue;
    const AppImage& im = b.img[img];
    const uint32_t cy = idx / c.map_w, cx = idx - cy * c.map_w;
    ApplyCellIn in;
    apply_load_cell(c, im, cx, cy, in);
    const bool edge_x = cx + 1u == c.map_w, edge_y = cy + 1u == c.map_h;
    const int tbl = edge_x ? (edge_y ? 3 : 1) : (edge_y ? 2 : 0);
    if (__builtin_amdgcn_ballot_w64(tbl != 0) != 0ull) lut_cell<FMT, false>(c, im, cx, cy, in, tbl, s_srgb, s_gain, s_code);
    else if (c.lut_plain_div != 0u) lut_cell_pk<FMT>(c, im, cx, cy, in, s_srgb, s_gain, s_code);
    else lut_cell<FMT, true>(c, im, cx, cy, in, 0, s_srgb, s_gain, s_code);
  }
}

template <int FMT>
static hipError_t launch_apply_t(const AppConsts& c, const AppBatch& b, int n, int mode,
                                 bool fast_s4, hipStream_t s) {
  if (n == 0 || c.width == 0 || c.height == 0) return hipSuccess;
  const bool exact = mode == 1 || mode == 3;
  if (mode == 2) {
    if (c.lut == nullptr) return hipErrorInvalidValue;
    if (fast_s4) {
      constexpr bool kOetf = FMT == 2 || FMT == 3;
      const uint32_t lds = (kOetf ? kLutHlgN * 2u : 0u) + (kLutSrgbInvN + kGainLutN) * 4u;
      const uint32_t chunks = (c.map_w * c.map_h + kLutS4Block - 1u) / kLutS4Block * (uint32_t)n;
      // more than 64 KiB of LDS has to be allowed per function (and per device: set on every launch, it is a host-side flag)
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_apply_lut_s4<FMT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess) {
        hipLaunchKernelGGL((k_apply_lut_s4<FMT>), dim3(chunks < 256u ? chunks : 256u), dim3(kLutS4Block), lds, s, c, b, (uint32_t)n);
        return hipGetLastError();
      }
      (void)hipGetLastError();   // (a device that does not grant the LDS: the per-pixel kernel below gives the same bytes)
    }
    const size_t total = (size_t)c.width * c.height, per_block = 256u * (size_t)kLutPixelsPerThread;
    hipLaunchKernelGGL((k_apply_lut<FMT>), dim3((unsigned)((total + per_block - 1u) / per_block), n), dim3(256), 0, s, c, b);
    return hipGetLastError();
  }
  if (fast_s4 && !exact && c.map_w >= 2u) {   // (a map one cell wide: the per-pixel kernel; apply_load_cell_pk reads two columns)
    // Cells per thread: a block copies 15-37 KB of tables into LDS before its first pixel, so it should walk many cells -- but a
    // launch also has to fill 256 CUs x 2 resident blocks, or a single 4K image (2025 blocks of 256 cells) would leave three
    // quarters of the chip idle with 8 cells per thread.
    AppConsts cc = c;
    const uint32_t total = c.map_w * c.map_h;
    uint32_t cpt = kApplyMaxCellsPerThread;
    auto blocks = [&](uint32_t k) { return (uint64_t)((total + kApplyBlock * k - 1u) / (kApplyBlock * k)) * (uint64_t)n; };
    while (cpt > 1u && blocks(cpt) < 448u) cpt >>= 1;   // (a single 4K image: 507 blocks of 2 cells per thread, all resident at once)
    cc.cells_per_thread = cpt;
    cc.step_x = kApplyBlock % c.map_w;
    cc.step_y = kApplyBlock / c.map_w;
    const dim3 grid(n, (unsigned)((total + kApplyBlock * cpt - 1u) / (kApplyBlock * cpt)));
    // Channels can only exceed 1.0 (reach code 1024 and wrap through the reference's & 0x3ff; leave the stage-2 table) when the
    // display boost is capped below the content boost -- and then only if the largest factor the call can produce,
    // max(minBoost, maxBoost)^(display / max) / display, is above 1: a display boost of 2 under a content boost of 4.9 stays
    // below (0.955), a display boost of 1 does not (1.38).
    // The factor alone decides: with minContentBoost > maxContentBoost (nothing on the apply path refuses it, ultrahdr.cpp:360-425;
    // a crafted XMP gets here through decodeJPEGR) `top` is far above 1 although display_boost == max_boost, and stage 2's cell
    // number -- byte 2 of 2 + 2u, unclamped -- would index past its 129 cells.
    const double top = std::exp2(std::fmax(c.log2_min_d, c.log2_max_d) * (double)c.display_boost / (double)c.max_boost) / (double)c.display_boost;
    const bool mask = !(top <= 1.0 + 1e-6);
    if (!mask && ApplyTab<FMT, false>::kOetf) {
      // u = T(c) * 2^(g E), handed to stage 2 as 2 + 2u: the exponent's constants times g (1/2 for HLG: sqrt; m1 for PQ), plus 1
      const float g = FMT == 3 ? 0.5f : UHDR_PQ_M1;
      cc.fast.A *= g; cc.fast.A255 *= g; cc.fast.B = cc.fast.B * g + 1.0f;
    }
    // sampleMap's standard weights times A / 255: ONE float product per weight, the same one the kernel forms for the per-lane
    // tables of the last column / row (its prologue, round to nearest) -- a cell gets the same bytes whichever form computes it
    for (int i = 0; i < 4 * 2 * 3 * 2; ++i) (&cc.fast.wD[0][0][0][0])[i] *= cc.fast.A255;
    if (mask) hipLaunchKernelGGL((k_apply_s4<FMT, true>), grid, dim3(kApplyBlock), 0, s, cc, b);
    else hipLaunchKernelGGL((k_apply_s4<FMT, false>), grid, dim3(kApplyBlock), 0, s, cc, b);
  } else {
    const dim3 grid = px_grid(c.width, c.height, n);
    if (exact && c.ex_ws != nullptr) {
      if (fast_s4) hipLaunchKernelGGL((k_apply_s4_est<FMT>), dim3((c.map_w * c.map_h + 255u) / 256u, n), dim3(256), 0, s, c, b);
      else hipLaunchKernelGGL((k_apply_px_est<FMT>), grid, dim3(256), 0, s, c, b);
      hipLaunchKernelGGL((k_apply_resolve<FMT>), dim3(n, kExSlices), dim3(256), 0, s, c, b);
    } else if (exact) hipLaunchKernelGGL((k_apply_px<FMT, true>), grid, dim3(256), 0, s, c, b);
    else hipLaunchKernelGGL((k_apply_px<FMT, false>), grid, dim3(256), 0, s, c, b);
  }
  return hipGetLastError();
}

hipError_t launch_apply(const AppConsts& c, const AppBatch& b, int n, int fmt, int mode,
                        bool fast_s4, hipStream_t s) {
  switch (fmt) {
    case 1: return launch_apply_t<1>(c, b, n, mode, fast_s4, s);
    case 2: return launch_apply_t<2>(c, b, n, mode, fast_s4, s);
    case 3: return launch_apply_t<3>(c, b, n, mode, fast_s4, s);
    case 4: return launch_apply_t<4>(c, b, n, mode, fast_s4, s);
    default: return hipErrorInvalidValue;
  }
}

// =================================================================================================
// The SDR rendition of decodeJPEGR (jpegr.cpp:768-786): what JpegDecoderHelper::decompressImage(..., DECODE_TO_RGBA) makes of the
// primary image (jpegdecoderhelper.cpp:251-281), i.e. libjpeg-turbo's own path from the decoded 4:2:0 planes to RGBA:
//   jdsample.c h2v2_fancy_upsample  triangle filter over the four nearest chroma samples, weights 9 3 3 1 / 16, rounding constant 8
//                                    on even and 7 on odd columns, edges replicated; chroma planes at most 2 samples wide are
//                                    replicated instead (jinit_upsampler)
//   jdcolor.c ycc_rgb_convert        R = Y + ((91881 (Cr-128) + 32768) >> 16) etc., the tables of build_ycc_rgb_table evaluated in
//                                    place, clamp to 0..255, alpha 0xFF
// One thread per horizontal pixel pair (they share the centre chroma column).  Restated in oracle/jpeg_oracle.c and pinned there
// against libjpeg-turbo itself (Pillow's).
// =================================================================================================
__global__ void __launch_bounds__(256) k_ycc420_rgba(const uint8_t* __restrict__ yp, const uint8_t* __restrict__ cbp, const uint8_t* __restrict__ crp,
                                                     uint32_t w, uint32_t h, uint32_t ys, uint32_t cs, uint8_t* __restrict__ rgba) {
  const uint32_t cw = w >> 1, ch = h >> 1;
  const uint32_t c = blockIdx.x * 256u + threadIdx.x, r = blockIdx.y;
  if (c >= cw) return;
  const uint32_t i = r >> 1;
  const uint32_t o = (r & 1u) ? min(i + 1u, ch - 1u) : (i ? i - 1u : 0u);
  const uint32_t cl = c ? c - 1u : 0u, cr_ = min(c + 1u, cw - 1u);
  int cb[2], cr[2];
  if (cw > 2u) {
    auto colsum = [&](const uint8_t* p, uint32_t col) { return 3 * (int)p[i * cs + col] + (int)p[o * cs + col]; };
    const int b0 = colsum(cbp, c), r0 = colsum(crp, c);
    cb[0] = (3 * b0 + colsum(cbp, cl) + 8) >> 4; cb[1] = (3 * b0 + colsum(cbp, cr_) + 7) >> 4;
    cr[0] = (3 * r0 + colsum(crp, cl) + 8) >> 4; cr[1] = (3 * r0 + colsum(crp, cr_) + 7) >> 4;
  } else {
    cb[0] = cb[1] = cbp[i * cs + c];
    cr[0] = cr[1] = crp[i * cs + c];
  }
  const uint8_t* yrow = yp + (size_t)r * ys + 2u * c;
  const uint32_t y2 = (uint32_t)yrow[0] | ((uint32_t)yrow[1] << 8);
  uint32_t px[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int y = (int)((y2 >> (8 * k)) & 0xffu), xb = cb[k] - 128, xr = cr[k] - 128;
    int R = y + ((91881 * xr + 32768) >> 16);
    int G = y + ((-22554 * xb + 32768 - 46802 * xr) >> 16);
    int B = y + ((116130 * xb + 32768) >> 16);
    R = min(max(R, 0), 255); G = min(max(G, 0), 255); B = min(max(B, 0), 255);
    px[k] = (uint32_t)R | ((uint32_t)G << 8) | ((uint32_t)B << 16) | 0xFF000000u;
  }
  *reinterpret_cast<uint2*>(rgba + ((size_t)r * w + 2u * c) * 4u) = make_uint2(px[0], px[1]);
}

hipError_t launch_ycc420_to_rgba(const uint8_t* y, const uint8_t* cb, const uint8_t* cr, uint32_t w, uint32_t h, uint32_t y_stride,
                                 uint32_t c_stride, uint8_t* rgba, hipStream_t s) {
  if (w == 0 || h == 0) return hipSuccess;
  hipLaunchKernelGGL(k_ycc420_rgba, dim3((w / 2u + 255u) / 256u, h), dim3(256), 0, s, y, cb, cr, w, h, y_stride, c_stride, rgba);
  return hipGetLastError();
}

// =================================================================================================
// toneMap (ultrahdr.cpp:517-558): Y8 = (Y16 >> 6 >> 2) & 0xff == bits 15..8 of the P010 word
// =================================================================================================

// luma: grid.y = row, grid.z = image; one thread per 8 (ALIGNED) or 1 destination bytes across [0, dy_stride)
template <bool ALIGNED>
__global__ void __launch_bounds__(256) k_tonemap_luma(const ToneBatch b) {
  const ToneImage& t = b.img[blockIdx.z];
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  for (uint32_t row = blockIdx.y; row < t.height; row += gridDim.y) {   // (rows beyond the grid's 65535 by striding)
    const uint16_t* src = t.sy + (size_t)row * t.sy_stride;
    uint8_t* dst = t.dy + (size_t)row * t.dy_stride;
    if (ALIGNED) {
      const uint32_t x = i * 8u;
      if (x >= t.dy_stride) return;
      uint2 o = make_uint2(0u, 0u);
      if (x < t.width) {  // width % 8 == 0 on this path
        const uint4 q = *reinterpret_cast<const uint4*>(src + x);
        o.x = ((q.x >> 8) & 0xffu) | ((q.x >> 24) << 8) | (((q.y >> 8) & 0xffu) << 16) | ((q.y >> 24) << 24);
        o.y = ((q.z >> 8) & 0xffu) | ((q.z >> 24) << 8) | (((q.w >> 8) & 0xffu) << 16) | ((q.w >> 24) << 24);
      }
      *reinterpret_cast<uint2*>(dst + x) = o;
    } else {
      if (i >= t.dy_stride && i >= t.width) return;
      if (i < t.width) dst[i] = (uint8_t)((src[i] >> 6 >> 2) & 0xff);
      else dst[i] = 0;  // memset(dst_y_row + width, 0, luma_stride - width)
    }
  }
}

// chroma: grid.y = chroma row; U and V de-interleaved, padding [width/2, dc_stride) zeroed
template <bool ALIGNED>
__global__ void __launch_bounds__(256) k_tonemap_chroma(const ToneBatch b) {
  const ToneImage& t = b.img[blockIdx.z];
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  const uint32_t cw = t.width / 2u;
  for (uint32_t row = blockIdx.y; row < t.height / 2u; row += gridDim.y) {
    const uint16_t* src = t.suv + (size_t)row * t.suv_stride;
    uint8_t* du = t.du + (size_t)row * t.dc_stride;
    uint8_t* dv = t.dv + (size_t)row * t.dc_stride;
    if (ALIGNED) {
      const uint32_t x = i * 8u;  // chroma sample index; cw % 8 == 0 on this path
      if (x >= t.dc_stride) return;
      uint2 ou = make_uint2(0u, 0u), ov = make_uint2(0u, 0u);
      if (x < cw) {
        const uint4 a = *reinterpret_cast<const uint4*>(src + 2u * x);
        const uint4 bq = *reinterpret_cast<const uint4*>(src + 2u * x + 8u);
        ou.x = ((a.x >> 8) & 0xffu) | (((a.y >> 8) & 0xffu) << 8) | (((a.z >> 8) & 0xffu) << 16) | (((a.w >> 8) & 0xffu) << 24);
        ov.x = (a.x >> 24) | ((a.y >> 24) << 8) | ((a.z >> 24) << 16) | ((a.w >> 24) << 24);
        ou.y = ((bq.x >> 8) & 0xffu) | (((bq.y >> 8) & 0xffu) << 8) | (((bq.z >> 8) & 0xffu) << 16) | (((bq.w >> 8) & 0xffu) << 24);
        ov.y = (bq.x >> 24) | ((bq.y >> 24) << 8) | ((bq.z >> 24) << 16) | ((bq.w >> 24) << 24);
      }
      *reinterpret_cast<uint2*>(du + x) = ou;
      *reinterpret_cast<uint2*>(dv + x) = ov;
    } else {
      if (i >= t.dc_stride && i >= cw) return;
      if (i < cw) {
        du[i] = (uint8_t)((src[2u * i] >> 6 >> 2) & 0xff);
        dv[i] = (uint8_t)((src[2u * i + 1u] >> 6 >> 2) & 0xff);
      } else {
        du[i] = 0; dv[i] = 0;
      }
    }
  }
}

hipError_t launch_tonemap(const ToneBatch& b, int n, bool aligned, hipStream_t s) {
  const ToneImage& t = b.img[0];
  if (n <= 0 || t.width == 0 || t.height == 0) return hipSuccess;
  // the grid covers the widest destination row of the launch (every thread checks its own image's strides)
  uint32_t lcols = t.width, ccols = t.width / 2u;
  for (int i = 0; i < n; ++i) { lcols = b.img[i].dy_stride > lcols ? b.img[i].dy_stride : lcols; ccols = b.img[i].dc_stride > ccols ? b.img[i].dc_stride : ccols; }
  if (aligned) {
    hipLaunchKernelGGL((k_tonemap_luma<true>), dim3((lcols / 8u + 255u) / 256u, t.height < 65535u ? t.height : 65535u, n), dim3(256), 0, s, b);
    if (t.height / 2u)
      hipLaunchKernelGGL((k_tonemap_chroma<true>), dim3((ccols / 8u + 255u) / 256u, t.height / 2u < 65535u ? t.height / 2u : 65535u, n), dim3(256), 0, s, b);
  } else {
    hipLaunchKernelGGL((k_tonemap_luma<false>), dim3((lcols + 255u) / 256u, t.height < 65535u ? t.height : 65535u, n), dim3(256), 0, s, b);
    if (t.height / 2u && ccols)
      hipLaunchKernelGGL((k_tonemap_chroma<false>), dim3((ccols + 255u) / 256u, t.height / 2u < 65535u ? t.height / 2u : 65535u, n), dim3(256), 0, s, b);
  }
  return hipGetLastError();
}

// =================================================================================================
// convertYuv (jpegr.cpp:1199-1203 over transformYuv420, gainmapmath.cpp:483-520), in place.
// Every 2x2 block reads and writes only its own 4 luma + 1 U + 1 V bytes, so blocks are independent.
// =================================================================================================

struct Yuv { float y, u, v; };
__device__ __forceinline__ Yuv yuv_mat(const float (&m)[9], float y, float u, float v) {
  Yuv o;
  o.y = m[0] * y + m[1] * u + m[2] * v;
  o.u = m[3] * y + m[4] * u + m[5] * v;
  o.v = m[6] * y + m[7] * u + m[8] * v;
  return o;
}
__device__ __forceinline__ uint32_t round_u8(float x) {  // (uint8_t)CLIP3(x, 0, 255), x already + 0.5f
  x = (x < 0.0f) ? 0.0f : (x > 255.0f) ? 255.0f : x;
  return (uint32_t)x;
}
// one 2x2 block; y00..y11 luma bytes, ub/vb chroma bytes.  The kernel is bound by VALU issue, not by its 3 bytes per pixel, so
// the four pixels run as two packed pairs: every float operation below is the reference's own (the products and sums of its yuvXToY
// matrices in their order, gainmapmath.cpp:447-481; the chroma average and the +0.5 / CLIP3 of transformYuv420, :483-520), two at a
// time; the products of the block's single (u, v) with the matrix are formed once.
__device__ __forceinline__ void cvt_block(const float (&m)[9], uint32_t (&yb)[4], uint32_t& ub, uint32_t& vb) {
  const float u = (float)((int)ub - 128) * k255, v = (float)((int)vb - 128) * k255;
  const f2 y01 = (f2){(float)yb[0], (float)yb[1]} * splat(k255), y23 = (f2){(float)yb[2], (float)yb[3]} * splat(k255);
  const float yu = m[1] * u, yv = m[2] * v, uu = m[4] * u, uv = m[5] * v, vu = m[7] * u, vv = m[8] * v;
  const f2 py01 = (splat(m[0]) * y01 + splat(yu)) + splat(yv), py23 = (splat(m[0]) * y23 + splat(yu)) + splat(yv);
  const f2 pu01 = (splat(m[3]) * y01 + splat(uu)) + splat(uv), pu23 = (splat(m[3]) * y23 + splat(uu)) + splat(uv);
  const f2 pv01 = (splat(m[6]) * y01 + splat(vu)) + splat(vv), pv23 = (splat(m[6]) * y23 + splat(vu)) + splat(vv);
  // (((p0 + p1) + p2) + p3) / 4 for u and v side by side
  f2 c = (f2){pu01.x, pv01.x} + (f2){pu01.y, pv01.y};
  c = c + (f2){pu23.x, pv23.x};
  c = c + (f2){pu23.y, pv23.y};
  c = c * splat(0.25f);                                    // == / 4.0f
  c = (c * splat(255.0f) + splat(128.0f)) + splat(0.5f);
  const f2 o01 = py01 * splat(255.0f) + splat(0.5f), o23 = py23 * splat(255.0f) + splat(0.5f);
  // (uint8_t)CLIP3(x, 0, 255): one v_med3_f32 (the same value as the reference's compare chain for every x but NaN, which
  // bytes times a finite matrix cannot produce) and a truncating conversion
  yb[0] = (uint32_t)__builtin_amdgcn_fmed3f(o01.x, 0.0f, 255.0f); yb[1] = (uint32_t)__builtin_amdgcn_fmed3f(o01.y, 0.0f, 255.0f);
  yb[2] = (uint32_t)__builtin_amdgcn_fmed3f(o23.x, 0.0f, 255.0f); yb[3] = (uint32_t)__builtin_amdgcn_fmed3f(o23.y, 0.0f, 255.0f);
  ub = (uint32_t)__builtin_amdgcn_fmed3f(c.x, 0.0f, 255.0f);
  vb = (uint32_t)__builtin_amdgcn_fmed3f(c.y, 0.0f, 255.0f);
}

template <bool ALIGNED>
__global__ void __launch_bounds__(256) k_convert_yuv(const CvtBatch b) {
  const CvtImage& t = b.img[blockIdx.z];
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  const uint32_t cw = t.width / 2u;
  for (uint32_t cyr = blockIdx.y; cyr < t.height / 2u; cyr += gridDim.y) {   // chroma row (rows beyond the grid's 65535 by striding)
    uint8_t* y0 = t.y + (size_t)(2u * cyr) * t.y_stride;
    uint8_t* y1 = y0 + t.y_stride;
    uint8_t* ur = t.u + (size_t)cyr * t.c_stride;
    uint8_t* vr = t.v + (size_t)cyr * t.c_stride;
    const uint8_t* sy0 = t.sy + (size_t)(2u * cyr) * t.sy_stride;
    const uint8_t* sy1 = sy0 + t.sy_stride;
    const uint8_t* sur = t.su + (size_t)cyr * t.sc_stride;
    const uint8_t* svr = t.sv + (size_t)cyr * t.sc_stride;
    if (ALIGNED) {  // 4 chroma samples = 8 luma columns per thread; cw % 4 == 0
      const uint32_t cx = i * 4u;
      if (cx >= cw) return;
      uint2 a = *reinterpret_cast<const uint2*>(sy0 + 2u * cx);
      uint2 bq = *reinterpret_cast<const uint2*>(sy1 + 2u * cx);
      uint32_t uw = *reinterpret_cast<const uint32_t*>(sur + cx);
      uint32_t vw = *reinterpret_cast<const uint32_t*>(svr + cx);
      const uint32_t top[2] = {a.x, a.y}, bot[2] = {bq.x, bq.y};
      uint32_t otop[2] = {0u, 0u}, obot[2] = {0u, 0u}, ou = 0u, ov = 0u;
  #pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int sh = 16 * (k & 1);
        uint32_t yb[4] = {(top[k >> 1] >> sh) & 0xffu, (top[k >> 1] >> (sh + 8)) & 0xffu,
                          (bot[k >> 1] >> sh) & 0xffu, (bot[k >> 1] >> (sh + 8)) & 0xffu};
        uint32_t ub = (uw >> (8 * k)) & 0xffu, vb = (vw >> (8 * k)) & 0xffu;
        cvt_block(b.img[0].m, yb, ub, vb);
        otop[k >> 1] |= (yb[0] << sh) | (yb[1] << (sh + 8));
        obot[k >> 1] |= (yb[2] << sh) | (yb[3] << (sh + 8));
        ou |= ub << (8 * k); ov |= vb << (8 * k);
      }
      *reinterpret_cast<uint2*>(y0 + 2u * cx) = make_uint2(otop[0], otop[1]);
      *reinterpret_cast<uint2*>(y1 + 2u * cx) = make_uint2(obot[0], obot[1]);
      *reinterpret_cast<uint32_t*>(ur + cx) = ou;
      *reinterpret_cast<uint32_t*>(vr + cx) = ov;
    } else {
      if (i >= cw) return;
      uint32_t yb[4] = {sy0[2u * i], sy0[2u * i + 1u], sy1[2u * i], sy1[2u * i + 1u]};
      uint32_t ub = sur[i], vb = svr[i];
      cvt_block(b.img[0].m, yb, ub, vb);
      y0[2u * i] = (uint8_t)yb[0]; y0[2u * i + 1u] = (uint8_t)yb[1];
      y1[2u * i] = (uint8_t)yb[2]; y1[2u * i + 1u] = (uint8_t)yb[3];
      ur[i] = (uint8_t)ub; vr[i] = (uint8_t)vb;
    }
  }
}

hipError_t launch_convert_yuv(const CvtBatch& b, int n, bool aligned, hipStream_t s) {
  const uint32_t cw = b.img[0].width / 2u, ch = b.img[0].height / 2u;
  if (n <= 0 || cw == 0 || ch == 0) return hipSuccess;
  const uint32_t rows = ch < 65535u ? ch : 65535u;
  if (aligned) hipLaunchKernelGGL((k_convert_yuv<true>), dim3((cw / 4u + 255u) / 256u, rows, n), dim3(256), 0, s, b);
  else hipLaunchKernelGGL((k_convert_yuv<false>), dim3((cw + 255u) / 256u, rows, n), dim3(256), 0, s, b);
  return hipGetLastError();
}

// =================================================================================================
// editorhelper effects (ref lib/src/editorhelper.cpp:26-360): pure byte gathers.  One thread produces 4
// consecutive output bytes of one plane row (one dword store when the row is aligned), grid.z = plane job.
// =================================================================================================
__device__ __forceinline__ size_t fx_src_index(const FxJob& j, uint32_t i, uint32_t c) {
  switch (j.op) {
    case FX_FLIP_V: return (size_t)(j.rows - i - 1u) * j.src_stride + c;
    case FX_FLIP_H: return (size_t)i * j.src_stride + (j.in_w - c - 1u);
    case FX_ROT90: return (size_t)(j.in_h - c - 1u) * j.src_stride + i;
    case FX_ROT180: return (size_t)(j.in_h - i - 1u) * j.src_stride + (j.in_w - c - 1u);
    case FX_ROT270: return (size_t)c * j.src_stride + (j.in_w - i - 1u);
    case FX_RESIZE: return ((size_t)i * j.row_num / j.row_den) * j.src_stride + (size_t)c * j.col_num / j.col_den;
    default: return (size_t)i * j.src_stride + c;  // FX_COPY
  }
}

// 4 output bytes at column c0 of output row i, gathered one by one (any operation, any alignment)
__device__ __forceinline__ void fx_quad(const FxJob& j, uint32_t i, uint32_t c0, uint8_t* d) {
  if (c0 + 4u <= j.cols && ((reinterpret_cast<uintptr_t>(d) & 3u) == 0u)) {
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) v |= (uint32_t)j.src[fx_src_index(j, i, c0 + k)] << (8 * k);
    *reinterpret_cast<uint32_t*>(d) = v;
  } else {
    for (uint32_t k = 0; k < 4u && c0 + k < j.cols; ++k) d[k] = j.src[fx_src_index(j, i, c0 + k)];
  }
}
__device__ __forceinline__ uint32_t bswap32(uint32_t v) { return __builtin_amdgcn_perm(v, v, 0x00010203u); }

// One thread produces 16 consecutive output bytes of one plane row.  Copies, flips and the half turn move whole 16-byte pieces
// when both sides are aligned (the mirrored ones with their bytes reversed: four v_perm_b32); resize gathers from a copy of the
// source stretch in LDS.
__global__ void __launch_bounds__(256) k_effect(const FxJobs jobs) {
  const FxJob& j = jobs.job[blockIdx.z];
  const uint32_t cb = blockIdx.x * 4096u;                 // the block's first output column
  if (cb >= j.cols) return;                               // (block-uniform)
  const uint32_t c0 = cb + threadIdx.x * 16u;
  const bool active = c0 < j.cols;
  // resize, at most 4 source bytes per output byte: the block's stretch of the source row is copied into LDS with dword loads and
  // gathered from there (16 byte loads per thread made the kernel load-instruction bound: 44 us for a 4K -> 5760x3240 resize)
  __shared__ uint32_t s_row[(4096u * 4u + 16u) / 4u];
  const bool lds_resize = j.op == FX_RESIZE && j.col_num <= 4u * j.col_den && (uint64_t)j.rows * j.row_num < (1ull << 32) &&
                          (uint64_t)j.cols * j.col_num < (1ull << 32);
  for (uint32_t i = blockIdx.y; i < j.rows; i += gridDim.y) {   // (rows beyond the grid's 65535 by striding)
    uint8_t* d = j.dst + (size_t)i * j.dst_stride + c0;
    const bool whole = active && c0 + 16u <= j.cols && (reinterpret_cast<uintptr_t>(d) & 15u) == 0u;
    if (lds_resize) {
      const uint8_t* srow = j.src + (size_t)(i * j.row_num / j.row_den) * j.src_stride;
      const uint32_t c_last = min(cb + 4095u, j.cols - 1u);
      const uint32_t s_lo = cb * j.col_num / j.col_den, s_hi = min(c_last * j.col_num / j.col_den, j.in_w - 1u);
      const uint8_t* first = srow + s_lo;
      const uint32_t off0 = (uint32_t)(reinterpret_cast<uintptr_t>(first) & 3u);
      const uint8_t* p_al = first - off0;
      const uint32_t nbytes = s_hi - s_lo + 1u + off0;
      const uint8_t* row_end = srow + j.in_w;
      __syncthreads();                                      // (the previous row's gathers are done)
      for (uint32_t k = threadIdx.x; k < (nbytes + 3u) / 4u; k += 256u) {
        const uint8_t* p = p_al + 4u * k;
        uint32_t v;
        if (p >= srow && p + 4 <= row_end) v = *reinterpret_cast<const uint32_t*>(p);
        else {                                              // the row's first / last bytes: never read outside [srow, row_end)
          v = 0;
          for (int b = 0; b < 4; ++b) if (p + b >= srow && p + b < row_end) v |= (uint32_t)p[b] << (8 * b);
        }
        s_row[k] = v;
      }
      __syncthreads();
      if (active) {
        const uint8_t* lrow = reinterpret_cast<const uint8_t*>(s_row) + off0;
        const uint32_t step = j.col_num / j.col_den, rem = j.col_num - step * j.col_den;
        const uint32_t p0 = c0 * j.col_num;
        uint32_t q = p0 / j.col_den, r = p0 - q * j.col_den;
        q -= s_lo;
        if (whole) {
          uint32_t w[4];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            uint32_t v = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              v |= (uint32_t)lrow[q] << (8 * k);
              q += step; r += rem;
              if (r >= j.col_den) { r -= j.col_den; ++q; }
            }
            w[g] = v;
          }
          *reinterpret_cast<uint4*>(d) = make_uint4(w[0], w[1], w[2], w[3]);
        } else {
          for (uint32_t k = 0; k < 16u && c0 + k < j.cols; ++k) {
            d[k] = lrow[q];
            q += step; r += rem;
            if (r >= j.col_den) { r -= j.col_den; ++q; }
          }
        }
      }
      continue;
    }
    if (!active) continue;
    if (whole && (j.op == FX_COPY || j.op == FX_FLIP_V || j.op == FX_FLIP_H || j.op == FX_ROT180)) {
      const bool rev = j.op == FX_FLIP_H || j.op == FX_ROT180;
      const uint32_t sr = j.op == FX_COPY || j.op == FX_FLIP_H ? i : (j.op == FX_FLIP_V ? j.rows - i - 1u : j.in_h - i - 1u);
      const uint8_t* sp = j.src + (size_t)sr * j.src_stride + (rev ? j.in_w - c0 - 16u : c0);
      if ((reinterpret_cast<uintptr_t>(sp) & 15u) == 0u) {
        const uint4 v = *reinterpret_cast<const uint4*>(sp);
        *reinterpret_cast<uint4*>(d) = rev ? make_uint4(bswap32(v.w), bswap32(v.z), bswap32(v.y), bswap32(v.x)) : v;
        continue;
      }
    }
#pragma unroll 1
    for (uint32_t q = 0; q < 4u; ++q)
      if (c0 + 4u * q < j.cols) fx_quad(j, i, c0 + 4u * q, d + 4u * q);
  }
}

// 90 / 270 degree rotations are transposes: a 64x64 source tile is read row-wise (coalesced), parked in LDS and
// written row-wise in the rotated orientation, so neither side of the copy walks a column of the image in HBM
// (the plain gather above reaches 0.45 TB/s on a 4K frame, this one is limited by launch latency).
__global__ void __launch_bounds__(256) k_effect_rot(const FxJobs jobs) {
  const FxJob& j = jobs.job[blockIdx.z];
  __shared__ __attribute__((aligned(4))) uint8_t tile[64][68];
  const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;  // output tile origin (row, col)
  if (i0 >= (int)j.rows || j0 >= (int)j.cols) return;
  const bool r90 = j.op == FX_ROT90;
  // source tile origin: ROT90 out[i][j] = src[in_h-1-j][i];  ROT270 out[i][j] = src[j][in_w-1-i]
  const int row_base = r90 ? (int)j.in_h - 64 - j0 : j0;
  const int col_base = r90 ? i0 : (int)j.in_w - 64 - i0;
  const int t = threadIdx.x;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int ry = p * 16 + (t >> 4), cx = 4 * (t & 15);
    const int sr = row_base + ry;
    const int sc0 = col_base + cx;
    const uint8_t* sp = j.src + (size_t)(sr < 0 ? 0 : sr) * j.src_stride + sc0;
    if (sr >= 0 && sr < (int)j.in_h && sc0 >= 0 && sc0 + 3 < (int)j.in_w && (reinterpret_cast<uintptr_t>(sp) & 3u) == 0u) {
      *reinterpret_cast<uint32_t*>(&tile[ry][cx]) = *reinterpret_cast<const uint32_t*>(sp);   // (a row of the tile is 68 bytes: cx a multiple of 4)
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int sc = sc0 + k;
        uint8_t v = 0;
        if (sr >= 0 && sr < (int)j.in_h && sc >= 0 && sc < (int)j.in_w) v = j.src[(size_t)sr * j.src_stride + sc];
        tile[ry][cx + k] = v;
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int a = p * 16 + (t >> 4), b0 = 4 * (t & 15);  // output row a, columns b0..b0+3 of the tile
    const int oi = i0 + a;
    if (oi >= (int)j.rows) continue;
    uint8_t* d = j.dst + (size_t)oi * j.dst_stride + j0 + b0;
    uint8_t v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = r90 ? tile[63 - (b0 + k)][a] : tile[b0 + k][63 - a];
    if (j0 + b0 + 4 <= (int)j.cols && ((reinterpret_cast<uintptr_t>(d) & 3u) == 0u)) {
      *reinterpret_cast<uint32_t*>(d) = v[0] | (v[1] << 8) | (v[2] << 16) | ((uint32_t)v[3] << 24);
    } else {
      for (int k = 0; k < 4 && j0 + b0 + k < (int)j.cols; ++k) d[k] = v[k];
    }
  }
}

hipError_t launch_effect(const FxJobs& j, hipStream_t s) {
  uint32_t rows = 0, cols = 0;
  for (int k = 0; k < j.n; ++k) { rows = rows > j.job[k].rows ? rows : j.job[k].rows; cols = cols > j.job[k].cols ? cols : j.job[k].cols; }
  if (j.n == 0 || rows == 0 || cols == 0) return hipSuccess;
  if (j.job[0].op == FX_ROT90 || j.job[0].op == FX_ROT270) {
    hipLaunchKernelGGL(k_effect_rot, dim3((cols + 63u) / 64u, (rows + 63u) / 64u, (unsigned)j.n), dim3(256), 0, s, j);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(k_effect, dim3(((cols + 15u) / 16u + 255u) / 256u, rows < 65535u ? rows : 65535u, (unsigned)j.n), dim3(256), 0, s, j);
  return hipGetLastError();
}

// =================================================================================================
// diagnostics: evaluate one transfer function over an array (tests/test_gpu_transfer_exhaustive.py)
// =================================================================================================
__global__ void __launch_bounds__(256) k_eval_transfer(int fn, const float* in, float* out, size_t n, EvalConsts ec) {
  const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const float x = in[i];
  float y = 0.0f;
  switch (fn) {
    case 0: y = srgb_inv_oetf_guarded(x); break;
    case 1: y = hlg_inv_oetf_guarded(x); break;
    case 2: y = pq_inv_oetf_guarded(x); break;
    case 3: y = (float)encode_gain_guarded(x, ec.min_boost, ec.max_boost, ec.log2_min, ec.log2_max, ec.enc_scale,
                                           ec.enc_byte_min, ec.enc_byte_max); break;
    case 4: { float v[1] = {x}; hlg_oetf_guarded_n<1>(v); y = v[0]; break; }
    case 5: { float v[1] = {x}; pq_oetf_guarded_n<1>(v); y = v[0]; break; }
    case 6: y = exp2_to_float_guarded(x); break;
    case 16: y = (float)exp2((double)x); break;
    case 10: y = srgb_inv_oetf_exact(x); break;
    case 11: y = hlg_inv_oetf_exact(x); break;
    case 12: y = pq_inv_oetf_exact(x); break;
    case 13: y = (float)encode_gain(x, ec.min_boost, ec.max_boost, ec.log2_min, ec.log2_max); break;
    case 14: y = hlg_oetf_exact(x); break;
    case 15: y = pq_oetf_exact(x); break;
    case 24: y = hlg_oetf_fast(x); break;
    case 26: y = __builtin_amdgcn_exp2f(x); break;
    case 27: y = pq_oetf_est(x); break;
    case 25: y = pq_oetf_fast(x); break;
    case 20: y = srgb_inv_oetf_fast(x); break;
    case 21: y = hlg_inv_oetf_fast(x); break;
    case 22: y = pq_inv_oetf_fast(x); break;
    case 23: y = __builtin_amdgcn_logf(x); break;
    // the reference's LUT accessors over the device tables; 46: GainLUT(min, max, displayBoost = max).getGainFactor
    case 40: y = ec.lut[kLutSrgbInv + lut_index(x, kLutSrgbInvN)]; break;
    case 41: y = ec.lut[kLutHlgInv + lut_index(x, kLutHlgInvN)]; break;
    case 42: y = ec.lut[kLutPqInv + lut_index(x, kLutPqInvN)]; break;
    case 44: y = ec.lut[kLutHlg + lut_index(x, kLutHlgN)]; break;
    case 45: y = ec.lut[kLutPq + lut_index(x, kLutPqN)]; break;
    case 46: y = gain_lut_entry(lut_index(x, kGainLutN), ec.log2_min_d, ec.log2_max_d, 1.0f); break;
    case 47: y = round_half_up(x) == (uint32_t)((double)x + 0.5) ? 1.0f : 0.0f; break;   // (the accessors' one-instruction index: +0 <= x < 2^31 and -0)
    // FAST apply's line-segment tables, read from the device buffer with the kernel's own index arithmetic.  50 / 53 / 54: stage 1,
    // T(x) = srgbInvOetf(x)^g for g = 1, 1/2, m1.  51 / 52: the 10-bit code stage 2 yields for u in [0, 1] (HLG: u = sqrt(e), PQ:
    // u = e^m1); the kernel's fma rounds toward zero, which is the truncation of the exact sum formed here in double.
    case 50: case 53: case 54: {
      const uint32_t base = fn == 50 ? kTabS1Lin : fn == 53 ? kTabS1Hlg : kTabS1Pq;
      const float2 t = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(ec.lut + base) + (fn == 50 ? tab_offset(x) : pow_offset(x)));
      y = __builtin_fmaf(t.y, x, t.x); break; }
    case 51: case 52: {
      const float s = __builtin_fmaf(x, 2.0f, 2.0f);
      const float2 t = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(ec.lut + (fn == 51 ? kTabS2Hlg : kTabS2Pq)) + (s2_address(s, 0u) >> 5));
      const double v = -(((double)t.y * (double)s + (double)t.x) + 2.0) * 4194304.0;
      y = (float)(long long)v + (v < 0.0 ? -1000.0f : 0.0f); break; }   // (a negative sum would leave the binade: flagged)
    case 30: y = map_to_float_fast((uint32_t)x); break;
    case 31: y = map_to_float((uint32_t)x); break;
    // 1.0 where the lean f64 path was accepted by the rounding test, 0.0 where the exact path ran
    case 100: { const float t = div_const(x + 0.055f, 1.055f, 1.0f / 1.055f);
                float v1[1] = {x};
                srgb_inv_oetf_guarded_n<1>(v1);   // (kept for symmetry; the acceptance rate is measured below)
                const double xd = (double)t, z0 = (double)__builtin_amdgcn_exp2f(0.4f * __builtin_amdgcn_logf(t));
                const double X = xd * xd, z2 = z0 * z0, z4 = z2 * z2;
                const double yy = X * __builtin_fma(-__builtin_fma(z4, z0, -X), (double)(0.2f * __builtin_amdgcn_rcpf((float)z4)), z0);
                y = (x <= 0.04045f || ziv_safe(yy)) ? 1.0f : 0.0f; break; }
    case 101: { const float v = div_const(x - UHDR_HLG_C, UHDR_HLG_A, 1.0f / UHDR_HLG_A);
                y = (x <= 0.5f || ziv_safe((fast_exp2((double)v * 0x1.71547652b82fep+0) + (double)UHDR_HLG_B) * (1.0 / 12.0))) ? 1.0f : 0.0f; break; }
    default: break;
  }
  out[i] = y;
}

// The deterministic frame pair of SURVEY.md section 8(d), written straight into HBM: s <- s * 1664525 + 1013904223 (mod 2^32),
// draw = s >> 8, draws alternating between the P010 word and the 8-bit sample of element i.  Thread i jumps to state 2 i + 1 by
// composing the affine map with itself (binary exponentiation on (a, c)), so the elements are independent.
__global__ void __launch_bounds__(256) k_synth_lcg(uint16_t* p010, uint8_t* yuv, uint32_t n_luma, uint32_t n, uint32_t seed) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  uint32_t ar = 1u, cr = 0u, ap = 1664525u, cp = 1013904223u;
  for (uint32_t k = 2u * i + 1u; k != 0u; k >>= 1) {
    if (k & 1u) { ar *= ap; cr = cr * ap + cp; }
    cp = cp * ap + cp;
    ap *= ap;
  }
  const uint32_t s1 = ar * seed + cr, s2 = s1 * 1664525u + 1013904223u;
  // Both remainders by constant divisors (multiply-high forms).  With a divisor chosen at run time hipcc, knowing the draw to be
  // below 2^24, expands the division through v_rcp_iflag_f32 + one upward correction, which returns quotient + 1 where the true
  // quotient lies within 2^-23 below an integer (15172754 / 897 = 16914.9989 -> 16915: 18 wrong words in a 640x480 frame).
  const uint32_t d = s1 >> 8, r = i < n_luma ? d % 877u : d % 897u;
  p010[i] = (uint16_t)((64u + r) << 6);
  yuv[i] = (uint8_t)(s2 >> 8);
}
hipError_t launch_synth_lcg(uint16_t* p010, uint8_t* yuv, uint32_t n_luma, uint32_t n, uint32_t seed, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_synth_lcg, dim3((n + 255u) / 256u), dim3(256), 0, s, p010, yuv, n_luma, n, seed);
  return hipGetLastError();
}

hipError_t launch_eval_transfer(int fn, const float* in, float* out, size_t n, const EvalConsts& ec, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_eval_transfer, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, fn, in, out, n, ec);
  return hipGetLastError();
}

}  // namespace uhdr
