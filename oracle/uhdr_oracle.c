/*
 * uhdr_oracle.c -- TEST INFRASTRUCTURE ONLY (see uhdr_oracle.h).
 *
 * Plain-C99 restatement of the reference's gain-map pixel path.  Every function
 * cites the reference file:line it follows (paths relative to /root/reference).
 *
 * Arithmetic rules made explicit:
 *  - the reference calls unqualified pow/exp/log/log2/exp2/sqrt, which bind to
 *    glibc's DOUBLE versions; float operands are promoted, the result is rounded to
 *    float only at the assignment/return.  In C that is exactly what <math.h> gives.
 *  - float sub-expressions stay float (FLT_EVAL_METHOD == 0 on x86-64/SSE2).
 *  - build with -ffp-contract=off (x86-64 baseline has no FMA; keep it that way).
 *  - float -> integer conversions truncate.
 */
#include "uhdr_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

/* ------------------------------------------------------------------------------------------
 * Color helpers (gainmapmath.h:38-134)
 * ---------------------------------------------------------------------------------------- */
static orc_color c3(float a, float b, float c) { orc_color r = {a, b, c}; return r; }

/* gainmapmath.cpp:115-118 */
static float clampPixelFloat(float v) { return (v < 0.0f) ? 0.0f : (v > 1.0f) ? 1.0f : v; }

/* gainmapmath.cpp:121,177,208 luminance coefficients; :129,185,215 Cb/Cr */
static const float kSrgbR = 0.2126f, kSrgbG = 0.7152f, kSrgbB = 0.0722f;
static const float kSrgbCb = 1.8556f, kSrgbCr = 1.5748f;
static const float kP3R = 0.20949f, kP3G = 0.72160f, kP3B = 0.06891f;
static const float kP3YR = 0.299f, kP3YG = 0.587f, kP3YB = 0.114f;
static const float kP3Cb = 1.772f, kP3Cr = 1.402f;
static const float kBt2100R = 0.2627f, kBt2100G = 0.6780f, kBt2100B = 0.0593f;
static const float kBt2100Cb = 1.8814f, kBt2100Cr = 1.4746f;

/* gainmapmath.cpp:139-140,195-196,247-248 : float (B*Cb)/G, (R*Cr)/G */
static float gcb(int gamut) {
  switch (gamut) {
    case ORC_CG_BT709: return kSrgbB * kSrgbCb / kSrgbG;
    case ORC_CG_P3: return kP3YB * kP3Cb / kP3YG;
    default: return kBt2100B * kBt2100Cb / kBt2100G;
  }
}
static float gcr(int gamut) {
  switch (gamut) {
    case ORC_CG_BT709: return kSrgbR * kSrgbCr / kSrgbG;
    case ORC_CG_P3: return kP3YR * kP3Cr / kP3YG;
    default: return kBt2100R * kBt2100Cr / kBt2100G;
  }
}

/* gainmapmath.cpp:123,179,210 */
float orc_luminance(int gamut, orc_color e) {
  switch (gamut) {
    case ORC_CG_BT709: return kSrgbR * e.r + kSrgbG * e.g + kSrgbB * e.b;
    case ORC_CG_P3: return kP3R * e.r + kP3G * e.g + kP3B * e.b;
    default: return kBt2100R * e.r + kBt2100G * e.g + kBt2100B * e.b;
  }
}

/* gainmapmath.cpp:142-146,198-202,250-254 ; e = (y,u,v) */
orc_color orc_yuvToRgb(int gamut, orc_color e) {
  float cr, cb;
  switch (gamut) {
    case ORC_CG_BT709: cr = kSrgbCr; cb = kSrgbCb; break;
    case ORC_CG_P3: cr = kP3Cr; cb = kP3Cb; break;
    default: cr = kBt2100Cr; cb = kBt2100Cb; break;
  }
  float g_cb = gcb(gamut), g_cr = gcr(gamut);
  return c3(clampPixelFloat(e.r + cr * e.b), clampPixelFloat(e.r - g_cb * e.g - g_cr * e.b),
            clampPixelFloat(e.r + cb * e.g));
}

/* gainmapmath.cpp:131-134,187-190,217-220 */
orc_color orc_rgbToYuv(int gamut, orc_color e) {
  float y, cb, cr;
  switch (gamut) {
    case ORC_CG_BT709: y = kSrgbR * e.r + kSrgbG * e.g + kSrgbB * e.b; cb = kSrgbCb; cr = kSrgbCr; break;
    case ORC_CG_P3: y = kP3YR * e.r + kP3YG * e.g + kP3YB * e.b; cb = kP3Cb; cr = kP3Cr; break;
    default: y = kBt2100R * e.r + kBt2100G * e.g + kBt2100B * e.b; cb = kBt2100Cb; cr = kBt2100Cr; break;
  }
  return c3(y, (e.b - y) / cb, (e.r - y) / cr);
}

/* gainmapmath.cpp:149-155 */
float orc_srgbInvOetf(float e) {
  if (e <= 0.04045f) return e / 12.92f;
  return (float)pow((double)((e + 0.055f) / 1.055f), 2.4);
}

/* gainmapmath.cpp:257 (kHlgC is a double literal narrowed to float) */
static const float kHlgA = 0.17883277f, kHlgB = 0.28466892f, kHlgC = 0.55991073;

/* gainmapmath.cpp:259-265 */
float orc_hlgOetf(float e) {
  if (e <= 1.0f / 12.0f) return (float)sqrt((double)(3.0f * e));
  return (float)((double)kHlgA * log((double)(12.0f * e - kHlgB)) + (double)kHlgC);
}

/* gainmapmath.cpp:280-286 */
float orc_hlgInvOetf(float e) {
  if (e <= 0.5f) return (float)(pow((double)e, (double)2.0f) / (double)3.0f);
  return (float)((exp((double)((e - kHlgC) / kHlgA)) + (double)kHlgB) / (double)12.0f);
}

/* gainmapmath.cpp:305-307 */
static const float kPqM1 = 2610.0f / 16384.0f, kPqM2 = 2523.0f / 4096.0f * 128.0f;
static const float kPqC1 = 3424.0f / 4096.0f, kPqC2 = 2413.0f / 4096.0f * 32.0f,
                   kPqC3 = 2392.0f / 4096.0f * 32.0f;

/* gainmapmath.cpp:309-312 */
float orc_pqOetf(float e) {
  if (e <= 0.0f) return 0.0f;
  return (float)pow(((double)kPqC1 + (double)kPqC2 * pow((double)e, (double)kPqM1)) /
                        (1 + (double)kPqC3 * pow((double)e, (double)kPqM1)),
                    (double)kPqM2);
}

/* gainmapmath.cpp:327-338 */
static const float kPqInvA = 128.0f, kPqInvB = 107.0f, kPqInvC = 2413.0f, kPqInvD = 2392.0f,
                   kPqInvE = 6.2773946361f, kPqInvF = 0.0126833f;
float orc_pqInvOetf(float e) {
  if (e <= 0.0001f) return 0.0f;
  return (float)pow(((double)kPqInvA * pow((double)e, (double)kPqInvF) - (double)kPqInvB) /
                        ((double)kPqInvC - (double)kPqInvD * pow((double)e, (double)kPqInvF)),
                    (double)kPqInvE);
}

static orc_color map3(float (*f)(float), orc_color e) { return c3(f(e.r), f(e.g), f(e.b)); }

/* ------------------------------------------------------------------------------------------
 * LUT variants (gainmapmath.cpp:21-64 tables, :162-171,269-277,292-302,316-324,344-354 accessors;
 * GainLUT gainmapmath.h:149-182).  Dead in the fork's live path (its USE_*_LUT macros are only
 * defined in jpegr.cpp:33-38, after which ultrahdr.cpp never sees them) but exported and tested
 * (gainmapmath_test.cpp:808-939); upstream libultrahdr runs with them ON.
 * ---------------------------------------------------------------------------------------- */
#define ORC_SRGB_INV_N 1024u  /* gainmapmath.h:268-269 */
#define ORC_HLG_N 65536u      /* :329-330 */
#define ORC_HLG_INV_N 4096u   /* :342-343 */
#define ORC_PQ_N 65536u       /* :355-356 */
#define ORC_PQ_INV_N 4096u    /* :368-369 */
static float t_srgb_inv[ORC_SRGB_INV_N], t_hlg[ORC_HLG_N], t_hlg_inv[ORC_HLG_INV_N], t_pq[ORC_PQ_N],
    t_pq_inv[ORC_PQ_INV_N];
static pthread_once_t lut_once = PTHREAD_ONCE_INIT;
static void lut_fill(float* t, size_t n, float (*f)(float)) { /* gainmapmath.cpp:21-64 */
  for (size_t idx = 0; idx < n; idx++) {
    float value = (float)idx / (float)(n - 1);
    t[idx] = f(value);
  }
}
static void lut_build(void) {
  lut_fill(t_pq, ORC_PQ_N, orc_pqOetf);
  lut_fill(t_pq_inv, ORC_PQ_INV_N, orc_pqInvOetf);
  lut_fill(t_hlg, ORC_HLG_N, orc_hlgOetf);
  lut_fill(t_hlg_inv, ORC_HLG_INV_N, orc_hlgInvOetf);
  lut_fill(t_srgb_inv, ORC_SRGB_INV_N, orc_srgbInvOetf);
}
/* `uint32_t value = static_cast<uint32_t>(e * (N - 1) + 0.5); value = CLIP3(value, 0, N - 1);`:
 * e * size_t -> float product, + 0.5 (double literal) -> double sum, truncated.  A negative or
 * huge sum is undefined behaviour in C++; on x86-64 the conversion goes through the 64-bit
 * cvttsd2si and keeps the low 32 bits, which is what this restates (int64 -> uint32). */
static uint32_t lut_index(float e, uint32_t n) {
  double pos = (double)(e * (float)(n - 1)) + 0.5;
  uint32_t value;
  if (pos != pos || pos >= 9223372036854775808.0 || pos < -9223372036854775808.0) value = 0u; /* "integer indefinite" low word */
  else value = (uint32_t)(int64_t)pos;
  return value > n - 1 ? n - 1 : value; /* CLIP3 on an unsigned: only the upper bound can bite */
}
float orc_srgbInvOetfLUT(float e) { pthread_once(&lut_once, lut_build); return t_srgb_inv[lut_index(e, ORC_SRGB_INV_N)]; }
float orc_hlgOetfLUT(float e) { pthread_once(&lut_once, lut_build); return t_hlg[lut_index(e, ORC_HLG_N)]; }
float orc_hlgInvOetfLUT(float e) { pthread_once(&lut_once, lut_build); return t_hlg_inv[lut_index(e, ORC_HLG_INV_N)]; }
float orc_pqOetfLUT(float e) { pthread_once(&lut_once, lut_build); return t_pq[lut_index(e, ORC_PQ_N)]; }
float orc_pqInvOetfLUT(float e) { pthread_once(&lut_once, lut_build); return t_pq_inv[lut_index(e, ORC_PQ_INV_N)]; }
const float* orc_lut_table(int which, size_t* n) { /* 0 srgbInv 1 hlgInv 2 pqInv 4 hlg 5 pq */
  pthread_once(&lut_once, lut_build);
  switch (which) {
    case 0: *n = ORC_SRGB_INV_N; return t_srgb_inv;
    case 1: *n = ORC_HLG_INV_N; return t_hlg_inv;
    case 2: *n = ORC_PQ_INV_N; return t_pq_inv;
    case 4: *n = ORC_HLG_N; return t_hlg;
    case 5: *n = ORC_PQ_N; return t_pq;
    default: *n = 0; return NULL;
  }
}

/* GainLUT (gainmapmath.h:151-182).  `log2(float)` / `exp2(float)` there are unqualified calls from
 * namespace ultrahdr with only <cmath> included: they bind to the C double functions, so the
 * weighted sum runs in double and is rounded into the float `logBoost`; the product with
 * boostFactor is a float product.  with_display_boost == 0 is the one-argument constructor. */
void orc_gainLutBuild(float minBoost, float maxBoost, int with_display_boost, float displayBoost,
                      float* table) {
  float boostFactor = 1.0f;
  if (with_display_boost) boostFactor = displayBoost > 0 ? displayBoost / maxBoost : 1.0f;
  for (size_t idx = 0; idx < ORC_GAIN_LUT_N; idx++) {
    float value = (float)idx / (float)(ORC_GAIN_LUT_N - 1);
    float logBoost = (float)(log2((double)minBoost) * (double)(1.0f - value) +
                             log2((double)maxBoost) * (double)value);
    if (with_display_boost) table[idx] = (float)exp2((double)(logBoost * boostFactor));
    else table[idx] = (float)exp2((double)logBoost);
  }
}
float orc_gainLutFactor(const float* table, float gain) { /* getGainFactor, :173-178 */
  return table[lut_index(gain, ORC_GAIN_LUT_N)];
}
orc_color orc_applyGainLUT(orc_color e, float gain, const float* table) { /* gainmapmath.cpp:557-560 */
  float f = orc_gainLutFactor(table, gain);
  return c3(e.r * f, e.g * f, e.b * f);
}

/* gainmapmath.cpp:359-393 */
static orc_color mat3(const float m[9], orc_color e) {
  return c3(m[0] * e.r + m[1] * e.g + m[2] * e.b, m[3] * e.r + m[4] * e.g + m[5] * e.b,
            m[6] * e.r + m[7] * e.g + m[8] * e.b);
}
static const float kBt709ToP3[9] = {0.82254f, 0.17755f, 0.00006f, 0.03312f, 0.96684f, -0.00001f,
                                    0.01706f, 0.07240f, 0.91049f};
static const float kBt709ToBt2100[9] = {0.62740f, 0.32930f, 0.04332f, 0.06904f, 0.91958f,
                                        0.01138f, 0.01636f, 0.08799f, 0.89555f};
static const float kP3ToBt709[9] = {1.22482f, -0.22490f, -0.00007f, -0.04196f, 1.04199f,
                                    0.00001f, -0.01961f, -0.07865f, 1.09831f};
static const float kP3ToBt2100[9] = {0.75378f, 0.19862f, 0.04754f, 0.04576f, 0.94177f,
                                     0.01250f, -0.00121f, 0.01757f, 0.98359f};
static const float kBt2100ToBt709[9] = {1.66045f, -0.58764f, -0.07286f, -0.12445f, 1.13282f,
                                        -0.00837f, -0.01811f, -0.10057f, 1.11878f};
static const float kBt2100ToP3[9] = {1.34369f, -0.28223f, -0.06135f, -0.06533f, 1.07580f,
                                     -0.01051f, 0.00283f, -0.01957f, 1.01679f};

/* gainmapmath.cpp:397-440 getHdrConversionFn: returns matrix, NULL for identity; *is_null=1 when the
 * reference would return nullptr (an UNSPECIFIED gamut) */
static const float* hdr_conv_matrix(int sdr_gamut, int hdr_gamut, int* is_null) {
  *is_null = 0;
  if (sdr_gamut < ORC_CG_BT709 || sdr_gamut > ORC_CG_BT2100 || hdr_gamut < ORC_CG_BT709 ||
      hdr_gamut > ORC_CG_BT2100) {
    *is_null = 1;
    return NULL;
  }
  if (sdr_gamut == hdr_gamut) return NULL; /* identityConversion */
  switch (sdr_gamut) {
    case ORC_CG_BT709: return hdr_gamut == ORC_CG_P3 ? kP3ToBt709 : kBt2100ToBt709;
    case ORC_CG_P3: return hdr_gamut == ORC_CG_BT709 ? kBt709ToP3 : kBt2100ToP3;
    default: return hdr_gamut == ORC_CG_BT709 ? kBt709ToBt2100 : kP3ToBt2100;
  }
}
orc_color orc_gamutConv(int sdr_gamut, int hdr_gamut, orc_color e, int* is_null) {
  int n;
  const float* m = hdr_conv_matrix(sdr_gamut, hdr_gamut, &n);
  if (is_null) *is_null = n;
  return m ? mat3(m, e) : e;
}

/* gainmapmath.cpp:447-481 (the 1.0f*y and 0.0f*y terms are kept) */
static const float kYuv709To601[9] = {1.0f, 0.101579f, 0.196076f, 0.0f, 0.989854f, -0.110653f,
                                      0.0f, -0.072453f, 0.983398f};
static const float kYuv709To2100[9] = {1.0f, -0.016969f, 0.096312f, 0.0f, 0.995306f, -0.051192f,
                                       0.0f, 0.011507f, 1.002637f};
static const float kYuv601To709[9] = {1.0f, -0.118188f, -0.212685f, 0.0f, 1.018640f, 0.114618f,
                                      0.0f, 0.075049f, 1.025327f};
static const float kYuv601To2100[9] = {1.0f, -0.128245f, -0.115879f, 0.0f, 1.010016f, 0.061592f,
                                       0.0f, 0.086969f, 1.029350f};
static const float kYuv2100To709[9] = {1.0f, 0.018149f, -0.095132f, 0.0f, 1.004123f, 0.051267f,
                                       0.0f, -0.011524f, 0.996782f};
static const float kYuv2100To601[9] = {1.0f, 0.117887f, 0.105521f, 0.0f, 0.995211f, -0.059549f,
                                       0.0f, -0.084085f, 0.976518f};
/* jpegr.cpp:1142-1192 selection table; NULL = no conversion */
static const float* yuv_conv_matrix(int src, int dst) {
  if (src == dst) return NULL;
  switch (src) {
    case ORC_CG_BT709: return dst == ORC_CG_P3 ? kYuv709To601 : kYuv709To2100;
    case ORC_CG_P3: return dst == ORC_CG_BT709 ? kYuv601To709 : kYuv601To2100;
    default: return dst == ORC_CG_BT709 ? kYuv2100To709 : kYuv2100To601;
  }
}
orc_color orc_yuvToYuv(int src, int dst, orc_color e) {
  const float* m = yuv_conv_matrix(src, dst);
  return m ? mat3(m, e) : e;
}

/* gainmapmath.cpp:529-541 */
uint8_t orc_encodeGain(float y_sdr, float y_hdr, float minBoost, float maxBoost, float log2Min,
                       float log2Max) {
  float gain = 1.0f;
  if (y_sdr > 0.0f) gain = y_hdr / y_sdr;
  if (gain < minBoost) gain = minBoost;
  if (gain > maxBoost) gain = maxBoost;
  return (uint8_t)((log2((double)gain) - (double)log2Min) / (double)(log2Max - log2Min) *
                   (double)255.0f);
}
/* gainmapmath.cpp:524-527 */
uint8_t orc_encodeGain3(float y_sdr, float y_hdr, float minBoost, float maxBoost) {
  return orc_encodeGain(y_sdr, y_hdr, minBoost, maxBoost, (float)log2((double)minBoost),
                        (float)log2((double)maxBoost));
}

/* gainmapmath.cpp:543-548 */
orc_color orc_applyGain3(orc_color e, float gain, float minBoost, float maxBoost) {
  float logBoost = (float)(log2((double)minBoost) * (double)(1.0f - gain) +
                           log2((double)maxBoost) * (double)gain);
  float gainFactor = (float)exp2((double)logBoost);
  return c3(e.r * gainFactor, e.g * gainFactor, e.b * gainFactor);
}
/* gainmapmath.cpp:550-555 */
orc_color orc_applyGain4(orc_color e, float gain, float minBoost, float maxBoost,
                         float displayBoost) {
  float logBoost = (float)(log2((double)minBoost) * (double)(1.0f - gain) +
                           log2((double)maxBoost) * (double)gain);
  float gainFactor = (float)exp2((double)(logBoost * displayBoost / maxBoost));
  return c3(e.r * gainFactor, e.g * gainFactor, e.b * gainFactor);
}

/* gainmapmath.cpp:562-581 */
orc_color orc_getYuv420Pixel(const orc_image* img, size_t x, size_t y) {
  const uint8_t* luma = (const uint8_t*)img->data;
  const uint8_t* chroma = (const uint8_t*)img->chroma_data;
  size_t offset_cr = img->chroma_stride * (img->height / 2);
  size_t yi = x + y * img->luma_stride;
  size_t ci = x / 2 + (y / 2) * img->chroma_stride;
  uint8_t yv = luma[yi], uv = chroma[ci], vv = chroma[offset_cr + ci];
  return c3((float)yv * (1 / 255.0f), (float)(uv - 128) * (1 / 255.0f),
            (float)(vv - 128) * (1 / 255.0f));
}

/* gainmapmath.cpp:583-601 */
orc_color orc_getP010Pixel(const orc_image* img, size_t x, size_t y) {
  const uint16_t* luma = (const uint16_t*)img->data;
  size_t luma_stride = img->luma_stride == 0 ? img->width : img->luma_stride;
  const uint16_t* chroma = (const uint16_t*)img->chroma_data;
  size_t yi = y * luma_stride + x;
  size_t ui = (y >> 1) * img->chroma_stride + (x & ~(size_t)0x1);
  size_t vi = ui + 1;
  uint16_t yv = luma[yi] >> 6, uv = chroma[ui] >> 6, vv = chroma[vi] >> 6;
  return c3((float)(yv - 64) * (1 / 876.0f), (float)(uv - 64) * (1 / 896.0f) - 0.5f,
            (float)(vv - 64) * (1 / 896.0f) - 0.5f);
}

/* gainmapmath.cpp:605-615 : dy outer, dx inner, sequential float accumulation */
typedef orc_color (*get_pixel_fn)(const orc_image*, size_t, size_t);
static orc_color samplePixels(const orc_image* img, size_t s, size_t x, size_t y, get_pixel_fn f) {
  orc_color e = {0.0f, 0.0f, 0.0f};
  for (size_t dy = 0; dy < s; ++dy)
    for (size_t dx = 0; dx < s; ++dx) {
      orc_color p = f(img, x * s + dx, y * s + dy);
      e.r += p.r; e.g += p.g; e.b += p.b;
    }
  float d = (float)(s * s);
  return c3(e.r / d, e.g / d, e.b / d);
}
orc_color orc_sampleYuv420(const orc_image* img, size_t s, size_t x, size_t y) {
  return samplePixels(img, s, x, y, orc_getYuv420Pixel);
}
orc_color orc_sampleP010(const orc_image* img, size_t s, size_t x, size_t y) {
  return samplePixels(img, s, x, y, orc_getP010Pixel);
}

/* gainmapmath.cpp:69-110 (+ gainmapmath.h:184-195) */
static float euclideanDistance(float x1, float x2, float y1, float y2) {
  return (float)sqrt((double)(((y2 - y1) * (y2 - y1)) + (x2 - x1) * (x2 - x1)));
}
void orc_fillShepardsIDW(float* weights, int scale, int incR, int incB) {
  for (int y = 0; y < scale; y++) {
    for (int x = 0; x < scale; x++) {
      float pos_x = ((float)x) / scale;
      float pos_y = ((float)y) / scale;
      int curr_x = (int)floor((double)pos_x);
      int curr_y = (int)floor((double)pos_y);
      int next_x = curr_x + incR;
      int next_y = curr_y + incB;
      float e1_distance = euclideanDistance(pos_x, curr_x, pos_y, curr_y);
      int index = y * scale * 4 + x * 4;
      if (e1_distance == 0) {
        weights[index++] = 1.f; weights[index++] = 0.f; weights[index++] = 0.f; weights[index++] = 0.f;
      } else {
        float e1_weight = 1.f / e1_distance;
        float e2_distance = euclideanDistance(pos_x, curr_x, pos_y, next_y);
        float e2_weight = 1.f / e2_distance;
        float e3_distance = euclideanDistance(pos_x, next_x, pos_y, curr_y);
        float e3_weight = 1.f / e3_distance;
        float e4_distance = euclideanDistance(pos_x, next_x, pos_y, next_y);
        float e4_weight = 1.f / e4_distance;
        float total_weight = e1_weight + e2_weight + e3_weight + e4_weight;
        weights[index++] = e1_weight / total_weight;
        weights[index++] = e2_weight / total_weight;
        weights[index++] = e3_weight / total_weight;
        weights[index++] = e4_weight / total_weight;
      }
    }
  }
}

typedef struct { int scale; float *w, *wnr, *wnb, *wc; } idw_tables;
static void idw_init(idw_tables* t, int scale) {
  size_t n = (size_t)scale * scale * 4;
  t->scale = scale;
  t->w = (float*)malloc(4 * n * sizeof(float));
  t->wnr = t->w + n; t->wnb = t->wnr + n; t->wc = t->wnb + n;
  orc_fillShepardsIDW(t->w, scale, 1, 1);
  orc_fillShepardsIDW(t->wnr, scale, 0, 1);
  orc_fillShepardsIDW(t->wnb, scale, 1, 0);
  orc_fillShepardsIDW(t->wc, scale, 0, 0);
}
static void idw_free(idw_tables* t) { free(t->w); }

static float mapUintToFloat(uint8_t v) { return (float)v / 255.0f; } /* gainmapmath.cpp:632 */
static size_t zmin(size_t a, size_t b) { return a < b ? a : b; }

/* gainmapmath.cpp:686-720 ; NOTE indexes with map->width, not the stride */
static float sampleMapIdw(const orc_image* map, size_t s, size_t x, size_t y, const idw_tables* t) {
  size_t xl = x / s, xu = xl + 1, yl = y / s, yu = yl + 1;
  xl = zmin(xl, map->width - 1); xu = zmin(xu, map->width - 1);
  yl = zmin(yl, map->height - 1); yu = zmin(yu, map->height - 1);
  const uint8_t* d = (const uint8_t*)map->data;
  float e1 = mapUintToFloat(d[xl + yl * map->width]);
  float e2 = mapUintToFloat(d[xl + yu * map->width]);
  float e3 = mapUintToFloat(d[xu + yl * map->width]);
  float e4 = mapUintToFloat(d[xu + yu * map->width]);
  int ox = (int)(x % s), oy = (int)(y % s);
  const float* w = t->w;
  if (xl == xu && yl == yu) w = t->wc;
  else if (xl == xu) w = t->wnr;
  else if (yl == yu) w = t->wnb;
  w += oy * s * 4 + ox * 4;
  return e1 * w[0] + e2 * w[1] + e3 * w[2] + e4 * w[3];
}
float orc_sampleMapIdw(const orc_image* map, size_t s, size_t x, size_t y) {
  idw_tables t; idw_init(&t, (int)s);
  float r = sampleMapIdw(map, s, x, y, &t);
  idw_free(&t);
  return r;
}

/* gainmapmath.cpp:628-684 (float-scale overload; dead in apply, kept for the reference's
 * table==float test; preserves the `e4_dist == 0 -> return e2` quirk at :674) */
static size_t zclamp(size_t v, size_t lo, size_t hi) { return v < lo ? lo : (hi < v ? hi : v); }
static float pythDistance(float xd, float yd) {
  return (float)sqrt(pow((double)xd, (double)2.0f) + pow((double)yd, (double)2.0f));
}
float orc_sampleMapFloat(const orc_image* map, float s, size_t x, size_t y) {
  float x_map = (float)x / s, y_map = (float)y / s;
  size_t xl = (size_t)floor((double)x_map), xu = xl + 1;
  size_t yl = (size_t)floor((double)y_map), yu = yl + 1;
  xl = zclamp(xl, 0, map->width - 1); xu = zclamp(xu, 0, map->width - 1);
  yl = zclamp(yl, 0, map->height - 1); yu = zclamp(yu, 0, map->height - 1);
  const uint8_t* d = (const uint8_t*)map->data;
  float e1 = mapUintToFloat(d[xl + yl * map->width]);
  float e1_dist = pythDistance(x_map - (float)xl, y_map - (float)yl);
  if (e1_dist == 0.0f) return e1;
  float e2 = mapUintToFloat(d[xl + yu * map->width]);
  float e2_dist = pythDistance(x_map - (float)xl, y_map - (float)yu);
  if (e2_dist == 0.0f) return e2;
  float e3 = mapUintToFloat(d[xu + yl * map->width]);
  float e3_dist = pythDistance(x_map - (float)xu, y_map - (float)yl);
  if (e3_dist == 0.0f) return e3;
  float e4 = mapUintToFloat(d[xu + yu * map->width]);
  float e4_dist = pythDistance(x_map - (float)xu, y_map - (float)yu);
  if (e4_dist == 0.0f) return e2;
  float w1 = 1.0f / e1_dist, w2 = 1.0f / e2_dist, w3 = 1.0f / e3_dist, w4 = 1.0f / e4_dist;
  float tw = w1 + w2 + w3 + w4;
  return e1 * (w1 / tw) + e2 * (w2 / tw) + e3 * (w3 / tw) + e4 * (w4 / tw);
}

/* gainmapmath.cpp:722-727 */
uint32_t orc_colorToRgba1010102(orc_color e) {
  return (0x3ff & (uint32_t)(e.r * 1023.0f)) | ((0x3ff & (uint32_t)(e.g * 1023.0f)) << 10) |
         ((0x3ff & (uint32_t)(e.b * 1023.0f)) << 20) | (0x3u << 30);
}

/* gainmapmath.h:136-147 */
uint16_t orc_floatToHalf(float f) {
  uint32_t bits;
  memcpy(&bits, &f, 4);
  const uint32_t b = bits + 0x00001000;
  const int32_t e = (b & 0x7F800000) >> 23;
  const uint32_t m = b & 0x007FFFFF;
  return (uint16_t)((b & 0x80000000) >> 16 | (e > 112) * ((((e - 112) << 10) & 0x7C00) | m >> 13) |
                    ((e < 113) & (e > 101)) * ((((0x007FF000 + m) >> (125 - e)) + 1) >> 1) |
                    (e > 143) * 0x7FFF);
}
/* gainmapmath.cpp:729-732 */
uint64_t orc_colorToRgbaF16(orc_color e) {
  return (uint64_t)orc_floatToHalf(e.r) | (((uint64_t)orc_floatToHalf(e.g)) << 16) |
         (((uint64_t)orc_floatToHalf(e.b)) << 32) | (((uint64_t)orc_floatToHalf(1.0f)) << 48);
}

/* gainmapmath.cpp:483-520 */
#define CLIP3(x, lo, hi) ((x) < (lo)) ? (lo) : ((x) > (hi)) ? (hi) : (x)
static void transformYuv420(orc_image* img, size_t xc, size_t yc, const float* m) {
  orc_color p1 = orc_getYuv420Pixel(img, xc * 2, yc * 2);
  orc_color p2 = orc_getYuv420Pixel(img, xc * 2 + 1, yc * 2);
  orc_color p3 = orc_getYuv420Pixel(img, xc * 2, yc * 2 + 1);
  orc_color p4 = orc_getYuv420Pixel(img, xc * 2 + 1, yc * 2 + 1);
  if (m) { p1 = mat3(m, p1); p2 = mat3(m, p2); p3 = mat3(m, p3); p4 = mat3(m, p4); }
  /* (yuv1 + yuv2 + yuv3 + yuv4) / 4.0f, left to right */
  float nu = (((p1.g + p2.g) + p3.g) + p4.g) / 4.0f;
  float nv = (((p1.b + p2.b) + p3.b) + p4.b) / 4.0f;
  uint8_t* Y = (uint8_t*)img->data;
  uint8_t* C = (uint8_t*)img->chroma_data;
  size_t i1 = xc * 2 + yc * 2 * img->luma_stride, i2 = (xc * 2 + 1) + yc * 2 * img->luma_stride;
  size_t i3 = xc * 2 + (yc * 2 + 1) * img->luma_stride,
         i4 = (xc * 2 + 1) + (yc * 2 + 1) * img->luma_stride;
  size_t pixel_count = img->chroma_stride * img->height / 2;
  size_t iuv = xc + yc * img->chroma_stride;
  Y[i1] = (uint8_t)(CLIP3((p1.r * 255.0f + 0.5f), 0, 255));
  Y[i2] = (uint8_t)(CLIP3((p2.r * 255.0f + 0.5f), 0, 255));
  Y[i3] = (uint8_t)(CLIP3((p3.r * 255.0f + 0.5f), 0, 255));
  Y[i4] = (uint8_t)(CLIP3((p4.r * 255.0f + 0.5f), 0, 255));
  C[iuv] = (uint8_t)(CLIP3((nu * 255.0f + 128.0f + 0.5f), 0, 255));
  C[pixel_count + iuv] = (uint8_t)(CLIP3((nv * 255.0f + 128.0f + 0.5f), 0, 255));
}
void orc_transformYuv420(orc_image* img, size_t xc, size_t yc, int src, int dst) {
  transformYuv420(img, xc, yc, yuv_conv_matrix(src, dst));
}

/* jpegr.cpp:1132-1206 */
int orc_convertYuv(orc_image* image, int src, int dst) {
  if (image == NULL) return ORC_ERR_BAD_PTR;
  if (src == ORC_CG_UNSPECIFIED || dst == ORC_CG_UNSPECIFIED) return ORC_ERR_INVALID_COLORGAMUT;
  if (src < ORC_CG_BT709 || src > ORC_CG_BT2100 || dst < ORC_CG_BT709 || dst > ORC_CG_BT2100)
    return ORC_ERR_INVALID_COLORGAMUT;
  if (src == dst) return ORC_OK;
  const float* m = yuv_conv_matrix(src, dst);
  for (size_t y = 0; y < image->height / 2; ++y)
    for (size_t x = 0; x < image->width / 2; ++x) transformYuv420(image, x, y, m);
  return ORC_OK;
}

/* ------------------------------------------------------------------------------------------
 * Row-band threading (replaces JobQueue, ultrahdr.cpp:131-183; results do not depend on it)
 * ---------------------------------------------------------------------------------------- */
static int ref_thread_count(void) { /* ultrahdr.cpp:42-59,304 */
  long n = sysconf(_SC_NPROCESSORS_ONLN);
  if (n <= 0) n = 1;
  return n < 4 ? (int)n : 4;
}
typedef void (*band_fn)(void* ctx, size_t row0, size_t row1, int tid);
typedef struct { band_fn fn; void* ctx; size_t rows, step; int tid; size_t* next; pthread_mutex_t* mu; } band_job;
static void* band_worker(void* p) {
  band_job* j = (band_job*)p;
  for (;;) {
    pthread_mutex_lock(j->mu);
    size_t r0 = *j->next;
    size_t r1 = r0 + j->step < j->rows ? r0 + j->step : j->rows;
    *j->next = r1;
    pthread_mutex_unlock(j->mu);
    if (r0 >= j->rows) break;
    j->fn(j->ctx, r0, r1, j->tid);
  }
  return NULL;
}
#define ORC_MAX_THREADS 256
static void run_bands(band_fn fn, void* ctx, size_t rows, size_t step, int threads) {
  if (threads <= 0) threads = ref_thread_count();
  if (threads > ORC_MAX_THREADS) threads = ORC_MAX_THREADS;
  if (threads == 1) { fn(ctx, 0, rows, 0); return; }
  pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
  size_t next = 0;
  pthread_t th[ORC_MAX_THREADS];
  band_job jobs[ORC_MAX_THREADS];
  for (int t = 0; t < threads; ++t) {
    band_job j = {fn, ctx, rows, step, t, &next, &mu};
    jobs[t] = j;
  }
  for (int t = 1; t < threads; ++t) pthread_create(&th[t], NULL, band_worker, &jobs[t]);
  band_worker(&jobs[0]);
  for (int t = 1; t < threads; ++t) pthread_join(th[t], NULL);
}

/* ------------------------------------------------------------------------------------------
 * generateGainMap (ultrahdr.cpp:185-358)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  const orc_image *yuv, *p010;
  uint8_t* map;
  size_t map_w;
  int hdr_tf, lum_gamut, sdr_yuv_gamut, hdr_yuv_gamut, lut;
  const float* gamut_m;
  float hdr_white_nits, minBoost, maxBoost, log2Min, log2Max;
  float tmin[ORC_MAX_THREADS], tmax[ORC_MAX_THREADS];
} gen_ctx;

static void gen_band(void* p, size_t r0, size_t r1, int tid) {
  gen_ctx* c = (gen_ctx*)p;
  float gmin = c->tmin[tid], gmax = c->tmax[tid];
  for (size_t y = r0; y < r1; ++y) {
    for (size_t x = 0; x < c->map_w; ++x) { /* ultrahdr.cpp:315-335 */
      orc_color sdr_yuv_gamma = orc_sampleYuv420(c->yuv, 4, x, y);
      orc_color sdr_rgb_gamma = orc_yuvToRgb(c->sdr_yuv_gamut, sdr_yuv_gamma);
      orc_color sdr_rgb = map3(c->lut ? orc_srgbInvOetfLUT : orc_srgbInvOetf, sdr_rgb_gamma); /* :319-323 */
      float sdr_y_nits = orc_luminance(c->lum_gamut, sdr_rgb) * 203.0f;

      orc_color hdr_yuv_gamma = orc_sampleP010(c->p010, 4, x, y);
      orc_color hdr_rgb_gamma = orc_yuvToRgb(c->hdr_yuv_gamut, hdr_yuv_gamma);
      orc_color hdr_rgb = hdr_rgb_gamma;
      if (c->hdr_tf == ORC_TF_HLG) hdr_rgb = map3(c->lut ? orc_hlgInvOetfLUT : orc_hlgInvOetf, hdr_rgb_gamma); /* :230-234 */
      else if (c->hdr_tf == ORC_TF_PQ) hdr_rgb = map3(c->lut ? orc_pqInvOetfLUT : orc_pqInvOetf, hdr_rgb_gamma); /* :238-242 */
      if (c->gamut_m) hdr_rgb = mat3(c->gamut_m, hdr_rgb);
      float hdr_y_nits = orc_luminance(c->lum_gamut, hdr_rgb) * c->hdr_white_nits;

      c->map[x + y * c->map_w] =
          orc_encodeGain(sdr_y_nits, hdr_y_nits, c->minBoost, c->maxBoost, c->log2Min, c->log2Max);

      /* extra statistic (no reference counterpart): unclamped gain of gainmapmath.cpp:531-534 */
      float gain = 1.0f;
      if (sdr_y_nits > 0.0f) gain = hdr_y_nits / sdr_y_nits;
      if (gain < gmin) gmin = gain;
      if (gain > gmax) gmax = gain;
    }
  }
  c->tmin[tid] = gmin; c->tmax[tid] = gmax;
}

static int generate_impl(const orc_image* yuv, const orc_image* p010, int hdr_tf,
                         orc_metadata* md, uint8_t* map_out, int sdr_is_601, int threads,
                         float* minmax_out, int lut) {
  /* ultrahdr.cpp:189-202 */
  if (yuv == NULL || p010 == NULL || md == NULL || map_out == NULL || yuv->data == NULL ||
      yuv->chroma_data == NULL || p010->data == NULL || p010->chroma_data == NULL)
    return ORC_ERR_BAD_PTR;
  if (yuv->width != p010->width || yuv->height != p010->height) return ORC_ERR_RESOLUTION_MISMATCH;
  if (yuv->colorGamut == ORC_CG_UNSPECIFIED || p010->colorGamut == ORC_CG_UNSPECIFIED)
    return ORC_ERR_INVALID_COLORGAMUT;

  gen_ctx c;
  c.yuv = yuv; c.p010 = p010; c.map = map_out;
  c.map_w = yuv->width / 4;
  size_t map_h = yuv->height / 4;
  c.hdr_tf = hdr_tf;
  c.lut = lut;
  switch (hdr_tf) { /* ultrahdr.cpp:222-248 */
    case ORC_TF_LINEAR: c.hdr_white_nits = 1000.0f; break;
    case ORC_TF_HLG: c.hdr_white_nits = 1000.0f; break;
    case ORC_TF_PQ: c.hdr_white_nits = 10000.0f; break;
    default: return ORC_ERR_INVALID_TRANS_FUNC;
  }
  /* ultrahdr.cpp:250-260 */
  md->version_ok = 1;
  md->maxContentBoost = c.hdr_white_nits / 203.0f;
  md->minContentBoost = 1.0f;
  md->gamma = 1.0f; md->offsetSdr = 0.0f; md->offsetHdr = 0.0f;
  md->hdrCapacityMin = 1.0f;
  md->hdrCapacityMax = md->maxContentBoost;
  c.minBoost = md->minContentBoost; c.maxBoost = md->maxContentBoost;
  c.log2Min = (float)log2((double)md->minContentBoost);
  c.log2Max = (float)log2((double)md->maxContentBoost);
  int is_null;
  c.gamut_m = hdr_conv_matrix(yuv->colorGamut, p010->colorGamut, &is_null); /* :262-263 */
  if (yuv->colorGamut < ORC_CG_BT709 || yuv->colorGamut > ORC_CG_BT2100) /* :267-283 */
    return ORC_ERR_INVALID_COLORGAMUT;
  c.lum_gamut = yuv->colorGamut;
  c.sdr_yuv_gamut = sdr_is_601 ? ORC_CG_P3 : yuv->colorGamut; /* :284-286 */
  if (p010->colorGamut < ORC_CG_BT709 || p010->colorGamut > ORC_CG_BT2100) /* :289-302 */
    return ORC_ERR_INVALID_COLORGAMUT;
  c.hdr_yuv_gamut = p010->colorGamut;
  for (int t = 0; t < ORC_MAX_THREADS; ++t) { c.tmin[t] = INFINITY; c.tmax[t] = -INFINITY; }

  run_bands(gen_band, &c, map_h, 4, threads); /* 16 image rows = 4 map rows per job, :346 */
  if (minmax_out) {
    float gmin = INFINITY, gmax = -INFINITY;
    for (int t = 0; t < ORC_MAX_THREADS; ++t) {
      if (c.tmin[t] < gmin) gmin = c.tmin[t];
      if (c.tmax[t] > gmax) gmax = c.tmax[t];
    }
    minmax_out[0] = gmin; minmax_out[1] = gmax;
  }
  return ORC_OK;
}
int orc_generateGainMapStats(const orc_image* yuv, const orc_image* p010, int hdr_tf,
                             orc_metadata* md, uint8_t* map_out, int sdr_is_601, int threads,
                             float* minmax_out) {
  return generate_impl(yuv, p010, hdr_tf, md, map_out, sdr_is_601, threads, minmax_out, 0);
}
int orc_generateGainMap(const orc_image* yuv, const orc_image* p010, int hdr_tf, orc_metadata* md,
                        uint8_t* map_out, int sdr_is_601, int threads) {
  return generate_impl(yuv, p010, hdr_tf, md, map_out, sdr_is_601, threads, NULL, 0);
}
/* the same loop as upstream builds it: USE_SRGB/HLG/PQ_INVOETF_LUT = 1 (ultrahdr.cpp:230,238,319) */
int orc_generateGainMapLUT(const orc_image* yuv, const orc_image* p010, int hdr_tf, orc_metadata* md,
                           uint8_t* map_out, int sdr_is_601, int threads) {
  pthread_once(&lut_once, lut_build);
  return generate_impl(yuv, p010, hdr_tf, md, map_out, sdr_is_601, threads, NULL, 1);
}

/* ------------------------------------------------------------------------------------------
 * applyGainMap (ultrahdr.cpp:360-515)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  const orc_image *yuv, *map;
  orc_image* dest;
  const orc_metadata* md;
  idw_tables idw;
  size_t scale;
  int fmt, lut;
  float display_boost;
  float gain_lut[ORC_GAIN_LUT_N];
} app_ctx;

static void app_band(void* p, size_t r0, size_t r1, int tid) {
  (void)tid;
  app_ctx* c = (app_ctx*)p;
  size_t width = c->yuv->width, height = c->yuv->height;
  for (size_t y = r0; y < r1; ++y) {
    for (size_t x = 0; x < width; ++x) { /* ultrahdr.cpp:428-494 */
      orc_color yuv_gamma_sdr = orc_getYuv420Pixel(c->yuv, x, y);
      orc_color rgb_gamma_sdr = orc_yuvToRgb(ORC_CG_P3, yuv_gamma_sdr); /* always BT.601, :431 */
      orc_color rgb_sdr = map3(c->lut ? orc_srgbInvOetfLUT : orc_srgbInvOetf, rgb_gamma_sdr); /* :433-437 */
      float gain = sampleMapIdw(c->map, c->scale, x, y, &c->idw);
      orc_color rgb_hdr;
      if (c->lut) rgb_hdr = orc_applyGainLUT(rgb_sdr, gain, c->gain_lut); /* :446-447 */
      else rgb_hdr = orc_applyGain4(rgb_sdr, gain, c->md->minContentBoost, c->md->maxContentBoost,
                                    c->display_boost);
      rgb_hdr = c3(rgb_hdr.r / c->display_boost, rgb_hdr.g / c->display_boost,
                   rgb_hdr.b / c->display_boost);
      size_t idx = x + y * width;
      switch (c->fmt) {
        case ORC_OUT_HDR_LINEAR:
          ((uint64_t*)c->dest->data)[idx] = orc_colorToRgbaF16(rgb_hdr);
          break;
        case ORC_OUT_HDR_LINEAR_RGB_10BIT: {
          uint16_t r = 0x3ff & (uint32_t)(rgb_hdr.r * 1023.0f);
          uint16_t g = 0x3ff & (uint32_t)(rgb_hdr.g * 1023.0f);
          uint16_t b = 0x3ff & (uint32_t)(rgb_hdr.b * 1023.0f);
          ((uint16_t*)c->dest->data)[idx] = r;
          ((uint16_t*)c->dest->data)[width * height + idx] = g;
          ((uint16_t*)c->dest->data)[width * height * 2 + idx] = b;
          break;
        }
        case ORC_OUT_HDR_HLG:
          ((uint32_t*)c->dest->data)[idx] =
              orc_colorToRgba1010102(map3(c->lut ? orc_hlgOetfLUT : orc_hlgOetf, rgb_hdr)); /* :470-474 */
          break;
        case ORC_OUT_HDR_PQ:
          ((uint32_t*)c->dest->data)[idx] =
              orc_colorToRgba1010102(map3(c->lut ? orc_pqOetfLUT : orc_pqOetf, rgb_hdr)); /* :481-485 */
          break;
        default: break; /* nothing written, :491-493 */
      }
    }
  }
}

static int apply_impl(const orc_image* yuv, const orc_image* map, const orc_metadata* md, int fmt,
                      float max_display_boost, orc_image* dest, int threads, int lut) {
  /* ultrahdr.cpp:364-406 */
  if (yuv == NULL || map == NULL || md == NULL || dest == NULL || yuv->data == NULL ||
      yuv->chroma_data == NULL || map->data == NULL)
    return ORC_ERR_BAD_PTR;
  if (!md->version_ok) return ORC_ERR_BAD_METADATA;
  if (md->gamma != 1.0f) return ORC_ERR_BAD_METADATA;
  if (md->offsetSdr != 0.0f || md->offsetHdr != 0.0f) return ORC_ERR_BAD_METADATA;
  if (md->hdrCapacityMin != md->minContentBoost || md->hdrCapacityMax != md->maxContentBoost)
    return ORC_ERR_BAD_METADATA;
  if (yuv->width % map->width != 0 || yuv->height % map->height != 0)
    return ORC_ERR_UNSUPPORTED_MAP_SCALE_FACTOR;
  if (yuv->width * map->height != yuv->height * map->width)
    return ORC_ERR_UNSUPPORTED_MAP_SCALE_FACTOR;
  app_ctx c;
  c.yuv = yuv; c.map = map; c.dest = dest; c.md = md; c.fmt = fmt;
  c.scale = yuv->width / map->width; /* :409 */
  dest->width = yuv->width; dest->height = yuv->height; dest->colorGamut = yuv->colorGamut;
  idw_init(&c.idw, (int)c.scale);
  c.display_boost = max_display_boost < md->maxContentBoost ? max_display_boost
                                                            : md->maxContentBoost; /* :415 */
  c.lut = lut;
  if (lut) orc_gainLutBuild(md->minContentBoost, md->maxContentBoost, 1, c.display_boost, c.gain_lut); /* :416 */
  run_bands(app_band, &c, yuv->height, c.scale, threads); /* jobs of `scale` rows, :505 */
  idw_free(&c.idw);
  return ORC_OK;
}
int orc_applyGainMap(const orc_image* yuv, const orc_image* map, const orc_metadata* md, int fmt,
                     float max_display_boost, orc_image* dest, int threads) {
  return apply_impl(yuv, map, md, fmt, max_display_boost, dest, threads, 0);
}
/* the same loop as upstream builds it: USE_SRGB_INVOETF_LUT, USE_APPLY_GAIN_LUT, USE_HLG/PQ_OETF_LUT = 1 */
int orc_applyGainMapLUT(const orc_image* yuv, const orc_image* map, const orc_metadata* md, int fmt,
                        float max_display_boost, orc_image* dest, int threads) {
  pthread_once(&lut_once, lut_build);
  return apply_impl(yuv, map, md, fmt, max_display_boost, dest, threads, 1);
}

/* ------------------------------------------------------------------------------------------
 * toneMap (ultrahdr.cpp:517-558)
 * ---------------------------------------------------------------------------------------- */
int orc_toneMap(const orc_image* src, orc_image* dest) {
  if (src == NULL || dest == NULL) return ORC_ERR_BAD_PTR;
  if (src->width != dest->width || src->height != dest->height) return ORC_ERR_RESOLUTION_MISMATCH;
  const uint16_t* sy = (const uint16_t*)src->data;
  uint8_t* dy = (uint8_t*)dest->data;
  for (size_t y = 0; y < src->height; ++y) {
    const uint16_t* srow = sy + y * src->luma_stride;
    uint8_t* drow = dy + y * dest->luma_stride;
    for (size_t x = 0; x < src->width; ++x) {
      uint16_t v = srow[x] >> 6;
      drow[x] = (uint8_t)((v >> 2) & 0xff);
    }
    if (dest->width != dest->luma_stride) memset(drow + dest->width, 0, dest->luma_stride - dest->width);
  }
  const uint16_t* suv = (const uint16_t*)src->chroma_data;
  uint8_t* du = (uint8_t*)dest->chroma_data;
  size_t v_off = dest->chroma_stride * dest->height / 2;
  uint8_t* dv = du + v_off;
  for (size_t y = 0; y < src->height / 2; ++y) {
    const uint16_t* srow = suv + y * src->chroma_stride;
    uint8_t* urow = du + y * dest->chroma_stride;
    uint8_t* vrow = dv + y * dest->chroma_stride;
    for (size_t x = 0; x < src->width / 2; ++x) {
      uint16_t u = srow[x << 1] >> 6, v = srow[(x << 1) + 1] >> 6;
      urow[x] = (uint8_t)((u >> 2) & 0xff);
      vrow[x] = (uint8_t)((v >> 2) & 0xff);
    }
    if (dest->width / 2 != dest->chroma_stride) {
      memset(urow + dest->width / 2, 0, dest->chroma_stride - dest->width / 2);
      memset(vrow + dest->width / 2, 0, dest->chroma_stride - dest->width / 2);
    }
  }
  dest->colorGamut = src->colorGamut;
  return ORC_OK;
}

/* ------------------------------------------------------------------------------------------
 * editorhelper effects: lib/src/editorhelper.cpp:26-360, quirks included
 * ---------------------------------------------------------------------------------------- */
#define ORC_ERR_INVALID_CROPPING_PARAMETERS (-10011)
#define ORC_ERR_UNSUPPORTED_FEATURE (-30000)
static int fx_check(const orc_image* in, orc_image* out) {
  if (in == NULL || in->data == NULL || out == NULL || out->data == NULL) return ORC_ERR_BAD_PTR;
  return ORC_OK;
}
static int fx_fmt_ok(const orc_image* in) { return in->pixelFormat == 1 || in->pixelFormat == 2; } /* YUV420 | MONO */
static const uint8_t* fx_chroma(const orc_image* in, int ls) { /* :66-69 */
  return in->chroma_data ? (const uint8_t*)in->chroma_data : (const uint8_t*)in->data + (size_t)ls * in->height;
}

/* editorhelper.cpp:26-76.  NOTE the chroma loop runs over out->height rows (:72): the V plane of the result is
 * only the cropped V when nothing is cropped vertically -- kept as is */
int orc_crop(const orc_image* in, int left, int right, int top, int bottom, orc_image* out) {
  int rc = fx_check(in, out);
  if (rc) return rc;
  if (left < 0 || (size_t)right >= in->width || top < 0 || (size_t)bottom >= in->height)
    return ORC_ERR_INVALID_CROPPING_PARAMETERS;
  if (!fx_fmt_ok(in)) return ORC_ERR_UNSUPPORTED_FEATURE;
  out->colorGamut = in->colorGamut; out->pixelFormat = in->pixelFormat;
  int ls = in->luma_stride != 0 ? (int)in->luma_stride : (int)in->width;
  out->width = right - left + 1; out->height = bottom - top + 1; out->luma_stride = out->width;
  const uint8_t* src = (const uint8_t*)in->data + ls * top + left;
  uint8_t* dst = (uint8_t*)out->data;
  for (int i = 0; i < (int)out->height; i++) memcpy(dst + i * out->luma_stride, src + i * ls, out->width);
  if (in->pixelFormat == 2) return ORC_OK;
  int cs = in->chroma_stride != 0 ? (int)in->chroma_stride : (ls >> 1);
  out->chroma_stride = out->luma_stride / 2;
  out->chroma_data = (uint8_t*)out->data + out->luma_stride * out->height;
  src = fx_chroma(in, ls) + cs * (top / 2) + (left / 2);
  dst = (uint8_t*)out->chroma_data;
  for (int i = 0; i < (int)out->height; i++) memcpy(dst + i * out->chroma_stride, src + i * cs, out->width / 2);
  return ORC_OK;
}

/* editorhelper.cpp:78-170; dir 0 = vertical, 1 = horizontal.  Output strides follow the INPUT luma stride (:92) */
int orc_mirror(const orc_image* in, int dir, orc_image* out) {
  int rc = fx_check(in, out);
  if (rc) return rc;
  if (!fx_fmt_ok(in)) return ORC_ERR_UNSUPPORTED_FEATURE;
  out->colorGamut = in->colorGamut; out->pixelFormat = in->pixelFormat;
  int ls = in->luma_stride != 0 ? (int)in->luma_stride : (int)in->width;
  out->width = in->width; out->height = in->height; out->luma_stride = ls;
  int w = (int)out->width, h = (int)out->height;
  const uint8_t* sy = (const uint8_t*)in->data;
  uint8_t* dy = (uint8_t*)out->data;
  for (int i = 0; i < h; i++)
    for (int j = 0; j < w; j++)
      dy[(dir == 0 ? (h - i - 1) : i) * ls + (dir == 0 ? j : j)] =
          dir == 0 ? sy[i * ls + j] : sy[i * ls + (w - j - 1)];
  if (dir != 0) { /* horizontal written as dest-indexed loop in the reference: same bytes */ }
  if (in->pixelFormat == 2) return ORC_OK;
  int cs = in->chroma_stride != 0 ? (int)in->chroma_stride : (ls >> 1);
  out->chroma_stride = out->luma_stride / 2;
  out->chroma_data = (uint8_t*)out->data + out->luma_stride * out->height;
  int ocs = (int)out->chroma_stride;
  const uint8_t* su = fx_chroma(in, ls);
  uint8_t* du = (uint8_t*)out->chroma_data;
  for (int p = 0; p < 2; p++) {
    const uint8_t* sp = su + (p ? cs * ((int)in->height / 2) : 0);
    uint8_t* dp = du + (p ? ocs * (h / 2) : 0);
    for (int i = 0; i < h / 2; i++)
      for (int j = 0; j < w / 2; j++) {
        if (dir == 0) dp[(h / 2 - i - 1) * ocs + j] = sp[i * cs + j];
        else dp[i * ocs + j] = sp[i * cs + ((int)in->width / 2 - j - 1)];
      }
  }
  return ORC_OK;
}

/* editorhelper.cpp:172-306 */
int orc_rotate(const orc_image* in, int deg, orc_image* out) {
  int rc = fx_check(in, out);
  if (rc) return rc;
  if (deg != 90 && deg != 180 && deg != 270) return ORC_ERR_INVALID_CROPPING_PARAMETERS;
  if (!fx_fmt_ok(in)) return ORC_ERR_UNSUPPORTED_FEATURE;
  out->colorGamut = in->colorGamut; out->pixelFormat = in->pixelFormat;
  int ls = in->luma_stride != 0 ? (int)in->luma_stride : (int)in->width;
  int iw = (int)in->width, ih = (int)in->height;
  if (deg == 180) { out->width = iw; out->height = ih; out->luma_stride = ls; }
  else { out->width = ih; out->height = iw; out->luma_stride = out->width; }
  int ow = (int)out->width, oh = (int)out->height, ols = (int)out->luma_stride;
  const uint8_t* sy = (const uint8_t*)in->data;
  uint8_t* dy = (uint8_t*)out->data;
  for (int i = 0; i < oh; i++)
    for (int j = 0; j < ow; j++)
      dy[i * ols + j] = deg == 90 ? sy[(ih - j - 1) * ls + i]
                      : deg == 180 ? sy[(ih - i - 1) * ls + (iw - j - 1)] : sy[j * ls + (iw - i - 1)];
  if (in->pixelFormat == 2) return ORC_OK;
  int cs = in->chroma_stride != 0 ? (int)in->chroma_stride : (ls >> 1);
  out->chroma_stride = out->luma_stride / 2;
  out->chroma_data = (uint8_t*)out->data + out->luma_stride * out->height;
  int ocs = (int)out->chroma_stride;
  const uint8_t* su = fx_chroma(in, ls);
  uint8_t* du = (uint8_t*)out->chroma_data;
  for (int p = 0; p < 2; p++) {
    const uint8_t* sp = su + (p ? cs * (ih / 2) : 0);
    uint8_t* dp = du + (p ? ocs * (oh / 2) : 0);
    for (int i = 0; i < oh / 2; i++)
      for (int j = 0; j < ow / 2; j++)
        dp[i * ocs + j] = deg == 90 ? sp[(ih / 2 - j - 1) * cs + i]
                        : deg == 180 ? sp[(ih / 2 - i - 1) * cs + (iw / 2 - j - 1)] : sp[j * cs + (iw / 2 - i - 1)];
  }
  return ORC_OK;
}

/* editorhelper.cpp:308-360: nearest neighbour; the chroma loop covers U and V in one go (rows < out height, :350) */
int orc_resize(const orc_image* in, int ow, int oh, orc_image* out) {
  int rc = fx_check(in, out);
  if (rc) return rc;
  if (!fx_fmt_ok(in)) return ORC_ERR_UNSUPPORTED_FEATURE;
  out->colorGamut = in->colorGamut; out->pixelFormat = in->pixelFormat;
  int ls = in->luma_stride != 0 ? (int)in->luma_stride : (int)in->width;
  out->width = ow; out->height = oh; out->luma_stride = out->width;
  const uint8_t* sy = (const uint8_t*)in->data;
  uint8_t* dy = (uint8_t*)out->data;
  for (int i = 0; i < oh; i++)
    for (int j = 0; j < ow; j++)
      dy[i * out->luma_stride + j] = sy[(size_t)i * in->height / out->height * ls + (size_t)j * in->width / out->width];
  if (in->pixelFormat == 2) return ORC_OK;
  int cs = in->chroma_stride != 0 ? (int)in->chroma_stride : (ls >> 1);
  out->chroma_stride = out->luma_stride / 2;
  out->chroma_data = (uint8_t*)out->data + out->luma_stride * out->height;
  const uint8_t* sc = fx_chroma(in, ls);
  uint8_t* dc = (uint8_t*)out->chroma_data;
  for (int i = 0; i < oh; i++)
    for (int j = 0; j < ow / 2; j++)
      dc[i * out->chroma_stride + j] = sc[(size_t)i * in->height / out->height * cs + (size_t)j * in->width / out->width];
  return ORC_OK;
}

/* batch evaluation of the scalar functions (tests compare the device functions against these) */
/* addEffects (editorhelper.cpp:362-446): the effects applied in order, each into a fresh tightly sized buffer whose
 * bytes are then copied into out->data; out's fields follow the last effect and its chroma pointer is re-derived as
 * data + luma_stride * height.  Restated for the cases in which the reference is defined: it ignores the status of the
 * individual effects (and then copies uninitialised fields); here the first failing effect's status is returned. */
int orc_add_effects(const orc_image* in, const orc_effect* fx, int n, orc_image* out) {
  if (in == NULL || in->data == NULL || out == NULL || out->data == NULL) return ORC_ERR_BAD_PTR;
  const int mono = in->pixelFormat == 2;
  size_t size = in->width * in->height;
  if (!mono) size = size * 3 / 2;
  out->width = in->width; out->height = in->height; out->colorGamut = in->colorGamut; out->pixelFormat = in->pixelFormat;
  out->luma_stride = in->luma_stride; out->chroma_stride = in->chroma_stride;
  memcpy(out->data, in->data, size);
  const orc_image* last = in;
  for (int i = 0; i < n; ++i) {
    orc_image tmp;
    memset(&tmp, 0, sizeof(tmp));
    int rc;
    switch (fx[i].type) {
      case 0: size = (size_t)(fx[i].d - fx[i].c + 1) * (size_t)(fx[i].b - fx[i].a + 1); break;
      case 1: case 2: size = last->width * last->height; break;
      case 3: size = (size_t)fx[i].a * (size_t)fx[i].b; break;
      default: return ORC_ERR_BAD_PTR;
    }
    if (!mono) size = size * 3 / 2;
    /* (the reference allocates exactly `size`; mirror / rotate-180 of a padded image write more than that there) */
    size_t cap = size;
    if (fx[i].type == 1 || fx[i].type == 2) {
      size_t ls = last->luma_stride ? last->luma_stride : last->width;
      cap = ls * last->height * 2 + 64;
    }
    uint8_t* buf = (uint8_t*)malloc(cap ? cap : 1);
    tmp.data = buf;
    switch (fx[i].type) {
      case 0: rc = orc_crop(last, fx[i].a, fx[i].b, fx[i].c, fx[i].d, &tmp); break;
      case 1: rc = orc_mirror(last, fx[i].a, &tmp); break;
      case 2: rc = orc_rotate(last, fx[i].a, &tmp); break;
      default: rc = orc_resize(last, fx[i].a, fx[i].b, &tmp); break;
    }
    if (rc != ORC_OK) { free(buf); return rc; }
    out->width = tmp.width; out->height = tmp.height; out->colorGamut = tmp.colorGamut; out->pixelFormat = tmp.pixelFormat;
    out->luma_stride = tmp.luma_stride; out->chroma_stride = tmp.chroma_stride;
    memcpy(out->data, tmp.data, size);
    if (!mono) out->chroma_data = (uint8_t*)out->data + out->luma_stride * out->height;
    free(buf);
    last = out;
  }
  return ORC_OK;
}

void orc_eval_transfer(int fn, const float* in, float* out, size_t n, float minBoost, float maxBoost) {
  float l2min = (float)log2((double)minBoost), l2max = (float)log2((double)maxBoost);
  float gain_lut[ORC_GAIN_LUT_N];
  if (fn == 46) orc_gainLutBuild(minBoost, maxBoost, 1, maxBoost, gain_lut);
  for (size_t i = 0; i < n; ++i) {
    float x = in[i], y = 0.0f;
    switch (fn) {
      case 40: y = orc_srgbInvOetfLUT(x); break;
      case 41: y = orc_hlgInvOetfLUT(x); break;
      case 42: y = orc_pqInvOetfLUT(x); break;
      case 44: y = orc_hlgOetfLUT(x); break;
      case 45: y = orc_pqOetfLUT(x); break;
      case 46: y = orc_gainLutFactor(gain_lut, x); break;
      case 0: y = orc_srgbInvOetf(x); break;
      case 1: y = orc_hlgInvOetf(x); break;
      case 2: y = orc_pqInvOetf(x); break;
      case 3: y = (float)orc_encodeGain(1.0f, x, minBoost, maxBoost, l2min, l2max); break; /* gain = x/1 */
      case 4: y = orc_hlgOetf(x); break;
      case 5: y = orc_pqOetf(x); break;
      default: break;
    }
    out[i] = y;
  }
}

/* ------------------------------------------------------------------------------------------
 * Synthetic input + checksum
 * ---------------------------------------------------------------------------------------- */
void orc_fill_lcg(uint16_t* p010, uint8_t* yuv, size_t w, size_t h, uint32_t seed) {
  uint32_t s = seed;
#define ORC_NEXT() (s = s * 1664525u + 1013904223u, s >> 8)
  size_t n = w * h;
  for (size_t i = 0; i < n; ++i) {
    p010[i] = (uint16_t)((64 + ORC_NEXT() % 877) << 6);
    yuv[i] = (uint8_t)(ORC_NEXT() & 255);
  }
  for (size_t i = n; i < n * 3 / 2; ++i) {
    p010[i] = (uint16_t)((64 + ORC_NEXT() % 897) << 6);
    yuv[i] = (uint8_t)(ORC_NEXT() & 255);
  }
#undef ORC_NEXT
}
uint64_t orc_checksum_u8(const uint8_t* p, size_t n) {
  uint64_t cs = 0;
  for (size_t i = 0; i < n; ++i) cs = cs * 131 + p[i];
  return cs;
}
uint64_t orc_checksum_u32(const uint32_t* p, size_t n) {
  uint64_t cs = 0;
  for (size_t i = 0; i < n; ++i) cs = cs * 131 + p[i];
  return cs;
}
