/*
 * ref_harness.cpp -- TEST INFRASTRUCTURE ONLY.  Built only where /root/reference exists.
 *
 * Thin extern "C" harness around the REFERENCE's own pixel math: it is linked with
 * /root/reference/lib/src/gainmapmath.cpp compiled in place (see oracle/Makefile) and calls the
 * reference's public functions (ultrahdr::sampleYuv420, srgbInvOetf, encodeGain, sampleMap, ...).
 *
 * What is and is not the reference here:
 *  - every per-pixel function is the reference's compiled code;
 *  - lib/src/ultrahdr.cpp (generateGainMap/applyGainMap/toneMap member functions) is NOT buildable
 *    in this image: it #includes a forked, un-vendored libheif (ultrahdr.cpp:39-40) and stand-in
 *    headers are not allowed.  The row/column loops below therefore restate
 *    ultrahdr.cpp:220-336 and :414-494 (function-pointer selection + call order) while every
 *    arithmetic step is a call into the reference object.  convertYuv (jpegr.cpp:1199-1203) is
 *    reached through the reference's public transformYuv420 + yuvXToY functions.
 *  - the loops are additionally pinned by the md5s/checksums the full reference produced
 *    (SURVEY.md 8(c)/(d)); see tests/test_oracle_pins.py.
 */
#include <cmath>
#include <cstdint>
#include <cstring>

#include "ultrahdr/editorhelper.h"
#include "ultrahdr/gainmapmath.h"
#include <memory>
#include <vector>

#include "uhdr_oracle.h"

using namespace ultrahdr;

namespace {
ultrahdr_uncompressed_struct to_ref(const orc_image* i) {
  ultrahdr_uncompressed_struct r;
  r.data = i->data;
  r.width = i->width;
  r.height = i->height;
  r.colorGamut = static_cast<ultrahdr_color_gamut>(i->colorGamut);
  r.chroma_data = i->chroma_data;
  r.luma_stride = i->luma_stride;
  r.chroma_stride = i->chroma_stride;
  r.pixelFormat = static_cast<ultrahdr_pixel_format>(i->pixelFormat);
  return r;
}
ultrahdr_metadata_struct to_ref(const orc_metadata* m) {
  ultrahdr_metadata_struct r;
  r.version = m->version_ok ? kGainMapVersion : "bad";
  r.maxContentBoost = m->maxContentBoost;
  r.minContentBoost = m->minContentBoost;
  r.gamma = m->gamma;
  r.offsetSdr = m->offsetSdr;
  r.offsetHdr = m->offsetHdr;
  r.hdrCapacityMin = m->hdrCapacityMin;
  r.hdrCapacityMax = m->hdrCapacityMax;
  return r;
}
Color C(orc_color c) { return {{{c.r, c.g, c.b}}}; }
orc_color O(Color c) { return {c.r, c.g, c.b}; }
ColorTransformFn yuv2rgb(int g) {
  return g == ORC_CG_BT709 ? srgbYuvToRgb : g == ORC_CG_P3 ? p3YuvToRgb : bt2100YuvToRgb;
}
ColorTransformFn rgb2yuv(int g) {
  return g == ORC_CG_BT709 ? srgbRgbToYuv : g == ORC_CG_P3 ? p3RgbToYuv : bt2100RgbToYuv;
}
ColorCalculationFn lum(int g) {
  return g == ORC_CG_BT709 ? srgbLuminance : g == ORC_CG_P3 ? p3Luminance : bt2100Luminance;
}
ColorTransformFn yuv2yuv(int s, int d) {
  if (s == d) return nullptr;
  if (s == ORC_CG_BT709) return d == ORC_CG_P3 ? yuv709To601 : yuv709To2100;
  if (s == ORC_CG_P3) return d == ORC_CG_BT709 ? yuv601To709 : yuv601To2100;
  return d == ORC_CG_BT709 ? yuv2100To709 : yuv2100To601;
}
}  // namespace

extern "C" {

float ref_srgbInvOetf(float e) { return srgbInvOetf(e); }
float ref_hlgOetf(float e) { return hlgOetf(e); }
float ref_hlgInvOetf(float e) { return hlgInvOetf(e); }
float ref_pqOetf(float e) { return pqOetf(e); }
float ref_pqInvOetf(float e) { return pqInvOetf(e); }
float ref_srgbInvOetfLUT(float e) { return srgbInvOetfLUT(e); }
float ref_hlgOetfLUT(float e) { return hlgOetfLUT(e); }
float ref_hlgInvOetfLUT(float e) { return hlgInvOetfLUT(e); }
float ref_pqOetfLUT(float e) { return pqOetfLUT(e); }
float ref_pqInvOetfLUT(float e) { return pqInvOetfLUT(e); }
float ref_luminance(int g, orc_color e) { return lum(g)(C(e)); }
orc_color ref_yuvToRgb(int g, orc_color e) { return O(yuv2rgb(g)(C(e))); }
orc_color ref_rgbToYuv(int g, orc_color e) { return O(rgb2yuv(g)(C(e))); }
orc_color ref_gamutConv(int sdr, int hdr, orc_color e, int* is_null) {
  ColorTransformFn f = getHdrConversionFn(static_cast<ultrahdr_color_gamut>(sdr),
                                          static_cast<ultrahdr_color_gamut>(hdr));
  if (is_null) *is_null = f == nullptr;
  return f ? O(f(C(e))) : e;
}
orc_color ref_yuvToYuv(int s, int d, orc_color e) {
  ColorTransformFn f = yuv2yuv(s, d);
  return f ? O(f(C(e))) : e;
}
uint8_t ref_encodeGain(float y_sdr, float y_hdr, float minB, float maxB, float l2min, float l2max) {
  ultrahdr_metadata_struct m;
  m.minContentBoost = minB;
  m.maxContentBoost = maxB;
  return encodeGain(y_sdr, y_hdr, &m, l2min, l2max);
}
uint8_t ref_encodeGain3(float y_sdr, float y_hdr, float minB, float maxB) {
  ultrahdr_metadata_struct m;
  m.minContentBoost = minB;
  m.maxContentBoost = maxB;
  return encodeGain(y_sdr, y_hdr, &m);
}
orc_color ref_applyGain3(orc_color e, float gain, float minB, float maxB) {
  ultrahdr_metadata_struct m;
  m.minContentBoost = minB;
  m.maxContentBoost = maxB;
  return O(applyGain(C(e), gain, &m));
}
orc_color ref_applyGain4(orc_color e, float gain, float minB, float maxB, float db) {
  ultrahdr_metadata_struct m;
  m.minContentBoost = minB;
  m.maxContentBoost = maxB;
  return O(applyGain(C(e), gain, &m, db));
}
orc_color ref_applyGainLUT(orc_color e, float gain, float minB, float maxB, float db) {
  ultrahdr_metadata_struct m;
  m.minContentBoost = minB;
  m.maxContentBoost = maxB;
  GainLUT lut(&m, db);
  return O(applyGainLUT(C(e), gain, lut));
}
orc_color ref_getYuv420Pixel(const orc_image* i, size_t x, size_t y) {
  auto r = to_ref(i);
  return O(getYuv420Pixel(&r, x, y));
}
orc_color ref_getP010Pixel(const orc_image* i, size_t x, size_t y) {
  auto r = to_ref(i);
  return O(getP010Pixel(&r, x, y));
}
orc_color ref_sampleYuv420(const orc_image* i, size_t s, size_t x, size_t y) {
  auto r = to_ref(i);
  return O(sampleYuv420(&r, s, x, y));
}
orc_color ref_sampleP010(const orc_image* i, size_t s, size_t x, size_t y) {
  auto r = to_ref(i);
  return O(sampleP010(&r, s, x, y));
}
void ref_fillShepardsIDW(float* w, int scale, int incR, int incB) {
  ShepardsIDW t(scale);
  t.fillShepardsIDW(w, incR, incB);
}
float ref_sampleMapIdw(const orc_image* i, size_t s, size_t x, size_t y) {
  auto r = to_ref(i);
  ShepardsIDW t(static_cast<int>(s));
  return sampleMap(&r, s, x, y, t);
}
float ref_sampleMapFloat(const orc_image* i, float s, size_t x, size_t y) {
  auto r = to_ref(i);
  return sampleMap(&r, s, x, y);
}
uint32_t ref_colorToRgba1010102(orc_color e) { return colorToRgba1010102(C(e)); }
uint64_t ref_colorToRgbaF16(orc_color e) { return colorToRgbaF16(C(e)); }
uint16_t ref_floatToHalf(float f) { return floatToHalf(f); }
void ref_transformYuv420(orc_image* i, size_t xc, size_t yc, int s, int d) {
  auto r = to_ref(i);
  ColorTransformFn f = yuv2yuv(s, d);
  if (f) transformYuv420(&r, xc, yc, f);
}

/* the reference's editorhelper.cpp, compiled in place */
static void fx_back(const ultrahdr_uncompressed_struct& r, orc_image* o) {
  o->width = r.width; o->height = r.height; o->colorGamut = r.colorGamut; o->chroma_data = r.chroma_data;
  o->luma_stride = r.luma_stride; o->chroma_stride = r.chroma_stride; o->pixelFormat = r.pixelFormat;
}
int ref_crop(const orc_image* in, int l, int r, int t, int b, orc_image* out) {
  if (!in || !out) return crop(nullptr, l, r, t, b, nullptr);
  auto i = to_ref(in); auto o = to_ref(out);
  int rc = crop(&i, l, r, t, b, &o);
  if (rc == 0) fx_back(o, out);
  return rc;
}
int ref_mirror(const orc_image* in, int dir, orc_image* out) {
  if (!in || !out) return mirror(nullptr, ULTRAHDR_MIRROR_VERTICAL, nullptr);
  auto i = to_ref(in); auto o = to_ref(out);
  int rc = mirror(&i, dir == 0 ? ULTRAHDR_MIRROR_VERTICAL : ULTRAHDR_MIRROR_HORIZONTAL, &o);
  if (rc == 0) fx_back(o, out);
  return rc;
}
int ref_rotate(const orc_image* in, int deg, orc_image* out) {
  if (!in || !out) return rotate(nullptr, deg, nullptr);
  auto i = to_ref(in); auto o = to_ref(out);
  int rc = rotate(&i, deg, &o);
  if (rc == 0) fx_back(o, out);
  return rc;
}
int ref_resize(const orc_image* in, int w, int h, orc_image* out) {
  if (!in || !out) return resize(nullptr, w, h, nullptr);
  auto i = to_ref(in); auto o = to_ref(out);
  int rc = resize(&i, w, h, &o);
  if (rc == 0) fx_back(o, out);
  return rc;
}

/* the reference's addEffects (editorhelper.cpp:362-446) over its own effect structs */
int ref_add_effects(const orc_image* in, const orc_effect* fx, int n, orc_image* out) {
  if (!in || !out) return addEffects(nullptr, *(new std::vector<ultrahdr_effect*>()), nullptr);
  std::vector<std::unique_ptr<ultrahdr_effect>> own;
  std::vector<ultrahdr_effect*> effects;
  for (int i = 0; i < n; ++i) {
    switch (fx[i].type) {
      case 0: { auto* e = new ultrahdr_crop_effect; e->left = fx[i].a; e->right = fx[i].b; e->top = fx[i].c; e->bottom = fx[i].d; own.emplace_back(e); break; }
      case 1: { auto* e = new ultrahdr_mirror_effect; e->mirror_dir = fx[i].a == 0 ? ULTRAHDR_MIRROR_VERTICAL : ULTRAHDR_MIRROR_HORIZONTAL; own.emplace_back(e); break; }
      case 2: { auto* e = new ultrahdr_rotate_effect; e->clockwise_degree = fx[i].a; own.emplace_back(e); break; }
      default: { auto* e = new ultrahdr_resize_effect; e->new_width = fx[i].a; e->new_height = fx[i].b; own.emplace_back(e); break; }
    }
    effects.push_back(own.back().get());
  }
  auto i = to_ref(in); auto o = to_ref(out);
  int rc = addEffects(&i, effects, &o);
  if (rc == 0) fx_back(o, out);
  return rc;
}

/* jpegr.cpp:1199-1203 loop order over the reference's transformYuv420 */
int ref_convertYuv(orc_image* i, int s, int d) {
  if (!i) return ORC_ERR_BAD_PTR;
  if (s == ORC_CG_UNSPECIFIED || d == ORC_CG_UNSPECIFIED) return ORC_ERR_INVALID_COLORGAMUT;
  ColorTransformFn f = yuv2yuv(s, d);
  if (!f) return ORC_OK;
  auto r = to_ref(i);
  for (size_t y = 0; y < r.height / 2; ++y)
    for (size_t x = 0; x < r.width / 2; ++x) transformYuv420(&r, x, y, f);
  return ORC_OK;
}

/* ultrahdr.cpp:220-336 call order, single-threaded */
static int ref_generate_impl(const orc_image* yuv_, const orc_image* p010_, int hdr_tf, orc_metadata* md,
                             uint8_t* map_out, int sdr_is_601, bool lut) {
  if (!yuv_ || !p010_ || !md || !map_out || !yuv_->data || !yuv_->chroma_data || !p010_->data ||
      !p010_->chroma_data)
    return ORC_ERR_BAD_PTR;
  if (yuv_->width != p010_->width || yuv_->height != p010_->height)
    return ORC_ERR_RESOLUTION_MISMATCH;
  if (yuv_->colorGamut == ORC_CG_UNSPECIFIED || p010_->colorGamut == ORC_CG_UNSPECIFIED)
    return ORC_ERR_INVALID_COLORGAMUT;
  auto yuv = to_ref(yuv_);
  auto p010 = to_ref(p010_);
  size_t map_w = yuv.width / kMapDimensionScaleFactor, map_h = yuv.height / kMapDimensionScaleFactor;
  ColorTransformFn hdrInvOetf;
  float hdr_white_nits;
  switch (hdr_tf) {
    case ORC_TF_LINEAR: hdrInvOetf = identityConversion; hdr_white_nits = kHlgMaxNits; break;
    case ORC_TF_HLG: hdrInvOetf = lut ? (ColorTransformFn)hlgInvOetfLUT : (ColorTransformFn)hlgInvOetf; hdr_white_nits = kHlgMaxNits; break;
    case ORC_TF_PQ: hdrInvOetf = lut ? (ColorTransformFn)pqInvOetfLUT : (ColorTransformFn)pqInvOetf; hdr_white_nits = kPqMaxNits; break;
    default: return ORC_ERR_INVALID_TRANS_FUNC;
  }
  ultrahdr_metadata_struct m;
  m.version = kGainMapVersion;
  m.maxContentBoost = hdr_white_nits / kSdrWhiteNits;
  m.minContentBoost = 1.0f;
  m.gamma = 1.0f;
  m.offsetSdr = 0.0f;
  m.offsetHdr = 0.0f;
  m.hdrCapacityMin = 1.0f;
  m.hdrCapacityMax = m.maxContentBoost;
  float log2MinBoost = log2(m.minContentBoost);
  float log2MaxBoost = log2(m.maxContentBoost);
  ColorTransformFn gamutFn = getHdrConversionFn(yuv.colorGamut, p010.colorGamut);
  if (yuv.colorGamut < 0 || yuv.colorGamut > 2 || p010.colorGamut < 0 || p010.colorGamut > 2)
    return ORC_ERR_INVALID_COLORGAMUT;
  ColorCalculationFn luminanceFn = lum(yuv.colorGamut);
  ColorTransformFn sdrYuvToRgbFn = sdr_is_601 ? p3YuvToRgb : yuv2rgb(yuv.colorGamut);
  ColorTransformFn hdrYuvToRgbFn = yuv2rgb(p010.colorGamut);
  for (size_t y = 0; y < map_h; ++y)
    for (size_t x = 0; x < map_w; ++x) {
      Color sdr_yuv_gamma = sampleYuv420(&yuv, kMapDimensionScaleFactor, x, y);
      Color sdr_rgb_gamma = sdrYuvToRgbFn(sdr_yuv_gamma);
      Color sdr_rgb = lut ? srgbInvOetfLUT(sdr_rgb_gamma) : srgbInvOetf(sdr_rgb_gamma);
      float sdr_y_nits = luminanceFn(sdr_rgb) * kSdrWhiteNits;
      Color hdr_yuv_gamma = sampleP010(&p010, kMapDimensionScaleFactor, x, y);
      Color hdr_rgb_gamma = hdrYuvToRgbFn(hdr_yuv_gamma);
      Color hdr_rgb = hdrInvOetf(hdr_rgb_gamma);
      hdr_rgb = gamutFn(hdr_rgb);
      float hdr_y_nits = luminanceFn(hdr_rgb) * hdr_white_nits;
      map_out[x + y * map_w] = encodeGain(sdr_y_nits, hdr_y_nits, &m, log2MinBoost, log2MaxBoost);
    }
  md->version_ok = 1;
  md->maxContentBoost = m.maxContentBoost;
  md->minContentBoost = m.minContentBoost;
  md->gamma = m.gamma;
  md->offsetSdr = m.offsetSdr;
  md->offsetHdr = m.offsetHdr;
  md->hdrCapacityMin = m.hdrCapacityMin;
  md->hdrCapacityMax = m.hdrCapacityMax;
  return ORC_OK;
}

int ref_generateGainMap(const orc_image* yuv, const orc_image* p010, int hdr_tf, orc_metadata* md, uint8_t* map_out,
                        int sdr_is_601, int /*threads*/) {
  return ref_generate_impl(yuv, p010, hdr_tf, md, map_out, sdr_is_601, false);
}
/* the branches ultrahdr.cpp:230,238,319 take when USE_*_LUT is 1 */
int ref_generateGainMapLUT(const orc_image* yuv, const orc_image* p010, int hdr_tf, orc_metadata* md, uint8_t* map_out,
                           int sdr_is_601, int /*threads*/) {
  return ref_generate_impl(yuv, p010, hdr_tf, md, map_out, sdr_is_601, true);
}

/* ultrahdr.cpp:364-494 call order, single-threaded */
static int ref_apply_impl(const orc_image* yuv_, const orc_image* map_, const orc_metadata* md, int fmt,
                          float max_display_boost, orc_image* dest, bool lut) {
  if (!yuv_ || !map_ || !md || !dest || !yuv_->data || !yuv_->chroma_data || !map_->data)
    return ORC_ERR_BAD_PTR;
  ultrahdr_metadata_struct m = to_ref(md);
  ultrahdr_metadata_ptr metadata = &m;
  if (m.version.compare(kGainMapVersion)) return ORC_ERR_BAD_METADATA;
  if (m.gamma != 1.0f) return ORC_ERR_BAD_METADATA;
  if (m.offsetSdr != 0.0f || m.offsetHdr != 0.0f) return ORC_ERR_BAD_METADATA;
  if (m.hdrCapacityMin != m.minContentBoost || m.hdrCapacityMax != m.maxContentBoost)
    return ORC_ERR_BAD_METADATA;
  auto yuv = to_ref(yuv_);
  auto map = to_ref(map_);
  if (yuv.width % map.width != 0 || yuv.height % map.height != 0)
    return ORC_ERR_UNSUPPORTED_MAP_SCALE_FACTOR;
  if (yuv.width * map.height != yuv.height * map.width) return ORC_ERR_UNSUPPORTED_MAP_SCALE_FACTOR;
  size_t map_scale_factor = yuv.width / map.width;
  dest->width = yuv.width;
  dest->height = yuv.height;
  dest->colorGamut = yuv.colorGamut;
  ShepardsIDW idwTable(static_cast<int>(map_scale_factor));
  float display_boost = (std::min)(max_display_boost, m.maxContentBoost);
  GainLUT gainLUT(metadata, display_boost);
  size_t width = yuv.width, height = yuv.height;
  for (size_t y = 0; y < height; ++y)
    for (size_t x = 0; x < width; ++x) {
      Color yuv_gamma_sdr = getYuv420Pixel(&yuv, x, y);
      Color rgb_gamma_sdr = p3YuvToRgb(yuv_gamma_sdr);
      Color rgb_sdr = lut ? srgbInvOetfLUT(rgb_gamma_sdr) : srgbInvOetf(rgb_gamma_sdr);
      float gain = sampleMap(&map, map_scale_factor, x, y, idwTable);
      Color rgb_hdr = lut ? applyGainLUT(rgb_sdr, gain, gainLUT) : applyGain(rgb_sdr, gain, metadata, display_boost);
      rgb_hdr = rgb_hdr / display_boost;
      size_t pixel_idx = x + y * width;
      switch (fmt) {
        case ORC_OUT_HDR_LINEAR:
          reinterpret_cast<uint64_t*>(dest->data)[pixel_idx] = colorToRgbaF16(rgb_hdr);
          break;
        case ORC_OUT_HDR_LINEAR_RGB_10BIT: {
          uint16_t r = 0x3ff & static_cast<uint32_t>(rgb_hdr.r * 1023.0f);
          uint16_t g = 0x3ff & static_cast<uint32_t>(rgb_hdr.g * 1023.0f);
          uint16_t b = 0x3ff & static_cast<uint32_t>(rgb_hdr.b * 1023.0f);
          reinterpret_cast<uint16_t*>(dest->data)[pixel_idx] = r;
          reinterpret_cast<uint16_t*>(dest->data)[width * height + pixel_idx] = g;
          reinterpret_cast<uint16_t*>(dest->data)[width * height * 2 + pixel_idx] = b;
          break;
        }
        case ORC_OUT_HDR_HLG:
          reinterpret_cast<uint32_t*>(dest->data)[pixel_idx] = colorToRgba1010102(lut ? hlgOetfLUT(rgb_hdr) : hlgOetf(rgb_hdr));
          break;
        case ORC_OUT_HDR_PQ:
          reinterpret_cast<uint32_t*>(dest->data)[pixel_idx] = colorToRgba1010102(lut ? pqOetfLUT(rgb_hdr) : pqOetf(rgb_hdr));
          break;
        default: break;
      }
    }
  return ORC_OK;
}
int ref_applyGainMap(const orc_image* yuv, const orc_image* map, const orc_metadata* md, int fmt, float max_display_boost,
                     orc_image* dest, int /*threads*/) {
  return ref_apply_impl(yuv, map, md, fmt, max_display_boost, dest, false);
}
/* the branches ultrahdr.cpp:433,446,470,481 take when USE_*_LUT is 1 */
int ref_applyGainMapLUT(const orc_image* yuv, const orc_image* map, const orc_metadata* md, int fmt,
                        float max_display_boost, orc_image* dest, int /*threads*/) {
  return ref_apply_impl(yuv, map, md, fmt, max_display_boost, dest, true);
}

/* GainLUT tables as the reference's header builds them (gainmapmath.h:151-182) */
void ref_gainLutBuild(float minB, float maxB, int with_display_boost, float db, float* table) {
  ultrahdr_metadata_struct m;
  m.minContentBoost = minB;
  m.maxContentBoost = maxB;
  // the table is private: read it back through getGainFactor at the knots idx/(N-1)... which rounds to idx only
  // if float(idx/1023f*1023f+0.5) truncates to idx; use the exact inverse instead: (idx)/1023 as the nearest float
  // whose product lands in [idx-0.5, idx+0.5)
  GainLUT a(&m), b(&m, db);
  GainLUT& g = with_display_boost ? b : a;
  for (size_t idx = 0; idx < kGainFactorNumEntries; ++idx) {
    float gain = static_cast<float>(idx) / static_cast<float>(kGainFactorNumEntries - 1);
    table[idx] = g.getGainFactor(gain);
  }
}
float ref_gainLutFactor(float minB, float maxB, float db, float gain) {
  ultrahdr_metadata_struct m;
  m.minContentBoost = minB;
  m.maxContentBoost = maxB;
  GainLUT g(&m, db);
  return g.getGainFactor(gain);
}

}  // extern "C"
