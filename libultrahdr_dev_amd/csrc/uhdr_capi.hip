// uhdr_capi.hip -- the C-ABI of include/uhdr_hip.h: argument validation in the reference's order,
// per-call constants, host staging, batching.  All pixel work happens in uhdr_kernels.hip on the
// GPU; there is deliberately no CPU fallback -- without a usable device every compute entry point
// returns UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <memory>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <chrono>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/uhdr_hip.h"
#include "uhdr_kernels.h"
#include "uhdr_jpeg.h"
#include "uhdr_jpegr.h"

namespace {

using namespace uhdr;

thread_local char t_err[256] = "";

void set_err(const char* where, hipError_t e) {
  snprintf(t_err, sizeof(t_err), "%s: %s", where, hipGetErrorString(e));
}

#define HIP_TRY(expr)                          \
  do {                                         \
    hipError_t _e = (expr);                    \
    if (_e != hipSuccess) {                    \
      set_err(#expr, _e);                      \
      return UHDR_HIP_UNKNOWN_ERROR;           \
    }                                          \
  } while (0)

// ---- per-device state --------------------------------------------------------------------------
// grow-only device staging buffers of one UHDR_HIP_MEM_HOST call.  The four pixel-path entry points (generate, apply, toneMap,
// convertYuv) lease a set of their own for the duration of a call, so host callers on different streams overlap their copies and
// kernels; the codec entry points lease a whole context (CodecLease).
struct StageSet {
  void* stage[14] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t stage_bytes[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
};
struct DeviceState : StageSet {
  bool ready = false;
  std::map<int, float*> idw;  // scale -> device tables (4 * scale*scale*4 floats)
  float* lut = nullptr;       // the five static transfer-function tables (kLutTotal floats), built at init
  std::vector<std::unique_ptr<StageSet>> sets;   // every set ever leased ...
  std::vector<StageSet*> free_sets;              // ... and those not in use (both under g_mu)
  std::vector<std::pair<hipStream_t, void*>> retired;   // workspaces that were outgrown while a launch may still have named them, by stream
  // uhdr_hip_jpegr_decode[_batch]: per-file decoder workspaces and planes
  std::vector<void*> pool;
  std::vector<size_t> pool_bytes;
  // encodeJPEGR: the gain-map JPEG is compressed on a stream of its own, next to the SDR image's conversion and compression
  hipStream_t aux = nullptr;
  hipEvent_t map_ready = nullptr;
  // generate with statistics: the candidate lists of one launch (kStatWsBytes), one workspace per stream the caller has used --
  // launches of one stream follow each other, launches of different streams may overlap
  std::map<hipStream_t, uint32_t*> stat_ws;
  // EXACT apply behind its pre-filter: the lists of pixels in doubt (uhdr_kernels.h: ex_ws_bytes), per stream, grown on demand
  struct ExWs { uint32_t* p = nullptr; size_t bytes = 0; };
  std::map<hipStream_t, ExWs> ex_ws;
  // the codec entry points (jpeg_*, jpegr_*, effects and tables through host memory) each lease a context of their own for the
  // duration of a call -- staging slots, decoder pool, side stream and event -- so that callers on different streams overlap
  // (CodecLease); a context is a DeviceState that borrows this one's tables
  std::vector<std::unique_ptr<DeviceState>> codec_sets;   // every context ever leased ...
  std::vector<DeviceState*> free_codec;                    // ... and those not in use (both under g_mu)
};
std::mutex g_mu;                    // guards g_dev (init / table cache)
// a kernel and the resolve kernel behind it share a per-stream workspace: the pair is enqueued as one unit, so that two host threads
// using the same stream cannot interleave their launches (they would read each other's lists)
std::mutex g_pair_mu;
std::map<int, DeviceState> g_dev;

// gainmapmath.cpp:69-110: sqrt runs in double on a float expression, weights are float divisions
float euclid(float x1, float x2, float y1, float y2) {
  return (float)std::sqrt((double)(((y2 - y1) * (y2 - y1)) + (x2 - x1) * (x2 - x1)));
}
void fill_idw(float* w, int scale, int incR, int incB) {
  for (int y = 0; y < scale; y++)
    for (int x = 0; x < scale; x++) {
      const float pos_x = ((float)x) / scale, pos_y = ((float)y) / scale;
      const int curr_x = (int)std::floor((double)pos_x), curr_y = (int)std::floor((double)pos_y);
      const int next_x = curr_x + incR, next_y = curr_y + incB;
      const float d1 = euclid(pos_x, curr_x, pos_y, curr_y);
      float* o = w + y * scale * 4 + x * 4;
      if (d1 == 0) {
        o[0] = 1.f; o[1] = 0.f; o[2] = 0.f; o[3] = 0.f;
      } else {
        const float w1 = 1.f / d1;
        const float w2 = 1.f / euclid(pos_x, curr_x, pos_y, next_y);
        const float w3 = 1.f / euclid(pos_x, next_x, pos_y, curr_y);
        const float w4 = 1.f / euclid(pos_x, next_x, pos_y, next_y);
        const float total = w1 + w2 + w3 + w4;
        o[0] = w1 / total; o[1] = w2 / total; o[2] = w3 / total; o[3] = w4 / total;
      }
    }
}
// table order: std (1,1), no-right (0,1), no-bottom (1,0), corner (0,0)  gainmapmath.h:191-194
void build_idw_tables(int scale, std::vector<float>& out) {
  const size_t n = (size_t)scale * scale * 4;
  out.assign(4 * n, 0.f);
  fill_idw(out.data(), scale, 1, 1);
  fill_idw(out.data() + n, scale, 0, 1);
  fill_idw(out.data() + 2 * n, scale, 1, 0);
  fill_idw(out.data() + 3 * n, scale, 0, 0);
}

int current_state(DeviceState** st) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) return UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE;
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_dev.find(dev);
  if (it == g_dev.end() || !it->second.ready) {
    snprintf(t_err, sizeof(t_err), "uhdr_hip_init(%d) has not been called", dev);
    return UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE;
  }
  *st = &it->second;
  return UHDR_HIP_NO_ERROR;
}

// ShepardsIDW(scale) on the device.  The scale comes from the caller's images (a file, on the decode path): the table holds
// 16 scale^2 floats -- 4.3 GB for an 8192-wide image over a 1-pixel map, which the reference allocates per call and frees.  Tables up
// to kIdwCacheMaxScale (1 MiB) are built once per device and kept; larger ones live for the call (*transient: the caller frees it
// once its launches have finished).  The table is built outside the lock; running out of memory is a status, not an exception.
constexpr int kIdwCacheMaxScale = 128;
int idw_for_scale(DeviceState* st, int scale, const float** dptr, float** transient) {
  *transient = nullptr;
  if (scale <= kIdwCacheMaxScale) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = st->idw.find(scale);
    if (it != st->idw.end()) { *dptr = it->second; return UHDR_HIP_NO_ERROR; }
  }
  std::vector<float> t;
  try {
    build_idw_tables(scale, t);
  } catch (const std::bad_alloc&) {
    snprintf(t_err, sizeof(t_err), "no memory for the weight table of map scale %d", scale);
    return UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE;
  }
  float* d = nullptr;
  if (hipMalloc(&d, t.size() * sizeof(float)) != hipSuccess) {
    (void)hipGetLastError();
    snprintf(t_err, sizeof(t_err), "no device memory for the weight table of map scale %d", scale);
    return UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE;
  }
  if (hipMemcpy(d, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return UHDR_HIP_UNKNOWN_ERROR; }
  if (scale > kIdwCacheMaxScale) { *transient = d; *dptr = d; return UHDR_HIP_NO_ERROR; }
  std::lock_guard<std::mutex> lk(g_mu);
  auto ins = st->idw.emplace(scale, d);
  if (!ins.second) (void)hipFree(d);   // another thread was faster
  *dptr = ins.first->second;
  return UHDR_HIP_NO_ERROR;
}

// ---- colour constants (gainmapmath.cpp:121-248, 359-393, 447-481) ------------------------------
struct YuvRgb { float cr, gcb, gcr, cb; };
YuvRgb yuv_rgb_coeffs(int gamut) {
  // G coefficients are float (B*Cb)/G and (R*Cr)/G exactly as the reference's static initialisers
  switch (gamut) {
    case UHDR_HIP_CG_BT709: {
      const float R = 0.2126f, G = 0.7152f, B = 0.0722f, Cb = 1.8556f, Cr = 1.5748f;
      return {Cr, B * Cb / G, R * Cr / G, Cb};
    }
    case UHDR_HIP_CG_P3: {
      const float R = 0.299f, G = 0.587f, B = 0.114f, Cb = 1.772f, Cr = 1.402f;
      return {Cr, B * Cb / G, R * Cr / G, Cb};
    }
    default: {
      const float R = 0.2627f, G = 0.6780f, B = 0.0593f, Cb = 1.8814f, Cr = 1.4746f;
      return {Cr, B * Cb / G, R * Cr / G, Cb};
    }
  }
}
void luminance_coeffs(int gamut, float* o) {
  switch (gamut) {
    case UHDR_HIP_CG_BT709: o[0] = 0.2126f; o[1] = 0.7152f; o[2] = 0.0722f; break;
    case UHDR_HIP_CG_P3: o[0] = 0.20949f; o[1] = 0.72160f; o[2] = 0.06891f; break;
    default: o[0] = 0.2627f; o[1] = 0.6780f; o[2] = 0.0593f; break;
  }
}
const float kBt709ToP3[9] = {0.82254f, 0.17755f, 0.00006f, 0.03312f, 0.96684f, -0.00001f, 0.01706f, 0.07240f, 0.91049f};
const float kBt709ToBt2100[9] = {0.62740f, 0.32930f, 0.04332f, 0.06904f, 0.91958f, 0.01138f, 0.01636f, 0.08799f, 0.89555f};
const float kP3ToBt709[9] = {1.22482f, -0.22490f, -0.00007f, -0.04196f, 1.04199f, 0.00001f, -0.01961f, -0.07865f, 1.09831f};
const float kP3ToBt2100[9] = {0.75378f, 0.19862f, 0.04754f, 0.04576f, 0.94177f, 0.01250f, -0.00121f, 0.01757f, 0.98359f};
const float kBt2100ToBt709[9] = {1.66045f, -0.58764f, -0.07286f, -0.12445f, 1.13282f, -0.00837f, -0.01811f, -0.10057f, 1.11878f};
const float kBt2100ToP3[9] = {1.34369f, -0.28223f, -0.06135f, -0.06533f, 1.07580f, -0.01051f, 0.00283f, -0.01957f, 1.01679f};
// getHdrConversionFn(sdr_gamut, hdr_gamut): nullptr <=> identity
const float* hdr_conversion(int sdr, int hdr) {
  if (sdr == hdr) return nullptr;
  switch (sdr) {
    case UHDR_HIP_CG_BT709: return hdr == UHDR_HIP_CG_P3 ? kP3ToBt709 : kBt2100ToBt709;
    case UHDR_HIP_CG_P3: return hdr == UHDR_HIP_CG_BT709 ? kBt709ToP3 : kBt2100ToP3;
    default: return hdr == UHDR_HIP_CG_BT709 ? kBt709ToBt2100 : kP3ToBt2100;
  }
}
const float kYuv709To601[9] = {1.0f, 0.101579f, 0.196076f, 0.0f, 0.989854f, -0.110653f, 0.0f, -0.072453f, 0.983398f};
const float kYuv709To2100[9] = {1.0f, -0.016969f, 0.096312f, 0.0f, 0.995306f, -0.051192f, 0.0f, 0.011507f, 1.002637f};
const float kYuv601To709[9] = {1.0f, -0.118188f, -0.212685f, 0.0f, 1.018640f, 0.114618f, 0.0f, 0.075049f, 1.025327f};
const float kYuv601To2100[9] = {1.0f, -0.128245f, -0.115879f, 0.0f, 1.010016f, 0.061592f, 0.0f, 0.086969f, 1.029350f};
const float kYuv2100To709[9] = {1.0f, 0.018149f, -0.095132f, 0.0f, 1.004123f, 0.051267f, 0.0f, -0.011524f, 0.996782f};
const float kYuv2100To601[9] = {1.0f, 0.117887f, 0.105521f, 0.0f, 0.995211f, -0.059549f, 0.0f, -0.084085f, 0.976518f};

bool valid_gamut(int g) { return g >= UHDR_HIP_CG_BT709 && g <= UHDR_HIP_CG_BT2100; }
bool al(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

// ---- generate ------------------------------------------------------------------------------------
// checks of ultrahdr.cpp:189-202 + the switch defaults of :222-302, in the reference's order
int validate_generate(const uhdr_hip_image_t* yuv, const uhdr_hip_image_t* p010, int hdr_tf,
                      const uhdr_hip_metadata_t* md, const uhdr_hip_image_t* dest) {
  if (yuv == nullptr || p010 == nullptr || md == nullptr || dest == nullptr || yuv->data == nullptr ||
      yuv->chroma_data == nullptr || p010->data == nullptr || p010->chroma_data == nullptr)
    return UHDR_HIP_ERROR_BAD_PTR;
  if (yuv->width != p010->width || yuv->height != p010->height) return UHDR_HIP_ERROR_RESOLUTION_MISMATCH;
  if (yuv->colorGamut == UHDR_HIP_CG_UNSPECIFIED || p010->colorGamut == UHDR_HIP_CG_UNSPECIFIED)
    return UHDR_HIP_ERROR_INVALID_COLORGAMUT;
  if (hdr_tf != UHDR_HIP_TF_LINEAR && hdr_tf != UHDR_HIP_TF_HLG && hdr_tf != UHDR_HIP_TF_PQ)
    return UHDR_HIP_ERROR_INVALID_TRANS_FUNC;
  if (!valid_gamut(yuv->colorGamut) || !valid_gamut(p010->colorGamut)) return UHDR_HIP_ERROR_INVALID_COLORGAMUT;
  return UHDR_HIP_NO_ERROR;
}

void fill_generate_metadata(int hdr_tf, uhdr_hip_metadata_t* md) {  // ultrahdr.cpp:250-257
  const float white = hdr_tf == UHDR_HIP_TF_PQ ? 10000.0f : 1000.0f;
  memset(md->version, 0, sizeof(md->version));
  strcpy(md->version, "1.0");
  md->maxContentBoost = white / 203.0f;
  md->minContentBoost = 1.0f;
  md->gamma = 1.0f;
  md->offsetSdr = 0.0f;
  md->offsetHdr = 0.0f;
  md->hdrCapacityMin = 1.0f;
  md->hdrCapacityMax = md->maxContentBoost;
}

void fill_generate_dest(const uhdr_hip_image_t* yuv, uhdr_hip_image_t* dest) {  // ultrahdr.cpp:210-216
  dest->width = yuv->width / 4;
  dest->height = yuv->height / 4;
  dest->colorGamut = UHDR_HIP_CG_UNSPECIFIED;
  dest->luma_stride = dest->width;
  dest->chroma_data = nullptr;
  dest->chroma_stride = 0;
  dest->pixelFormat = UHDR_HIP_PIX_FMT_MONOCHROME;
}

// bytes encodeGain (gainmapmath.cpp:529-541) yields for a gain clamped to min / max, evaluated with the
// host libm exactly as the reference does, and the scale of the in-range fast path
void encode_constants(float min_boost, float max_boost, float l2min, float l2max, double* scale, uint32_t* bmin,
                      uint32_t* bmax) {
  const double den = (double)(l2max - l2min);
  *scale = (double)255.0f / den;
  *bmin = (uint8_t)((std::log2((double)min_boost) - (double)l2min) / den * (double)255.0f);
  *bmax = (uint8_t)((std::log2((double)max_boost) - (double)l2min) / den * (double)255.0f);
}

GenConsts generate_consts(int sdr_gamut, int hdr_gamut, int hdr_tf, int sdr_is_601, size_t w, size_t h,
                          const uhdr_hip_metadata_t& md) {
  GenConsts c;
  const YuvRgb s = yuv_rgb_coeffs(sdr_is_601 ? UHDR_HIP_CG_P3 : sdr_gamut);  // ultrahdr.cpp:267-286
  const YuvRgb hh = yuv_rgb_coeffs(hdr_gamut);                               // :288-302
  c.sdr_cr = s.cr; c.sdr_gcb = s.gcb; c.sdr_gcr = s.gcr; c.sdr_cb = s.cb;
  c.hdr_cr = hh.cr; c.hdr_gcb = hh.gcb; c.hdr_gcr = hh.gcr; c.hdr_cb = hh.cb;
  float l[3];
  luminance_coeffs(sdr_gamut, l);  // the SDR gamut's luminance is used for BOTH images (:324,330)
  c.lum_r = l[0]; c.lum_g = l[1]; c.lum_b = l[2];
  const float* gm = hdr_conversion(sdr_gamut, hdr_gamut);
  c.gm_identity = gm == nullptr;
  for (int i = 0; i < 9; ++i) c.gm[i] = gm ? gm[i] : (i % 4 == 0 ? 1.0f : 0.0f);
  c.hdr_white_nits = hdr_tf == UHDR_HIP_TF_PQ ? 10000.0f : 1000.0f;
  c.min_boost = md.minContentBoost;
  c.max_boost = md.maxContentBoost;
  c.log2_min = (float)std::log2((double)md.minContentBoost);  // ultrahdr.cpp:259-260
  c.log2_max = (float)std::log2((double)md.maxContentBoost);
  encode_constants(c.min_boost, c.max_boost, c.log2_min, c.log2_max, &c.enc_scale, &c.enc_byte_min, &c.enc_byte_max);
  c.width = (uint32_t)w; c.height = (uint32_t)h;
  c.map_w = (uint32_t)(w / 4); c.map_h = (uint32_t)(h / 4);
  c.stat_keys = nullptr;
  c.stat_stride = 2;
  c.stat_ws = nullptr;
  c.stat_out = nullptr;
  c.stat_spread = 0u;
  c.stat_slots = 0u;
  c.lut = nullptr;
  c.bias4096 = 4096.0f;
  // f32 pre-filter (gen_pair; error budget in DESIGN.md section 5): the fast gain is within kRel of the exact one
  // (3.3e-6 by analysis on top of the exhaustively measured transfer-function errors), v_log_f32 within kLogAbs of
  // log2 on [0.25, 64]; the float evaluation of the code value adds < 2.5e-5.  x1.25 on top.
  // PQ: the f32 inverse OETF (pq_inv_oetf_fast) is within 3.4e-6 instead of 3.4e-7 (the outer power multiplies every error by 6.28):
  // HDR luminance <= 1.82 x 3.4e-6 + 1.5e-6, ratio <= 9e-6.  Its code scale is smaller (log2 range 5.6 instead of 2.3), so the
  // distance in code units comes out the same.
  const double kRel = hdr_tf == UHDR_HIP_TF_PQ ? 1.0e-5 : 4.0e-6, kLogAbs = 5.0e-7;
  c.flt_scale = (float)c.enc_scale;
  c.flt_delta = (float)(1.25 * (c.enc_scale * (kRel * 1.4426950408889634 + kLogAbs) + 4.0e-5));
  c.flt_lo = (float)((double)c.min_boost * (1.0 - 2.0 * kRel));
  c.flt_hi = (float)((double)c.max_boost * (1.0 + 2.0 * kRel));
  c.flt_gain_rel = (float)(2.0 * kRel);
  return c;
}

GenImage gen_image(const uhdr_hip_image_t& yuv, const uhdr_hip_image_t& p010, void* map) {
  GenImage g;
  g.y = static_cast<const uint8_t*>(yuv.data);
  g.u = static_cast<const uint8_t*>(yuv.chroma_data);
  g.hy = static_cast<const uint16_t*>(p010.data);
  g.huv = static_cast<const uint16_t*>(p010.chroma_data);
  g.map = static_cast<uint8_t*>(map);
  g.y_stride = (uint32_t)yuv.luma_stride;
  g.c_stride = (uint32_t)yuv.chroma_stride;
  g.hy_stride = (uint32_t)(p010.luma_stride == 0 ? p010.width : p010.luma_stride);  // gainmapmath.cpp:585
  g.huv_stride = (uint32_t)p010.chroma_stride;
  return g;
}
bool gen_aligned(const GenImage& g, uint32_t w, uint32_t h) {
  const uint8_t* v = g.u + (size_t)g.c_stride * (h / 2u);
  if ((uint64_t)g.hy_stride * h >= (1ull << 31) || (uint64_t)g.y_stride * h >= (1ull << 32)) return false;  // 32-bit offsets
  return (w % 8u == 0) && al(g.hy, 16) && g.hy_stride % 8u == 0 && al(g.huv, 16) && g.huv_stride % 8u == 0 &&
         al(g.y, 8) && g.y_stride % 8u == 0 && al(g.u, 4) && al(v, 4) && g.c_stride % 4u == 0 && al(g.map, 2);
}

// ---- FAST apply's line-segment tables (uhdr_kernels.h, k_apply_s4) ---------------------------------
// An entry is the chord of the function over its cell, lowered by half its largest deviation, as (c0, c1): f ~ c0 + c1 * argument.
template <class F>
void chord(F f, double lo, double hi, double* c0, double* c1) {
  *c1 = (f(hi) - f(lo)) / (hi - lo);
  *c0 = f(lo) - *c1 * lo;
  double dev = 0.0;
  for (int k = 1; k < 32; ++k) { const double x = lo + (hi - lo) * k / 32.0; const double d = f(x) - (*c0 + *c1 * x); if (std::fabs(d) > std::fabs(dev)) dev = d; }
  *c0 += 0.5 * dev;
}
// Stage 1: T(c) = srgbInvOetf(c)^g (gainmapmath.cpp:149-155).  Cell j holds the arguments whose half-precision conversion, rounded
// toward zero, has the bits j << 3 | 0..7: e = j >> 7, m = j & 127: [2^(e-15) (1 + m/128), 2^(e-15) (1 + (m+1)/128)) for e >= 1, the
// subnormal slots [m, m+1) 2^-21 for e = 0 (no input of the kernel other than 0 falls below 2^-17: (y + k dv) / 255).
void f16_cell(uint32_t j, double* lo, double* hi) {
  const uint32_t e = j >> 7, m = j & 127u;
  if (e == 0) { *lo = std::ldexp((double)m, -21); *hi = std::ldexp((double)(m + 1u), -21); return; }
  *lo = std::ldexp(1.0 + m / 128.0, (int)e - 15);
  *hi = std::ldexp(1.0 + (m + 1u) / 128.0, (int)e - 15);
}
// ... and for g < 1 cell j = (e & 31) << 4 | m with e the biased exponent and m the top four mantissa bits:
// [2^(e-127) (1 + m/16), 2^(e-127) (1 + (m+1)/16)), e = 96 .. 127.  Five exponent bits are enough: a channel is y / 255 plus chroma
// terms, operands of magnitude >= 2^-8, so a non-zero result is a multiple of 2^-31 -- the kernel's inputs are 0 or lie in
// [2^-31, 1].  Cell 0 serves the input 0 (and the sliver above 2^-31): a line through the origin.
void f32_cell16(uint32_t j, double* lo, double* hi) {
  const uint32_t e = 96u + (j >> 4), m = j & 15u;
  *lo = std::ldexp(1.0 + m / 16.0, (int)e - 127);
  *hi = std::ldexp(1.0 + (m + 1u) / 16.0, (int)e - 127);
}
void build_stage1(double g, float* out) {
  const double thr = (double)0.04045f;
  auto eotf = [&](double x) { return x <= thr ? x / (double)12.92f : std::pow((x + (double)0.055f) / (double)1.055f, 2.4); };
  auto f = [&](double x) { return g == 1.0 ? eotf(x) : std::pow(eotf(x), g); };
  const uint32_t cells = (g == 1.0 ? kTabS1Bytes : kTabS1PowBytes) / 8u;
  for (uint32_t j = 0; j < cells; ++j) {
    double lo, hi, c0, c1;
    if (g == 1.0) f16_cell(j, &lo, &hi); else f32_cell16(j, &lo, &hi);
    if (g == 1.0 && hi <= thr) { c0 = 0.0; c1 = 1.0 / (double)12.92f; }
    else if (lo >= 1.0) { c0 = 1.0; c1 = 0.0; }
    else if (j == 0) { c0 = 0.0; c1 = f(hi) / hi; }   // T(0) = 0 exactly
    else chord(f, lo, hi, &c0, &c1);
    out[2 * j] = (float)c0; out[2 * j + 1] = (float)c1;
  }
}
// Stage 2: the 10-bit code (colorToRgba1010102, gainmapmath.cpp:722-727: truncation of OETF * 1023) as a function of
// s = 2 + 2u, cell k = [2 + k/64, 2 + (k+1)/64), cell 128 = s >= 4 (u = 1, nothing above it is reached).  The entry evaluates
// to -(2 + code 2^-22); the kernel's fma rounds toward zero, i.e. truncates the code, and the bits of the result are
// 0xC0000000 | code.  c0 must then be a whole number of half codes (an ulp of [1, 2) is 2^-23), so its fraction is moved into the
// slope: that tilts the line by < 0.002 codes over a cell (the argument stays within 1/256 of the cell's middle, relatively).
// A line never evaluates below 0 (the result would leave the [2, 4) binade): cell 0 passes through (2, 0) exactly.
template <class F>
void build_stage2(F code_of_u, float* out) {
  for (uint32_t k = 0; k < kTabS2Cells; ++k) {
    const double lo = 2.0 + k / 64.0, hi = 2.0 + (k + 1u) / 64.0;
    auto f = [&](double s) { return code_of_u(std::min(1.0, 0.5 * (s - 2.0))); };
    float c0f = -2.0f, c1f = 0.0f;                       // a cell whose codes all truncate to 0
    if (k == kTabS2Cells - 1u) c0f = (float)(-2.0 - std::ldexp(1023.0, -22));
    else if (f(hi) >= 0.99) {
      double c0, c1;
      chord(f, lo, hi, &c0, &c1);
      c0 += 1.0 / 256.0;                                 // margin: the line stays above 0 after the steps below
      // -2 - n 2^-22 must be a float: n a whole number of codes when the magnitude is in [2, 4), of half codes when in [1, 2)
      const double n = c0 < 0.0 ? std::nearbyint(2.0 * c0) / 2.0 : std::nearbyint(c0);
      c1 += (c0 - n) / (0.5 * (lo + hi));
      c0f = (float)(-2.0 - std::ldexp(n, -22));
      c1f = (float)(-std::ldexp(c1, -22));
      auto at = [&](double s) { return -(((double)c1f * s + (double)c0f) + 2.0) * 4194304.0; };
      for (int it = 0; it < 4096 && at(lo) < 0.0; ++it) c1f = std::nextafter(c1f, -INFINITY);
      if (at(lo) < 0.0) { c0f = -2.0f; c1f = 0.0f; }
    }
    out[2 * k] = c0f; out[2 * k + 1] = c1f;
  }
}
void build_line_tables(std::vector<float>& tab) {
  tab.assign(kTabEnd - kTabS1Lin, 0.0f);
  const double m1 = (double)(2610.0f / 16384.0f), m2 = (double)(2523.0f / 4096.0f * 128.0f);
  const double k1 = (double)(3424.0f / 4096.0f), k2 = (double)(2413.0f / 4096.0f * 32.0f), k3 = (double)(2392.0f / 4096.0f * 32.0f);
  build_stage1(1.0, tab.data() + (kTabS1Lin - kTabS1Lin));
  build_stage1(0.5, tab.data() + (kTabS1Hlg - kTabS1Lin));
  build_stage1(m1, tab.data() + (kTabS1Pq - kTabS1Lin));
  // HLG OETF (gainmapmath.cpp:257-267) of x = u^2; PQ OETF (:305-314) of x = u^(1/m1), i.e. of p = u directly
  auto hlg = [](double u) { const double x = u * u; return 1023.0 * (x <= 1.0 / 12.0 ? std::sqrt(3.0) * u : (double)0.17883277f * std::log(12.0 * x - (double)0.28466892f) + (double)0.55991073f); };
  auto pq = [&](double p) { return p <= 0.0 ? 0.0 : 1023.0 * std::pow((k1 + k2 * p) / (1.0 + k3 * p), m2); };
  build_stage2(hlg, tab.data() + (kTabS2Hlg - kTabS1Lin));
  build_stage2(pq, tab.data() + (kTabS2Pq - kTabS1Lin));
}

// ---- apply ---------------------------------------------------------------------------------------
// checks of ultrahdr.cpp:364-406 in order
int validate_apply(const uhdr_hip_image_t* yuv, const uhdr_hip_image_t* map, const uhdr_hip_metadata_t* md,
                   const uhdr_hip_image_t* dest) {
  if (yuv == nullptr || map == nullptr || md == nullptr || dest == nullptr || yuv->data == nullptr ||
      yuv->chroma_data == nullptr || map->data == nullptr)
    return UHDR_HIP_ERROR_BAD_PTR;
  if (strncmp(md->version, "1.0", sizeof(md->version)) != 0) return UHDR_HIP_ERROR_BAD_METADATA;
  if (md->gamma != 1.0f) return UHDR_HIP_ERROR_BAD_METADATA;
  if (md->offsetSdr != 0.0f || md->offsetHdr != 0.0f) return UHDR_HIP_ERROR_BAD_METADATA;
  if (md->hdrCapacityMin != md->minContentBoost || md->hdrCapacityMax != md->maxContentBoost)
    return UHDR_HIP_ERROR_BAD_METADATA;
  if (map->width == 0 || map->height == 0) return UHDR_HIP_ERROR_UNSUPPORTED_MAP_SCALE_FACTOR;  // (ref: division by zero)
  if (yuv->width % map->width != 0 || yuv->height % map->height != 0)
    return UHDR_HIP_ERROR_UNSUPPORTED_MAP_SCALE_FACTOR;
  if (yuv->width * map->height != yuv->height * map->width) return UHDR_HIP_ERROR_UNSUPPORTED_MAP_SCALE_FACTOR;
  return UHDR_HIP_NO_ERROR;
}
bool apply_writes(int fmt) {
  return fmt == UHDR_HIP_OUTPUT_HDR_LINEAR || fmt == UHDR_HIP_OUTPUT_HDR_PQ || fmt == UHDR_HIP_OUTPUT_HDR_HLG ||
         fmt == UHDR_HIP_OUTPUT_HDR_LINEAR_RGB_10BIT;
}
size_t apply_bpp(int fmt) {
  return fmt == UHDR_HIP_OUTPUT_HDR_LINEAR ? 8 : fmt == UHDR_HIP_OUTPUT_HDR_LINEAR_RGB_10BIT ? 6 : 4;
}
AppConsts apply_consts(const uhdr_hip_image_t& yuv, const uhdr_hip_image_t& map, const uhdr_hip_metadata_t& md,
                       float max_display_boost, const float* idw) {
  AppConsts c;
  c.width = (uint32_t)yuv.width; c.height = (uint32_t)yuv.height;
  c.map_w = (uint32_t)map.width; c.map_h = (uint32_t)map.height;
  c.scale = (uint32_t)(yuv.width / map.width);                                    // ultrahdr.cpp:409
  c.display_boost = (std::min)(max_display_boost, md.maxContentBoost);            // :415
  c.inv_display_boost = 1.0f / c.display_boost;
  c.max_boost = md.maxContentBoost;
  c.inv_max_boost = 1.0f / md.maxContentBoost;
  c.log2_min_d = std::log2((double)md.minContentBoost);                           // gainmapmath.cpp:551-552
  c.log2_max_d = std::log2((double)md.maxContentBoost);
  c.idw = idw;
  c.lut = nullptr;
  c.lut_boost_factor = c.display_boost > 0 ? c.display_boost / md.maxContentBoost : 1.0f;  // gainmapmath.h:162
  // LUT-mode apply divides (sRGB table value x GainLUT entry) by display_boost.  The table values are 0 or lie in [7e-5, 1], the
  // GainLUT entries between 2^(log2 min x factor) and 2^(log2 max x factor): with the divisor in [2^-20, 2^20] and those exponents
  // within +-40 no operand, quotient or remainder of the division's IEEE expansion leaves the normal range, v_div_scale_f32 rescales
  // nothing and the expansion can run as it stands, two quotients per instruction (k_apply_lut_s4: lut_cell_pk).
  // The quotient is then at most 2^(the larger exponent) / display_boost; below 32768 its index product into a 65536-entry table
  // stays under 2^31 (lut_index_pos: no test for the wild range).
  const double lut_top = std::exp2(std::fmax(c.log2_min_d, c.log2_max_d) * (double)c.lut_boost_factor) / (double)c.display_boost;
  c.lut_plain_div = (c.display_boost >= 0x1p-20f && c.display_boost <= 0x1p20f && std::fabs(c.log2_min_d * (double)c.lut_boost_factor) <= 40.0 &&
                     std::fabs(c.log2_max_d * (double)c.lut_boost_factor) <= 40.0 && lut_top <= 32768.0) ? 1u : 0u;
  // FAST scale-4 kernel: factor/display_boost = 2^(gain*A + B); weights pre-multiplied by A / 255 (k_apply_s4)
  const double ratio = (double)c.display_boost / (double)md.maxContentBoost;
  c.fast.A = (float)((c.log2_max_d - c.log2_min_d) * ratio);
  c.fast.B = (float)(c.log2_min_d * ratio - std::log2((double)c.display_boost));
  c.fast.A255 = (float)((c.log2_max_d - c.log2_min_d) * ratio / 255.0);
  std::vector<float> t;
  build_idw_tables(4, t);
  for (int oy = 0; oy < 4; ++oy)
    for (int pr = 0; pr < 2; ++pr)
      for (int k = 1; k < 4; ++k)
        for (int j = 0; j < 2; ++j)
          c.fast.wD[oy][pr][k - 1][j] = t[oy * 16 + (2 * pr + j) * 4 + k];   // the weight itself: launch_apply_t multiplies by the final A / 255
  c.tab = nullptr;
  c.ex_ws = nullptr;
  c.ex_cap = 0;
  c.cells_per_thread = 8;
  c.step_x = c.step_y = 0;
  return c;
}
AppImage app_image(const uhdr_hip_image_t& yuv, const uhdr_hip_image_t& map, void* dst) {
  AppImage a;
  a.y = static_cast<const uint8_t*>(yuv.data);
  a.u = static_cast<const uint8_t*>(yuv.chroma_data);
  a.v = a.u + yuv.chroma_stride * (yuv.height / 2);
  a.map = static_cast<const uint8_t*>(map.data);
  a.dst = dst;
  a.y_stride = (uint32_t)yuv.luma_stride;
  a.c_stride = (uint32_t)yuv.chroma_stride;
  return a;
}
bool app_fast_s4(const AppConsts& c, const AppImage& a) {
  // the fast kernels index planes with 32-bit byte offsets (the widest output is 8 bytes per pixel)
  if ((uint64_t)a.y_stride * c.height >= (1ull << 32) || (uint64_t)c.width * c.height >= (1ull << 29)) return false;
  return c.scale == 4 && c.width == 4u * c.map_w && c.height == 4u * c.map_h && al(a.y, 4) && a.y_stride % 4u == 0 &&
         al(a.u, 2) && al(a.v, 2) && a.c_stride % 2u == 0 && al(a.dst, 16);
}
void fill_apply_dest(const uhdr_hip_image_t* yuv, uhdr_hip_image_t* dest) {  // ultrahdr.cpp:411-413
  dest->width = yuv->width;
  dest->height = yuv->height;
  dest->colorGamut = yuv->colorGamut;
}

// ---- host staging ----------------------------------------------------------------------------------
int stage_reserve(StageSet* st, int slot, size_t bytes) {
  if (bytes == 0) bytes = 256;
  if (st->stage_bytes[slot] >= bytes) return UHDR_HIP_NO_ERROR;
  if (st->stage[slot]) HIP_TRY(hipFree(st->stage[slot]));
  st->stage[slot] = nullptr;
  st->stage_bytes[slot] = 0;
  HIP_TRY(hipMalloc(&st->stage[slot], bytes));
  st->stage_bytes[slot] = bytes;
  return UHDR_HIP_NO_ERROR;
}
size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// one staging set for the duration of a host-memory call
class StageLease {
 public:
  explicit StageLease(DeviceState* st) : st_(st) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (st_->free_sets.empty()) {
      try {
        st_->sets.emplace_back(new StageSet());
        set_ = st_->sets.back().get();
      } catch (const std::bad_alloc&) {
        set_ = nullptr;   // (the caller returns INSUFFICIENT_RESOURCE: nothing may throw through the C boundary)
      }
    } else {
      set_ = st_->free_sets.back();
      st_->free_sets.pop_back();
    }
  }
  ~StageLease() {
    if (set_ == nullptr) return;
    std::lock_guard<std::mutex> lk(g_mu);
    try { st_->free_sets.push_back(set_); } catch (const std::bad_alloc&) {}   // (the set stays owned by st_->sets)
  }
  StageLease(const StageLease&) = delete;
  StageLease& operator=(const StageLease&) = delete;
  StageSet* get() const { return set_; }   // nullptr: out of host memory
 private:
  DeviceState* st_;
  StageSet* set_;
};

// A codec call's private context.  Round 2 serialised every jpeg_* / jpegr_* call of a process behind two mutexes because the calls
// shared the device's staging slots, decoder pool and side stream; a JPEG decode is latency-bound (tens of synchronisation rounds
// of one lane's work each), so two callers on their own streams now overlap almost completely (tests/test_gpu_async.py).
class CodecLease {
 public:
  explicit CodecLease(DeviceState* st) : st_(st) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!st_->free_codec.empty()) {
      ctx_ = st_->free_codec.back();
      st_->free_codec.pop_back();
    } else {
      try {
        st_->free_codec.reserve(st_->codec_sets.size() + 1);   // (so that giving it back cannot fail)
        st_->codec_sets.emplace_back(new DeviceState());
        ctx_ = st_->codec_sets.back().get();
      } catch (const std::bad_alloc&) {
        ctx_ = nullptr;
      }
    }
    if (ctx_) { ctx_->ready = true; ctx_->lut = st_->lut; }
  }
  ~CodecLease() {
    if (ctx_ == nullptr) return;
    std::lock_guard<std::mutex> lk(g_mu);
    st_->free_codec.push_back(ctx_);
  }
  CodecLease(const CodecLease&) = delete;
  CodecLease& operator=(const CodecLease&) = delete;
  DeviceState* get() const { return ctx_; }

 private:
  DeviceState* st_;
  DeviceState* ctx_ = nullptr;
};

// copy `rows` rows of `row_elems` elements of `esz` bytes from a strided host plane into a device
// plane with pitch dpitch_elems.  Only bytes the reference itself would touch are read.
int h2d_plane(void* d, size_t dpitch_elems, const void* h, size_t hstride_elems, size_t row_elems, size_t rows,
              size_t esz, hipStream_t s) {
  if (rows == 0 || row_elems == 0) return UHDR_HIP_NO_ERROR;
  if (hstride_elems == 0) {  // degenerate stride: every row aliases row 0
    for (size_t r = 0; r < rows; ++r)
      HIP_TRY(hipMemcpyAsync(static_cast<char*>(d) + r * dpitch_elems * esz, h, row_elems * esz, hipMemcpyHostToDevice, s));
    return UHDR_HIP_NO_ERROR;
  }
  HIP_TRY(hipMemcpy2DAsync(d, dpitch_elems * esz, h, hstride_elems * esz, row_elems * esz, rows, hipMemcpyHostToDevice, s));
  return UHDR_HIP_NO_ERROR;
}
int d2h_plane(void* h, size_t hstride_elems, const void* d, size_t dpitch_elems, size_t row_elems, size_t rows,
              size_t esz, hipStream_t s) {
  if (rows == 0 || row_elems == 0) return UHDR_HIP_NO_ERROR;
  HIP_TRY(hipMemcpy2DAsync(h, hstride_elems * esz, d, dpitch_elems * esz, row_elems * esz, rows, hipMemcpyDeviceToHost, s));
  return UHDR_HIP_NO_ERROR;
}

// device copy of a host YUV420 image in slots [slot, slot+1]: Y then U|V (pitch = 64-aligned)
int stage_yuv420_in(StageSet* st, int slot, const uhdr_hip_image_t& h, uhdr_hip_image_t* d, hipStream_t s) {
  const size_t w = h.width, hh = h.height, cw = (w + 1) / 2, ch = (hh + 1) / 2;
  const size_t lp = round_up(w ? w : 1, 64), cp = round_up(cw ? cw : 1, 64);
  // the V plane must sit at u + cp*(hh/2) for the kernels (gainmapmath.cpp:568)
  int rc;
  if ((rc = stage_reserve(st, slot, lp * (hh ? hh : 1))) != 0) return rc;
  if ((rc = stage_reserve(st, slot + 1, cp * ((hh / 2) + ch + 1))) != 0) return rc;
  *d = h;
  d->data = st->stage[slot];
  d->chroma_data = st->stage[slot + 1];
  d->luma_stride = lp;
  d->chroma_stride = cp;
  if ((rc = h2d_plane(d->data, lp, h.data, h.luma_stride, w, hh, 1, s)) != 0) return rc;
  const uint8_t* hu = static_cast<const uint8_t*>(h.chroma_data);
  const uint8_t* hv = hu + h.chroma_stride * (hh / 2);
  uint8_t* du = static_cast<uint8_t*>(d->chroma_data);
  if ((rc = h2d_plane(du, cp, hu, h.chroma_stride, cw, ch, 1, s)) != 0) return rc;
  if ((rc = h2d_plane(du + cp * (hh / 2), cp, hv, h.chroma_stride, cw, ch, 1, s)) != 0) return rc;
  return UHDR_HIP_NO_ERROR;
}
int stage_p010_in(StageSet* st, int slot, const uhdr_hip_image_t& h, uhdr_hip_image_t* d, hipStream_t s) {
  const size_t w = h.width, hh = h.height, cw2 = ((w + 1) / 2) * 2, ch = (hh + 1) / 2;
  const size_t lp = round_up(w ? w : 1, 64), cp = round_up(cw2 ? cw2 : 2, 64);
  int rc;
  if ((rc = stage_reserve(st, slot, lp * (hh ? hh : 1) * 2)) != 0) return rc;
  if ((rc = stage_reserve(st, slot + 1, cp * (ch ? ch : 1) * 2)) != 0) return rc;
  *d = h;
  d->data = st->stage[slot];
  d->chroma_data = st->stage[slot + 1];
  d->luma_stride = lp;
  d->chroma_stride = cp;
  const size_t hls = h.luma_stride == 0 ? h.width : h.luma_stride;  // gainmapmath.cpp:585
  if ((rc = h2d_plane(d->data, lp, h.data, hls, w, hh, 2, s)) != 0) return rc;
  if ((rc = h2d_plane(d->chroma_data, cp, h.chroma_data, h.chroma_stride, cw2, ch, 2, s)) != 0) return rc;
  return UHDR_HIP_NO_ERROR;
}

}  // namespace

// =====================================================================================================
extern "C" {

int uhdr_hip_abi_version(void) { return UHDR_HIP_ABI_VERSION; }

int uhdr_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

const char* uhdr_hip_last_error(void) { return t_err; }

int uhdr_hip_init(int device) {
  if (device < 0 || device >= uhdr_hip_device_count()) {
    snprintf(t_err, sizeof(t_err), "uhdr_hip_init: no HIP device %d (this library has no CPU path)", device);
    return UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE;
  }
  HIP_TRY(hipSetDevice(device));
  std::lock_guard<std::mutex> lk(g_mu);
  DeviceState& st = g_dev[device];
  if (!st.ready) {
    std::vector<float> t;
    build_idw_tables(4, t);
    HIP_TRY(upload_idw4(t.data()));
    // the reference fills its LUTs at static-initialisation time (gainmapmath.cpp:21-64); here: once per device
    HIP_TRY(hipMalloc(&st.lut, sizeof(float) * kLutBufferFloats));
    HIP_TRY(launch_build_luts(st.lut, nullptr));
    {
      std::vector<float> tab;
      build_line_tables(tab);
      HIP_TRY(hipMemcpy(st.lut + kTabS1Lin, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    HIP_TRY(launch_build_lut_codes(st.lut, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    HIP_TRY(jpeg::upload_tables());
    st.ready = true;
  }
  return UHDR_HIP_NO_ERROR;
}

// the workspaces device-memory calls keep per stream (uhdr_hip_stream_reserve / _release hand them out ahead of time / take them back)
int stat_workspace(DeviceState* st, hipStream_t s, uint32_t** out) {
  std::lock_guard<std::mutex> lk(g_mu);
  uint32_t** wp = nullptr;
  try { wp = &st->stat_ws[s]; } catch (const std::bad_alloc&) { return UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE; }
  uint32_t*& w = *wp;
  if (w == nullptr) {   // cleared once, in stream order: every launch leaves the headers cleared behind it
    uint32_t* fresh = nullptr;
    HIP_TRY(hipMalloc(&fresh, kStatWsBytes));
    const hipError_t me = hipMemsetAsync(fresh, 0, kStatWsBytes, s);
    if (me != hipSuccess) {   // never publish a workspace whose headers were not cleared
      (void)hipFree(fresh);
      set_err("hipMemsetAsync(statistics workspace)", me);
      return UHDR_HIP_UNKNOWN_ERROR;
    }
    w = fresh;
  }
  *out = w;
  return UHDR_HIP_NO_ERROR;
}
int exact_workspace(DeviceState* st, hipStream_t s, int images, uint32_t cap, uint32_t** out) {
  const size_t need = ((size_t)kMaxChunk * kExHdrWords + (size_t)images * kExLists * cap) * 4u;
  std::lock_guard<std::mutex> lk(g_mu);
  DeviceState::ExWs* wp = nullptr;
  try {
    wp = &st->ex_ws[s];
    if (wp->bytes < need && wp->p) st->retired.reserve(st->retired.size() + 1);
  } catch (const std::bad_alloc&) {
    return UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE;
  }
  DeviceState::ExWs& w = *wp;
  if (w.bytes < need) {
    // another caller of this stream may be about to launch with the old one: it stays allocated (sizes at least double)
    if (w.p) { st->retired.emplace_back(s, static_cast<void*>(w.p)); w.p = nullptr; }
    const size_t grown = std::max(need, 2 * w.bytes);
    w.bytes = 0;
    uint32_t* fresh = nullptr;
    HIP_TRY(hipMalloc(&fresh, grown));
    const hipError_t me = hipMemsetAsync(fresh, 0, (size_t)kMaxChunk * kExHdrWords * 4u, s);   // the headers: cleared once, left cleared by every launch
    if (me != hipSuccess) {                                                                    // (never published uncleared)
      (void)hipFree(fresh);
      set_err("hipMemsetAsync(EXACT workspace)", me);
      return UHDR_HIP_UNKNOWN_ERROR;
    }
    w.p = fresh;
    w.bytes = grown;
  }
  *out = w.p;
  return UHDR_HIP_NO_ERROR;
}

int uhdr_hip_stream_reserve(void* stream, int exact_images, size_t width, size_t height, int map_scale_factor) {
  DeviceState* st = nullptr;
  int rc0 = current_state(&st);
  if (rc0 != UHDR_HIP_NO_ERROR) return rc0;
  if (exact_images < 0 || map_scale_factor < 0 || (exact_images > 0 && (width == 0 || height == 0 || (uint64_t)width * height > 0xFFFFFFFFull)))
    return UHDR_HIP_ERROR_RESOLUTION_MISMATCH;
  hipStream_t s = static_cast<hipStream_t>(stream);
  uint32_t* w = nullptr;
  int rc = stat_workspace(st, s, &w);
  if (rc != UHDR_HIP_NO_ERROR) return rc;
  if (exact_images > 0) {
    const int m = exact_images < kMaxChunk ? exact_images : kMaxChunk;
    if ((rc = exact_workspace(st, s, m, ex_list_cap((uint64_t)width * height), &w)) != UHDR_HIP_NO_ERROR) return rc;
  }
  if (map_scale_factor > 0) {
    const float* idw = nullptr;
    float* transient = nullptr;
    if ((rc = idw_for_scale(st, map_scale_factor, &idw, &transient)) != UHDR_HIP_NO_ERROR) return rc;
    if (transient) { (void)hipStreamSynchronize(s); (void)hipFree(transient); }   // (a table too large to keep: nothing to reserve)
  }
  HIP_TRY(hipStreamSynchronize(s));   // the clears are done: what follows on the stream only enqueues
  return UHDR_HIP_NO_ERROR;
}

int uhdr_hip_stream_release(void* stream) {
  DeviceState* st = nullptr;
  const int rc0 = current_state(&st);
  if (rc0 != UHDR_HIP_NO_ERROR) return rc0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // the pair lock first, THEN the wait: a generate / EXACT-apply pair another thread enqueues on `s` holds g_pair_mu from the moment
  // it takes its workspace until its kernels are enqueued, so whatever names a workspace of `s` is on the stream before the wait
  std::lock_guard<std::mutex> pl(g_pair_mu);
  HIP_TRY(hipStreamSynchronize(s));
  std::lock_guard<std::mutex> lk(g_mu);
  auto a = st->stat_ws.find(s);
  if (a != st->stat_ws.end()) { if (a->second) (void)hipFree(a->second); st->stat_ws.erase(a); }
  auto b = st->ex_ws.find(s);
  if (b != st->ex_ws.end()) { if (b->second.p) (void)hipFree(b->second.p); st->ex_ws.erase(b); }
  // ... and the lists of this stream that EXACT launches outgrew (kept allocated until nothing on the stream can name them: now)
  size_t keep = 0;
  for (size_t i = 0; i < st->retired.size(); ++i) {
    if (st->retired[i].first == s) (void)hipFree(st->retired[i].second);
    else st->retired[keep++] = st->retired[i];
  }
  st->retired.resize(keep);
  return UHDR_HIP_NO_ERROR;
}

// ---- placement pools (include/uhdr_hip.h: "where resident images lie in device memory") ---------------------------------------
// Chunks are created one after another, so on a device whose memory is mostly free they walk through it; an allocation takes chunks
// spaced evenly over the free ones in that order, i.e. it is spread over the whole pool and interleaves with its neighbours.
struct uhdr_hip_mem_pool {
  int device = 0;
  size_t chunk = 0;
  std::vector<hipMemGenericAllocationHandle_t> handle;
  std::vector<uint8_t> state;   // 0 free, 1 in an allocation, 2 given back to the device (trim)
  struct Range { void* va; size_t chunks; std::vector<size_t> member; };
  std::vector<Range> ranges;
  std::mutex mu;
};
namespace {
struct DeviceGuard {   // the pool's device is current inside a call and the caller's again afterwards
  int prev = -1;
  bool ok = false;
  explicit DeviceGuard(int dev) { ok = hipGetDevice(&prev) == hipSuccess && (prev == dev || hipSetDevice(dev) == hipSuccess); if (prev == dev) prev = -1; }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
void pool_unmap(uhdr_hip_mem_pool* p, uhdr_hip_mem_pool::Range& r) {
  (void)hipMemUnmap(r.va, r.chunks * p->chunk);
  (void)hipMemAddressFree(r.va, r.chunks * p->chunk);
  for (size_t k : r.member) p->state[k] = 0;
}
}  // namespace

int uhdr_hip_mem_pool_create(int device, size_t bytes, size_t chunk_bytes, uhdr_hip_mem_pool_t** pool) {
  if (pool == nullptr) return UHDR_HIP_ERROR_BAD_PTR;
  *pool = nullptr;
  if (chunk_bytes == 0) chunk_bytes = (size_t)16 << 20;
  if (bytes == 0 || chunk_bytes % ((size_t)2 << 20) != 0) return UHDR_HIP_ERROR_UNSUPPORTED_FEATURE;
  if (device < 0 || device >= uhdr_hip_device_count()) return UHDR_HIP_ERROR_UNSUPPORTED_FEATURE;
  DeviceGuard g(device);
  if (!g.ok) { snprintf(t_err, sizeof(t_err), "uhdr_hip_mem_pool_create: cannot make device %d current", device); return UHDR_HIP_UNKNOWN_ERROR; }
  std::unique_ptr<uhdr_hip_mem_pool> p(new (std::nothrow) uhdr_hip_mem_pool());
  if (!p) return UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE;
  p->device = device;
  p->chunk = chunk_bytes;
  const size_t n = (bytes + chunk_bytes - 1) / chunk_bytes;
  size_t dev_free = 0, dev_total = 0;
  HIP_TRY(hipMemGetInfo(&dev_free, &dev_total));
  if (n > dev_free / chunk_bytes) return UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE;   // (before a single chunk is taken)
  try { p->handle.reserve(n); p->state.reserve(n); } catch (const std::bad_alloc&) { return UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE; }
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = device;
  for (size_t i = 0; i < n; ++i) {
    hipMemGenericAllocationHandle_t h;
    const hipError_t e = hipMemCreate(&h, chunk_bytes, &prop, 0);
    if (e != hipSuccess) {
      set_err("hipMemCreate(pool chunk)", e);
      for (auto& hh : p->handle) (void)hipMemRelease(hh);
      return e == hipErrorOutOfMemory ? UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE : UHDR_HIP_UNKNOWN_ERROR;
    }
    p->handle.push_back(h);
    p->state.push_back(0);
  }
  *pool = p.release();
  return UHDR_HIP_NO_ERROR;
}

int uhdr_hip_mem_pool_alloc(uhdr_hip_mem_pool_t* p, size_t bytes, void** ptr) {
  if (p == nullptr || ptr == nullptr) return UHDR_HIP_ERROR_BAD_PTR;
  *ptr = nullptr;
  if (bytes == 0) return UHDR_HIP_ERROR_UNSUPPORTED_FEATURE;
  std::lock_guard<std::mutex> lk(p->mu);
  DeviceGuard g(p->device);
  if (!g.ok) return UHDR_HIP_UNKNOWN_ERROR;
  const size_t m = (bytes + p->chunk - 1) / p->chunk;
  std::vector<size_t> freec;
  uhdr_hip_mem_pool::Range r;
  try {
    for (size_t i = 0; i < p->state.size(); ++i) if (p->state[i] == 0) freec.push_back(i);
    if (freec.size() < m) return UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE;
    // member j = the free chunk at position (j + 1/2) F / m: evenly spaced over the free ones, distinct because F >= m
    for (size_t j = 0; j < m; ++j) r.member.push_back(freec[(size_t)(((2 * (unsigned long long)j + 1) * freec.size()) / (2 * (unsigned long long)m))]);
    p->ranges.reserve(p->ranges.size() + 1);
  } catch (const std::bad_alloc&) { return UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE; }
  r.chunks = m;
  r.va = nullptr;
  HIP_TRY(hipMemAddressReserve(&r.va, m * p->chunk, 0, nullptr, 0));
  size_t mapped = 0;
  hipError_t e = hipSuccess;
  for (; mapped < m; ++mapped) {
    e = hipMemMap(static_cast<char*>(r.va) + mapped * p->chunk, p->chunk, 0, p->handle[r.member[mapped]], 0);
    if (e != hipSuccess) break;
  }
  if (e == hipSuccess) {
    hipMemAccessDesc a = {};
    a.location.type = hipMemLocationTypeDevice;
    a.location.id = p->device;
    a.flags = hipMemAccessFlagsProtReadWrite;
    e = hipMemSetAccess(r.va, m * p->chunk, &a, 1);
  }
  if (e != hipSuccess) {
    set_err("hipMemMap / hipMemSetAccess(pool allocation)", e);
    if (mapped) (void)hipMemUnmap(r.va, mapped * p->chunk);
    (void)hipMemAddressFree(r.va, m * p->chunk);
    return UHDR_HIP_UNKNOWN_ERROR;
  }
  for (size_t k : r.member) p->state[k] = 1;
  *ptr = r.va;
  p->ranges.push_back(std::move(r));
  return UHDR_HIP_NO_ERROR;
}

int uhdr_hip_mem_pool_free(uhdr_hip_mem_pool_t* p, void* ptr) {
  if (p == nullptr || ptr == nullptr) return UHDR_HIP_ERROR_BAD_PTR;
  std::lock_guard<std::mutex> lk(p->mu);
  DeviceGuard g(p->device);
  if (!g.ok) return UHDR_HIP_UNKNOWN_ERROR;
  for (size_t i = 0; i < p->ranges.size(); ++i)
    if (p->ranges[i].va == ptr) {
      pool_unmap(p, p->ranges[i]);
      p->ranges.erase(p->ranges.begin() + (long)i);
      return UHDR_HIP_NO_ERROR;
    }
  return UHDR_HIP_ERROR_UNSUPPORTED_FEATURE;
}

int uhdr_hip_mem_pool_trim(uhdr_hip_mem_pool_t* p) {
  if (p == nullptr) return UHDR_HIP_ERROR_BAD_PTR;
  std::lock_guard<std::mutex> lk(p->mu);
  DeviceGuard g(p->device);
  if (!g.ok) return UHDR_HIP_UNKNOWN_ERROR;
  for (size_t i = 0; i < p->state.size(); ++i)
    if (p->state[i] == 0) { (void)hipMemRelease(p->handle[i]); p->state[i] = 2; }
  return UHDR_HIP_NO_ERROR;
}

int uhdr_hip_mem_pool_stats(uhdr_hip_mem_pool_t* p, size_t* chunks, size_t* free_chunks) {
  if (p == nullptr) return UHDR_HIP_ERROR_BAD_PTR;
  std::lock_guard<std::mutex> lk(p->mu);
  size_t held = 0, fr = 0;
  for (uint8_t s : p->state) { held += s != 2; fr += s == 0; }
  if (chunks) *chunks = held;
  if (free_chunks) *free_chunks = fr;
  return UHDR_HIP_NO_ERROR;
}

int uhdr_hip_mem_pool_destroy(uhdr_hip_mem_pool_t* p) {
  if (p == nullptr) return UHDR_HIP_ERROR_BAD_PTR;
  {
    std::lock_guard<std::mutex> lk(p->mu);
    DeviceGuard g(p->device);
    if (!g.ok) return UHDR_HIP_UNKNOWN_ERROR;
    (void)hipDeviceSynchronize();
    for (auto& r : p->ranges) pool_unmap(p, r);
    p->ranges.clear();
    for (size_t i = 0; i < p->state.size(); ++i)
      if (p->state[i] != 2) (void)hipMemRelease(p->handle[i]);
  }
  delete p;
  return UHDR_HIP_NO_ERROR;
}

int uhdr_hip_shutdown(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  int prev = -1;
  (void)hipGetDevice(&prev);
  for (auto& kv : g_dev) {
    if (hipSetDevice(kv.first) != hipSuccess) continue;
    (void)hipDeviceSynchronize();
    for (auto& t : kv.second.idw) (void)hipFree(t.second);
    if (kv.second.lut) (void)hipFree(kv.second.lut);
    for (void* q : kv.second.pool) if (q) (void)hipFree(q);
    for (auto& w : kv.second.stat_ws) if (w.second) (void)hipFree(w.second);
    for (auto& w : kv.second.ex_ws) if (w.second.p) (void)hipFree(w.second.p);
    for (int i = 0; i < 14; ++i)
      if (kv.second.stage[i]) (void)hipFree(kv.second.stage[i]);
    for (auto& set : kv.second.sets)
      for (int i = 0; i < 14; ++i)
        if (set->stage[i]) (void)hipFree(set->stage[i]);
    for (auto& q : kv.second.retired) (void)hipFree(q.second);
    if (kv.second.map_ready) (void)hipEventDestroy(kv.second.map_ready);
    if (kv.second.aux) (void)hipStreamDestroy(kv.second.aux);
    for (auto& cx : kv.second.codec_sets) {   // the leased codec contexts (their tables are this state's: not freed here)
      for (void* q : cx->pool) if (q) (void)hipFree(q);
      for (int i = 0; i < 14; ++i)
        if (cx->stage[i]) (void)hipFree(cx->stage[i]);
      if (cx->map_ready) (void)hipEventDestroy(cx->map_ready);
      if (cx->aux) (void)hipStreamDestroy(cx->aux);
    }
  }
  g_dev.clear();
  if (prev >= 0) (void)hipSetDevice(prev);
  return UHDR_HIP_NO_ERROR;
}

int uhdr_hip_synth_lcg_frame(size_t width, size_t height, unsigned int seed, void* p010, void* yuv, void* stream) {
  if (p010 == nullptr || yuv == nullptr) return UHDR_HIP_ERROR_BAD_PTR;
  if ((width | height) & 1u) return UHDR_HIP_ERROR_UNSUPPORTED_WIDTH_HEIGHT;
  const uint64_t n_luma = (uint64_t)width * height, n = n_luma + n_luma / 2u;
  if (n >= (1ull << 31)) return UHDR_HIP_ERROR_UNSUPPORTED_WIDTH_HEIGHT;   // state indices 2 i + 2 stay below 2^32
  DeviceState* st = nullptr;
  const int rc = current_state(&st);
  if (rc != UHDR_HIP_NO_ERROR) return rc;
  HIP_TRY(launch_synth_lcg(static_cast<uint16_t*>(p010), static_cast<uint8_t*>(yuv), (uint32_t)n_luma, (uint32_t)n, seed,
                           static_cast<hipStream_t>(stream)));
  return UHDR_HIP_NO_ERROR;
}

int uhdr_hip_eval_transfer(int fn, const float* in, float* out, size_t n, float min_boost, float max_boost,
                           void* stream) {
  if (in == nullptr || out == nullptr) return UHDR_HIP_ERROR_BAD_PTR;
  DeviceState* st = nullptr;
  const int rc = current_state(&st);
  if (rc != UHDR_HIP_NO_ERROR) return rc;
  EvalConsts ec;
  ec.min_boost = min_boost;
  ec.max_boost = max_boost;
  ec.log2_min = (float)std::log2((double)min_boost);
  ec.log2_max = (float)std::log2((double)max_boost);
  encode_constants(min_boost, max_boost, ec.log2_min, ec.log2_max, &ec.enc_scale, &ec.enc_byte_min, &ec.enc_byte_max);
  ec.lut = st->lut;
  ec.log2_min_d = std::log2((double)min_boost);
  ec.log2_max_d = std::log2((double)max_boost);
  HIP_TRY(launch_eval_transfer(fn, in, out, n, ec, static_cast<hipStream_t>(stream)));
  return UHDR_HIP_NO_ERROR;
}

}  // extern "C"

namespace {
// block geometry and quantisation tables of one image (jpeg_set_quality(q, TRUE), jpegencoderhelper.cpp:119-136)
void encode_job_tables(size_t w, size_t h, bool gray, int quality, jpeg::Job* jp) {
  jpeg::Job& j = *jp;
  memset(&j, 0, sizeof(j));
  j.gray = gray ? 1 : 0;
  j.ybw = (uint32_t)((w + 7) / 8); j.ybh = (uint32_t)((h + 7) / 8);
  j.mcus_x = (uint32_t)((w + 15) / 16);
  j.nblk = gray ? j.ybw * j.ybh : j.mcus_x * (uint32_t)((h + 15) / 16) * 6u;
  uint16_t qn[64];
  jpeg::quant_table(quality, false, qn); jpeg::zigzag_table(qn, j.q_lum);
  jpeg::quant_table(quality, true, qn); jpeg::zigzag_table(qn, j.q_chr);
  for (int i = 0; i < 64; ++i) {
    j.m_lum[i] = (uint32_t)((1ull << 32) / ((uint32_t)j.q_lum[i] << 3)) + 1u;
    j.m_chr[i] = (uint32_t)((1ull << 32) / ((uint32_t)j.q_chr[i] << 3)) + 1u;
  }
}
jpeg::Plane encode_plane(const uint8_t* p, size_t pw, size_t ph, size_t stride, bool pad) {
  jpeg::Plane q;
  q.p = p; q.w = (int)pw; q.h = (int)ph; q.stride = (int)stride; q.pad_cols = pad ? 1 : 0;
  q.aligned4 = (reinterpret_cast<uintptr_t>(p) % 4 == 0 && stride % 4 == 0) ? 1 : 0;
  return q;
}
}  // namespace

extern "C" {

// JpegEncoderHelper::compressImage (jpegencoderhelper.cpp:39-52) on the device
int uhdr_hip_jpeg_encode(const uhdr_hip_image_t* image, int quality, const void* icc, size_t icc_size, void* out,
                         size_t out_capacity, size_t* out_size, int mem_space, void* stream) {
  if (image == nullptr || out_size == nullptr || image->data == nullptr) return UHDR_HIP_ERROR_BAD_PTR;
  const bool gray = image->pixelFormat == UHDR_HIP_PIX_FMT_MONOCHROME;
  if (!gray && image->chroma_data == nullptr) return UHDR_HIP_ERROR_BAD_PTR;
  if (out == nullptr && out_capacity != 0) return UHDR_HIP_ERROR_BAD_PTR;
  const size_t w = image->width, h = image->height;
  if (w == 0 || h == 0 || w > 65500 || h > 65500) return UHDR_HIP_ERROR_RESOLUTION_MISMATCH;   // libjpeg's JPEG_MAX_DIMENSION
  if (!gray && ((w | h) & 1)) return UHDR_HIP_ERROR_RESOLUTION_MISMATCH;
  DeviceState* st = nullptr;
  int rc = current_state(&st);
  if (rc != UHDR_HIP_NO_ERROR) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  CodecLease lease(st);   // the encoder workspace: this call's own
  if ((st = lease.get()) == nullptr) return UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE;

  const size_t ls = image->luma_stride ? image->luma_stride : w;
  const size_t cs = gray ? 0 : image->chroma_stride;
  const size_t aw = (w + 15) / 16 * 16, acw = (w / 2 + 7) / 8 * 8;
  jpeg::Job j;
  encode_job_tables(w, h, gray, quality, &j);

  const uint8_t* py = static_cast<const uint8_t*>(image->data);
  const uint8_t* pu = static_cast<const uint8_t*>(image->chroma_data);
  size_t dls = ls, dcs = cs;
  if (mem_space != UHDR_HIP_MEM_DEVICE) {
    // stage exactly the bytes the reference would touch: w columns when it pads, the 16-aligned width otherwise
    const size_t ycols = ls < aw ? w : aw, ccols = cs < acw ? w / 2 : acw;
    dls = round_up(ycols, 64); dcs = round_up(ccols ? ccols : 1, 64);
    if ((rc = stage_reserve(st, 0, dls * h)) != 0) return rc;
    if ((rc = h2d_plane(st->stage[0], dls, py, ls, ycols, h, 1, s)) != 0) return rc;
    py = static_cast<const uint8_t*>(st->stage[0]);
    if (!gray) {
      if ((rc = stage_reserve(st, 1, dcs * h + 64)) != 0) return rc;
      uint8_t* du = static_cast<uint8_t*>(st->stage[1]);
      if ((rc = h2d_plane(du, dcs, pu, cs, ccols, h / 2, 1, s)) != 0) return rc;
      if ((rc = h2d_plane(du + dcs * (h / 2), dcs, pu + cs * h / 2, cs, ccols, h / 2, 1, s)) != 0) return rc;
      pu = du;
    }
  }
  auto plane = encode_plane;
  j.plane[0] = plane(py, w, h, dls, ls < aw);
  if (!gray) {
    const size_t v_off = mem_space != UHDR_HIP_MEM_DEVICE ? dcs * (h / 2) : cs * h / 2;   // chromaStride * height / 2 (:140)
    j.plane[1] = plane(pu, w / 2, h / 2, dcs, cs < acw);
    j.plane[2] = plane(pu + v_off, w / 2, h / 2, dcs, cs < acw);
  }

  std::vector<uint8_t> header;
  jpeg::build_header((int)w, (int)h, gray, quality, icc, icc_size, header);
  jpeg::Layout l;
  const size_t ws_bytes = jpeg::workspace_bytes(j.nblk, &l);
  if ((rc = stage_reserve(st, 7, ws_bytes)) != 0) return rc;
  uint8_t* ws = static_cast<uint8_t*>(st->stage[7]);
  uint8_t* dout = static_cast<uint8_t*>(out);
  size_t dcap = out_capacity;
  if (mem_space != UHDR_HIP_MEM_DEVICE) {   // worst case: every stream byte stuffed
    dcap = header.size() + 2 * l.stream_bytes + 2;
    if ((rc = stage_reserve(st, 5, dcap)) != 0) return rc;
    dout = static_cast<uint8_t*>(st->stage[5]);
  }
  if (dcap >= header.size()) HIP_TRY(hipMemcpyAsync(dout, header.data(), header.size(), hipMemcpyHostToDevice, s));
  HIP_TRY(jpeg::encode_async(j, l, ws, dout, dcap >= header.size() ? dcap : 0, header.size(), s));
  uint64_t total = 0;
  HIP_TRY(hipMemcpyAsync(&total, ws + l.totals + 8, 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  *out_size = (size_t)total;
  if (total > out_capacity) return UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE;
  if (mem_space != UHDR_HIP_MEM_DEVICE) {
    HIP_TRY(hipMemcpyAsync(out, dout, total, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
  }
  return UHDR_HIP_NO_ERROR;
}

// Diagnostics (host only, no GPU): the quantised coefficients of a PROGRESSIVE file after all of its scans, as the host-side
// entropy decoder hands them to the device (csrc/uhdr_jpeg_prog.cpp) -- blocks in MCU order, zigzag order inside a block, DC as the
// value.  tests/test_jpeg_progressive.py compares them with libjpeg's jpeg_read_coefficients.  Returns the number of blocks through
// *blocks; baseline files: UNSUPPORTED_FEATURE (their entropy decoding runs on the device).
int uhdr_hip_jpeg_progressive_coefficients(const void* jpeg, size_t jpeg_size, int16_t* coef, size_t capacity_blocks, size_t* blocks, int* width,
                                           int* height, int* gray) {
  if (jpeg == nullptr || blocks == nullptr || width == nullptr || height == nullptr || gray == nullptr) return UHDR_HIP_ERROR_BAD_PTR;
  jpeg::DecInfo info;
  const int prc = jpeg::parse_header(static_cast<const uint8_t*>(jpeg), jpeg_size, &info);
  if (prc == -3) return UHDR_HIP_UNKNOWN_ERROR;   // host allocation failed; INSUFFICIENT_RESOURCE is the size-probe answer (*blocks set)
  if (prc == -2 || (prc == 0 && !info.progressive)) return UHDR_HIP_ERROR_UNSUPPORTED_FEATURE;
  if (prc != 0) return UHDR_HIP_UNKNOWN_ERROR;
  *width = info.w; *height = info.h; *gray = info.gray;
  *blocks = info.coef.size() / 64u;
  if (coef == nullptr || capacity_blocks < *blocks) return UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE;
  memcpy(coef, info.coef.data(), info.coef.size() * sizeof(int16_t));
  int prev[3] = {0, 0, 0};   // differences back to values
  for (size_t b = 0; b < *blocks; ++b) {
    const int c = info.gray ? 0 : ((b % 6u) < 4u ? 0 : (int)(b % 6u) - 3);
    prev[c] += coef[b * 64u];
    coef[b * 64u] = (int16_t)prev[c];
  }
  return UHDR_HIP_NO_ERROR;
}

// JpegDecoderHelper::decompressImage(..., DECODE_TO_YCBCR) (jpegdecoderhelper.cpp:188-327) on the device
int uhdr_hip_jpeg_decode(const void* jpeg, size_t jpeg_size, void* out, size_t out_capacity, uhdr_hip_image_t* desc,
                         int mem_space, void* stream) {
  if (jpeg == nullptr || desc == nullptr) return UHDR_HIP_ERROR_BAD_PTR;
  // INSUFFICIENT_RESOURCE is this call's size-probe answer (out == NULL / capacity too small: the header parsed and *desc is
  // filled); every status returned before that point leaves *desc zeroed, and a host allocation failure inside the parser
  // (std::bad_alloc, -3: a progressive file's coefficient array) is reported as UNKNOWN_ERROR so that it cannot be taken for it
  memset(desc, 0, sizeof(*desc));
  jpeg::DecInfo info;
#ifdef UHDR_JD_TIMING
  const auto T0 = std::chrono::steady_clock::now();
#endif
  const int prc = jpeg::parse_header(static_cast<const uint8_t*>(jpeg), jpeg_size, &info);
#ifdef UHDR_JD_TIMING
  const auto T1 = std::chrono::steady_clock::now();
#endif
  if (prc == -3) return UHDR_HIP_UNKNOWN_ERROR;
  if (prc == -2) return UHDR_HIP_ERROR_UNSUPPORTED_FEATURE;
  if (prc != 0 || info.w <= 0 || info.h <= 0) return UHDR_HIP_UNKNOWN_ERROR;
  const size_t w = (size_t)info.w, h = (size_t)info.h;
  if (w > 8192 || h > 8192) return UHDR_HIP_ERROR_RESOLUTION_MISMATCH;   // kMaxWidth / kMaxHeight, jpegdecoderhelper.h:42-43
  const size_t luma = w * h, chroma = luma / 4, need = info.gray ? luma : luma + 2 * chroma;
  desc->data = out;
  desc->width = w; desc->height = h;
  desc->colorGamut = UHDR_HIP_CG_UNSPECIFIED;
  desc->luma_stride = w;
  desc->chroma_data = info.gray ? nullptr : static_cast<uint8_t*>(out) + luma;
  desc->chroma_stride = info.gray ? 0 : w / 2;
  desc->pixelFormat = info.gray ? UHDR_HIP_PIX_FMT_MONOCHROME : UHDR_HIP_PIX_FMT_YUV420;
  if (out == nullptr || out_capacity < need) return UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE;
  DeviceState* st = nullptr;
  int rc = current_state(&st);
  if (rc != UHDR_HIP_NO_ERROR) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  CodecLease lease(st);
  if ((st = lease.get()) == nullptr) return UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE;

  jpeg::DecLayout l;
  const size_t ws_bytes = jpeg::dec_workspace_bytes(info, &l);
  if ((rc = stage_reserve(st, 7, ws_bytes)) != 0) return rc;
  uint8_t* ws = static_cast<uint8_t*>(st->stage[7]);
  HIP_TRY(hipMemcpyAsync(ws + l.src, static_cast<const uint8_t*>(jpeg) + info.scan_offset, info.scan_bytes, hipMemcpyHostToDevice, s));
#ifdef UHDR_JD_TIMING
  const auto T2 = std::chrono::steady_clock::now();
#endif
  uint8_t* dout = static_cast<uint8_t*>(out);
  if (mem_space != UHDR_HIP_MEM_DEVICE) {
    if ((rc = stage_reserve(st, 5, need)) != 0) return rc;
    dout = static_cast<uint8_t*>(st->stage[5]);
  }
  jpeg::DecPlane planes[3];
  memset(planes, 0, sizeof(planes));
  auto mk = [](uint8_t* p, size_t pw, size_t ph) {
    jpeg::DecPlane q;
    q.p = p; q.w = (int)pw; q.h = (int)ph; q.stride = (int)pw;
    q.aligned8 = (reinterpret_cast<uintptr_t>(p) % 8 == 0 && pw % 8 == 0) ? 1 : 0;
    return q;
  };
  planes[0] = mk(dout, w, h);
  if (!info.gray) { planes[1] = mk(dout + luma, w / 2, h / 2); planes[2] = mk(dout + luma + chroma, w / 2, h / 2); }
  hipError_t herr = hipSuccess;
  if ((rc = stage_reserve(st, 11, jpeg::dec_batch_scratch_bytes(1, &l))) != 0) return rc;
  const jpeg::DecInfo* infos[1] = {&info};
  uint8_t* wss[1] = {ws};
  jpeg::DecPlane (*pl[1])[3] = {&planes};
  const int drc = jpeg::decode_device_batch(1, infos, &l, wss, pl, s, static_cast<uint8_t*>(st->stage[11]), &herr, nullptr);
#ifdef UHDR_JD_TIMING
  {
    const auto T3 = std::chrono::steady_clock::now();
    auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    fprintf(stderr, "[jd] parse %.0f  lock+reserve+src copy %.0f  device %.0f us\n", us(T0, T1), us(T1, T2), us(T2, T3));
  }
#endif
  if (drc > 0) { set_err("uhdr_hip_jpeg_decode", herr); return UHDR_HIP_UNKNOWN_ERROR; }
  if (drc < 0) { snprintf(t_err, sizeof(t_err), "uhdr_hip_jpeg_decode: corrupt entropy-coded data"); return UHDR_HIP_UNKNOWN_ERROR; }
  if (mem_space != UHDR_HIP_MEM_DEVICE) {
    HIP_TRY(hipMemcpyAsync(out, dout, need, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
  }
  return UHDR_HIP_NO_ERROR;
}

}  // extern "C"

namespace {
// what the host learns from one JPEG/R file before anything is launched: the checks and the parsing of decodeJPEGR up to its
// first decompressImage call (jpegr.cpp:655-699) plus what it reads from the decoders afterwards (XMP :756-760, ICC :796-801)
struct JpegrFile {
  const uint8_t* jpg[2] = {nullptr, nullptr};   // primary image, gain map
  size_t len[2] = {0, 0};
  jpeg::DecInfo info[2];
  uhdr_hip_metadata_t md;
  int gamut = UHDR_HIP_CG_UNSPECIFIED;
};
// The image that starts at `begin`, found through its header: the header parser's walk over the (only) scan of a baseline file
// ends at EOI, and that is where the image ends.  Anything else -- more scans, other markers behind the scan, a header this decoder
// does not read -- is left to the container's own marker walk (jpegr::find_images), which then walks the scan a second time.
bool image_by_header(const uint8_t* file, size_t n, size_t begin, jpeg::DecInfo* info, size_t* len) {
  if (begin + 4 > n || jpeg::parse_header(file + begin, n - begin, info) != 0) return false;
  const size_t e = begin + info->scan_offset + info->scan_bytes;
  if (e + 2 > n || file[e] != 0xFF || file[e + 1] != 0xD9) return false;
  *len = e + 2 - begin;
  return true;
}

int parse_jpegr_file(const void* jpegr, size_t jpegr_size, int output_format, bool want_metadata, JpegrFile* f) {
  const uint8_t* file = static_cast<const uint8_t*>(jpegr);
  const bool sdr = output_format == UHDR_HIP_OUTPUT_SDR;   // the gain map is neither decompressed nor (unless asked for) read (:728, :754)
  jpegr::Range img[2];
  bool by_header = false;   // both ranges and both headers in one walk per image (the file the reference's encoder writes)
  bool have0 = false;       // the primary image at least (kept when the second image needs the container's walk: a progressive
                            // primary has had all its scans entropy-decoded by then, once is enough)
  size_t len0 = 0;
  if (file != nullptr && jpegr_size >= 4 && file[0] == 0xFF && file[1] == 0xD8 && image_by_header(file, jpegr_size, 0, &f->info[0], &img[0].len)) {
    have0 = true; len0 = img[0].len;
    img[0].begin = 0;
    size_t pos = img[0].len;
    while (pos + 1 < jpegr_size) {   // the next SOI, as find_images looks for it
      const void* q = memchr(file + pos, 0xFF, jpegr_size - 1 - pos);
      if (q == nullptr) break;
      pos = (size_t)(static_cast<const uint8_t*>(q) - file);
      if (file[pos + 1] == 0xD8) {
        by_header = image_by_header(file, jpegr_size, pos, &f->info[1], &img[1].len);
        img[1].begin = pos;
        break;
      }
      pos++;
    }
  }
  if (!by_header) {
    const int found = jpegr::find_images(file, jpegr_size, img);                                        // :823-876
    if (found == 0) return UHDR_HIP_ERROR_NO_IMAGES_FOUND;
    if (found == 1) return UHDR_HIP_ERROR_GAIN_MAP_IMAGE_NOT_FOUND;
  }
  for (int k = 0; k < (sdr ? 1 : 2); ++k) {   // the headers, parsed once (jpeg_read_header of either decompressImage call, :690-694 / :731-733)
    f->jpg[k] = file + img[k].begin; f->len[k] = img[k].len;
    const bool parsed = by_header || (k == 0 && have0 && img[0].begin == 0 && img[0].len == len0);
    const int prc = parsed ? 0 : jpeg::parse_header(f->jpg[k], f->len[k], &f->info[k]);
    if (prc == -3) return UHDR_HIP_UNKNOWN_ERROR;   // host allocation failed (never the capacity-probe status)
    if (prc == -2) return UHDR_HIP_ERROR_UNSUPPORTED_FEATURE;
    if (prc != 0 || f->info[k].w > 8192 || f->info[k].h > 8192) return UHDR_HIP_ERROR_DECODE_ERROR;
  }
  if (f->info[0].gray) return UHDR_HIP_ERROR_DECODE_ERROR;   // the primary image must come back as three planes
  // metadata from the gain map's XMP packet (:754-760)
  f->jpg[1] = file + img[1].begin; f->len[1] = img[1].len;
  if (!sdr || want_metadata) {
    const uint8_t* xmp = nullptr;
    size_t xmp_len = 0;
    if (!jpegr::first_xmp(f->jpg[1], f->len[1], &xmp, &xmp_len) || !jpegr::metadata_from_xmp(xmp, xmp_len, &f->md))
      return UHDR_HIP_ERROR_METADATA_ERROR;
  }
  const uint8_t* icc = nullptr;
  size_t icc_len = 0;
  f->gamut = jpegr::first_icc(f->jpg[0], f->len[0], &icc, &icc_len) ? jpegr::gamut_from_icc(icc, icc_len)
                                                                                                             : UHDR_HIP_CG_UNSPECIFIED;
  return UHDR_HIP_NO_ERROR;
}

// grow-only device buffers of the JPEG/R decode entry points (the caller's leased context)
int pool_reserve(DeviceState* st, size_t idx, size_t bytes) {
  if (st->pool.size() <= idx) { st->pool.resize(idx + 1, nullptr); st->pool_bytes.resize(idx + 1, 0); }
  if (bytes == 0) bytes = 256;
  if (st->pool_bytes[idx] >= bytes) return UHDR_HIP_NO_ERROR;
  if (st->pool[idx]) HIP_TRY(hipFree(st->pool[idx]));
  st->pool[idx] = nullptr; st->pool_bytes[idx] = 0;
  HIP_TRY(hipMalloc(&st->pool[idx], bytes));
  st->pool_bytes[idx] = bytes;
  return UHDR_HIP_NO_ERROR;
}
}  // namespace

extern "C" {

// JpegDecoderHelper::decompressImage(..., DECODE_TO_RGBA) (jpegdecoderhelper.cpp:251-281): a 4:2:0 JPEG -> RGBA8888
int uhdr_hip_jpeg_decode_rgba(const void* jpeg, size_t jpeg_size, void* out, size_t out_capacity, uhdr_hip_image_t* desc, int mem_space, void* stream) {
  if (jpeg == nullptr || desc == nullptr) return UHDR_HIP_ERROR_BAD_PTR;
  uhdr_hip_image_t planes;
  memset(&planes, 0, sizeof(planes));
  int rc = uhdr_hip_jpeg_decode(jpeg, jpeg_size, nullptr, 0, &planes, UHDR_HIP_MEM_DEVICE, stream);   // header probe
  if (rc != UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE) return rc;
  if (planes.width == 0 || planes.height == 0) return UHDR_HIP_UNKNOWN_ERROR;                        // (a probe answer always carries the size)
  if (planes.pixelFormat != UHDR_HIP_PIX_FMT_YUV420) return UHDR_HIP_UNKNOWN_ERROR;                    // :258-270: YCbCr 4:2:0 only
  const size_t w = planes.width, h = planes.height, need = w * h * 4;
  memset(desc, 0, sizeof(*desc));
  desc->data = out; desc->width = w; desc->height = h; desc->colorGamut = UHDR_HIP_CG_UNSPECIFIED; desc->luma_stride = w;
  desc->pixelFormat = UHDR_HIP_PIX_FMT_UNSPECIFIED;
  if (out == nullptr || out_capacity < need) return UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE;
  if ((w | h) & 1) return UHDR_HIP_ERROR_UNSUPPORTED_FEATURE;
  DeviceState* st = nullptr;
  if ((rc = current_state(&st)) != UHDR_HIP_NO_ERROR) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  CodecLease lease(st);
  if ((st = lease.get()) == nullptr) return UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE;
  const bool host = mem_space != UHDR_HIP_MEM_DEVICE;
  const size_t ybytes = w * h + 2 * (w * h / 4);
  if ((rc = stage_reserve(st, 8, ybytes + 64)) != 0) return rc;
  if (host && (rc = stage_reserve(st, 10, need)) != 0) return rc;
  if ((rc = uhdr_hip_jpeg_decode(jpeg, jpeg_size, st->stage[8], ybytes, &planes, UHDR_HIP_MEM_DEVICE, stream)) != UHDR_HIP_NO_ERROR) return rc;
  const uint8_t* yp = static_cast<const uint8_t*>(st->stage[8]);
  uint8_t* dst = static_cast<uint8_t*>(host ? st->stage[10] : out);
  HIP_TRY(launch_ycc420_to_rgba(yp, yp + w * h, yp + w * h + (w / 2) * (h / 2), (uint32_t)w, (uint32_t)h, (uint32_t)w, (uint32_t)(w / 2), dst, s));
  if (host) HIP_TRY(hipMemcpyAsync(out, dst, need, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return UHDR_HIP_NO_ERROR;
}

// JpegR::decodeJPEGR (jpegr.cpp:655-822) for n files at once.  A JPEG decode on the device is latency-bound (tens of synchronisation
// rounds of one lane's work each, uhdr_jpeg_dec.hip), so all images of the call share every kernel launch (blockIdx.y = image) and
// their rounds run side by side.
int uhdr_hip_jpegr_decode_batch(int n, const void* const* jpegr, const size_t* jpegr_size, int output_format, float max_display_boost,
                                void* const* dest_data, const size_t* dest_capacity, uhdr_hip_image_t* dests, uhdr_hip_metadata_t* metadata,
                                int* status, int apply_mode, int mem_space, void* stream) {
  if (n < 0 || (n > 0 && (jpegr == nullptr || jpegr_size == nullptr || dests == nullptr))) return UHDR_HIP_ERROR_BAD_PTR;
  if (max_display_boost < 1.0f) return UHDR_HIP_ERROR_INVALID_DISPLAY_BOOST;                             // :666-669
  if (output_format < UHDR_HIP_OUTPUT_SDR || output_format > UHDR_HIP_OUTPUT_HDR_LINEAR_RGB_10BIT) return UHDR_HIP_ERROR_INVALID_OUTPUT_FORMAT;
  std::vector<JpegrFile> files((size_t)n);
  std::vector<int> st_((size_t)n, UHDR_HIP_NO_ERROR);
  std::vector<size_t> out_bytes((size_t)n, 0);
  int live = 0;
  {
    // walking a 2 MB file's markers (memchr over the entropy-coded data, twice: container split and scan length) costs ~0.2 ms of
    // host time per file and touches nothing shared: the files of a batch are parsed by a few threads side by side
    auto parse_range = [&](int lo, int hi) {
      for (int i = lo; i < hi; ++i)
        st_[i] = jpegr[i] == nullptr ? UHDR_HIP_ERROR_BAD_PTR : parse_jpegr_file(jpegr[i], jpegr_size[i], output_format, metadata != nullptr, &files[i]);
    };
    const int nthreads = n >= 2 ? std::min(n, 8) : 1;
    if (nthreads <= 1) {
      parse_range(0, n);
    } else {
      std::vector<std::thread> workers;
      for (int t = 0; t < nthreads; ++t) workers.emplace_back(parse_range, (int)((long)n * t / nthreads), (int)((long)n * (t + 1) / nthreads));
      for (auto& w : workers) w.join();
    }
  }
  for (int i = 0; i < n; ++i) {
    if (st_[i] != UHDR_HIP_NO_ERROR) continue;
    if (metadata != nullptr) metadata[i] = files[i].md;
    dests[i].width = (size_t)files[i].info[0].w; dests[i].height = (size_t)files[i].info[0].h; dests[i].colorGamut = files[i].gamut;
    out_bytes[i] = dests[i].width * dests[i].height * apply_bpp(output_format);
    if (dest_data == nullptr || dest_capacity == nullptr || dest_data[i] == nullptr || dest_capacity[i] < out_bytes[i]) { st_[i] = UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE; continue; }
    ++live;
  }
  auto result = [&]() {
    int first = UHDR_HIP_NO_ERROR;
    for (int i = 0; i < n; ++i) { if (status) status[i] = st_[i]; if (first == UHDR_HIP_NO_ERROR) first = st_[i]; }
    return first;
  };
  if (live == 0) return result();

  DeviceState* st = nullptr;
  int rc;
  if ((rc = current_state(&st)) != UHDR_HIP_NO_ERROR) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  CodecLease lease(st);
  if ((st = lease.get()) == nullptr) return UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE;
  const bool host = mem_space != UHDR_HIP_MEM_DEVICE;
  // per file: two decoder workspaces, two sets of planes, (host callers) the rendition before it goes down
  std::vector<const jpeg::DecInfo*> infos;
  std::vector<jpeg::DecLayout> layouts;
  std::vector<uint8_t*> wss;
  std::vector<jpeg::DecPlane> planes;       // 3 per image
  std::vector<int> owner;
  std::vector<const uint8_t*> srcs;         // the entropy-coded segment of each image in the caller's file
  const bool sdr = output_format == UHDR_HIP_OUTPUT_SDR;
  const int per_file = sdr ? 1 : 2;          // the SDR rendition is the primary image alone (:768-786)
  infos.reserve(2 * live); layouts.reserve(2 * live); wss.reserve(2 * live); planes.reserve(6 * live);
  auto mk = [](uint8_t* p, size_t pw, size_t ph) {
    jpeg::DecPlane q;
    q.p = p; q.w = (int)pw; q.h = (int)ph; q.stride = (int)pw;
    q.aligned8 = (reinterpret_cast<uintptr_t>(p) % 8 == 0 && pw % 8 == 0) ? 1 : 0;
    return q;
  };
  for (int i = 0; i < n; ++i) {
    if (st_[i] != UHDR_HIP_NO_ERROR) continue;
    const JpegrFile& f = files[i];
    for (int k = 0; k < per_file; ++k) {
      const size_t w = (size_t)f.info[k].w, h = (size_t)f.info[k].h, luma = w * h, chroma = luma / 4;
      jpeg::DecLayout l;
      const size_t bytes = jpeg::dec_workspace_bytes(f.info[k], &l);
      if ((rc = pool_reserve(st, 5 * (size_t)i + k, bytes)) != 0) return rc;
      if ((rc = pool_reserve(st, 5 * (size_t)i + 2 + k, (f.info[k].gray ? luma : luma + 2 * chroma) + 64)) != 0) return rc;
      uint8_t* ws = static_cast<uint8_t*>(st->pool[5 * (size_t)i + k]);
      uint8_t* out = static_cast<uint8_t*>(st->pool[5 * (size_t)i + 2 + k]);
      infos.push_back(&f.info[k]); layouts.push_back(l); wss.push_back(ws); owner.push_back(i);
      srcs.push_back(f.jpg[k] + f.info[k].scan_offset);
      planes.push_back(mk(out, w, h));
      planes.push_back(f.info[k].gray ? jpeg::DecPlane{} : mk(out + luma, w / 2, h / 2));
      planes.push_back(f.info[k].gray ? jpeg::DecPlane{} : mk(out + luma + chroma, w / 2, h / 2));
    }
    if (host && (rc = pool_reserve(st, 5 * (size_t)i + 4, out_bytes[i])) != 0) return rc;
  }
  const int nimg = (int)infos.size();
  std::vector<jpeg::DecPlane (*)[3]> pl((size_t)nimg);
  for (int k = 0; k < nimg; ++k) pl[k] = reinterpret_cast<jpeg::DecPlane (*)[3]>(&planes[3 * (size_t)k]);
  std::vector<int> image_rc((size_t)nimg, 0);
  hipError_t herr = hipSuccess;
  // one launch per decoder step for all images of the call (jpeg::decode_device_batch): the files' latency-bound synchronisation
  // rounds run side by side and a batch costs the launches of one image plus its per-image prefix sums
  if ((rc = pool_reserve(st, 5 * (size_t)n, jpeg::dec_batch_scratch_bytes(nimg, layouts.data()))) != 0) return rc;
  for (int g = 0; g < nimg; ++g)
    HIP_TRY(hipMemcpyAsync(wss[g] + layouts[g].src, srcs[g], infos[g]->scan_bytes, hipMemcpyHostToDevice, s));
  const int drc = jpeg::decode_device_batch(nimg, infos.data(), layouts.data(), wss.data(), pl.data(), s, static_cast<uint8_t*>(st->pool[5 * (size_t)n]), &herr,
                                            image_rc.data());
  if (drc > 0) { set_err("uhdr_hip_jpegr_decode", herr); return UHDR_HIP_UNKNOWN_ERROR; }
  for (int k = 0; k < nimg; ++k)
    if (image_rc[k] != 0) st_[owner[k]] = UHDR_HIP_ERROR_DECODE_ERROR;

  // :796-801: the decoded planes as a YUV420 image with the ICC gamut; the gain map is the first plane of its JPEG
  for (int i = 0; i < n; ++i) {
    if (st_[i] != UHDR_HIP_NO_ERROR) continue;
    const JpegrFile& f = files[i];
    if (sdr) {   // libjpeg-turbo's DECODE_TO_RGBA on the planes just decoded (k_ycc420_rgba)
      const size_t w = (size_t)f.info[0].w, h = (size_t)f.info[0].h;
      if ((w | h) & 1) { st_[i] = UHDR_HIP_ERROR_UNSUPPORTED_FEATURE; continue; }
      const uint8_t* yp = static_cast<const uint8_t*>(st->pool[5 * (size_t)i + 2]);
      uint8_t* out = static_cast<uint8_t*>(host ? st->pool[5 * (size_t)i + 4] : dest_data[i]);
      HIP_TRY(launch_ycc420_to_rgba(yp, yp + w * h, yp + w * h + (w / 2) * (h / 2), (uint32_t)w, (uint32_t)h, (uint32_t)w, (uint32_t)(w / 2), out, s));
      dests[i].data = dest_data[i];
      if (host) HIP_TRY(hipMemcpyAsync(dest_data[i], out, out_bytes[i], hipMemcpyDeviceToHost, s));
      continue;
    }
    const size_t w = (size_t)f.info[0].w, h = (size_t)f.info[0].h, gw = (size_t)f.info[1].w, gh = (size_t)f.info[1].h;
    uhdr_hip_image_t ydesc, gimg;
    memset(&ydesc, 0, sizeof(ydesc));
    memset(&gimg, 0, sizeof(gimg));
    uint8_t* yp = static_cast<uint8_t*>(st->pool[5 * (size_t)i + 2]);
    ydesc.data = yp; ydesc.width = w; ydesc.height = h; ydesc.luma_stride = w; ydesc.colorGamut = f.gamut;
    ydesc.chroma_data = yp + w * h; ydesc.chroma_stride = w / 2; ydesc.pixelFormat = UHDR_HIP_PIX_FMT_YUV420;
    gimg.data = st->pool[5 * (size_t)i + 3]; gimg.width = gw; gimg.height = gh; gimg.luma_stride = gw; gimg.colorGamut = UHDR_HIP_CG_UNSPECIFIED;
    gimg.pixelFormat = UHDR_HIP_PIX_FMT_MONOCHROME;
    uhdr_hip_image_t ddev = dests[i];
    ddev.data = host ? st->pool[5 * (size_t)i + 4] : dest_data[i];
    st_[i] = uhdr_hip_apply_gainmap(&ydesc, &gimg, &f.md, output_format, max_display_boost, &ddev, apply_mode, UHDR_HIP_MEM_DEVICE, stream);
    if (st_[i] != UHDR_HIP_NO_ERROR) continue;
    dests[i].data = dest_data[i];
    dests[i].width = ddev.width; dests[i].height = ddev.height;
    if (host) HIP_TRY(hipMemcpyAsync(dest_data[i], ddev.data, out_bytes[i], hipMemcpyDeviceToHost, s));
  }
  HIP_TRY(hipStreamSynchronize(s));
  return result();
}

int uhdr_hip_jpegr_decode(const void* jpegr, size_t jpegr_size, int output_format, float max_display_boost, void* dest_data,
                          size_t dest_capacity, uhdr_hip_image_t* dest, uhdr_hip_metadata_t* metadata, int apply_mode,
                          int mem_space, void* stream) {
  if (jpegr == nullptr) return UHDR_HIP_ERROR_BAD_PTR;                                                   // :658-661
  if (dest == nullptr) return UHDR_HIP_ERROR_BAD_PTR;                                                    // :662-665
  return uhdr_hip_jpegr_decode_batch(1, &jpegr, &jpegr_size, output_format, max_display_boost, &dest_data, &dest_capacity, dest, metadata, nullptr,
                                     apply_mode, mem_space, stream);
}

int uhdr_hip_jpegr_append_gainmap(const void* primary_jpeg, size_t primary_size, const void* gainmap_jpeg, size_t gainmap_size,
                                  const void* exif, size_t exif_size, const void* icc, size_t icc_size,
                                  const uhdr_hip_metadata_t* metadata, void* out, size_t out_capacity, size_t* out_size) {
  if (primary_jpeg == nullptr || gainmap_jpeg == nullptr || metadata == nullptr || out_size == nullptr) return UHDR_HIP_ERROR_BAD_PTR;
  return jpegr::append_gainmap_to(static_cast<const uint8_t*>(primary_jpeg), primary_size, static_cast<const uint8_t*>(gainmap_jpeg), gainmap_size,
                                  static_cast<const uint8_t*>(exif), exif_size, static_cast<const uint8_t*>(icc), icc_size, *metadata,
                                  static_cast<uint8_t*>(out), out_capacity, out_size);
}

int uhdr_hip_icc_profile(int transfer_function, int color_gamut, void* out, size_t out_capacity, size_t* out_size) {
  if (out_size == nullptr) return UHDR_HIP_ERROR_BAD_PTR;
  if (transfer_function != UHDR_HIP_TF_SRGB) return UHDR_HIP_ERROR_UNSUPPORTED_FEATURE;   // HLG / PQ profiles (tone-map LUTs) are not built
  std::vector<uint8_t> icc;
  if (!jpegr::icc_profile_srgb_transfer(color_gamut, icc)) return UHDR_HIP_ERROR_INVALID_COLORGAMUT;
  *out_size = icc.size();
  if (out == nullptr || out_capacity < icc.size()) return UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE;
  memcpy(out, icc.data(), icc.size());
  return UHDR_HIP_NO_ERROR;
}

}  // extern "C"

namespace {

// JpegR::areInputArgumentsValid, the four-argument form (jpegr.cpp:75-173), checks in the reference's order
int check_encode_inputs(const uhdr_hip_image_t* p010, const uhdr_hip_image_t* yuv, int hdr_tf, const void* out, const size_t* out_size) {
  if (p010 == nullptr || p010->data == nullptr) return UHDR_HIP_ERROR_BAD_PTR;
  if ((p010->width | p010->height) & 1) return UHDR_HIP_ERROR_UNSUPPORTED_WIDTH_HEIGHT;
  if (p010->width < 8 || p010->height < 8 || p010->width > 8192 || p010->height > 8192) return UHDR_HIP_ERROR_UNSUPPORTED_WIDTH_HEIGHT;
  if (p010->colorGamut <= UHDR_HIP_CG_UNSPECIFIED || p010->colorGamut > UHDR_HIP_CG_BT2100) return UHDR_HIP_ERROR_INVALID_COLORGAMUT;
  if (p010->luma_stride != 0 && p010->luma_stride < p010->width) return UHDR_HIP_ERROR_INVALID_STRIDE;
  if (p010->chroma_data != nullptr && p010->chroma_stride < p010->width) return UHDR_HIP_ERROR_INVALID_STRIDE;
  if (out == nullptr || out_size == nullptr) return UHDR_HIP_ERROR_BAD_PTR;
  if (hdr_tf != UHDR_HIP_TF_LINEAR && hdr_tf != UHDR_HIP_TF_HLG && hdr_tf != UHDR_HIP_TF_PQ) return UHDR_HIP_ERROR_INVALID_TRANS_FUNC;
  if (yuv == nullptr) return UHDR_HIP_NO_ERROR;
  if (yuv->data == nullptr) return UHDR_HIP_ERROR_BAD_PTR;
  if (yuv->luma_stride != 0 && yuv->luma_stride < yuv->width) return UHDR_HIP_ERROR_INVALID_STRIDE;
  if (yuv->chroma_data != nullptr && yuv->chroma_stride < yuv->width / 2) return UHDR_HIP_ERROR_INVALID_STRIDE;
  if (p010->width != yuv->width || p010->height != yuv->height) return UHDR_HIP_ERROR_RESOLUTION_MISMATCH;
  if (yuv->colorGamut <= UHDR_HIP_CG_UNSPECIFIED || yuv->colorGamut > UHDR_HIP_CG_BT2100) return UHDR_HIP_ERROR_INVALID_COLORGAMUT;
  return UHDR_HIP_NO_ERROR;
}

// "clean up input structure for later usage" (jpegr.cpp:261-275 and the same lines of every API)
void default_p010(uhdr_hip_image_t* im) {
  if (im->luma_stride == 0) im->luma_stride = im->width;
  if (im->chroma_data == nullptr) { im->chroma_data = static_cast<uint16_t*>(im->data) + im->luma_stride * im->height; im->chroma_stride = im->luma_stride; }
  im->pixelFormat = UHDR_HIP_PIX_FMT_P010;
}
void default_yuv(uhdr_hip_image_t* im) {
  if (im->luma_stride == 0) im->luma_stride = im->width;
  if (im->chroma_data == nullptr) { im->chroma_data = static_cast<uint8_t*>(im->data) + im->luma_stride * im->height; im->chroma_stride = im->luma_stride >> 1; }
  im->pixelFormat = UHDR_HIP_PIX_FMT_YUV420;
}

struct EncodeCtx {
  DeviceState* st;
  void* stream;
  int mem_space;
  bool host() const { return mem_space != UHDR_HIP_MEM_DEVICE; }
  hipStream_t s() const { return static_cast<hipStream_t>(stream); }
};

// Host bytes a compressed stream comes down into: page-locked (the copy runs at DMA speed instead of through the runtime's staging
// buffer) and kept between calls by their thread_local owners -- a fresh multi-megabyte std::vector costs a memset and a page fault
// per 4 KiB on every call, which was more than half of a 4K encodeJPEGR call.  Never freed: the process owns them until it ends.
struct HostBytes {
  uint8_t* p = nullptr;
  size_t n = 0, cap = 0;
  uint8_t* data() const { return p; }
  size_t size() const { return n; }
  void resize(size_t want) {
    if (want > cap) {
      if (p) (void)hipHostFree(p);
      p = nullptr; cap = 0;
      void* q = nullptr;
      if (hipHostMalloc(&q, want, hipHostMallocDefault) == hipSuccess) { p = static_cast<uint8_t*>(q); cap = want; }
    }
    n = p ? want : 0;
  }
};

// JpegEncoderHelper::compressImage with the bytes landing in host memory whichever side the planes live on
int jpeg_to_host(const EncodeCtx& c, const uhdr_hip_image_t& img, int q, const std::vector<uint8_t>* icc, HostBytes& dst, size_t* n) {
  for (int attempt = 0; attempt < 2; ++attempt) {
    if (dst.data() == nullptr) return UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE;
    int r;
    if (c.host()) {
      r = uhdr_hip_jpeg_encode(&img, q, icc ? icc->data() : nullptr, icc ? icc->size() : 0, dst.data(), dst.size(), n, UHDR_HIP_MEM_HOST, c.stream);
    } else {   // device planes in, bytes to a device buffer, then down
      int r2;
      if ((r2 = stage_reserve(c.st, 10, dst.size())) != 0) return r2;
      r = uhdr_hip_jpeg_encode(&img, q, icc ? icc->data() : nullptr, icc ? icc->size() : 0, c.st->stage[10], dst.size(), n, UHDR_HIP_MEM_DEVICE, c.stream);
      if (r == UHDR_HIP_NO_ERROR) {
        HIP_TRY(hipMemcpyAsync(dst.data(), c.st->stage[10], *n, hipMemcpyDeviceToHost, c.s()));
        HIP_TRY(hipStreamSynchronize(c.s()));
      }
    }
    if (r != UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE) return r;
    dst.resize(*n + 16);
  }
  return UHDR_HIP_ERROR_ENCODE_ERROR;
}

// The same for planes that live on the device, without a round trip: the kernels write the stream straight into the page-locked
// host buffer (the device reaches it over the bus; that replaces the device-to-host copy) and its size into a page-locked word, so a
// call can enqueue all its compressions and synchronise ONCE.  The size is valid after the stream has been synchronised; `dst` must
// not be touched before.  ws_slot: encoder workspace, 12 or 13 (two compressions in flight need two; uhdr_hip_jpeg_encode has its own).
constexpr size_t kPendingSize = ~(size_t)0;
struct PendingJpeg {
  HostBytes* dst = nullptr;
  uint64_t* total = nullptr;
  uhdr_hip_image_t img;
  int q = 0;
  std::vector<uint8_t> icc;
  bool has_icc = false;
};
uint64_t* pinned_totals() {   // two page-locked size words per host thread
  static thread_local HostBytes words;
  if (words.size() < 64) words.resize(64);
  return reinterpret_cast<uint64_t*>(words.data());
}
int jpeg_enqueue_device(const EncodeCtx& c, const uhdr_hip_image_t& img, int q, const std::vector<uint8_t>* icc, HostBytes& dst, int ws_slot,
                        uint64_t* total, PendingJpeg* pend) {
  if (dst.data() == nullptr || total == nullptr) return UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE;
  const bool gray = img.pixelFormat == UHDR_HIP_PIX_FMT_MONOCHROME;
  const size_t w = img.width, h = img.height;
  if (w == 0 || h == 0 || w > 65500 || h > 65500 || (!gray && ((w | h) & 1))) return UHDR_HIP_ERROR_RESOLUTION_MISMATCH;
  const size_t ls = img.luma_stride ? img.luma_stride : w, cs = gray ? 0 : img.chroma_stride;
  const size_t aw = (w + 15) / 16 * 16, acw = (w / 2 + 7) / 8 * 8;
  jpeg::Job j;
  encode_job_tables(w, h, gray, q, &j);
  const uint8_t* py = static_cast<const uint8_t*>(img.data);
  const uint8_t* pu = static_cast<const uint8_t*>(img.chroma_data);
  j.plane[0] = encode_plane(py, w, h, ls, ls < aw);
  if (!gray) {
    j.plane[1] = encode_plane(pu, w / 2, h / 2, cs, cs < acw);
    j.plane[2] = encode_plane(pu + cs * h / 2, w / 2, h / 2, cs, cs < acw);   // chromaStride * height / 2 (jpegencoderhelper.cpp:140)
  }
  std::vector<uint8_t> header;
  jpeg::build_header((int)w, (int)h, gray, q, icc ? icc->data() : nullptr, icc ? icc->size() : 0, header);
  if (dst.size() < header.size()) return UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE;
  jpeg::Layout l;
  const size_t ws_bytes = jpeg::workspace_bytes(j.nblk, &l);
  int rc;
  if ((rc = stage_reserve(c.st, ws_slot, ws_bytes)) != 0) return rc;   // slots 12 / 13 of the caller's leased context
  uint8_t* ws = static_cast<uint8_t*>(c.st->stage[ws_slot]);
  memcpy(dst.data(), header.data(), header.size());   // host memory: no copy to enqueue
  *total = 0;
  HIP_TRY(jpeg::encode_async(j, l, ws, dst.data(), dst.size(), header.size(), c.s(), total));   // (the size goes where the bytes go: page-locked host memory)
  pend->dst = &dst; pend->total = total; pend->img = img; pend->q = q;
  pend->has_icc = icc != nullptr;
  if (icc) pend->icc = *icc;
  return UHDR_HIP_NO_ERROR;
}
// after the stream has been synchronised: the size, or (a stream larger than the buffer: it was cut off) the compression again
int jpeg_collect(const EncodeCtx& c, PendingJpeg& p, size_t* n) {
  const uint64_t total = *p.total;
  if (total != 0 && total <= p.dst->size()) { *n = (size_t)total; return UHDR_HIP_NO_ERROR; }
  if (total == 0) return UHDR_HIP_ERROR_ENCODE_ERROR;
  p.dst->resize((size_t)total + 16);
  return jpeg_to_host(c, p.img, p.q, p.has_icc ? &p.icc : nullptr, *p.dst, n);
}
PendingJpeg& pending_gainmap() { static thread_local PendingJpeg p; return p; }

// compressGainMap (jpegr.cpp:806-821): one plane at kMapCompressQuality = 85
int gainmap_to_jpeg(const EncodeCtx& c, const uhdr_hip_image_t& map, HostBytes& jpeg, size_t* n) {
  uhdr_hip_image_t g = map;
  g.chroma_data = nullptr; g.chroma_stride = 0; g.pixelFormat = UHDR_HIP_PIX_FMT_MONOCHROME;
  // an earlier call that left on an error path may still have kernels writing into this (page-locked, thread-local) buffer on
  // the side stream: they must have finished before a growing resize frees it
  if (c.st->aux != nullptr) HIP_TRY(hipStreamSynchronize(c.st->aux));
  jpeg.resize(map.width * map.height + 65536);
  if (!c.host()) {   // enqueued; *n == kPendingSize until resolve_gainmap_jpeg() (or finish_from_planes) has synchronised
    // On a stream of its own behind the kernel that wrote the map: nothing the caller's stream does next (the SDR image's BT.601
    // re-encode and compression) depends on it, and its ~12 small launches fit next to those (the context is this call's own).
    if (c.st->aux == nullptr) {
      HIP_TRY(hipStreamCreateWithFlags(&c.st->aux, hipStreamNonBlocking));
      HIP_TRY(hipEventCreateWithFlags(&c.st->map_ready, hipEventDisableTiming));
    }
    HIP_TRY(hipEventRecord(c.st->map_ready, c.s()));
    HIP_TRY(hipStreamWaitEvent(c.st->aux, c.st->map_ready, 0));
    EncodeCtx side = c;
    side.stream = c.st->aux;
    *n = kPendingSize;
    return jpeg_enqueue_device(side, g, 85, nullptr, jpeg, 13, pinned_totals(), &pending_gainmap()) == UHDR_HIP_NO_ERROR ? UHDR_HIP_NO_ERROR : UHDR_HIP_ERROR_ENCODE_ERROR;
  }
  return jpeg_to_host(c, g, 85, nullptr, jpeg, n) == UHDR_HIP_NO_ERROR ? UHDR_HIP_NO_ERROR : UHDR_HIP_ERROR_ENCODE_ERROR;
}

// generateGainMap followed by compressGainMap: the block every one of API-0..3 contains (e.g. jpegr.cpp:277-292)
int make_gainmap_jpeg(const EncodeCtx& c, const uhdr_hip_image_t& yuv, const uhdr_hip_image_t& p010, int hdr_tf, int sdr_is_601,
                      uhdr_hip_metadata_t* md, HostBytes& jpeg, size_t* n) {
  const size_t mw = yuv.width / 4, mh = yuv.height / 4;
  std::vector<uint8_t> host_map;
  uhdr_hip_image_t map = yuv;
  int rc;
  if ((rc = stage_reserve(c.st, 9, mw * mh + 64)) != 0) return rc;
  if (c.host()) { host_map.resize(mw * mh ? mw * mh : 1); map.data = host_map.data(); } else map.data = c.st->stage[9];
  rc = uhdr_hip_generate_gainmap(&yuv, &p010, hdr_tf, md, &map, sdr_is_601, c.mem_space, c.stream);
  if (rc != UHDR_HIP_NO_ERROR) return rc;
  map.width = mw; map.height = mh; map.luma_stride = mw;
  return gainmap_to_jpeg(c, map, jpeg, n);
}

// the 3x3 of JpegR::convertYuv for a pair of different encodings (jpegr.cpp:1134-1197)
const float* yuv_matrix(int src_encoding, int dest_encoding) {
  switch (src_encoding) {
    case UHDR_HIP_CG_BT709: return dest_encoding == UHDR_HIP_CG_P3 ? kYuv709To601 : kYuv709To2100;
    case UHDR_HIP_CG_P3: return dest_encoding == UHDR_HIP_CG_BT709 ? kYuv601To709 : kYuv601To2100;
    default: return dest_encoding == UHDR_HIP_CG_BT709 ? kYuv2100To709 : kYuv2100To601;
  }
}
// convertYuv of `src` into the planes of `dst` (device memory, same size): the private copy API-1 converts is written by the
// conversion itself instead of by three plane copies in front of it
// descriptor of one convertYuv: `src`'s samples through m into `dst`'s planes (the same image for the in-place form)
CvtImage cvt_image(const uhdr_hip_image_t& src, const uhdr_hip_image_t& dst, const float* m, bool* aligned) {
  CvtImage t;
  t.y = static_cast<uint8_t*>(dst.data);
  t.u = static_cast<uint8_t*>(dst.chroma_data);
  t.v = t.u + dst.chroma_stride * (dst.height / 2);
  t.y_stride = (uint32_t)dst.luma_stride; t.c_stride = (uint32_t)dst.chroma_stride;
  t.width = (uint32_t)dst.width; t.height = (uint32_t)dst.height;
  for (int i = 0; i < 9; ++i) t.m[i] = m[i];
  t.sy = static_cast<const uint8_t*>(src.data);
  t.su = static_cast<const uint8_t*>(src.chroma_data);
  t.sv = t.su + src.chroma_stride * (src.height / 2);
  t.sy_stride = (uint32_t)src.luma_stride; t.sc_stride = (uint32_t)src.chroma_stride;
  *aligned = t.width % 8u == 0 && al(t.y, 8) && t.y_stride % 8u == 0 && al(t.u, 4) && al(t.v, 4) && t.c_stride % 4u == 0 &&
             al(t.sy, 8) && t.sy_stride % 8u == 0 && al(t.su, 4) && al(t.sv, 4) && t.sc_stride % 4u == 0;
  return t;
}
int convert_yuv_into(const uhdr_hip_image_t& src, const uhdr_hip_image_t& dst, const float* m, hipStream_t s) {
  CvtBatch b;
  bool aligned;
  b.img[0] = cvt_image(src, dst, m, &aligned);
  HIP_TRY(launch_convert_yuv(b, 1, aligned, s));
  return UHDR_HIP_NO_ERROR;
}

// the tail API-0 and API-1 share (jpegr.cpp:210-247 / :294-380): ICC for the SDR gamut, BT.601 re-encode unless P3, JPEG at `quality`,
// appendGainMap.  `enc` must be private to the call when it is not P3 (it is converted in place).
int finish_from_planes(const EncodeCtx& c, uhdr_hip_image_t enc, int quality, const void* exif, size_t exif_size,
                       const HostBytes& gm_jpeg, size_t gm_n, const uhdr_hip_metadata_t& md, void* out, size_t out_capacity,
                       size_t* out_size, bool converted = false) {   // gm_n may be kPendingSize: the gain-map JPEG is still being written (device callers)
  std::vector<uint8_t> icc;
  if (!jpegr::icc_profile_srgb_transfer(enc.colorGamut, icc)) return UHDR_HIP_ERROR_INVALID_COLORGAMUT;
  int rc;
  if (!converted && enc.colorGamut != UHDR_HIP_CG_P3 &&
      (rc = uhdr_hip_convert_yuv(&enc, enc.colorGamut, UHDR_HIP_CG_P3, c.mem_space, c.stream)) != UHDR_HIP_NO_ERROR)
    return rc;
  static thread_local HostBytes sdr_jpeg;
  sdr_jpeg.resize(enc.width * enc.height + 65536);
  size_t sdr_n = 0;
  if (!c.host()) {   // both compressions in flight, one synchronisation for the call
    PendingJpeg sdr;
    if (jpeg_enqueue_device(c, enc, quality, &icc, sdr_jpeg, 12, pinned_totals() + 1, &sdr) != UHDR_HIP_NO_ERROR) return UHDR_HIP_ERROR_ENCODE_ERROR;
    HIP_TRY(hipStreamSynchronize(c.s()));
    if (gm_n == kPendingSize) HIP_TRY(hipStreamSynchronize(c.st->aux));
    if (gm_n == kPendingSize && jpeg_collect(c, pending_gainmap(), &gm_n) != UHDR_HIP_NO_ERROR) return UHDR_HIP_ERROR_ENCODE_ERROR;
    if (jpeg_collect(c, sdr, &sdr_n) != UHDR_HIP_NO_ERROR) return UHDR_HIP_ERROR_ENCODE_ERROR;
  } else if (jpeg_to_host(c, enc, quality, &icc, sdr_jpeg, &sdr_n) != UHDR_HIP_NO_ERROR) {
    return UHDR_HIP_ERROR_ENCODE_ERROR;
  }
  return jpegr::append_gainmap_to(sdr_jpeg.data(), sdr_n, gm_jpeg.data(), gm_n, static_cast<const uint8_t*>(exif), exif_size, nullptr, 0, md,
                                  static_cast<uint8_t*>(out), out_capacity, out_size);
}

// for the callers that need the gain-map JPEG at once (API-2 / API-3 / API-x): wait for it
int resolve_gainmap_jpeg(const EncodeCtx& c, size_t* n) {
  if (*n != kPendingSize) return UHDR_HIP_NO_ERROR;
  HIP_TRY(hipStreamSynchronize(c.st->aux));
  return jpeg_collect(c, pending_gainmap(), n) == UHDR_HIP_NO_ERROR ? UHDR_HIP_NO_ERROR : UHDR_HIP_ERROR_ENCODE_ERROR;
}

}  // namespace

extern "C" {

// JpegR::encodeJPEGR API-0 (jpegr.cpp:186-247)
int uhdr_hip_jpegr_encode_api0(const uhdr_hip_image_t* p010_in, int hdr_tf, int quality, const void* exif, size_t exif_size, void* out,
                               size_t out_capacity, size_t* out_size, int mem_space, void* stream) {
  if (quality < 0 || quality > 100) return UHDR_HIP_ERROR_INVALID_QUALITY_FACTOR;                                 // :175-183
  int rc = check_encode_inputs(p010_in, nullptr, hdr_tf, out, out_size);
  if (rc != UHDR_HIP_NO_ERROR) return rc;
  if (exif == nullptr && exif_size != 0) return UHDR_HIP_ERROR_BAD_PTR;                                          // :190-193
  uhdr_hip_image_t p010 = *p010_in;
  default_p010(&p010);
  EncodeCtx c{nullptr, stream, mem_space};
  if ((rc = current_state(&c.st)) != UHDR_HIP_NO_ERROR) return rc;
  CodecLease lease(c.st);
  if ((c.st = lease.get()) == nullptr) return UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE;

  // :208-223: the tone-mapped SDR image, luma stride rounded up to the encoder's 16-column batch, zero-initialised
  const size_t w = p010.width, h = p010.height, ls = (w + 15) / 16 * 16, total = ls * h * 3 / 2;
  std::vector<uint8_t> host_yuv;
  uhdr_hip_image_t yuv;
  memset(&yuv, 0, sizeof(yuv));
  yuv.width = w; yuv.height = h; yuv.colorGamut = p010.colorGamut;
  yuv.luma_stride = ls; yuv.chroma_stride = ls >> 1; yuv.pixelFormat = UHDR_HIP_PIX_FMT_YUV420;
  if (c.host()) {
    host_yuv.assign(total, 0);
    yuv.data = host_yuv.data();
  } else {
    if ((rc = stage_reserve(c.st, 8, total + 64)) != 0) return rc;
    HIP_TRY(hipMemsetAsync(c.st->stage[8], 0, total, c.s()));
    yuv.data = c.st->stage[8];
  }
  yuv.chroma_data = static_cast<uint8_t*>(yuv.data) + ls * h;
  if ((rc = uhdr_hip_tonemap(&p010, &yuv, mem_space, stream)) != UHDR_HIP_NO_ERROR) return rc;                    // :226

  uhdr_hip_metadata_t md;
  static thread_local HostBytes gm_jpeg;
  size_t gm_n = 0;
  if ((rc = make_gainmap_jpeg(c, yuv, p010, hdr_tf, 0, &md, gm_jpeg, &gm_n)) != UHDR_HIP_NO_ERROR) return rc;     // :228-244
  return finish_from_planes(c, yuv, quality, exif, exif_size, gm_jpeg, gm_n, md, out, out_capacity, out_size);
}

// JpegR::encodeJPEGR API-1 (jpegr.cpp:249-381)
int uhdr_hip_jpegr_encode_api1(const uhdr_hip_image_t* p010_in, const uhdr_hip_image_t* yuv_in, int hdr_tf, int quality, const void* exif,
                               size_t exif_size, void* out, size_t out_capacity, size_t* out_size, int mem_space, void* stream) {
  if (yuv_in == nullptr) return UHDR_HIP_ERROR_BAD_PTR;                                                           // :253-256
  if (quality < 0 || quality > 100) return UHDR_HIP_ERROR_INVALID_QUALITY_FACTOR;                                 // :175-183
  int rc = check_encode_inputs(p010_in, yuv_in, hdr_tf, out, out_size);
  if (rc != UHDR_HIP_NO_ERROR) return rc;
  if (exif == nullptr && exif_size != 0) return UHDR_HIP_ERROR_BAD_PTR;                                          // :258-261
  uhdr_hip_image_t p010 = *p010_in, yuv = *yuv_in;
  default_p010(&p010);
  default_yuv(&yuv);
  EncodeCtx c{nullptr, stream, mem_space};
  if ((rc = current_state(&c.st)) != UHDR_HIP_NO_ERROR) return rc;
  CodecLease lease(c.st);
  if ((c.st = lease.get()) == nullptr) return UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE;
  const size_t w = yuv.width, h = yuv.height;

  uhdr_hip_metadata_t md;
  static thread_local HostBytes gm_jpeg;
  size_t gm_n = 0;
  if ((rc = make_gainmap_jpeg(c, yuv, p010, hdr_tf, 0, &md, gm_jpeg, &gm_n)) != UHDR_HIP_NO_ERROR) return rc;     // :277-292

  // :297-358: unless the SDR image is P3 (= BT.601 encoding) already, a copy with 16-aligned strides, zero padded, is what gets converted
  uhdr_hip_image_t enc = yuv;
  std::vector<uint8_t> host_601;
  bool converted = false;
  if (yuv.colorGamut != UHDR_HIP_CG_P3) {
    const size_t ls = (w + 15) / 16 * 16, cs = ls >> 1, total = ls * h * 3 / 2;
    enc.luma_stride = ls; enc.chroma_stride = cs;
    const uint8_t* src_u = static_cast<const uint8_t*>(yuv.chroma_data);
    const uint8_t* src_v = src_u + yuv.chroma_stride * h / 2;
    if (c.host()) {
      host_601.assign(total ? total : 1, 0);
      enc.data = host_601.data();
      uint8_t* du = host_601.data() + ls * h;
      uint8_t* dv = du + cs * h / 2;
      for (size_t r = 0; r < h; ++r) memcpy(host_601.data() + r * ls, static_cast<const uint8_t*>(yuv.data) + r * yuv.luma_stride, w);
      for (size_t r = 0; r < h / 2; ++r) { memcpy(du + r * cs, src_u + r * yuv.chroma_stride, w / 2); memcpy(dv + r * cs, src_v + r * yuv.chroma_stride, w / 2); }
    } else {
      if ((rc = stage_reserve(c.st, 8, total + 64)) != 0) return rc;
      uint8_t* d = static_cast<uint8_t*>(c.st->stage[8]);
      if (ls != w) HIP_TRY(hipMemsetAsync(d, 0, total, c.s()));   // the padding columns (a width of whole 16-column batches has none)
      enc.data = d;
      enc.chroma_data = d + ls * h;
      // the copy is written by the conversion (the same arithmetic on the same samples as a copy followed by the in-place form)
      if (!valid_gamut(yuv.colorGamut)) return UHDR_HIP_ERROR_INVALID_COLORGAMUT;
      if ((rc = convert_yuv_into(yuv, enc, yuv_matrix(yuv.colorGamut, UHDR_HIP_CG_P3), c.s())) != UHDR_HIP_NO_ERROR) return rc;
      converted = true;
    }
    enc.chroma_data = static_cast<uint8_t*>(enc.data) + ls * h;
  }
  return finish_from_planes(c, enc, quality, exif, exif_size, gm_jpeg, gm_n, md, out, out_capacity, out_size, converted);
}

// JpegR::encodeJPEGR API-4 (jpegr.cpp:502-560): host bytes only, nothing runs on the device
int uhdr_hip_jpegr_encode_api4(const void* sdr_jpeg, size_t sdr_jpeg_size, int sdr_jpeg_gamut, const void* gainmap_jpeg, size_t gainmap_jpeg_size,
                               const uhdr_hip_metadata_t* metadata, void* out, size_t out_capacity, size_t* out_size) {
  if (sdr_jpeg == nullptr || gainmap_jpeg == nullptr) return UHDR_HIP_ERROR_BAD_PTR;                              // :505-512
  if (out == nullptr || out_size == nullptr) return UHDR_HIP_ERROR_BAD_PTR;                                       // :513-516
  const uint8_t* pj = static_cast<const uint8_t*>(sdr_jpeg);
  if (!jpegr::has_valid_header(pj, sdr_jpeg_size)) return UHDR_HIP_ERROR_DECODE_ERROR;                            // :520-524
  const uint8_t* have = nullptr;
  size_t have_len = 0;
  std::vector<uint8_t> icc;
  if (!jpegr::first_icc(pj, sdr_jpeg_size, &have, &have_len)) {          // :527-541
    if (sdr_jpeg_gamut <= UHDR_HIP_CG_UNSPECIFIED || sdr_jpeg_gamut > UHDR_HIP_CG_BT2100) return UHDR_HIP_ERROR_INVALID_COLORGAMUT;
    jpegr::icc_profile_srgb_transfer(sdr_jpeg_gamut, icc);
  }
  if (metadata == nullptr) return UHDR_HIP_ERROR_BAD_PTR;                                                         // :955-958
  return jpegr::append_gainmap_to(pj, sdr_jpeg_size, static_cast<const uint8_t*>(gainmap_jpeg), gainmap_jpeg_size, nullptr, 0,
                                  icc.empty() ? nullptr : icc.data(), icc.size(), *metadata, static_cast<uint8_t*>(out), out_capacity, out_size);
}

// JpegR::encodeJPEGR API-2 (jpegr.cpp:384-437)
int uhdr_hip_jpegr_encode_api2(const uhdr_hip_image_t* p010_in, const uhdr_hip_image_t* yuv_in, const void* sdr_jpeg, size_t sdr_jpeg_size,
                               int sdr_jpeg_gamut, int hdr_tf, void* out, size_t out_capacity, size_t* out_size, int mem_space, void* stream) {
  if (yuv_in == nullptr) return UHDR_HIP_ERROR_BAD_PTR;                                                           // :390-393
  if (sdr_jpeg == nullptr) return UHDR_HIP_ERROR_BAD_PTR;                                                         // :394-397
  int rc = check_encode_inputs(p010_in, yuv_in, hdr_tf, out, out_size);
  if (rc != UHDR_HIP_NO_ERROR) return rc;
  uhdr_hip_image_t p010 = *p010_in, yuv = *yuv_in;
  default_p010(&p010);
  default_yuv(&yuv);
  EncodeCtx c{nullptr, stream, mem_space};
  if ((rc = current_state(&c.st)) != UHDR_HIP_NO_ERROR) return rc;
  uhdr_hip_metadata_t md;
  static thread_local HostBytes gm_jpeg;
  size_t gm_n = 0;
  {
    CodecLease lease(c.st);
    if ((c.st = lease.get()) == nullptr) return UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE;
    if ((rc = make_gainmap_jpeg(c, yuv, p010, hdr_tf, 0, &md, gm_jpeg, &gm_n)) != UHDR_HIP_NO_ERROR) return rc;   // :416-434
    if ((rc = resolve_gainmap_jpeg(c, &gm_n)) != UHDR_HIP_NO_ERROR) return rc;
  }
  return uhdr_hip_jpegr_encode_api4(sdr_jpeg, sdr_jpeg_size, sdr_jpeg_gamut, gm_jpeg.data(), gm_n, &md, out, out_capacity, out_size);
}

// JpegR::encodeJPEGR API-3 (jpegr.cpp:439-500): the SDR rendition arrives as a JPEG only and is decoded on the device
int uhdr_hip_jpegr_encode_api3(const uhdr_hip_image_t* p010_in, const void* sdr_jpeg, size_t sdr_jpeg_size, int sdr_jpeg_gamut, int hdr_tf,
                               void* out, size_t out_capacity, size_t* out_size, int mem_space, void* stream) {
  if (sdr_jpeg == nullptr) return UHDR_HIP_ERROR_BAD_PTR;                                                         // :443-446
  int rc = check_encode_inputs(p010_in, nullptr, hdr_tf, out, out_size);
  if (rc != UHDR_HIP_NO_ERROR) return rc;
  uhdr_hip_image_t p010 = *p010_in;
  default_p010(&p010);
  const uint8_t* pj = static_cast<const uint8_t*>(sdr_jpeg);

  // :457-462 decode; a header probe first for the size (host work: an unreadable file is reported without a device)
  uhdr_hip_image_t ydesc;
  memset(&ydesc, 0, sizeof(ydesc));
  rc = uhdr_hip_jpeg_decode(pj, sdr_jpeg_size, nullptr, 0, &ydesc, UHDR_HIP_MEM_DEVICE, stream);
  if (rc == UHDR_HIP_ERROR_UNSUPPORTED_FEATURE) return rc;
  if (rc != UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE || ydesc.pixelFormat != UHDR_HIP_PIX_FMT_YUV420 || ydesc.width == 0 || ydesc.height == 0)
    return UHDR_HIP_ERROR_DECODE_ERROR;
  EncodeCtx c{nullptr, stream, mem_space};
  if ((rc = current_state(&c.st)) != UHDR_HIP_NO_ERROR) return rc;
  const size_t w = ydesc.width, h = ydesc.height, ybytes = w * h + 2 * (w * h / 4);
  uhdr_hip_metadata_t md;
  static thread_local HostBytes gm_jpeg;
  size_t gm_n = 0;
  {
    CodecLease lease(c.st);
    if ((c.st = lease.get()) == nullptr) return UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE;
    std::vector<uint8_t> host_yuv;
    void* planes;
    if (c.host()) { host_yuv.resize(ybytes); planes = host_yuv.data(); }
    else { if ((rc = stage_reserve(c.st, 8, ybytes + 64)) != 0) return rc; planes = c.st->stage[8]; }
    if (uhdr_hip_jpeg_decode(pj, sdr_jpeg_size, planes, ybytes, &ydesc, mem_space, stream) != UHDR_HIP_NO_ERROR) return UHDR_HIP_ERROR_DECODE_ERROR;
    // :467-488 the gamut: the ICC profile's when there is one (and it must agree with a configured gamut), else the configured one
    const uint8_t* icc = nullptr;
    size_t icc_len = 0;
    if (jpegr::first_icc(pj, sdr_jpeg_size, &icc, &icc_len)) {
      const int cg = jpegr::gamut_from_icc(icc, icc_len);
      if (cg == UHDR_HIP_CG_UNSPECIFIED || (sdr_jpeg_gamut != UHDR_HIP_CG_UNSPECIFIED && sdr_jpeg_gamut != cg)) return UHDR_HIP_ERROR_INVALID_COLORGAMUT;
      ydesc.colorGamut = cg;
    } else {
      if (sdr_jpeg_gamut <= UHDR_HIP_CG_UNSPECIFIED || sdr_jpeg_gamut > UHDR_HIP_CG_BT2100) return UHDR_HIP_ERROR_INVALID_COLORGAMUT;
      ydesc.colorGamut = sdr_jpeg_gamut;
    }
    if (p010.width != w || p010.height != h) return UHDR_HIP_ERROR_RESOLUTION_MISMATCH;                           // :496-499
    if ((rc = make_gainmap_jpeg(c, ydesc, p010, hdr_tf, 1 /* sdr_is_601 */, &md, gm_jpeg, &gm_n)) != UHDR_HIP_NO_ERROR) return rc;
    if ((rc = resolve_gainmap_jpeg(c, &gm_n)) != UHDR_HIP_NO_ERROR) return rc;
  }
  return uhdr_hip_jpegr_encode_api4(sdr_jpeg, sdr_jpeg_size, sdr_jpeg_gamut, gm_jpeg.data(), gm_n, &md, out, out_capacity, out_size);
}

// JpegR::encodeJPEGR "API-x" (jpegr.cpp:562-631): SDR planes + a ready gain map + its metadata; no BT.601 re-encode on this path
int uhdr_hip_jpegr_encode_apix(const uhdr_hip_image_t* yuv_in, const uhdr_hip_image_t* gainmap, const uhdr_hip_metadata_t* metadata, int quality,
                               const void* exif, size_t exif_size, void* out, size_t out_capacity, size_t* out_size, int mem_space, void* stream) {
  if (quality < 0 || quality > 100) return UHDR_HIP_ERROR_INVALID_QUALITY_FACTOR;                                 // :566-568
  if (yuv_in == nullptr || yuv_in->data == nullptr) return UHDR_HIP_ERROR_BAD_PTR;
  if (gainmap == nullptr || gainmap->data == nullptr) return UHDR_HIP_ERROR_BAD_PTR;
  if (metadata == nullptr) return UHDR_HIP_ERROR_BAD_PTR;
  if (out == nullptr || out_size == nullptr) return UHDR_HIP_ERROR_BAD_PTR;
  uhdr_hip_image_t yuv = *yuv_in;
  default_yuv(&yuv);
  EncodeCtx c{nullptr, stream, mem_space};
  int rc;
  if ((rc = current_state(&c.st)) != UHDR_HIP_NO_ERROR) return rc;
  CodecLease lease(c.st);
  if ((c.st = lease.get()) == nullptr) return UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE;
  static thread_local HostBytes gm_jpeg;
  std::vector<uint8_t> icc;
  size_t gm_n = 0;
  if ((rc = gainmap_to_jpeg(c, *gainmap, gm_jpeg, &gm_n)) != UHDR_HIP_NO_ERROR) return rc;                        // :590-597
  if ((rc = resolve_gainmap_jpeg(c, &gm_n)) != UHDR_HIP_NO_ERROR) return rc;
  if (!jpegr::icc_profile_srgb_transfer(yuv.colorGamut, icc)) return UHDR_HIP_ERROR_INVALID_COLORGAMUT;           // :599-600
  static thread_local HostBytes sdr_jpeg;
  sdr_jpeg.resize(yuv.width * yuv.height + 65536);
  size_t sdr_n = 0;
  if (jpeg_to_host(c, yuv, quality, &icc, sdr_jpeg, &sdr_n) != UHDR_HIP_NO_ERROR) return UHDR_HIP_ERROR_ENCODE_ERROR;   // :602-611
  return jpegr::append_gainmap_to(sdr_jpeg.data(), sdr_n, gm_jpeg.data(), gm_n, static_cast<const uint8_t*>(exif), exif_size, nullptr, 0, *metadata,
                                  static_cast<uint8_t*>(out), out_capacity, out_size);
}

// JpegR::getJPEGRInfo (jpegr.cpp:633-653)
int uhdr_hip_jpegr_info(const void* jpegr, size_t jpegr_size, uhdr_hip_jpeg_info_t* primary, uhdr_hip_jpeg_info_t* gainmap) {
  if (jpegr == nullptr || primary == nullptr) return UHDR_HIP_ERROR_BAD_PTR;
  const uint8_t* file = static_cast<const uint8_t*>(jpegr);
  jpegr::Range img[2];
  const int found = jpegr::find_images(file, jpegr_size, img);
  if (found == 0) return UHDR_HIP_ERROR_NO_IMAGES_FOUND;
  if (found == 1) return UHDR_HIP_ERROR_GAIN_MAP_IMAGE_NOT_FOUND;
  auto parse = [&](const jpegr::Range& r, uhdr_hip_jpeg_info_t* info) -> int {   // parseJpegInfo :878-915
    const uint8_t* j = file + r.begin;
    struct { int w, h; } di;
    if (!jpegr::has_valid_header(j, r.len) || !jpegr::dimensions(j, r.len, &di.w, &di.h)) return UHDR_HIP_ERROR_DECODE_ERROR;
    if (di.w > 8192 || di.h > 8192) return UHDR_HIP_ERROR_DECODE_ERROR;                                         // jpegdecoderhelper.cpp:251-256
    memset(info, 0, sizeof(*info));
    info->offset = r.begin; info->size = r.len;
    info->width = (size_t)di.w; info->height = (size_t)di.h;
    jpegr::first_packets(j, r.len, &info->xmp_offset, &info->xmp_size, &info->exif_offset, &info->exif_size, &info->icc_offset, &info->icc_size);
    if (info->xmp_size) info->xmp_offset += r.begin;
    if (info->exif_size) info->exif_offset += r.begin;
    if (info->icc_size) info->icc_offset += r.begin;
    return UHDR_HIP_NO_ERROR;
  };
  int rc = parse(img[0], primary);
  if (rc != UHDR_HIP_NO_ERROR) return rc;
  return gainmap != nullptr ? parse(img[1], gainmap) : UHDR_HIP_NO_ERROR;
}

int uhdr_hip_jpegr_metadata(const void* jpegr, size_t jpegr_size, uhdr_hip_metadata_t* metadata) {
  if (jpegr == nullptr || metadata == nullptr) return UHDR_HIP_ERROR_BAD_PTR;
  const uint8_t* file = static_cast<const uint8_t*>(jpegr);
  jpegr::Range img[2];
  const int found = jpegr::find_images(file, jpegr_size, img);
  if (found == 0) return UHDR_HIP_ERROR_NO_IMAGES_FOUND;
  if (found == 1) return UHDR_HIP_ERROR_GAIN_MAP_IMAGE_NOT_FOUND;
  const uint8_t* xmp = nullptr;
  size_t xmp_len = 0;
  if (!jpegr::first_xmp(file + img[1].begin, img[1].len, &xmp, &xmp_len) ||
      !jpegr::metadata_from_xmp(xmp, xmp_len, metadata))
    return UHDR_HIP_ERROR_METADATA_ERROR;
  return UHDR_HIP_NO_ERROR;
}

int uhdr_hip_lut_table(int which, float* out, size_t capacity, size_t* count) {
  if (out == nullptr || count == nullptr) return UHDR_HIP_ERROR_BAD_PTR;
  DeviceState* st = nullptr;
  const int rc = current_state(&st);
  if (rc != UHDR_HIP_NO_ERROR) return rc;
  uint32_t off = 0, n = 0;
  switch (which) {
    case 0: off = kLutSrgbInv; n = kLutSrgbInvN; break;
    case 1: off = kLutHlgInv; n = kLutHlgInvN; break;
    case 2: off = kLutPqInv; n = kLutPqInvN; break;
    case 4: off = kLutHlg; n = kLutHlgN; break;
    case 5: off = kLutPq; n = kLutPqN; break;
    default: return UHDR_HIP_ERROR_UNSUPPORTED_FEATURE;
  }
  *count = n;
  if (capacity < n) return UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE;
  HIP_TRY(hipMemcpy(out, st->lut + off, sizeof(float) * n, hipMemcpyDeviceToHost));
  return UHDR_HIP_NO_ERROR;
}

int uhdr_hip_gain_lut(const uhdr_hip_metadata_t* metadata, int with_display_boost, float display_boost, float* out) {
  if (metadata == nullptr || out == nullptr) return UHDR_HIP_ERROR_BAD_PTR;
  DeviceState* st = nullptr;
  int rc = current_state(&st);
  if (rc != UHDR_HIP_NO_ERROR) return rc;
  float factor = 1.0f;  // gainmapmath.h:152-159 (no display boost) | :161-169
  if (with_display_boost) factor = display_boost > 0 ? display_boost / metadata->maxContentBoost : 1.0f;
  CodecLease lease(st);
  if ((st = lease.get()) == nullptr) return UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE;
  if ((rc = stage_reserve(st, 6, sizeof(float) * kGainLutN)) != 0) return rc;
  float* d = static_cast<float*>(st->stage[6]);
  HIP_TRY(launch_build_gain_lut(d, std::log2((double)metadata->minContentBoost), std::log2((double)metadata->maxContentBoost),
                                factor, nullptr));
  HIP_TRY(hipMemcpy(out, d, sizeof(float) * kGainLutN, hipMemcpyDeviceToHost));
  return UHDR_HIP_NO_ERROR;
}

int uhdr_hip_idw_tables(int scale, float* out) {
  if (out == nullptr || scale <= 0) return UHDR_HIP_ERROR_BAD_PTR;
  std::vector<float> t;
  build_idw_tables(scale, t);
  memcpy(out, t.data(), t.size() * sizeof(float));
  return UHDR_HIP_NO_ERROR;
}

// ---------------------------------------------------------------------------------------------------
int uhdr_hip_generate_gainmap_batch(int n, const uhdr_hip_image_t* yuvs, const uhdr_hip_image_t* p010s, int hdr_tf,
                                    uhdr_hip_metadata_t* metadata, uhdr_hip_image_t* dests, int sdr_is_601,
                                    float* content_minmax, void* stream) {
  return uhdr_hip_generate_gainmap_batch_ex(n, yuvs, p010s, hdr_tf, metadata, dests, sdr_is_601, UHDR_HIP_GENERATE_EXACT,
                                            content_minmax, stream);
}

int uhdr_hip_generate_gainmap_batch_ex(int n, const uhdr_hip_image_t* yuvs, const uhdr_hip_image_t* p010s, int hdr_tf,
                                       uhdr_hip_metadata_t* metadata, uhdr_hip_image_t* dests, int sdr_is_601,
                                       int generate_mode, float* content_minmax, void* stream) {
  if (generate_mode != UHDR_HIP_GENERATE_EXACT && generate_mode != UHDR_HIP_GENERATE_LUT &&
      generate_mode != UHDR_HIP_GENERATE_UNFILTERED)
    return UHDR_HIP_ERROR_UNSUPPORTED_FEATURE;
  const bool lut = generate_mode == UHDR_HIP_GENERATE_LUT;
  if (n < 0 || (n > 0 && (yuvs == nullptr || p010s == nullptr || dests == nullptr)) || metadata == nullptr)
    return UHDR_HIP_ERROR_BAD_PTR;
  for (int i = 0; i < n; ++i) {
    const int rc = validate_generate(&yuvs[i], &p010s[i], hdr_tf, metadata, &dests[i]);
    if (rc != UHDR_HIP_NO_ERROR) return rc;
    if (dests[i].data == nullptr) return UHDR_HIP_ERROR_BAD_PTR;  // C-ABI: caller provides the map buffer
  }
  if (hdr_tf != UHDR_HIP_TF_LINEAR && hdr_tf != UHDR_HIP_TF_HLG && hdr_tf != UHDR_HIP_TF_PQ)
    return UHDR_HIP_ERROR_INVALID_TRANS_FUNC;
  DeviceState* st = nullptr;
  int rc = current_state(&st);
  if (rc != UHDR_HIP_NO_ERROR) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);

  fill_generate_metadata(hdr_tf, metadata);
  // Content min / max.  The filtered kernel (the usual case) works in a workspace of the library: its candidates are resolved, the
  // result written to content_minmax and the workspace cleared by k_stats_resolve.  The other kernels keep exact keys in
  // content_minmax itself: cleared here, turned into floats by k_stats_finalize at the end.
  uint32_t* keys = reinterpret_cast<uint32_t*>(content_minmax);

  int i = 0;
  while (i < n) {
    // chunk = up to kMaxChunk consecutive images of identical size, gamuts and alignment class
    const uhdr_hip_image_t& y0 = yuvs[i];
    GenConsts c = generate_consts(y0.colorGamut, p010s[i].colorGamut, hdr_tf, sdr_is_601, y0.width, y0.height, *metadata);
    c.stat_keys = keys ? keys + 2 * i : nullptr;
    c.stat_stride = 2;
    c.lut = lut ? st->lut : nullptr;
    GenBatch b;
    int m = 0;
    bool aligned = true;
    while (i + m < n && m < kMaxChunk) {
      const uhdr_hip_image_t& y = yuvs[i + m];
      if (y.width != y0.width || y.height != y0.height || y.colorGamut != y0.colorGamut ||
          p010s[i + m].colorGamut != p010s[i].colorGamut)
        break;
      b.img[m] = gen_image(y, p010s[i + m], dests[i + m].data);
      const bool a = gen_aligned(b.img[m], c.width, c.height);
      if (m == 0) aligned = a;
      else if (a != aligned) break;
      fill_generate_dest(&y, &dests[i + m]);
      ++m;
    }
    bool filter = generate_mode == UHDR_HIP_GENERATE_EXACT && c.flt_delta < 0.25f && c.min_boost >= 0.25f && c.max_boost <= 64.0f;
    // The filtered kernel of a large launch leaves its pixels in doubt and the exact extremes to k_generate_resolve.  A small
    // launch (one 4K image) would pay that second kernel's latency with nothing to hide it behind: without statistics it runs the
    // filtered kernel that falls back to the exact path in place.  With statistics the choice is between the two kernels and the
    // exact kernel (every pixel on the f64 path: 5.5 us per megapixel against 0.75 + the resolve kernel's ~20 us): the pair from
    // about half a 4K frame up (round 3: 8 x 1080p 91 -> 30 us, 8 x 4K 141 -> 80), the exact kernel below.
    const bool small = generate_is_small(c, m);
    const bool pair = keys != nullptr ? generate_resolve_pays(c, m) : !small;
    if (keys != nullptr && !pair) filter = false;
    const bool resolve = filter && aligned && !lut && pair;
    std::unique_lock<std::mutex> pair_lk(g_pair_mu, std::defer_lock);
    if (resolve) pair_lk.lock();
    if (resolve) {
      uint32_t* w = nullptr;
      const int wrc = stat_workspace(st, s, &w);
      if (wrc != UHDR_HIP_NO_ERROR) return wrc;
      c.stat_ws = w;
      c.stat_keys = keys ? w + 4 : nullptr;
      c.stat_stride = kStatWords;
      c.stat_out = keys ? content_minmax + 2 * i : nullptr;
      c.stat_spread = (m <= 16 && (uint64_t)((c.map_w + 1u) >> 1) * c.map_h >= 512u * 64u) ? 1u : 0u;
      c.stat_slots = generate_slot_waves(c, m);
    } else if (keys != nullptr) {
      HIP_TRY(launch_stats_init(keys + 2 * i, m, s));
    }
    HIP_TRY(launch_generate(c, b, m, hdr_tf, aligned, lut, filter, s));
    if (resolve) HIP_TRY(launch_stats_resolve(c, b, m, hdr_tf, aligned, s));
    else if (keys != nullptr) HIP_TRY(launch_stats_finalize(keys + 2 * i, m, s));
    i += m;
  }
  return UHDR_HIP_NO_ERROR;
}

int uhdr_hip_apply_gainmap_batch(int n, const uhdr_hip_image_t* yuvs, const uhdr_hip_image_t* maps,
                                 const uhdr_hip_metadata_t* metadata, int output_format, float max_display_boost,
                                 uhdr_hip_image_t* dests, int apply_mode, void* stream) {
  if (n < 0 || (n > 0 && (yuvs == nullptr || maps == nullptr || dests == nullptr)) || metadata == nullptr)
    return UHDR_HIP_ERROR_BAD_PTR;
  for (int i = 0; i < n; ++i) {
    const int rc = validate_apply(&yuvs[i], &maps[i], metadata, &dests[i]);
    if (rc != UHDR_HIP_NO_ERROR) return rc;
  }
  if (apply_mode != UHDR_HIP_APPLY_FAST && apply_mode != UHDR_HIP_APPLY_EXACT && apply_mode != UHDR_HIP_APPLY_LUT &&
      apply_mode != UHDR_HIP_APPLY_EXACT_UNFILTERED)
    return UHDR_HIP_ERROR_UNSUPPORTED_FEATURE;
  DeviceState* st = nullptr;
  int rc = current_state(&st);
  if (rc != UHDR_HIP_NO_ERROR) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool writes = apply_writes(output_format);
  for (int i = 0; i < n; ++i)
    if (writes && dests[i].data == nullptr) return UHDR_HIP_ERROR_BAD_PTR;

  int i = 0;
  while (i < n) {
    const uhdr_hip_image_t& y0 = yuvs[i];
    const uhdr_hip_image_t& m0 = maps[i];
    const int scale = (int)(y0.width / m0.width);
    const float* idw = nullptr;
    float* idw_transient = nullptr;
    if ((rc = idw_for_scale(st, scale, &idw, &idw_transient)) != UHDR_HIP_NO_ERROR) return rc;
    struct FreeAfter {   // a table too large to keep: freed when this chunk's launches have finished
      float* p; hipStream_t s;
      ~FreeAfter() { if (p) { (void)hipStreamSynchronize(s); (void)hipFree(p); } }
    } free_after{idw_transient, s};
    AppConsts c = apply_consts(y0, m0, *metadata, max_display_boost, idw);
    c.lut = apply_mode == UHDR_HIP_APPLY_LUT ? st->lut : nullptr;
    c.tab = st->lut;
    AppBatch b;
    int m = 0;
    bool fast = true;
    while (i + m < n && m < kMaxChunk) {
      const uhdr_hip_image_t& y = yuvs[i + m];
      const uhdr_hip_image_t& mp = maps[i + m];
      if (y.width != y0.width || y.height != y0.height || mp.width != m0.width || mp.height != m0.height) break;
      b.img[m] = app_image(y, mp, dests[i + m].data);
      const bool f = app_fast_s4(c, b.img[m]);
      if (m == 0) fast = f;
      else if (f != fast) break;
      fill_apply_dest(&y, &dests[i + m]);
      ++m;
    }
    // EXACT: f32 estimate, then the exact path on the pixels it leaves in doubt.  The estimate's error bounds are measured for
    // |log2 boost| <= 32 (tests/test_gpu_exact_filter.py); beyond that every pixel takes the exact path.
    std::unique_lock<std::mutex> pair_lk(g_pair_mu, std::defer_lock);
    if (writes && apply_mode == UHDR_HIP_APPLY_EXACT && (uint64_t)c.width * c.height <= 0xFFFFFFFFull &&
        std::fabs(c.log2_min_d) <= 32.0 && std::fabs(c.log2_max_d) <= 32.0) {
      pair_lk.lock();
      const uint32_t cap = ex_list_cap((uint64_t)c.width * c.height);
      uint32_t* w = nullptr;
      const int wrc = exact_workspace(st, s, m, cap, &w);
      if (wrc != UHDR_HIP_NO_ERROR) return wrc;
      c.ex_ws = w;
      c.ex_cap = cap;
    }
    if (writes) HIP_TRY(launch_apply(c, b, m, output_format, apply_mode, fast, s));
    i += m;
  }
  return UHDR_HIP_NO_ERROR;
}

// ---------------------------------------------------------------------------------------------------
int uhdr_hip_generate_gainmap(const uhdr_hip_image_t* yuv, const uhdr_hip_image_t* p010, int hdr_tf,
                              uhdr_hip_metadata_t* metadata, uhdr_hip_image_t* dest, int sdr_is_601, int mem_space,
                              void* stream) {
  return uhdr_hip_generate_gainmap_ex(yuv, p010, hdr_tf, metadata, dest, sdr_is_601, UHDR_HIP_GENERATE_EXACT, mem_space, stream);
}

int uhdr_hip_generate_gainmap_ex(const uhdr_hip_image_t* yuv, const uhdr_hip_image_t* p010, int hdr_tf,
                                 uhdr_hip_metadata_t* metadata, uhdr_hip_image_t* dest, int sdr_is_601, int generate_mode,
                                 int mem_space, void* stream) {
  int rc = validate_generate(yuv, p010, hdr_tf, metadata, dest);
  if (rc != UHDR_HIP_NO_ERROR) return rc;
  if (dest->data == nullptr) return UHDR_HIP_ERROR_BAD_PTR;
  if (mem_space == UHDR_HIP_MEM_DEVICE)
    return uhdr_hip_generate_gainmap_batch_ex(1, yuv, p010, hdr_tf, metadata, dest, sdr_is_601, generate_mode, nullptr, stream);

  DeviceState* st = nullptr;
  if ((rc = current_state(&st)) != UHDR_HIP_NO_ERROR) return rc;
  StageLease lease(st);
  StageSet* ss = lease.get();
  if (ss == nullptr) return UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  uhdr_hip_image_t dy, dp, dm = *dest;
  if ((rc = stage_yuv420_in(ss, 0, *yuv, &dy, s)) != 0) return rc;
  if ((rc = stage_p010_in(ss, 2, *p010, &dp, s)) != 0) return rc;
  const size_t mw = yuv->width / 4, mh = yuv->height / 4;
  if ((rc = stage_reserve(ss, 4, mw * mh)) != 0) return rc;
  dm.data = ss->stage[4];
  rc = uhdr_hip_generate_gainmap_batch_ex(1, &dy, &dp, hdr_tf, metadata, &dm, sdr_is_601, generate_mode, nullptr, stream);
  if (rc != UHDR_HIP_NO_ERROR) return rc;
  if (mw * mh) HIP_TRY(hipMemcpyAsync(dest->data, dm.data, mw * mh, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  fill_generate_dest(yuv, dest);
  return UHDR_HIP_NO_ERROR;
}

int uhdr_hip_apply_gainmap(const uhdr_hip_image_t* yuv, const uhdr_hip_image_t* map, const uhdr_hip_metadata_t* metadata,
                           int output_format, float max_display_boost, uhdr_hip_image_t* dest, int apply_mode,
                           int mem_space, void* stream) {
  int rc = validate_apply(yuv, map, metadata, dest);
  if (rc != UHDR_HIP_NO_ERROR) return rc;
  if (mem_space == UHDR_HIP_MEM_DEVICE)
    return uhdr_hip_apply_gainmap_batch(1, yuv, map, metadata, output_format, max_display_boost, dest, apply_mode, stream);

  const bool writes = apply_writes(output_format);
  if (writes && dest->data == nullptr) return UHDR_HIP_ERROR_BAD_PTR;
  DeviceState* st = nullptr;
  if ((rc = current_state(&st)) != UHDR_HIP_NO_ERROR) return rc;
  StageLease lease(st);
  StageSet* ss = lease.get();
  if (ss == nullptr) return UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  uhdr_hip_image_t dy, dm = *map, dd = *dest;
  if ((rc = stage_yuv420_in(ss, 0, *yuv, &dy, s)) != 0) return rc;
  const size_t map_bytes = map->width * map->height;  // the reference reads the map with stride == width
  if ((rc = stage_reserve(ss, 4, map_bytes)) != 0) return rc;
  HIP_TRY(hipMemcpyAsync(ss->stage[4], map->data, map_bytes, hipMemcpyHostToDevice, s));
  dm.data = ss->stage[4];
  const size_t out_bytes = writes ? yuv->width * yuv->height * apply_bpp(output_format) : 0;
  if ((rc = stage_reserve(ss, 5, out_bytes)) != 0) return rc;
  dd.data = ss->stage[5];
  rc = uhdr_hip_apply_gainmap_batch(1, &dy, &dm, metadata, output_format, max_display_boost, &dd, apply_mode, stream);
  if (rc != UHDR_HIP_NO_ERROR) return rc;
  if (out_bytes) HIP_TRY(hipMemcpyAsync(dest->data, dd.data, out_bytes, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  fill_apply_dest(yuv, dest);
  return UHDR_HIP_NO_ERROR;
}

namespace {
ToneImage tone_image(const uhdr_hip_image_t& sd, const uhdr_hip_image_t& dd, bool* aligned) {
  ToneImage t;
  t.sy = static_cast<const uint16_t*>(sd.data);
  t.suv = static_cast<const uint16_t*>(sd.chroma_data);
  t.dy = static_cast<uint8_t*>(dd.data);
  t.du = static_cast<uint8_t*>(dd.chroma_data);
  t.dv = t.du + (dd.chroma_stride * dd.height / 2);  // ultrahdr.cpp:539
  t.sy_stride = (uint32_t)sd.luma_stride; t.suv_stride = (uint32_t)sd.chroma_stride;
  t.dy_stride = (uint32_t)dd.luma_stride; t.dc_stride = (uint32_t)dd.chroma_stride;
  t.width = (uint32_t)sd.width; t.height = (uint32_t)sd.height;
  *aligned = t.width % 16u == 0 && al(t.sy, 16) && t.sy_stride % 8u == 0 && al(t.suv, 16) &&
             t.suv_stride % 8u == 0 && al(t.dy, 8) && t.dy_stride % 8u == 0 && t.dy_stride >= t.width &&
             al(t.du, 8) && al(t.dv, 8) && t.dc_stride % 8u == 0 && t.dc_stride >= t.width / 2u;
  return t;
}
int tonemap_check(const uhdr_hip_image_t* src, const uhdr_hip_image_t* dest) {   // ultrahdr.cpp:518-523
  if (src == nullptr || dest == nullptr) return UHDR_HIP_ERROR_BAD_PTR;
  if (src->width != dest->width || src->height != dest->height) return UHDR_HIP_ERROR_RESOLUTION_MISMATCH;
  if (src->data == nullptr || src->chroma_data == nullptr || dest->data == nullptr || dest->chroma_data == nullptr)
    return UHDR_HIP_ERROR_BAD_PTR;  // (the reference would dereference them)
  return UHDR_HIP_NO_ERROR;
}
}  // namespace

// UltraHdr::toneMap over n images in device memory: images of equal size (and equal alignment class) share a launch, grid.z = image
int uhdr_hip_tonemap_batch(int n, const uhdr_hip_image_t* srcs, uhdr_hip_image_t* dests, void* stream) {
  if (n < 0 || (n > 0 && (srcs == nullptr || dests == nullptr))) return UHDR_HIP_ERROR_BAD_PTR;
  for (int i = 0; i < n; ++i) {
    const int rc = tonemap_check(&srcs[i], &dests[i]);
    if (rc != UHDR_HIP_NO_ERROR) return rc;
  }
  DeviceState* st = nullptr;
  int rc = current_state(&st);
  if (rc != UHDR_HIP_NO_ERROR) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  int i = 0;
  while (i < n) {
    ToneBatch b;
    bool aligned = true;
    int m = 0;
    while (i + m < n && m < kToneChunk) {
      bool a;
      const ToneImage t = tone_image(srcs[i + m], dests[i + m], &a);
      if (m == 0) aligned = a;
      else if (a != aligned || t.width != b.img[0].width || t.height != b.img[0].height) break;
      b.img[m++] = t;
    }
    HIP_TRY(launch_tonemap(b, m, aligned, s));
    for (int k = 0; k < m; ++k) dests[i + k].colorGamut = srcs[i + k].colorGamut;  // ultrahdr.cpp:556
    i += m;
  }
  return UHDR_HIP_NO_ERROR;
}

int uhdr_hip_tonemap(const uhdr_hip_image_t* src, uhdr_hip_image_t* dest, int mem_space, void* stream) {
  int rc = tonemap_check(src, dest);
  if (rc != UHDR_HIP_NO_ERROR) return rc;
  if (mem_space == UHDR_HIP_MEM_DEVICE) return uhdr_hip_tonemap_batch(1, src, dest, stream);
  DeviceState* st = nullptr;
  if ((rc = current_state(&st)) != UHDR_HIP_NO_ERROR) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);

  auto run = [&](const uhdr_hip_image_t& sd, const uhdr_hip_image_t& dd) -> int {
    ToneBatch b;
    bool aligned;
    b.img[0] = tone_image(sd, dd, &aligned);
    HIP_TRY(launch_tonemap(b, 1, aligned, s));
    return UHDR_HIP_NO_ERROR;
  };

  StageLease lease(st);
  StageSet* ss = lease.get();
  if (ss == nullptr) return UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE;
  uhdr_hip_image_t ds, dd = *dest;
  if ((rc = stage_p010_in(ss, 2, *src, &ds, s)) != 0) return rc;
  const size_t h = dest->height, ls = dest->luma_stride, cs = dest->chroma_stride;
  if ((rc = stage_reserve(ss, 0, ls * h)) != 0) return rc;
  if ((rc = stage_reserve(ss, 1, cs * h + cs)) != 0) return rc;
  dd.data = ss->stage[0];
  dd.chroma_data = ss->stage[1];
  if ((rc = run(ds, dd)) != 0) return rc;
  // the reference writes luma_stride bytes per luma row and chroma_stride bytes per chroma row
  if (ls * h) HIP_TRY(hipMemcpyAsync(dest->data, dd.data, ls * h, hipMemcpyDeviceToHost, s));
  const size_t v_off = cs * h / 2;
  if (cs * (h / 2)) {
    HIP_TRY(hipMemcpyAsync(dest->chroma_data, dd.chroma_data, cs * (h / 2), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(static_cast<uint8_t*>(dest->chroma_data) + v_off, static_cast<uint8_t*>(dd.chroma_data) + v_off,
                           cs * (h / 2), hipMemcpyDeviceToHost, s));
  }
  HIP_TRY(hipStreamSynchronize(s));
  dest->colorGamut = src->colorGamut;
  return UHDR_HIP_NO_ERROR;
}

namespace {
int convert_check(const uhdr_hip_image_t* image, int src_encoding, int dest_encoding) {   // jpegr.cpp:1134-1197
  if (image == nullptr) return UHDR_HIP_ERROR_BAD_PTR;
  if (src_encoding == UHDR_HIP_CG_UNSPECIFIED || dest_encoding == UHDR_HIP_CG_UNSPECIFIED)
    return UHDR_HIP_ERROR_INVALID_COLORGAMUT;
  if (!valid_gamut(src_encoding) || !valid_gamut(dest_encoding)) return UHDR_HIP_ERROR_INVALID_COLORGAMUT;
  return UHDR_HIP_NO_ERROR;
}
}  // namespace

// JpegR::convertYuv over n images in device memory, in place: images of equal size (and equal alignment class) share a launch
int uhdr_hip_convert_yuv_batch(int n, uhdr_hip_image_t* images, int src_encoding, int dest_encoding, void* stream) {
  if (n < 0 || (n > 0 && images == nullptr)) return UHDR_HIP_ERROR_BAD_PTR;
  int rc = convert_check(n > 0 ? &images[0] : nullptr, src_encoding, dest_encoding);
  if (n == 0) rc = (src_encoding == UHDR_HIP_CG_UNSPECIFIED || dest_encoding == UHDR_HIP_CG_UNSPECIFIED || !valid_gamut(src_encoding) ||
                    !valid_gamut(dest_encoding)) ? UHDR_HIP_ERROR_INVALID_COLORGAMUT : UHDR_HIP_NO_ERROR;
  if (rc != UHDR_HIP_NO_ERROR) return rc;
  if (src_encoding == dest_encoding) return UHDR_HIP_NO_ERROR;
  for (int i = 0; i < n; ++i)
    if (images[i].data == nullptr || images[i].chroma_data == nullptr) return UHDR_HIP_ERROR_BAD_PTR;
  const float* m = yuv_matrix(src_encoding, dest_encoding);
  DeviceState* st = nullptr;
  if ((rc = current_state(&st)) != UHDR_HIP_NO_ERROR) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  int i = 0;
  while (i < n) {
    CvtBatch b;
    bool aligned = true;
    int k = 0;
    while (i + k < n && k < kToneChunk) {
      bool a;
      const CvtImage t = cvt_image(images[i + k], images[i + k], m, &a);
      if (k == 0) aligned = a;
      else if (a != aligned || t.width != b.img[0].width || t.height != b.img[0].height) break;
      b.img[k++] = t;
    }
    HIP_TRY(launch_convert_yuv(b, k, aligned, s));
    i += k;
  }
  return UHDR_HIP_NO_ERROR;
}

int uhdr_hip_convert_yuv(uhdr_hip_image_t* image, int src_encoding, int dest_encoding, int mem_space, void* stream) {
  int rc = convert_check(image, src_encoding, dest_encoding);
  if (rc != UHDR_HIP_NO_ERROR) return rc;
  if (src_encoding == dest_encoding) return UHDR_HIP_NO_ERROR;
  if (image->data == nullptr || image->chroma_data == nullptr) return UHDR_HIP_ERROR_BAD_PTR;
  if (mem_space == UHDR_HIP_MEM_DEVICE) return uhdr_hip_convert_yuv_batch(1, image, src_encoding, dest_encoding, stream);
  const float* m = yuv_matrix(src_encoding, dest_encoding);
  DeviceState* st = nullptr;
  if ((rc = current_state(&st)) != UHDR_HIP_NO_ERROR) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  auto run = [&](const uhdr_hip_image_t& d) -> int { return convert_yuv_into(d, d, m, s); };
  StageLease lease(st);
  if (lease.get() == nullptr) return UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE;
  uhdr_hip_image_t d;
  if ((rc = stage_yuv420_in(lease.get(), 0, *image, &d, s)) != 0) return rc;
  if ((rc = run(d)) != 0) return rc;
  const size_t w = image->width, h = image->height, cw = w / 2, ch = h / 2;
  if ((rc = d2h_plane(image->data, image->luma_stride, d.data, d.luma_stride, cw * 2, ch * 2, 1, s)) != 0) return rc;
  uint8_t* hu = static_cast<uint8_t*>(image->chroma_data);
  const uint8_t* du = static_cast<const uint8_t*>(d.chroma_data);
  if ((rc = d2h_plane(hu, image->chroma_stride, du, d.chroma_stride, cw, ch, 1, s)) != 0) return rc;
  if ((rc = d2h_plane(hu + image->chroma_stride * (h / 2), image->chroma_stride, du + d.chroma_stride * (h / 2),
                      d.chroma_stride, cw, ch, 1, s)) != 0)
    return rc;
  HIP_TRY(hipStreamSynchronize(s));
  return UHDR_HIP_NO_ERROR;
}

// ---------------------------------------------------------------------------------------------------
// editorhelper effects.  fx_plan() restates the layout rules of editorhelper.cpp (output dims / strides /
// chroma placement, plane-by-plane index maps); the bytes are moved by k_effect.
// ---------------------------------------------------------------------------------------------------
namespace {
enum { FXK_CROP, FXK_MIRROR, FXK_ROTATE, FXK_RESIZE };

FxJob fx_job(const uint8_t* src, uint8_t* dst, size_t rows, size_t cols, size_t dst_stride, size_t src_stride, size_t in_w,
             size_t in_h, int op) {
  FxJob j;
  j.src = src; j.dst = dst; j.rows = (uint32_t)rows; j.cols = (uint32_t)cols;
  j.dst_stride = (uint32_t)dst_stride; j.src_stride = (uint32_t)src_stride;
  j.in_w = (uint32_t)in_w; j.in_h = (uint32_t)in_h;
  j.row_num = j.row_den = j.col_num = j.col_den = 1; j.op = op;
  return j;
}

// in/out hold pointers valid in the memory space the kernel will run in
int fx_plan(int kind, const uhdr_hip_image_t& in, int a, int b, int c, int d, uhdr_hip_image_t* out, FxJobs* jobs) {
  const bool mono = in.pixelFormat == UHDR_HIP_PIX_FMT_MONOCHROME;
  const size_t iw = in.width, ih = in.height;
  const size_t ls = in.luma_stride != 0 ? in.luma_stride : iw;                 // editorhelper.cpp:44
  const size_t cs = in.chroma_stride != 0 ? in.chroma_stride : (ls >> 1);      // :60-61
  const uint8_t* sy = static_cast<const uint8_t*>(in.data);
  const uint8_t* sc = in.chroma_data ? static_cast<const uint8_t*>(in.chroma_data) : sy + ls * ih;  // :66-69
  uint8_t* dy = static_cast<uint8_t*>(out->data);
  out->colorGamut = in.colorGamut;
  out->pixelFormat = in.pixelFormat;
  jobs->n = 0;
  size_t ow, oh, ols;
  if (kind == FXK_CROP) {           // a=left b=right c=top d=bottom   (:26-76)
    ow = (size_t)(b - a + 1); oh = (size_t)(d - c + 1); ols = ow;
    jobs->job[jobs->n++] = fx_job(sy + ls * c + a, dy, oh, ow, ols, ls, iw, ih, FX_COPY);
  } else if (kind == FXK_MIRROR) {  // a=direction                      (:78-170)
    ow = iw; oh = ih; ols = ls;
    jobs->job[jobs->n++] = fx_job(sy, dy, oh, ow, ols, ls, iw, ih, a == 0 ? FX_FLIP_V : FX_FLIP_H);
  } else if (kind == FXK_ROTATE) {  // a=degrees                        (:172-306)
    if (a == 180) { ow = iw; oh = ih; ols = ls; } else { ow = ih; oh = iw; ols = ow; }
    jobs->job[jobs->n++] = fx_job(sy, dy, oh, ow, ols, ls, iw, ih, a == 90 ? FX_ROT90 : a == 180 ? FX_ROT180 : FX_ROT270);
  } else {                          // a=out_width b=out_height         (:308-360)
    ow = (size_t)a; oh = (size_t)b; ols = ow;
    FxJob j = fx_job(sy, dy, oh, ow, ols, ls, iw, ih, FX_RESIZE);
    j.row_num = (uint32_t)ih; j.row_den = (uint32_t)oh; j.col_num = (uint32_t)iw; j.col_den = (uint32_t)ow;
    jobs->job[jobs->n++] = j;
  }
  out->width = ow; out->height = oh; out->luma_stride = ols;
  if (mono) return UHDR_HIP_NO_ERROR;
  const size_t ocs = ols / 2;
  uint8_t* dc = dy + ols * oh;
  out->chroma_stride = ocs;
  out->chroma_data = dc;
  if (kind == FXK_CROP) {           // one copy of `oh` rows starting in the U plane (:70-73)
    jobs->job[jobs->n++] = fx_job(sc + cs * (c / 2) + (a / 2), dc, oh, ow / 2, ocs, cs, iw / 2, ih, FX_COPY);
  } else if (kind == FXK_RESIZE) {  // one pass over U and V (:350-357): rows and columns use the LUMA ratios
    FxJob j = fx_job(sc, dc, oh, ow / 2, ocs, cs, iw / 2, ih, FX_RESIZE);
    j.row_num = (uint32_t)ih; j.row_den = (uint32_t)oh; j.col_num = (uint32_t)iw; j.col_den = (uint32_t)ow;
    jobs->job[jobs->n++] = j;
  } else {                          // U then V, each (ih/2) x (iw/2)
    const int op = jobs->job[0].op;
    for (int p = 0; p < 2; ++p)
      jobs->job[jobs->n++] = fx_job(sc + (p ? cs * (ih / 2) : 0), dc + (p ? ocs * (oh / 2) : 0), oh / 2, ow / 2, ocs, cs, iw / 2,
                                    ih / 2, op);
  }
  return UHDR_HIP_NO_ERROR;
}

int fx_run(int kind, const uhdr_hip_image_t* in, int a, int b, int c, int d, uhdr_hip_image_t* out, int mem_space, void* stream) {
  // argument checks in the reference's order (editorhelper.cpp:29-39, 81-88, 174-185, 310-317)
  if (in == nullptr || in->data == nullptr || out == nullptr || out->data == nullptr) return UHDR_HIP_ERROR_BAD_PTR;
  if (kind == FXK_CROP && (a < 0 || (size_t)b >= in->width || c < 0 || (size_t)d >= in->height))
    return UHDR_HIP_ERROR_INVALID_CROPPING_PARAMETERS;
  if (kind == FXK_ROTATE && a != 90 && a != 180 && a != 270) return UHDR_HIP_ERROR_INVALID_CROPPING_PARAMETERS;
  if (in->pixelFormat != UHDR_HIP_PIX_FMT_YUV420 && in->pixelFormat != UHDR_HIP_PIX_FMT_MONOCHROME)
    return UHDR_HIP_ERROR_UNSUPPORTED_FEATURE;
  if (kind == FXK_RESIZE && (a <= 0 || b <= 0)) return UHDR_HIP_ERROR_INVALID_CROPPING_PARAMETERS;  // (ref: division by zero)
  if (kind == FXK_CROP && (b < a || d < c)) return UHDR_HIP_ERROR_INVALID_CROPPING_PARAMETERS;      // (ref: negative memcpy size)
  DeviceState* st = nullptr;
  int rc = current_state(&st);
  if (rc != UHDR_HIP_NO_ERROR) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  FxJobs jobs;
  if (mem_space == UHDR_HIP_MEM_DEVICE) {
    if ((rc = fx_plan(kind, *in, a, b, c, d, out, &jobs)) != 0) return rc;
    HIP_TRY(launch_effect(jobs, s));
    return UHDR_HIP_NO_ERROR;
  }
  // host memory: stage exactly the bytes the reference touches
  CodecLease lease(st);
  if ((st = lease.get()) == nullptr) return UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE;
  const bool mono = in->pixelFormat == UHDR_HIP_PIX_FMT_MONOCHROME;
  const size_t iw = in->width, ih = in->height;
  const size_t ls = in->luma_stride != 0 ? in->luma_stride : iw;
  const size_t cs = in->chroma_stride != 0 ? in->chroma_stride : (ls >> 1);
  const size_t luma_bytes = ih ? ls * (ih - 1) + iw : 0;
  const size_t chroma_rows = mono ? 0 : ih;  // U and V stacked (crop / resize walk them as one plane)
  const size_t chroma_bytes = chroma_rows ? cs * (chroma_rows - 1) + iw / 2 : 0;
  if ((rc = stage_reserve(st, 0, luma_bytes)) != 0) return rc;
  if ((rc = stage_reserve(st, 1, chroma_bytes)) != 0) return rc;
  if (luma_bytes) HIP_TRY(hipMemcpyAsync(st->stage[0], in->data, luma_bytes, hipMemcpyHostToDevice, s));
  const uint8_t* hc = in->chroma_data ? static_cast<const uint8_t*>(in->chroma_data)
                                      : static_cast<const uint8_t*>(in->data) + ls * ih;
  if (chroma_bytes) HIP_TRY(hipMemcpyAsync(st->stage[1], hc, chroma_bytes, hipMemcpyHostToDevice, s));
  uhdr_hip_image_t din = *in, dout = *out;
  din.data = st->stage[0];
  din.chroma_data = mono ? nullptr : st->stage[1];
  din.luma_stride = ls; din.chroma_stride = cs;
  // upper bound of the output extent: every layout rule yields <= max(ls, ow) * oh * 3/2 bytes
  uhdr_hip_image_t probe = *out;
  uint8_t dummy = 0;
  probe.data = &dummy;
  FxJobs pj;
  fx_plan(kind, *in, a, b, c, d, &probe, &pj);
  const size_t out_luma = probe.luma_stride * probe.height;
  const size_t out_chroma_rows = mono ? 0 : ((kind == FXK_CROP || kind == FXK_RESIZE) ? probe.height : 2 * (probe.height / 2));
  const size_t out_bytes = out_luma + (out_chroma_rows ? probe.chroma_stride * (out_chroma_rows - 1) + probe.width / 2 : 0);
  if ((rc = stage_reserve(st, 5, out_bytes)) != 0) return rc;
  dout.data = st->stage[5];
  if ((rc = fx_plan(kind, din, a, b, c, d, &dout, &jobs)) != 0) return rc;
  // bytes the kernel does not write (stride padding of mirror / rotate-180) must keep the caller's content
  if (out_bytes) HIP_TRY(hipMemcpyAsync(st->stage[5], out->data, out_bytes, hipMemcpyHostToDevice, s));
  HIP_TRY(launch_effect(jobs, s));
  if (out_bytes) HIP_TRY(hipMemcpyAsync(out->data, st->stage[5], out_bytes, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  out->width = dout.width; out->height = dout.height; out->colorGamut = dout.colorGamut; out->pixelFormat = dout.pixelFormat;
  out->luma_stride = dout.luma_stride;
  if (!mono) {
    out->chroma_stride = dout.chroma_stride;
    out->chroma_data = static_cast<uint8_t*>(out->data) + out->luma_stride * out->height;
  }
  return UHDR_HIP_NO_ERROR;
}
}  // namespace

int uhdr_hip_crop(const uhdr_hip_image_t* in_img, int left, int right, int top, int bottom, uhdr_hip_image_t* out_img,
                  int mem_space, void* stream) {
  return fx_run(FXK_CROP, in_img, left, right, top, bottom, out_img, mem_space, stream);
}
int uhdr_hip_mirror(const uhdr_hip_image_t* in_img, int mirror_dir, uhdr_hip_image_t* out_img, int mem_space, void* stream) {
  return fx_run(FXK_MIRROR, in_img, mirror_dir, 0, 0, 0, out_img, mem_space, stream);
}
int uhdr_hip_rotate(const uhdr_hip_image_t* in_img, int clockwise_degree, uhdr_hip_image_t* out_img, int mem_space,
                    void* stream) {
  return fx_run(FXK_ROTATE, in_img, clockwise_degree, 0, 0, 0, out_img, mem_space, stream);
}
int uhdr_hip_resize(const uhdr_hip_image_t* in_img, int out_width, int out_height, uhdr_hip_image_t* out_img, int mem_space,
                    void* stream) {
  return fx_run(FXK_RESIZE, in_img, out_width, out_height, 0, 0, out_img, mem_space, stream);
}

// addEffects (editorhelper.cpp:362-446): the chain stays in device memory, two ping-pong temporaries
int uhdr_hip_add_effects(const uhdr_hip_image_t* in, const uhdr_hip_effect_t* effects, int n, uhdr_hip_image_t* out, int mem_space,
                         void* stream) {
  if (in == nullptr || in->data == nullptr || out == nullptr || out->data == nullptr || n < 0 || (n > 0 && effects == nullptr))
    return UHDR_HIP_ERROR_BAD_PTR;
  if (in->pixelFormat != UHDR_HIP_PIX_FMT_YUV420 && in->pixelFormat != UHDR_HIP_PIX_FMT_MONOCHROME)
    return UHDR_HIP_ERROR_UNSUPPORTED_FEATURE;
  const bool mono = in->pixelFormat == UHDR_HIP_PIX_FMT_MONOCHROME;
  const bool host = mem_space != UHDR_HIP_MEM_DEVICE;
  DeviceState* st = nullptr;
  int rc = current_state(&st);
  if (rc != UHDR_HIP_NO_ERROR) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  CodecLease lease(st);   // the temporaries: this call's own
  if ((st = lease.get()) == nullptr) return UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE;

  auto packed = [mono](size_t w, size_t h) { return mono ? w * h : w * h * 3 / 2; };
  size_t size = packed(in->width, in->height);
  const size_t size0 = size;
  // extents of every intermediate image (all tightly packed after the first effect)
  size_t max_bytes = size0;
  {
    size_t w = in->width, h = in->height;
    for (int i = 0; i < n; ++i) {
      const uhdr_hip_effect_t& e = effects[i];
      if (e.type == 0) { if (e.b < e.a || e.d < e.c) return UHDR_HIP_ERROR_INVALID_CROPPING_PARAMETERS; w = (size_t)(e.b - e.a + 1); h = (size_t)(e.d - e.c + 1); }
      else if (e.type == 2) { if (e.a == 90 || e.a == 270) std::swap(w, h); }
      else if (e.type == 3) { if (e.a <= 0 || e.b <= 0) return UHDR_HIP_ERROR_INVALID_CROPPING_PARAMETERS; w = (size_t)e.a; h = (size_t)e.b; }
      else if (e.type != 1) return UHDR_HIP_ERROR_BAD_PTR;
      max_bytes = std::max(max_bytes, packed(w, h));
    }
  }
  // The reference's out image is written after every step (:432-437), so beyond the last result it keeps the tails of the
  // earlier, larger ones.  dev_out plays that image: the caller's buffer itself, or its device stand-in for host calls.
  uint8_t* dev_out = static_cast<uint8_t*>(out->data);
  if (host) {
    if ((rc = stage_reserve(st, 5, max_bytes + 64)) != 0) return rc;
    dev_out = static_cast<uint8_t*>(st->stage[5]);
  }
  // :383-390: the descriptor and width*height(*3/2) bytes starting at the luma pointer are copied first
  out->width = in->width; out->height = in->height; out->colorGamut = in->colorGamut; out->pixelFormat = in->pixelFormat;
  out->luma_stride = in->luma_stride; out->chroma_stride = in->chroma_stride;
  if (size0) HIP_TRY(hipMemcpyAsync(dev_out, in->data, size0, host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, s));
  // the first effect reads the caller's image (any strides); host images are staged as fx_run does
  uhdr_hip_image_t last = *in;
  const size_t ls0 = in->luma_stride != 0 ? in->luma_stride : in->width;
  const size_t cs0 = in->chroma_stride != 0 ? in->chroma_stride : (ls0 >> 1);
  if (host && n > 0) {
    const size_t luma_bytes = in->height ? ls0 * (in->height - 1) + in->width : 0;
    const size_t chroma_bytes = (!mono && in->height) ? cs0 * (in->height - 1) + in->width / 2 : 0;
    if ((rc = stage_reserve(st, 0, luma_bytes)) != 0) return rc;
    if ((rc = stage_reserve(st, 1, chroma_bytes)) != 0) return rc;
    if (luma_bytes) HIP_TRY(hipMemcpyAsync(st->stage[0], in->data, luma_bytes, hipMemcpyHostToDevice, s));
    const uint8_t* hc = in->chroma_data ? static_cast<const uint8_t*>(in->chroma_data) : static_cast<const uint8_t*>(in->data) + ls0 * in->height;
    if (chroma_bytes) HIP_TRY(hipMemcpyAsync(st->stage[1], hc, chroma_bytes, hipMemcpyHostToDevice, s));
    last.data = st->stage[0];
    last.chroma_data = mono ? nullptr : st->stage[1];
    last.luma_stride = ls0; last.chroma_stride = cs0;
  }
  if ((rc = stage_reserve(st, 2, max_bytes + 64)) != 0) return rc;
  for (int i = 0; i < n; ++i) {
    const uhdr_hip_effect_t& e = effects[i];
    const size_t lls = last.luma_stride != 0 ? last.luma_stride : last.width;
    const bool keeps_stride = e.type == 1 || (e.type == 2 && e.a == 180);
    if (keeps_stride && (lls != last.width || (!mono && last.chroma_stride != 0 && last.chroma_stride != last.width / 2)))
      return UHDR_HIP_ERROR_UNSUPPORTED_FEATURE;   // the reference writes past its `size`-byte temporary here
    // same argument checks as the single effects (fx_run)
    if (e.type == 0 && (e.a < 0 || (size_t)e.b >= last.width || e.c < 0 || (size_t)e.d >= last.height)) return UHDR_HIP_ERROR_INVALID_CROPPING_PARAMETERS;
    if (e.type == 2 && e.a != 90 && e.a != 180 && e.a != 270) return UHDR_HIP_ERROR_INVALID_CROPPING_PARAMETERS;
    uhdr_hip_image_t tmp = *out;
    tmp.data = st->stage[2];
    FxJobs jobs;
    const int kind = e.type == 0 ? FXK_CROP : e.type == 1 ? FXK_MIRROR : e.type == 2 ? FXK_ROTATE : FXK_RESIZE;
    if ((rc = fx_plan(kind, last, e.a, e.b, e.c, e.d, &tmp, &jobs)) != 0) return rc;
    HIP_TRY(launch_effect(jobs, s));
    size = e.type == 0 ? packed((size_t)(e.b - e.a + 1), (size_t)(e.d - e.c + 1))
         : e.type == 3 ? packed((size_t)e.a, (size_t)e.b) : packed(last.width, last.height);   // :395-430
    if (size) HIP_TRY(hipMemcpyAsync(dev_out, tmp.data, size, hipMemcpyDeviceToDevice, s));    // the "deep copy", :437
    last = tmp;
    last.data = dev_out;                                                                         // last = out_img, :442
    last.chroma_data = mono ? nullptr : dev_out + last.luma_stride * last.height;               // :438-440
  }
  if (n > 0) {
    out->width = last.width; out->height = last.height; out->colorGamut = last.colorGamut; out->pixelFormat = last.pixelFormat;
    out->luma_stride = last.luma_stride; out->chroma_stride = last.chroma_stride;
    if (!mono) out->chroma_data = static_cast<uint8_t*>(out->data) + out->luma_stride * out->height;
  }
  if (host) {
    if (max_bytes) HIP_TRY(hipMemcpyAsync(out->data, dev_out, max_bytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
  }
  return UHDR_HIP_NO_ERROR;
}

}  // extern "C"
