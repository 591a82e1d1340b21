// uhdr_kernels.h -- internal host<->kernel interface (POD kernel arguments + launchers).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

namespace uhdr {

constexpr int kMaxChunk = 64;  // images per launch (descriptors travel in the 4 KiB kernarg segment: 64 x 56 B + consts)

// ---- LUT mode (gainmapmath.cpp:21-64 static tables; opt-in) --------------------
// one device buffer per device, filled once at uhdr_hip_init() by k_build_luts: table[i] = f((float)i / (float)(N - 1))
constexpr uint32_t kLutSrgbInvN = 1024, kLutHlgInvN = 4096, kLutPqInvN = 4096, kLutHlgN = 65536, kLutPqN = 65536;
constexpr uint32_t kLutSrgbInv = 0, kLutHlgInv = kLutSrgbInv + kLutSrgbInvN, kLutPqInv = kLutHlgInv + kLutHlgInvN,
                   kLutHlg = kLutPqInv + kLutPqInvN, kLutPq = kLutHlg + kLutHlgN, kLutTotal = kLutPq + kLutPqN;
// FAST apply only (not one of the reference's tables): the sRGB EOTF as 4096 line segments (c0, c1), cell = round(x * 32767) >> 3,
// appended to the LUT buffer by uhdr_hip_init (see k_apply_s4)
constexpr uint32_t kSrgbLineCells = 4096, kSrgbLine = kLutTotal, kLutBufferFloats = kLutTotal + 2 * kSrgbLineCells;
constexpr uint32_t kGainLutN = 1024;  // kGainFactorNumEntries, gainmapmath.h:149-150

// ---- generate ----------------------------------------------------------------------------------
struct GenConsts {
  float sdr_cr, sdr_gcb, sdr_gcr, sdr_cb;  // SDR YUV->RGB (gamut of the SDR image, or 601)
  float hdr_cr, hdr_gcb, hdr_gcr, hdr_cb;  // HDR YUV->RGB (gamut of the P010 image)
  float lum_r, lum_g, lum_b;               // luminance of the SDR gamut (used for both images)
  float gm[9];                             // HDR->SDR gamut matrix
  int gm_identity;
  float hdr_white_nits;
  float bias4096;                          // == 4096.0f (kept opaque to the optimizer, see gen_pair)
  float min_boost, max_boost, log2_min, log2_max;
  // encode_gain_guarded: 255/(double)(log2_max - log2_min) and the bytes of gain == min / gain == max
  double enc_scale;
  uint32_t enc_byte_min, enc_byte_max;
  uint32_t width, height, map_w, map_h;
  uint32_t* stat_keys;  // 2 words per image of the launch (min key, max key), or nullptr
  const float* lut;     // device LUT buffer (LUT mode only)
  // f32 pre-filter of the exact path (see gen_pair): code-value scale, half-width of the "too close to an integer"
  // band, and the gains below / above which the clamp certainly applies
  float flt_scale, flt_delta, flt_lo, flt_hi;
  float flt_gain_rel;   // 2 x kRel: how far (relatively) a filter estimate of the gain may sit from the exact one
};
struct EvalConsts {
  float min_boost, max_boost, log2_min, log2_max;
  double enc_scale;
  uint32_t enc_byte_min, enc_byte_max;
  const float* lut;
  double log2_min_d, log2_max_d;
};
struct GenImage {  // 56 bytes; the V plane is u + c_stride * (height / 2) (gainmapmath.cpp:568)
  const uint8_t* y;
  const uint8_t* u;
  const uint16_t* hy;
  const uint16_t* huv;
  uint8_t* map;
  uint32_t y_stride, c_stride, hy_stride, huv_stride;
};
struct GenBatch {
  GenImage img[kMaxChunk];
};

// ---- apply -------------------------------------------------------------------------------------
// constants of the FAST scale-4 kernel (see k_apply_s4): wA[oy][pair][k] = (w_k(ox=2*pair), w_k(ox=2*pair+1)) * A
struct AppFast {
  float A, B;
  float wA[4][2][4][2];
};
struct AppConsts {
  uint32_t width, height, map_w, map_h, scale;
  float display_boost, inv_display_boost, max_boost, inv_max_boost;
  double log2_min_d, log2_max_d;  // log2((double)minContentBoost), log2((double)maxContentBoost)
  const float* idw;               // device: 4 tables (std, NR, NB, C) of scale*scale*4 floats
  const float* lut;               // device LUT buffer (LUT mode only)
  const float* srgb_line;         // device: kSrgbLineCells x (c0, c1), FAST scale-4 kernel
  float lut_boost_factor;         // GainLUT(metadata, displayBoost): displayBoost > 0 ? displayBoost / max : 1 (gainmapmath.h:162)
  AppFast fast;
};
struct AppImage {
  const uint8_t* y;
  const uint8_t* u;
  const uint8_t* v;
  const uint8_t* map;
  void* dst;
  uint32_t y_stride, c_stride;
};
struct AppBatch {
  AppImage img[kMaxChunk];
};

// ---- toneMap / convertYuv ----------------------------------------------------------------------
struct ToneImage {
  const uint16_t* sy;
  const uint16_t* suv;
  uint8_t* dy;
  uint8_t* du;
  uint8_t* dv;
  uint32_t sy_stride, suv_stride, dy_stride, dc_stride;
  uint32_t width, height;
};
struct CvtImage {
  uint8_t* y;   // where the converted planes go ...
  uint8_t* u;
  uint8_t* v;
  uint32_t y_stride, c_stride, width, height;
  float m[9];
  const uint8_t* sy;   // ... and where the samples come from: the same planes (JpegR::convertYuv works in place), or the caller's
  const uint8_t* su;   // image when the conversion is also the private copy encodeJPEGR API-1 takes first (jpegr.cpp:297-358)
  const uint8_t* sv;
  uint32_t sy_stride, sc_stride;
};

// ---- editorhelper effects (crop / mirror / rotate / resize): byte gathers over planes ---------------
enum FxOp : int { FX_COPY = 0, FX_FLIP_V, FX_FLIP_H, FX_ROT90, FX_ROT180, FX_ROT270, FX_RESIZE };
struct FxJob {            // dst[i][j] = src[f(i, j)] for i < rows, j < cols
  const uint8_t* src;
  uint8_t* dst;
  uint32_t rows, cols, dst_stride, src_stride;
  uint32_t in_w, in_h;    // extent of the source plane the index maps refer to
  uint32_t row_num, row_den, col_num, col_den;  // FX_RESIZE: src row = i*row_num/row_den, src col = j*col_num/col_den
  int op;
};
struct FxJobs {
  FxJob job[3];
  int n;
};

static_assert(sizeof(GenConsts) + sizeof(GenBatch) <= 4096, "generate kernel arguments exceed the kernarg segment");
static_assert(sizeof(AppConsts) + sizeof(AppBatch) <= 4096, "apply kernel arguments exceed the kernarg segment");

// launchers (enqueue only; return hipError_t of the launch)
hipError_t launch_generate(const GenConsts& c, const GenBatch& b, int n, int hdr_tf, bool aligned, bool lut,
                           bool filter, hipStream_t s);
hipError_t launch_stats_init(uint32_t* keys, int n, hipStream_t s);
hipError_t launch_stats_finalize(uint32_t* keys, int n, hipStream_t s);
// mode: 0 FAST, 1 EXACT, 2 LUT
hipError_t launch_apply(const AppConsts& c, const AppBatch& b, int n, int fmt, int mode,
                        bool fast_s4, hipStream_t s);
hipError_t launch_build_luts(float* lut /* kLutTotal floats */, hipStream_t s);
// GainLUT table (kGainLutN floats, device) for (log2 min, log2 max, boost factor)
hipError_t launch_build_gain_lut(float* table, double log2_min, double log2_max, float boost_factor, hipStream_t s);
hipError_t launch_tonemap(const ToneImage& t, bool aligned, hipStream_t s);
hipError_t launch_convert_yuv(const CvtImage& t, bool aligned, hipStream_t s);
// decoded 4:2:0 planes -> RGBA8888 with libjpeg-turbo's arithmetic (k_ycc420_rgba); w, h even, strides in bytes
hipError_t launch_ycc420_to_rgba(const uint8_t* y, const uint8_t* cb, const uint8_t* cr, uint32_t w, uint32_t h, uint32_t y_stride,
                                 uint32_t c_stride, uint8_t* rgba, hipStream_t s);
hipError_t upload_idw4(const float* tables /* 4*64 floats */);
hipError_t launch_effect(const FxJobs& j, hipStream_t s);
hipError_t launch_eval_transfer(int fn, const float* in, float* out, size_t n, const EvalConsts& ec, hipStream_t s);

}  // namespace uhdr
