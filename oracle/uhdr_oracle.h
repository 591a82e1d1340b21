/*
 * uhdr_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C99) of the reference's gain-map pixel hot path:
 *   lib/src/gainmapmath.cpp, lib/include/ultrahdr/gainmapmath.h,
 *   lib/src/ultrahdr.cpp:185-558 (generateGainMap / applyGainMap / toneMap),
 *   lib/src/jpegr.cpp:1132-1206 (convertYuv).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call this.  The product (libultrahdr_dev_amd) never does.
 *
 * Parity status: PINNED.  See oracle/README.md -- checked against
 *   (1) the md5 / checksum known answers the real reference produced
 *       (SURVEY.md section 8(c)/(d)), on the reference's own 1280x720 fixture
 *       and on the LCG synthetic frames;
 *   (2) oracle/_ref (the reference's own gainmapmath.cpp compiled in place),
 *       function by function, when /root/reference is present;
 *   (3) the reference's gainmapmath_test.cpp known-answer values.
 */
#ifndef UHDR_ORACLE_H
#define UHDR_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* enum values follow lib/include/ultrahdr/ultrahdr.h:36-120 */
enum {
  ORC_CG_UNSPECIFIED = -1, ORC_CG_BT709 = 0, ORC_CG_P3 = 1, ORC_CG_BT2100 = 2,
  ORC_TF_LINEAR = 0, ORC_TF_HLG = 1, ORC_TF_PQ = 2, ORC_TF_SRGB = 3,
  ORC_OUT_SDR = 0, ORC_OUT_HDR_LINEAR = 1, ORC_OUT_HDR_PQ = 2, ORC_OUT_HDR_HLG = 3,
  ORC_OUT_HDR_LINEAR_RGB_10BIT = 4,
  ORC_OK = 0, ORC_ERR_BAD_PTR = -10001, ORC_ERR_INVALID_COLORGAMUT = -10003,
  ORC_ERR_INVALID_TRANS_FUNC = -10005, ORC_ERR_RESOLUTION_MISMATCH = -10006,
  ORC_ERR_BAD_METADATA = -10010, ORC_ERR_UNSUPPORTED_MAP_SCALE_FACTOR = -20008
};

/* POD mirror of ultrahdr_uncompressed_struct (ultrahdr.h:152-181); strides in pixels */
typedef struct {
  void* data;
  size_t width, height;
  int32_t colorGamut;
  void* chroma_data;
  size_t luma_stride, chroma_stride;
  int32_t pixelFormat;
} orc_image;

/* POD mirror of ultrahdr_metadata_struct (ultrahdr.h:129-147) minus the version string */
typedef struct {
  float maxContentBoost, minContentBoost, gamma, offsetSdr, offsetHdr, hdrCapacityMin,
      hdrCapacityMax;
  int32_t version_ok; /* 1 == version string equals "1.0" */
} orc_metadata;

typedef struct { float r, g, b; } orc_color; /* also y,u,v */

/* ---- scalar / per-pixel pieces (gainmapmath.cpp) ---- */
float orc_srgbInvOetf(float e);
float orc_hlgOetf(float e);
float orc_hlgInvOetf(float e);
float orc_pqOetf(float e);
float orc_pqInvOetf(float e);
float orc_luminance(int gamut, orc_color e);
orc_color orc_yuvToRgb(int gamut, orc_color e);
orc_color orc_rgbToYuv(int gamut, orc_color e);
orc_color orc_gamutConv(int sdr_gamut, int hdr_gamut, orc_color e, int* is_null);
orc_color orc_yuvToYuv(int src, int dst, orc_color e);
uint8_t orc_encodeGain(float y_sdr, float y_hdr, float minBoost, float maxBoost,
                       float log2Min, float log2Max);
uint8_t orc_encodeGain3(float y_sdr, float y_hdr, float minBoost, float maxBoost);
orc_color orc_applyGain3(orc_color e, float gain, float minBoost, float maxBoost);
orc_color orc_applyGain4(orc_color e, float gain, float minBoost, float maxBoost, float displayBoost);
orc_color orc_getYuv420Pixel(const orc_image* img, size_t x, size_t y);
orc_color orc_getP010Pixel(const orc_image* img, size_t x, size_t y);
orc_color orc_sampleYuv420(const orc_image* img, size_t scale, size_t x, size_t y);
orc_color orc_sampleP010(const orc_image* img, size_t scale, size_t x, size_t y);
void orc_fillShepardsIDW(float* weights, int scale, int incR, int incB);
float orc_sampleMapIdw(const orc_image* map, size_t scale, size_t x, size_t y);
float orc_sampleMapFloat(const orc_image* map, float scale, size_t x, size_t y);
uint32_t orc_colorToRgba1010102(orc_color e);
uint64_t orc_colorToRgbaF16(orc_color e);
uint16_t orc_floatToHalf(float f);
void orc_transformYuv420(orc_image* img, size_t x_chroma, size_t y_chroma, int src, int dst);

/* ---- LUT variants (gainmapmath.cpp:21-64,162-171,269-354; GainLUT gainmapmath.h:149-182) ---- */
#define ORC_GAIN_LUT_N 1024u /* kGainFactorNumEntries, gainmapmath.h:149-150 */
float orc_srgbInvOetfLUT(float e);
float orc_hlgOetfLUT(float e);
float orc_hlgInvOetfLUT(float e);
float orc_pqOetfLUT(float e);
float orc_pqInvOetfLUT(float e);
/* the static tables themselves: which = 0 srgbInv(1024) 1 hlgInv(4096) 2 pqInv(4096) 4 hlg(65536) 5 pq(65536) */
const float* orc_lut_table(int which, size_t* n);
/* GainLUT(metadata) when with_display_boost == 0, GainLUT(metadata, displayBoost) otherwise */
void orc_gainLutBuild(float minBoost, float maxBoost, int with_display_boost, float displayBoost,
                      float* table /* ORC_GAIN_LUT_N */);
float orc_gainLutFactor(const float* table, float gain);
orc_color orc_applyGainLUT(orc_color e, float gain, const float* table);

/* ---- whole-image functions (ultrahdr.cpp / jpegr.cpp) ---- */
/* map_out: caller-allocated (w/4)*(h/4) bytes.  threads<=0 -> min(ncpu,4) like the reference */
int orc_generateGainMap(const orc_image* yuv420, const orc_image* p010, int hdr_tf,
                        orc_metadata* metadata, uint8_t* map_out, int sdr_is_601, int threads);
/* same, additionally reporting min/max of the UNCLAMPED gain (no reference counterpart; F5) */
int orc_generateGainMapStats(const orc_image* yuv420, const orc_image* p010, int hdr_tf,
                             orc_metadata* metadata, uint8_t* map_out, int sdr_is_601,
                             int threads, float* minmax_out);
int orc_applyGainMap(const orc_image* yuv420, const orc_image* gainmap,
                     const orc_metadata* metadata, int output_format, float max_display_boost,
                     orc_image* dest, int threads);
/* the two loops as a build with jpegr.cpp:33-38's USE_*_LUT macros visible to ultrahdr.cpp compiles them
 * (upstream libultrahdr's configuration; dead code in this fork) */
int orc_generateGainMapLUT(const orc_image* yuv420, const orc_image* p010, int hdr_tf,
                           orc_metadata* metadata, uint8_t* map_out, int sdr_is_601, int threads);
int orc_applyGainMapLUT(const orc_image* yuv420, const orc_image* gainmap,
                        const orc_metadata* metadata, int output_format, float max_display_boost,
                        orc_image* dest, int threads);
int orc_toneMap(const orc_image* src, orc_image* dest);
int orc_convertYuv(orc_image* image, int src_encoding, int dest_encoding);

/* editorhelper effects (lib/src/editorhelper.cpp); out->data is caller-allocated; mirror dir: 0 vertical, 1 horizontal */
int orc_crop(const orc_image* in, int left, int right, int top, int bottom, orc_image* out);
int orc_mirror(const orc_image* in, int dir, orc_image* out);
int orc_rotate(const orc_image* in, int clockwise_degree, orc_image* out);
int orc_resize(const orc_image* in, int out_width, int out_height, orc_image* out);
/* addEffects (editorhelper.cpp:362-446).  type 0 crop(a=left,b=right,c=top,d=bottom) 1 mirror(a=dir) 2 rotate(a=degrees) 3 resize(a=w,b=h) */
typedef struct { int32_t type, a, b, c, d; } orc_effect;
int orc_add_effects(const orc_image* in, const orc_effect* effects, int n, orc_image* out);

/* ---- JPEG compression of the path's outputs (jpeg_oracle.c; SURVEY 8(f) rank 1, encode side) ----
 * JpegEncoderHelper::compressImage (lib/src/jpegencoderhelper.cpp:39-283): uv == NULL compresses one 8-bit plane
 * (the gain map), otherwise YUV 4:2:0 planar with V at uv + cs*h/2.  Returns the size of the JPEG; at most cap bytes
 * are written. */
long orc_jpeg_encode(const uint8_t* y, const uint8_t* uv, int w, int h, int ls, int cs, int quality,
                     const void* icc, unsigned icc_n, uint8_t* out, long cap);
long orc_jpeg_header(int w, int h, int gray, int quality, const void* icc, unsigned icc_n, uint8_t* out, long cap);
long orc_jpeg_block_count(int w, int h, int gray);
/* quantised coefficients (natural order) of every block in entropy-coding order, dummy edge blocks included */
long orc_jpeg_coefficients(const uint8_t* y, const uint8_t* uv, int w, int h, int ls, int cs, int quality, int16_t* coef);
void orc_jpeg_quant_table(int quality, int chroma, uint16_t out[64]);
void orc_jpeg_fdct_quant(const uint8_t samples[64], const uint16_t quant[64], int16_t coef[64]);
/* JpegDecoderHelper::decompressImage(..., DECODE_TO_YCBCR) (lib/src/jpegdecoderhelper.cpp:188-516) for baseline files:
 * out = w*h luma, then (4:2:0) Cb and Cr of (w/2)*(h/2) at w*h and w*h + w*h/4.  Returns bytes written; -1 malformed,
 * -2 process / sampling outside the restatement, -3 cap too small (*w, *h, *gray are set). */
long orc_jpeg_decode(const uint8_t* jpg, long n, uint8_t* out, long cap, int* w, int* h, int* gray);
/* decoded 4:2:0 planes -> RGBA8888 the way libjpeg-turbo does it for DECODE_TO_RGBA (fancy upsampling + fixed-point colour conversion) */
int orc_ycc420_to_rgba(const uint8_t* y, const uint8_t* cb, const uint8_t* cr, int w, int h, uint8_t* rgba);
void orc_jpeg_idct(const int16_t coef_natural[64], const uint16_t quant_natural[64], uint8_t samples[64]);

/* fn: 0 srgbInvOetf 1 hlgInvOetf 2 pqInvOetf 3 encodeGain(y_sdr=1,y_hdr=x) 4 hlgOetf 5 pqOetf;
 * 40/41/42/44/45 the LUT accessors of fn 0/1/2/4/5; 46 GainLUT(min,max,displayBoost=max).getGainFactor */
void orc_eval_transfer(int fn, const float* in, float* out, size_t n, float minBoost, float maxBoost);

/* ---- helpers for tests / bench ---- */
/* LCG synthetic frame of SURVEY.md 8(d): p010 and yuv each w*h*3/2 elements */
void orc_fill_lcg(uint16_t* p010, uint8_t* yuv, size_t w, size_t h, uint32_t seed);
/* cs = cs*131 + v  over n bytes / n uint32 words */
uint64_t orc_checksum_u8(const uint8_t* p, size_t n);
uint64_t orc_checksum_u32(const uint32_t* p, size_t n);

#ifdef __cplusplus
}
#endif
#endif
