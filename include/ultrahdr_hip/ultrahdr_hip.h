/*
 * ultrahdr_hip/ultrahdr_hip.h -- C++ host-side mirror of the reference's hot-path interface, implemented
 * on top of the C-ABI in include/uhdr_hip.h (MI355X kernels).
 *
 * The reference keeps the pixel path behind protected members of ultrahdr::UltraHdr
 * (lib/include/ultrahdr/ultrahdr.h:333-381) and private JpegR::convertYuv
 * (lib/include/ultrahdr/jpegr.h:329-330).  `ultrahdr::UltraHdrHip` exposes the same four operations
 * with the same names, argument meaning, ownership rules and status_t values, so that a caller written
 * like lib/src/jpegr.cpp (encodeJPEGR :200,:284,:421,:502; decodeJPEGR :801; :224,:360 for convertYuv)
 * compiles against it unchanged.  Type and enumerator names/values are the reference's.
 *
 * Ownership (same as the reference):
 *   generateGainMap  allocates dest->data with new uint8_t[]; the caller frees it with delete[]
 *                    (jpegr.cpp:207-208 wraps it in unique_ptr<uint8_t[]>);
 *   applyGainMap / toneMap / convertYuv: the caller owns every buffer.
 * All calls are synchronous on host memory (the image is staged through HBM and back).
 */
#ifndef ULTRAHDR_HIP_SHIM_H
#define ULTRAHDR_HIP_SHIM_H

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace ultrahdr {

// Enumerator names and numeric values are the reference's (lib/include/ultrahdr/ultrahdr.h:36-120), so that
// caller code compiles unchanged; only the codes the pixel path can return are listed for status_t.
enum ultrahdr_color_gamut : int {
  ULTRAHDR_COLORGAMUT_UNSPECIFIED = -1, ULTRAHDR_COLORGAMUT_BT709 = 0, ULTRAHDR_COLORGAMUT_P3 = 1,
  ULTRAHDR_COLORGAMUT_BT2100 = 2, ULTRAHDR_COLORGAMUT_MAX = 2
};
enum ultrahdr_transfer_function : int {
  ULTRAHDR_TF_UNSPECIFIED = -1, ULTRAHDR_TF_LINEAR = 0, ULTRAHDR_TF_HLG = 1, ULTRAHDR_TF_PQ = 2, ULTRAHDR_TF_SRGB = 3,
  ULTRAHDR_TF_MAX = 3
};
enum ultrahdr_output_format : int {
  ULTRAHDR_OUTPUT_UNSPECIFIED = -1, ULTRAHDR_OUTPUT_SDR = 0, ULTRAHDR_OUTPUT_HDR_LINEAR = 1, ULTRAHDR_OUTPUT_HDR_PQ = 2,
  ULTRAHDR_OUTPUT_HDR_HLG = 3, ULTRAHDR_OUTPUT_HDR_LINEAR_RGB_10BIT = 4, ULTRAHDR_OUTPUT_MAX = 4
};
enum ultrahdr_pixel_format : int {
  ULTRAHDR_PIX_FMT_UNSPECIFIED = -1, ULTRAHDR_PIX_FMT_P010 = 0, ULTRAHDR_PIX_FMT_YUV420 = 1,
  ULTRAHDR_PIX_FMT_MONOCHROME = 2, ULTRAHDR_PIX_FMT_RGBA8888 = 3, ULTRAHDR_PIX_FMT_RGBAF16 = 4,
  ULTRAHDR_PIX_FMT_RGBA1010102 = 5
};
enum status_t : int {
  ULTRAHDR_NO_ERROR = 0, ULTRAHDR_UNKNOWN_ERROR = -1, ERROR_ULTRAHDR_BAD_PTR = -10001,
  ERROR_ULTRAHDR_INVALID_COLORGAMUT = -10003, ERROR_ULTRAHDR_INVALID_TRANS_FUNC = -10005,
  ERROR_ULTRAHDR_RESOLUTION_MISMATCH = -10006, ERROR_ULTRAHDR_BAD_METADATA = -10010,
  ERROR_ULTRAHDR_INVALID_CROPPING_PARAMETERS = -10011, ERROR_ULTRAHDR_UNSUPPORTED_FEATURE = -30000,
  ERROR_ULTRAHDR_UNSUPPORTED_MAP_SCALE_FACTOR = -20008, ERROR_ULTRAHDR_INSUFFICIENT_RESOURCE = -20009,
  // the container path (JpegRHip below)
  ERROR_ULTRAHDR_UNSUPPORTED_WIDTH_HEIGHT = -10002, ERROR_ULTRAHDR_INVALID_STRIDE = -10004,
  ERROR_ULTRAHDR_INVALID_QUALITY_FACTOR = -10007, ERROR_ULTRAHDR_INVALID_DISPLAY_BOOST = -10008,
  ERROR_ULTRAHDR_INVALID_OUTPUT_FORMAT = -10009, ERROR_ULTRAHDR_ENCODE_ERROR = -20001, ERROR_ULTRAHDR_DECODE_ERROR = -20002,
  ERROR_ULTRAHDR_GAIN_MAP_IMAGE_NOT_FOUND = -20003, ERROR_ULTRAHDR_BUFFER_TOO_SMALL = -20004,
  ERROR_ULTRAHDR_METADATA_ERROR = -20005, ERROR_ULTRAHDR_NO_IMAGES_FOUND = -20006,
  ERROR_ULTRAHDR_MULTIPLE_EXIFS_RECEIVED = -20007
};

// ultrahdr_metadata_struct / ultrahdr_uncompressed_struct: same members, order and defaults as the reference
// (ultrahdr.h:129-147,152-181): linear boosts, strides in pixels, chroma_data == nullptr meaning "right after luma"
// is resolved by the CALLER (jpegr.cpp:265-278) before these functions are entered.
struct ultrahdr_metadata_struct {
  std::string version;
  float maxContentBoost, minContentBoost, gamma, offsetSdr, offsetHdr, hdrCapacityMin, hdrCapacityMax;
};
using ultrahdr_metadata_ptr = ultrahdr_metadata_struct*;

struct ultrahdr_uncompressed_struct {
  void* data;
  size_t width, height;
  ultrahdr_color_gamut colorGamut;
  void* chroma_data = nullptr;
  size_t luma_stride = 0, chroma_stride = 0;
  ultrahdr_pixel_format pixelFormat = ULTRAHDR_PIX_FMT_UNSPECIFIED;
};
using uhdr_uncompressed_ptr = ultrahdr_uncompressed_struct*;

static const char* const kGainMapVersion = "1.0";
static const size_t kMapDimensionScaleFactor = 4;

class UltraHdrHip {
 public:
  // device: HIP device ordinal this object runs on (uhdr_hip_init is called lazily, once)
  explicit UltraHdrHip(int device = 0);

  status_t generateGainMap(uhdr_uncompressed_ptr yuv420_image_ptr, uhdr_uncompressed_ptr p010_image_ptr,
                           ultrahdr_transfer_function hdr_tf, ultrahdr_metadata_ptr metadata,
                           uhdr_uncompressed_ptr dest, bool sdr_is_601 = false);

  status_t applyGainMap(uhdr_uncompressed_ptr yuv420_image_ptr, uhdr_uncompressed_ptr gainmap_image_ptr,
                        ultrahdr_metadata_ptr metadata, ultrahdr_output_format output_format,
                        float max_display_boost, uhdr_uncompressed_ptr dest);

  status_t toneMap(uhdr_uncompressed_ptr src, uhdr_uncompressed_ptr dest);

  status_t convertYuv(uhdr_uncompressed_ptr image, ultrahdr_color_gamut src_encoding,
                      ultrahdr_color_gamut dest_encoding);

  // UHDR_HIP_APPLY_FAST (default), UHDR_HIP_APPLY_EXACT or UHDR_HIP_APPLY_LUT, see include/uhdr_hip.h
  void setApplyMode(int mode) { mApplyMode = mode; }
  // UHDR_HIP_GENERATE_EXACT (default), UHDR_HIP_GENERATE_LUT or UHDR_HIP_GENERATE_UNFILTERED
  void setGenerateMode(int mode) { mGenerateMode = mode; }

 private:
  status_t ensureInit();
  int mDevice;
  bool mReady = false;
  int mApplyMode = 0;
  int mGenerateMode = 0;
};

// ---- editing effects: the free functions of lib/include/ultrahdr/editorhelper.h:49-63, same signatures ----------
enum ultrahdr_mirroring_direction : int { ULTRAHDR_MIRROR_VERTICAL = 0, ULTRAHDR_MIRROR_HORIZONTAL = 1 };

// host images in, host images out (out_img->data caller-allocated), executed on HIP device 0
status_t crop(uhdr_uncompressed_ptr const in_img, int left, int right, int top, int bottom, uhdr_uncompressed_ptr out_img);
status_t mirror(uhdr_uncompressed_ptr const in_img, ultrahdr_mirroring_direction mirror_dir, uhdr_uncompressed_ptr out_img);
status_t rotate(uhdr_uncompressed_ptr const in_img, int clockwise_degree, uhdr_uncompressed_ptr out_img);
status_t resize(uhdr_uncompressed_ptr const in_img, int out_width, int out_height, uhdr_uncompressed_ptr out_img);

// addEffects (lib/include/ultrahdr/editorhelper.h:27-47,62-63): the effect structs and the chaining call, same names
struct ultrahdr_effect {
  virtual ~ultrahdr_effect() = default;
};
struct ultrahdr_crop_effect : ultrahdr_effect { int left, right, top, bottom; };
struct ultrahdr_mirror_effect : ultrahdr_effect { ultrahdr_mirroring_direction mirror_dir; };
struct ultrahdr_rotate_effect : ultrahdr_effect { int clockwise_degree; };
struct ultrahdr_resize_effect : ultrahdr_effect { int new_width, new_height; };
status_t addEffects(uhdr_uncompressed_ptr const in_img, std::vector<ultrahdr_effect*>& effects, uhdr_uncompressed_ptr out_image);

// ---- JPEG helpers: the members callers use of JpegEncoderHelper (lib/include/ultrahdr/jpegencoderhelper.h:43-60) and of
// JpegDecoderHelper (lib/include/ultrahdr/jpegdecoderhelper.h:54-100), same names and meaning; host buffers in and out, the
// codec itself runs on HIP device 0 (uhdr_hip_jpeg_encode / uhdr_hip_jpeg_decode).
class JpegEncoderHelperHip {
 public:
  // uvBuffer == nullptr compresses the single plane yBuffer (the gain map, jpegr.cpp:294-297)
  bool compressImage(const uint8_t* yBuffer, const uint8_t* uvBuffer, int width, int height, int lumaStride, int chromaStride,
                     int quality, const void* iccBuffer, unsigned int iccSize);
  void* getCompressedImagePtr() { return mResultBuffer.data(); }
  size_t getCompressedImageSize() { return mResultBuffer.size(); }

 private:
  std::vector<uint8_t> mResultBuffer;
};

// jpegdecoderhelper.h:34-38 (PARSE_ONLY is served by getCompressedImageParameters there and by JpegRHip::getJPEGRInfo here)
enum decode_mode_t : int { DECODE_TO_RGBA = 1, DECODE_TO_YCBCR = 2 };

class JpegDecoderHelperHip {
 public:
  // DECODE_TO_YCBCR (what applyGainMap's caller asks for, jpegr.cpp:780-801): 4:2:0 -> Y, Cb, Cr planes; grayscale -> Y.
  // DECODE_TO_RGBA (the SDR rendition, jpegr.cpp:686-690): 4:2:0 -> RGBA8888 with libjpeg-turbo's arithmetic.
  bool decompressImage(const void* image, int length, decode_mode_t decodeTo = DECODE_TO_YCBCR);
  void* getDecompressedImagePtr() { return mResultBuffer.data(); }
  size_t getDecompressedImageSize() { return mResultBuffer.size(); }
  size_t getDecompressedImageWidth() { return mWidth; }
  size_t getDecompressedImageHeight() { return mHeight; }
  bool isSingleChannel() { return mSingleChannel; }

 private:
  std::vector<uint8_t> mResultBuffer;
  size_t mWidth = 0, mHeight = 0;
  bool mSingleChannel = false;
};

// ---- the JPEG/R codec: the public members of ultrahdr::JpegR (lib/include/ultrahdr/jpegr.h:59-265), same names, argument meaning,
// defaults and status values; structs as in ultrahdr.h:186-207 and jpegr.h:37-57.  Host buffers in and out; toneMap, gain-map
// generation / application and both JPEG codecs run on HIP device 0 (uhdr_hip_jpegr_*), the container is parsed / assembled on
// the host.  decodeJPEGR(ULTRAHDR_OUTPUT_SDR) returns the primary image as RGBA8888 with libjpeg-turbo's DECODE_TO_RGBA arithmetic.
struct ultrahdr_compressed_struct {
  void* data;
  int length;
  int maxLength;
  ultrahdr_color_gamut colorGamut;
};
using uhdr_compressed_ptr = ultrahdr_compressed_struct*;
struct ultrahdr_exif_struct {
  void* data;
  size_t length;
};
using uhdr_exif_ptr = ultrahdr_exif_struct*;
struct jpeg_info_struct {
  std::vector<uint8_t> imgData, iccData, exifData, xmpData;
  size_t width, height;
};
struct jpegr_info_struct {
  size_t width, height;
  jpeg_info_struct* primaryImgInfo = nullptr;
  jpeg_info_struct* gainmapImgInfo = nullptr;
};
using j_info_ptr = jpeg_info_struct*;
using uhdr_info_ptr = jpegr_info_struct*;

class JpegRHip {
 public:
  // API-0 (jpegr.h:81) and API-1 (:103)
  status_t encodeJPEGR(uhdr_uncompressed_ptr p010_image_ptr, ultrahdr_transfer_function hdr_tf, uhdr_compressed_ptr dest, int quality,
                       uhdr_exif_ptr exif);
  status_t encodeJPEGR(uhdr_uncompressed_ptr p010_image_ptr, uhdr_uncompressed_ptr yuv420_image_ptr, ultrahdr_transfer_function hdr_tf,
                       uhdr_compressed_ptr dest, int quality, uhdr_exif_ptr exif);
  // API-2 (:127), API-3 (:150), API-4 (:168)
  status_t encodeJPEGR(uhdr_uncompressed_ptr p010_image_ptr, uhdr_uncompressed_ptr yuv420_image_ptr, uhdr_compressed_ptr yuv420jpg_image_ptr,
                       ultrahdr_transfer_function hdr_tf, uhdr_compressed_ptr dest);
  status_t encodeJPEGR(uhdr_uncompressed_ptr p010_image_ptr, uhdr_compressed_ptr yuv420jpg_image_ptr, ultrahdr_transfer_function hdr_tf,
                       uhdr_compressed_ptr dest);
  status_t encodeJPEGR(uhdr_compressed_ptr yuv420jpg_image_ptr, uhdr_compressed_ptr gainmapjpg_image_ptr, ultrahdr_metadata_ptr metadata,
                       uhdr_compressed_ptr dest);
  // "API-x" (:263): SDR planes + ready gain map + metadata
  status_t encodeJPEGR(uhdr_uncompressed_ptr yuv420_image_ptr, uhdr_uncompressed_ptr gainmap_image_ptr, ultrahdr_metadata_ptr metadata,
                       uhdr_compressed_ptr dest, int quality, uhdr_exif_ptr exif);
  // (:213) dest->data, exif->data and gainmap_image_ptr->data are caller-allocated, as in the reference
  status_t decodeJPEGR(uhdr_compressed_ptr jpegr_image_ptr, uhdr_uncompressed_ptr dest, float max_display_boost = 3.4028234663852886e38f,
                       uhdr_exif_ptr exif = nullptr, ultrahdr_output_format output_format = ULTRAHDR_OUTPUT_HDR_LINEAR,
                       uhdr_uncompressed_ptr gainmap_image_ptr = nullptr, ultrahdr_metadata_ptr metadata = nullptr);
  // (:231)
  status_t getJPEGRInfo(uhdr_compressed_ptr jpegr_image_ptr, uhdr_info_ptr jpeg_image_info_ptr);

  void setApplyMode(int mode) { mApplyMode = mode; }   // UHDR_HIP_APPLY_EXACT (default here: decoded files equal the reference's), FAST, LUT

 private:
  int mApplyMode = 1;
};

}  // namespace ultrahdr

#endif  // ULTRAHDR_HIP_SHIM_H
