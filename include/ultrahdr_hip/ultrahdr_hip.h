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
  ERROR_ULTRAHDR_UNSUPPORTED_MAP_SCALE_FACTOR = -20008, ERROR_ULTRAHDR_INSUFFICIENT_RESOURCE = -20009
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

class JpegDecoderHelperHip {
 public:
  // DECODE_TO_YCBCR only (what applyGainMap's caller asks for, jpegr.cpp:780-801): 4:2:0 -> Y, Cb, Cr planes; grayscale -> Y
  bool decompressImage(const void* image, int length);
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

}  // namespace ultrahdr

#endif  // ULTRAHDR_HIP_SHIM_H
