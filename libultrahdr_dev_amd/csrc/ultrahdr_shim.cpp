// ultrahdr_shim.cpp -- ultrahdr::UltraHdrHip: the reference's C++ member signatures on top of the
// C-ABI (include/uhdr_hip.h).  Pure host glue: descriptor translation + the new[] ownership contract
// of generateGainMap (ref lib/src/ultrahdr.cpp:209,217-218,356).
#include "ultrahdr_hip/ultrahdr_hip.h"

#include <cstring>
#include <memory>
#include <vector>

#include "uhdr_hip.h"

namespace ultrahdr {
namespace {

uhdr_hip_image_t to_c(const ultrahdr_uncompressed_struct& s) {
  uhdr_hip_image_t c;
  c.data = s.data;
  c.width = s.width;
  c.height = s.height;
  c.colorGamut = static_cast<int32_t>(s.colorGamut);
  c.chroma_data = s.chroma_data;
  c.luma_stride = s.luma_stride;
  c.chroma_stride = s.chroma_stride;
  c.pixelFormat = static_cast<int32_t>(s.pixelFormat);
  return c;
}
void from_c(const uhdr_hip_image_t& c, ultrahdr_uncompressed_struct* s) {
  s->data = c.data;
  s->width = c.width;
  s->height = c.height;
  s->colorGamut = static_cast<ultrahdr_color_gamut>(c.colorGamut);
  s->chroma_data = c.chroma_data;
  s->luma_stride = c.luma_stride;
  s->chroma_stride = c.chroma_stride;
  s->pixelFormat = static_cast<ultrahdr_pixel_format>(c.pixelFormat);
}
uhdr_hip_metadata_t to_c(const ultrahdr_metadata_struct& m) {
  uhdr_hip_metadata_t c;
  std::memset(c.version, 0, sizeof(c.version));
  // a longer-than-7-character version can never equal "1.0"; keep it unequal after truncation
  std::strncpy(c.version, m.version.size() < sizeof(c.version) ? m.version.c_str() : "toolong", sizeof(c.version) - 1);
  c.maxContentBoost = m.maxContentBoost;
  c.minContentBoost = m.minContentBoost;
  c.gamma = m.gamma;
  c.offsetSdr = m.offsetSdr;
  c.offsetHdr = m.offsetHdr;
  c.hdrCapacityMin = m.hdrCapacityMin;
  c.hdrCapacityMax = m.hdrCapacityMax;
  return c;
}

}  // namespace

UltraHdrHip::UltraHdrHip(int device) : mDevice(device) {}

status_t UltraHdrHip::ensureInit() {
  if (!mReady) {
    const int rc = uhdr_hip_init(mDevice);
    if (rc != UHDR_HIP_NO_ERROR) return static_cast<status_t>(rc);
    mReady = true;
  }
  return ULTRAHDR_NO_ERROR;
}

status_t UltraHdrHip::generateGainMap(uhdr_uncompressed_ptr yuv420_image_ptr, uhdr_uncompressed_ptr p010_image_ptr,
                                      ultrahdr_transfer_function hdr_tf, ultrahdr_metadata_ptr metadata,
                                      uhdr_uncompressed_ptr dest, bool sdr_is_601) {
  // pointer checks first, exactly as ultrahdr.cpp:189-194, so a null argument never reaches the device
  if (yuv420_image_ptr == nullptr || p010_image_ptr == nullptr || metadata == nullptr || dest == nullptr ||
      yuv420_image_ptr->data == nullptr || yuv420_image_ptr->chroma_data == nullptr ||
      p010_image_ptr->data == nullptr || p010_image_ptr->chroma_data == nullptr)
    return ERROR_ULTRAHDR_BAD_PTR;
  if (yuv420_image_ptr->width != p010_image_ptr->width || yuv420_image_ptr->height != p010_image_ptr->height)
    return ERROR_ULTRAHDR_RESOLUTION_MISMATCH;
  if (yuv420_image_ptr->colorGamut == ULTRAHDR_COLORGAMUT_UNSPECIFIED ||
      p010_image_ptr->colorGamut == ULTRAHDR_COLORGAMUT_UNSPECIFIED)
    return ERROR_ULTRAHDR_INVALID_COLORGAMUT;
  status_t st = ensureInit();
  if (st != ULTRAHDR_NO_ERROR) return st;

  const size_t map_w = yuv420_image_ptr->width / kMapDimensionScaleFactor;
  const size_t map_h = yuv420_image_ptr->height / kMapDimensionScaleFactor;
  std::unique_ptr<uint8_t[]> map_data(new uint8_t[map_w * map_h > 0 ? map_w * map_h : 1]);  // :209,217-218

  uhdr_hip_image_t y = to_c(*yuv420_image_ptr), p = to_c(*p010_image_ptr), d = to_c(*dest);
  d.data = map_data.get();
  uhdr_hip_metadata_t md;
  std::memset(&md, 0, sizeof(md));
  const int rc = uhdr_hip_generate_gainmap_ex(&y, &p, static_cast<int>(hdr_tf), &md, &d, sdr_is_601 ? 1 : 0,
                                              mGenerateMode, UHDR_HIP_MEM_HOST, nullptr);
  if (rc != UHDR_HIP_NO_ERROR) return static_cast<status_t>(rc);
  metadata->version = md.version;
  metadata->maxContentBoost = md.maxContentBoost;
  metadata->minContentBoost = md.minContentBoost;
  metadata->gamma = md.gamma;
  metadata->offsetSdr = md.offsetSdr;
  metadata->offsetHdr = md.offsetHdr;
  metadata->hdrCapacityMin = md.hdrCapacityMin;
  metadata->hdrCapacityMax = md.hdrCapacityMax;
  from_c(d, dest);
  dest->data = map_data.release();  // :356 -- ownership passes to the caller (delete[])
  return ULTRAHDR_NO_ERROR;
}

status_t UltraHdrHip::applyGainMap(uhdr_uncompressed_ptr yuv420_image_ptr, uhdr_uncompressed_ptr gainmap_image_ptr,
                                   ultrahdr_metadata_ptr metadata, ultrahdr_output_format output_format,
                                   float max_display_boost, uhdr_uncompressed_ptr dest) {
  if (yuv420_image_ptr == nullptr || gainmap_image_ptr == nullptr || metadata == nullptr || dest == nullptr ||
      yuv420_image_ptr->data == nullptr || yuv420_image_ptr->chroma_data == nullptr ||
      gainmap_image_ptr->data == nullptr)
    return ERROR_ULTRAHDR_BAD_PTR;  // ultrahdr.cpp:364-368
  uhdr_hip_image_t y = to_c(*yuv420_image_ptr), g = to_c(*gainmap_image_ptr), d = to_c(*dest);
  const uhdr_hip_metadata_t md = to_c(*metadata);
  // metadata / scale checks do not need the device; let the C-ABI produce the reference's codes first
  int rc = uhdr_hip_apply_gainmap(&y, &g, &md, static_cast<int>(output_format), max_display_boost, &d, mApplyMode,
                                  UHDR_HIP_MEM_HOST, nullptr);
  if (rc == UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE && !mReady) {
    const status_t st = ensureInit();
    if (st != ULTRAHDR_NO_ERROR) return st;
    rc = uhdr_hip_apply_gainmap(&y, &g, &md, static_cast<int>(output_format), max_display_boost, &d, mApplyMode,
                                UHDR_HIP_MEM_HOST, nullptr);
  }
  if (rc != UHDR_HIP_NO_ERROR) return static_cast<status_t>(rc);
  dest->width = d.width;            // :411-413
  dest->height = d.height;
  dest->colorGamut = static_cast<ultrahdr_color_gamut>(d.colorGamut);
  return ULTRAHDR_NO_ERROR;
}

status_t UltraHdrHip::toneMap(uhdr_uncompressed_ptr src, uhdr_uncompressed_ptr dest) {
  if (src == nullptr || dest == nullptr) return ERROR_ULTRAHDR_BAD_PTR;
  if (src->width != dest->width || src->height != dest->height) return ERROR_ULTRAHDR_RESOLUTION_MISMATCH;
  const status_t st = ensureInit();
  if (st != ULTRAHDR_NO_ERROR) return st;
  uhdr_hip_image_t s = to_c(*src), d = to_c(*dest);
  const int rc = uhdr_hip_tonemap(&s, &d, UHDR_HIP_MEM_HOST, nullptr);
  if (rc != UHDR_HIP_NO_ERROR) return static_cast<status_t>(rc);
  dest->colorGamut = static_cast<ultrahdr_color_gamut>(d.colorGamut);  // :556
  return ULTRAHDR_NO_ERROR;
}

status_t UltraHdrHip::convertYuv(uhdr_uncompressed_ptr image, ultrahdr_color_gamut src_encoding,
                                 ultrahdr_color_gamut dest_encoding) {
  if (image == nullptr) return ERROR_ULTRAHDR_BAD_PTR;  // jpegr.cpp:1134-1140
  if (src_encoding == ULTRAHDR_COLORGAMUT_UNSPECIFIED || dest_encoding == ULTRAHDR_COLORGAMUT_UNSPECIFIED)
    return ERROR_ULTRAHDR_INVALID_COLORGAMUT;
  if (src_encoding == dest_encoding) return ULTRAHDR_NO_ERROR;
  const status_t st = ensureInit();
  if (st != ULTRAHDR_NO_ERROR) return st;
  uhdr_hip_image_t i = to_c(*image);
  return static_cast<status_t>(uhdr_hip_convert_yuv(&i, static_cast<int>(src_encoding), static_cast<int>(dest_encoding),
                                                    UHDR_HIP_MEM_HOST, nullptr));
}

// ---- editing effects (lib/src/editorhelper.cpp) ---------------------------------------------------------------
namespace {
template <typename Fn>
status_t run_effect(uhdr_uncompressed_ptr const in_img, uhdr_uncompressed_ptr out_img, Fn&& call) {
  if (in_img == nullptr || out_img == nullptr) return ERROR_ULTRAHDR_BAD_PTR;
  static const int init_rc = uhdr_hip_init(0);
  uhdr_hip_image_t i = to_c(*in_img), o = to_c(*out_img);
  int rc = call(&i, &o);
  if (rc == UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE && init_rc != UHDR_HIP_NO_ERROR) return static_cast<status_t>(init_rc);
  if (rc != UHDR_HIP_NO_ERROR) return static_cast<status_t>(rc);
  from_c(o, out_img);
  return ULTRAHDR_NO_ERROR;
}
}  // namespace

status_t crop(uhdr_uncompressed_ptr const in_img, int left, int right, int top, int bottom, uhdr_uncompressed_ptr out_img) {
  return run_effect(in_img, out_img, [&](uhdr_hip_image_t* i, uhdr_hip_image_t* o) {
    return uhdr_hip_crop(i, left, right, top, bottom, o, UHDR_HIP_MEM_HOST, nullptr);
  });
}
status_t mirror(uhdr_uncompressed_ptr const in_img, ultrahdr_mirroring_direction mirror_dir, uhdr_uncompressed_ptr out_img) {
  return run_effect(in_img, out_img, [&](uhdr_hip_image_t* i, uhdr_hip_image_t* o) {
    return uhdr_hip_mirror(i, static_cast<int>(mirror_dir), o, UHDR_HIP_MEM_HOST, nullptr);
  });
}
status_t rotate(uhdr_uncompressed_ptr const in_img, int clockwise_degree, uhdr_uncompressed_ptr out_img) {
  return run_effect(in_img, out_img, [&](uhdr_hip_image_t* i, uhdr_hip_image_t* o) {
    return uhdr_hip_rotate(i, clockwise_degree, o, UHDR_HIP_MEM_HOST, nullptr);
  });
}
status_t resize(uhdr_uncompressed_ptr const in_img, int out_width, int out_height, uhdr_uncompressed_ptr out_img) {
  return run_effect(in_img, out_img, [&](uhdr_hip_image_t* i, uhdr_hip_image_t* o) {
    return uhdr_hip_resize(i, out_width, out_height, o, UHDR_HIP_MEM_HOST, nullptr);
  });
}

status_t addEffects(uhdr_uncompressed_ptr const in_img, std::vector<ultrahdr_effect*>& effects, uhdr_uncompressed_ptr out_image) {
  if (in_img == nullptr || in_img->data == nullptr || out_image == nullptr || out_image->data == nullptr) return ERROR_ULTRAHDR_BAD_PTR;
  if (uhdr_hip_init(0) != UHDR_HIP_NO_ERROR) return ERROR_ULTRAHDR_INSUFFICIENT_RESOURCE;
  std::vector<uhdr_hip_effect_t> fx;
  for (ultrahdr_effect* e : effects) {   // the reference's dynamic_cast ladder (editorhelper.cpp:394-430); unknown effects are skipped there too
    if (auto* c = dynamic_cast<ultrahdr_crop_effect*>(e)) fx.push_back({0, c->left, c->right, c->top, c->bottom});
    else if (auto* m = dynamic_cast<ultrahdr_mirror_effect*>(e)) fx.push_back({1, (int)m->mirror_dir, 0, 0, 0});
    else if (auto* r = dynamic_cast<ultrahdr_rotate_effect*>(e)) fx.push_back({2, r->clockwise_degree, 0, 0, 0});
    else if (auto* z = dynamic_cast<ultrahdr_resize_effect*>(e)) fx.push_back({3, z->new_width, z->new_height, 0, 0});
  }
  uhdr_hip_image_t i = {in_img->data, in_img->width, in_img->height, (int32_t)in_img->colorGamut, in_img->chroma_data,
                        in_img->luma_stride, in_img->chroma_stride, (int32_t)in_img->pixelFormat};
  uhdr_hip_image_t o = {out_image->data, out_image->width, out_image->height, (int32_t)out_image->colorGamut, out_image->chroma_data,
                        out_image->luma_stride, out_image->chroma_stride, (int32_t)out_image->pixelFormat};
  const int rc = uhdr_hip_add_effects(&i, fx.data(), (int)fx.size(), &o, UHDR_HIP_MEM_HOST, nullptr);
  if (rc != UHDR_HIP_NO_ERROR) return static_cast<status_t>(rc);
  out_image->width = o.width; out_image->height = o.height;
  out_image->colorGamut = static_cast<ultrahdr_color_gamut>(o.colorGamut);
  out_image->pixelFormat = static_cast<ultrahdr_pixel_format>(o.pixelFormat);
  out_image->luma_stride = o.luma_stride; out_image->chroma_stride = o.chroma_stride;
  out_image->chroma_data = o.chroma_data;
  return ULTRAHDR_NO_ERROR;
}

// ---- JPEG helpers ----------------------------------------------------------------------------------------------------
bool JpegEncoderHelperHip::compressImage(const uint8_t* yBuffer, const uint8_t* uvBuffer, int width, int height, int lumaStride,
                                         int chromaStride, int quality, const void* iccBuffer, unsigned int iccSize) {
  mResultBuffer.clear();   // jpegencoderhelper.cpp:42
  if (width <= 0 || height <= 0 || uhdr_hip_init(0) != UHDR_HIP_NO_ERROR) return false;
  uhdr_hip_image_t img = {const_cast<uint8_t*>(yBuffer), (size_t)width, (size_t)height, UHDR_HIP_CG_UNSPECIFIED,
                          const_cast<uint8_t*>(uvBuffer), (size_t)lumaStride, (size_t)chromaStride,
                          uvBuffer ? UHDR_HIP_PIX_FMT_YUV420 : UHDR_HIP_PIX_FMT_MONOCHROME};
  mResultBuffer.resize((size_t)width * height + 65536);
  size_t n = 0;
  int rc = uhdr_hip_jpeg_encode(&img, quality, iccBuffer, iccSize, mResultBuffer.data(), mResultBuffer.size(), &n, UHDR_HIP_MEM_HOST, nullptr);
  if (rc == UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE) {   // n holds the size needed
    mResultBuffer.resize(n);
    rc = uhdr_hip_jpeg_encode(&img, quality, iccBuffer, iccSize, mResultBuffer.data(), n, &n, UHDR_HIP_MEM_HOST, nullptr);
  }
  mResultBuffer.resize(rc == UHDR_HIP_NO_ERROR ? n : 0);
  return rc == UHDR_HIP_NO_ERROR;
}

bool JpegDecoderHelperHip::decompressImage(const void* image, int length, decode_mode_t decodeTo) {
  mResultBuffer.clear();
  mWidth = mHeight = 0;
  if (image == nullptr || length <= 0 || uhdr_hip_init(0) != UHDR_HIP_NO_ERROR) return false;
  uhdr_hip_image_t desc = {};
  if (decodeTo == DECODE_TO_RGBA) {
    if (uhdr_hip_jpeg_decode_rgba(image, (size_t)length, nullptr, 0, &desc, UHDR_HIP_MEM_HOST, nullptr) != UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE) return false;
    if (desc.width == 0 || desc.height == 0) return false;   // (a probe answer always carries the size)
    mResultBuffer.resize(desc.width * desc.height * 4);   // jpegdecoderhelper.cpp:274
    if (uhdr_hip_jpeg_decode_rgba(image, (size_t)length, mResultBuffer.data(), mResultBuffer.size(), &desc, UHDR_HIP_MEM_HOST, nullptr) != UHDR_HIP_NO_ERROR) {
      mResultBuffer.clear();
      return false;
    }
    mWidth = desc.width; mHeight = desc.height; mSingleChannel = false;
    return true;
  }
  int rc = uhdr_hip_jpeg_decode(image, (size_t)length, nullptr, 0, &desc, UHDR_HIP_MEM_HOST, nullptr);
  if (rc != UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE || desc.width == 0 || desc.height == 0) return false;   // unreadable header, unsupported process, too large, out of memory
  mSingleChannel = desc.pixelFormat == UHDR_HIP_PIX_FMT_MONOCHROME;
  const size_t luma = desc.width * desc.height;
  mResultBuffer.resize(mSingleChannel ? luma : luma + 2 * (luma / 4));   // jpegdecoderhelper.cpp:264,266
  rc = uhdr_hip_jpeg_decode(image, (size_t)length, mResultBuffer.data(), mResultBuffer.size(), &desc, UHDR_HIP_MEM_HOST, nullptr);
  if (rc != UHDR_HIP_NO_ERROR) { mResultBuffer.clear(); return false; }
  mWidth = desc.width; mHeight = desc.height;
  return true;
}

// ---- JpegRHip: ultrahdr::JpegR's public members over uhdr_hip_jpegr_* ---------------------------------------------------------
namespace {
// Write()'s contract (jpegr.cpp:46-61): the file goes to dest->data, capacity dest->maxLength, dest->length receives the size
template <class Fn>
status_t encode_into(uhdr_compressed_ptr dest, Fn&& call) {
  void* out = dest != nullptr ? dest->data : nullptr;
  const size_t cap = dest != nullptr && dest->maxLength > 0 ? (size_t)dest->maxLength : 0;
  size_t n = 0;
  if (uhdr_hip_init(0) != UHDR_HIP_NO_ERROR) return ULTRAHDR_UNKNOWN_ERROR;
  const int rc = call(out, cap, &n);
  if (rc == UHDR_HIP_NO_ERROR) dest->length = (int)n;
  return static_cast<status_t>(rc);
}
const void* exif_ptr(uhdr_exif_ptr e) { return e ? e->data : nullptr; }
// "exif != nullptr && exif->data == nullptr -> BAD_PTR" (jpegr.cpp:190-193) travels as (NULL, nonzero size)
size_t exif_len(uhdr_exif_ptr e) { return e ? (e->data ? e->length : 1) : 0; }
}  // namespace

status_t JpegRHip::encodeJPEGR(uhdr_uncompressed_ptr p010_image_ptr, ultrahdr_transfer_function hdr_tf, uhdr_compressed_ptr dest, int quality,
                               uhdr_exif_ptr exif) {
  uhdr_hip_image_t p;
  if (p010_image_ptr) p = to_c(*p010_image_ptr);
  return encode_into(dest, [&](void* out, size_t cap, size_t* n) {
    return uhdr_hip_jpegr_encode_api0(p010_image_ptr ? &p : nullptr, (int)hdr_tf, quality, exif_ptr(exif), exif_len(exif), out, cap, n, UHDR_HIP_MEM_HOST, nullptr);
  });
}

status_t JpegRHip::encodeJPEGR(uhdr_uncompressed_ptr p010_image_ptr, uhdr_uncompressed_ptr yuv420_image_ptr, ultrahdr_transfer_function hdr_tf,
                               uhdr_compressed_ptr dest, int quality, uhdr_exif_ptr exif) {
  uhdr_hip_image_t p, y;
  if (p010_image_ptr) p = to_c(*p010_image_ptr);
  if (yuv420_image_ptr) y = to_c(*yuv420_image_ptr);
  return encode_into(dest, [&](void* out, size_t cap, size_t* n) {
    return uhdr_hip_jpegr_encode_api1(p010_image_ptr ? &p : nullptr, yuv420_image_ptr ? &y : nullptr, (int)hdr_tf, quality, exif_ptr(exif), exif_len(exif), out,
                                      cap, n, UHDR_HIP_MEM_HOST, nullptr);
  });
}

status_t JpegRHip::encodeJPEGR(uhdr_uncompressed_ptr p010_image_ptr, uhdr_uncompressed_ptr yuv420_image_ptr, uhdr_compressed_ptr yuv420jpg_image_ptr,
                               ultrahdr_transfer_function hdr_tf, uhdr_compressed_ptr dest) {
  uhdr_hip_image_t p, y;
  if (p010_image_ptr) p = to_c(*p010_image_ptr);
  if (yuv420_image_ptr) y = to_c(*yuv420_image_ptr);
  return encode_into(dest, [&](void* out, size_t cap, size_t* n) {
    return uhdr_hip_jpegr_encode_api2(p010_image_ptr ? &p : nullptr, yuv420_image_ptr ? &y : nullptr, yuv420jpg_image_ptr ? yuv420jpg_image_ptr->data : nullptr,
                                      yuv420jpg_image_ptr ? (size_t)yuv420jpg_image_ptr->length : 0,
                                      yuv420jpg_image_ptr ? (int)yuv420jpg_image_ptr->colorGamut : UHDR_HIP_CG_UNSPECIFIED, (int)hdr_tf, out, cap, n,
                                      UHDR_HIP_MEM_HOST, nullptr);
  });
}

status_t JpegRHip::encodeJPEGR(uhdr_uncompressed_ptr p010_image_ptr, uhdr_compressed_ptr yuv420jpg_image_ptr, ultrahdr_transfer_function hdr_tf,
                               uhdr_compressed_ptr dest) {
  uhdr_hip_image_t p;
  if (p010_image_ptr) p = to_c(*p010_image_ptr);
  return encode_into(dest, [&](void* out, size_t cap, size_t* n) {
    return uhdr_hip_jpegr_encode_api3(p010_image_ptr ? &p : nullptr, yuv420jpg_image_ptr ? yuv420jpg_image_ptr->data : nullptr,
                                      yuv420jpg_image_ptr ? (size_t)yuv420jpg_image_ptr->length : 0,
                                      yuv420jpg_image_ptr ? (int)yuv420jpg_image_ptr->colorGamut : UHDR_HIP_CG_UNSPECIFIED, (int)hdr_tf, out, cap, n,
                                      UHDR_HIP_MEM_HOST, nullptr);
  });
}

status_t JpegRHip::encodeJPEGR(uhdr_compressed_ptr yuv420jpg_image_ptr, uhdr_compressed_ptr gainmapjpg_image_ptr, ultrahdr_metadata_ptr metadata,
                               uhdr_compressed_ptr dest) {
  if (yuv420jpg_image_ptr == nullptr || gainmapjpg_image_ptr == nullptr) return ERROR_ULTRAHDR_BAD_PTR;   // jpegr.cpp:505-512
  uhdr_hip_metadata_t md;
  if (metadata) md = to_c(*metadata);
  void* out = dest ? dest->data : nullptr;
  size_t n = 0;
  const int rc = uhdr_hip_jpegr_encode_api4(yuv420jpg_image_ptr->data, (size_t)yuv420jpg_image_ptr->length, (int)yuv420jpg_image_ptr->colorGamut,
                                            gainmapjpg_image_ptr->data, (size_t)gainmapjpg_image_ptr->length, metadata ? &md : nullptr, out,
                                            dest && dest->maxLength > 0 ? (size_t)dest->maxLength : 0, &n);
  if (rc == UHDR_HIP_NO_ERROR) dest->length = (int)n;
  return static_cast<status_t>(rc);
}

status_t JpegRHip::encodeJPEGR(uhdr_uncompressed_ptr yuv420_image_ptr, uhdr_uncompressed_ptr gainmap_image_ptr, ultrahdr_metadata_ptr metadata,
                               uhdr_compressed_ptr dest, int quality, uhdr_exif_ptr exif) {
  uhdr_hip_image_t y, g;
  uhdr_hip_metadata_t md;
  if (yuv420_image_ptr) y = to_c(*yuv420_image_ptr);
  if (gainmap_image_ptr) { g = to_c(*gainmap_image_ptr); if (g.luma_stride == 0) g.luma_stride = g.width; }
  if (metadata) md = to_c(*metadata);
  return encode_into(dest, [&](void* out, size_t cap, size_t* n) {
    return uhdr_hip_jpegr_encode_apix(yuv420_image_ptr ? &y : nullptr, gainmap_image_ptr ? &g : nullptr, metadata ? &md : nullptr, quality, exif_ptr(exif),
                                      exif && exif->data ? exif->length : 0, out, cap, n, UHDR_HIP_MEM_HOST, nullptr);
  });
}

status_t JpegRHip::getJPEGRInfo(uhdr_compressed_ptr jpegr_image_ptr, uhdr_info_ptr info) {
  if (jpegr_image_ptr == nullptr || jpegr_image_ptr->data == nullptr || info == nullptr) return ERROR_ULTRAHDR_BAD_PTR;   // jpegr.cpp:634-641
  uhdr_hip_jpeg_info_t a, g;
  const int rc = uhdr_hip_jpegr_info(jpegr_image_ptr->data, (size_t)jpegr_image_ptr->length, &a, info->gainmapImgInfo ? &g : nullptr);
  if (rc != UHDR_HIP_NO_ERROR) return static_cast<status_t>(rc);
  const uint8_t* f = static_cast<const uint8_t*>(jpegr_image_ptr->data);
  auto fill = [&](const uhdr_hip_jpeg_info_t& s, jpeg_info_struct* d) {   // parseJpegInfo jpegr.cpp:888-909
    d->width = s.width; d->height = s.height;
    d->imgData.assign(f + s.offset, f + s.offset + s.size);
    if (s.icc_size) d->iccData.assign(f + s.icc_offset, f + s.icc_offset + s.icc_size);
    if (s.exif_size) d->exifData.assign(f + s.exif_offset, f + s.exif_offset + s.exif_size);
    if (s.xmp_size) { d->xmpData.assign(f + s.xmp_offset, f + s.xmp_offset + s.xmp_size); d->xmpData.push_back(0); }   // jpegdecoderhelper.cpp:235
  };
  info->width = a.width; info->height = a.height;
  if (info->primaryImgInfo) fill(a, info->primaryImgInfo);
  if (info->gainmapImgInfo) fill(g, info->gainmapImgInfo);
  return ULTRAHDR_NO_ERROR;
}

status_t JpegRHip::decodeJPEGR(uhdr_compressed_ptr jpegr_image_ptr, uhdr_uncompressed_ptr dest, float max_display_boost, uhdr_exif_ptr exif,
                               ultrahdr_output_format output_format, uhdr_uncompressed_ptr gainmap_image_ptr, ultrahdr_metadata_ptr metadata) {
  if (jpegr_image_ptr == nullptr || jpegr_image_ptr->data == nullptr) return ERROR_ULTRAHDR_BAD_PTR;      // jpegr.cpp:658-661
  if (dest == nullptr || dest->data == nullptr) return ERROR_ULTRAHDR_BAD_PTR;                            // :662-665
  if (max_display_boost < 1.0f) return ERROR_ULTRAHDR_INVALID_DISPLAY_BOOST;
  if (exif != nullptr && exif->data == nullptr) return ERROR_ULTRAHDR_BAD_PTR;                            // :670-673
  if (gainmap_image_ptr != nullptr && gainmap_image_ptr->data == nullptr) return ERROR_ULTRAHDR_BAD_PTR;  // :674-677
  if (output_format <= ULTRAHDR_OUTPUT_UNSPECIFIED || output_format > ULTRAHDR_OUTPUT_MAX) return ERROR_ULTRAHDR_INVALID_OUTPUT_FORMAT;
  if (uhdr_hip_init(0) != UHDR_HIP_NO_ERROR) return ULTRAHDR_UNKNOWN_ERROR;
  const void* file = jpegr_image_ptr->data;
  const size_t n = (size_t)jpegr_image_ptr->length;
  uhdr_hip_jpeg_info_t a, g;
  int rc = uhdr_hip_jpegr_info(file, n, &a, &g);
  if (rc != UHDR_HIP_NO_ERROR) return static_cast<status_t>(rc);
  if (exif != nullptr) {                                                                                   // :721-727
    if (exif->length < a.exif_size) return ERROR_ULTRAHDR_BUFFER_TOO_SMALL;
    std::memcpy(exif->data, static_cast<const uint8_t*>(file) + a.exif_offset, a.exif_size);
    exif->length = a.exif_size;
  }
  if (gainmap_image_ptr != nullptr) {                                                                      // :731-749: first plane of the gain map's JPEG
    JpegDecoderHelperHip dec;
    if (!dec.decompressImage(static_cast<const uint8_t*>(file) + g.offset, (int)g.size)) return ERROR_ULTRAHDR_DECODE_ERROR;
    gainmap_image_ptr->width = dec.getDecompressedImageWidth();
    gainmap_image_ptr->height = dec.getDecompressedImageHeight();
    std::memcpy(gainmap_image_ptr->data, dec.getDecompressedImagePtr(), gainmap_image_ptr->width * gainmap_image_ptr->height);
  }
  uhdr_hip_image_t d = to_c(*dest);
  uhdr_hip_metadata_t md;
  const size_t bpp = output_format == ULTRAHDR_OUTPUT_HDR_LINEAR ? 8 : output_format == ULTRAHDR_OUTPUT_HDR_LINEAR_RGB_10BIT ? 6 : 4;
  const bool want_md = metadata != nullptr || output_format != ULTRAHDR_OUTPUT_SDR;   // jpegr.cpp:754
  rc = uhdr_hip_jpegr_decode(file, n, (int)output_format, max_display_boost, dest->data, a.width * a.height * bpp, &d, want_md ? &md : nullptr, mApplyMode,
                             UHDR_HIP_MEM_HOST, nullptr);
  if (rc != UHDR_HIP_NO_ERROR) return static_cast<status_t>(rc);
  dest->width = d.width; dest->height = d.height;
  dest->colorGamut = static_cast<ultrahdr_color_gamut>(d.colorGamut);
  if (metadata != nullptr) {                                                                               // :757-766
    metadata->version = md.version;
    metadata->maxContentBoost = md.maxContentBoost; metadata->minContentBoost = md.minContentBoost; metadata->gamma = md.gamma;
    metadata->offsetSdr = md.offsetSdr; metadata->offsetHdr = md.offsetHdr;
    metadata->hdrCapacityMin = md.hdrCapacityMin; metadata->hdrCapacityMax = md.hdrCapacityMax;
  }
  return ULTRAHDR_NO_ERROR;
}

}  // namespace ultrahdr
