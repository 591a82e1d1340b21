/* oracle/ref_container_harness.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * Thin extern "C" harness around the REFERENCE's own container code, compiled in place from /root/reference (oracle/Makefile,
 * `make ref`): lib/src/jpegr.cpp, jpegrutils.cpp, multipictureformat.cpp, icc.cpp, jpegdecoderhelper.cpp, gainmapmath.cpp and the
 * vendored third_party/image_io sources, with the image's real libjpeg headers.  Not part of the build: lib/src/ultrahdr.cpp (it
 * includes an un-vendored libheif fork) and lib/src/jpegencoderhelper.cpp (does not compile against the image's IJG jpeglib.h:
 * `boolean` is an enum there) -- the functions they define stay undefined in the shared object and are never reached by the entry
 * points below, which are the ones of the reference that need neither:
 *   JpegR::encodeJPEGR API-4 (jpegr.cpp:608-653 -> appendGainMap :951-1130)   two JPEG streams + metadata -> the JPEG/R file
 *   JpegR::getJPEGRInfo / extractPrimaryImageAndGainMap (jpegr.cpp:655-690, 824-949)
 *   generateXmpForPrimaryImage / generateXmpForSecondaryImage / getMetadataFromXMP (jpegrutils.cpp)
 *   generateMpf (multipictureformat.cpp), IccHelper::writeIccProfile / readIccColorGamut (icc.cpp)
 * tests/test_ref_container.py compares the restatement (oracle/jpegr_oracle.py) and the product's host code with them.
 * Nothing here travels to the GPU box's product path; the .so is built only where /root/reference exists.
 */
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "ultrahdr/icc.h"
#include "ultrahdr/jpegr.h"
#include "ultrahdr/jpegrutils.h"
#include "ultrahdr/multipictureformat.h"

using namespace ultrahdr;

namespace {
ultrahdr_metadata_struct make_md(float max_boost, float min_boost, float gamma, float off_sdr, float off_hdr, float cap_min, float cap_max) {
  ultrahdr_metadata_struct md;
  md.version = kGainMapVersion;
  md.maxContentBoost = max_boost;
  md.minContentBoost = min_boost;
  md.gamma = gamma;
  md.offsetSdr = off_sdr;
  md.offsetHdr = off_hdr;
  md.hdrCapacityMin = cap_min;
  md.hdrCapacityMax = cap_max;
  return md;
}
long put(const void* src, size_t n, void* out, size_t cap) {
  if (n <= cap && n != 0) memcpy(out, src, n);
  return (long)n;
}
}  // namespace

extern "C" {

/* -> file length (> cap: nothing written), or the reference's negative status */
long refc_encode_api4(const void* sdr_jpeg, int sdr_len, int sdr_gamut, const void* gm_jpeg, int gm_len, float max_boost, float min_boost,
                      float gamma, float off_sdr, float off_hdr, float cap_min, float cap_max, void* out, int cap) {
  ultrahdr_compressed_struct a{const_cast<void*>(sdr_jpeg), sdr_len, sdr_len, (ultrahdr_color_gamut)sdr_gamut};
  ultrahdr_compressed_struct g{const_cast<void*>(gm_jpeg), gm_len, gm_len, ULTRAHDR_COLORGAMUT_UNSPECIFIED};
  ultrahdr_compressed_struct d{out, 0, cap, ULTRAHDR_COLORGAMUT_UNSPECIFIED};
  ultrahdr_metadata_struct md = make_md(max_boost, min_boost, gamma, off_sdr, off_hdr, cap_min, cap_max);
  JpegR codec;
  const status_t st = codec.encodeJPEGR(&a, &g, &md, &d);
  return st == ULTRAHDR_NO_ERROR ? (long)d.length : (long)st;
}

/* out8: primary offset, primary size, gain-map offset, gain-map size (bytes into the file) */
int refc_extract(const void* jpegr, int len, long* out4) {
  ultrahdr_compressed_struct in{const_cast<void*>(jpegr), len, len, ULTRAHDR_COLORGAMUT_UNSPECIFIED};
  ultrahdr_compressed_struct p{nullptr, 0, 0, ULTRAHDR_COLORGAMUT_UNSPECIFIED}, g = p;
  const status_t st = JpegR::extractPrimaryImageAndGainMap(&in, &p, &g);
  if (st != ULTRAHDR_NO_ERROR) return (int)st;
  out4[0] = (long)((const uint8_t*)p.data - (const uint8_t*)jpegr); out4[1] = p.length;
  out4[2] = (long)((const uint8_t*)g.data - (const uint8_t*)jpegr); out4[3] = g.length;
  return 0;
}

/* dims[4]: primary w, h, gain map w, h; the three payloads of each image are copied out when they fit; sizes[6] = primary icc,
 * exif, xmp, gain map icc, exif, xmp */
int refc_info(const void* jpegr, int len, long* dims, long* sizes, void* p_icc, void* p_exif, void* p_xmp, void* g_icc, void* g_exif,
              void* g_xmp, long cap) {
  ultrahdr_compressed_struct in{const_cast<void*>(jpegr), len, len, ULTRAHDR_COLORGAMUT_UNSPECIFIED};
  jpeg_info_struct pi, gi;
  jpegr_info_struct info;
  info.primaryImgInfo = &pi;
  info.gainmapImgInfo = &gi;
  JpegR codec;
  const status_t st = codec.getJPEGRInfo(&in, &info);
  if (st != ULTRAHDR_NO_ERROR) return (int)st;
  dims[0] = (long)pi.width; dims[1] = (long)pi.height; dims[2] = (long)gi.width; dims[3] = (long)gi.height;
  sizes[0] = put(pi.iccData.data(), pi.iccData.size(), p_icc, (size_t)cap);
  sizes[1] = put(pi.exifData.data(), pi.exifData.size(), p_exif, (size_t)cap);
  sizes[2] = put(pi.xmpData.data(), pi.xmpData.size(), p_xmp, (size_t)cap);
  sizes[3] = put(gi.iccData.data(), gi.iccData.size(), g_icc, (size_t)cap);
  sizes[4] = put(gi.exifData.data(), gi.exifData.size(), g_exif, (size_t)cap);
  sizes[5] = put(gi.xmpData.data(), gi.xmpData.size(), g_xmp, (size_t)cap);
  return 0;
}

long refc_xmp_primary(int secondary_length, float max_boost, float min_boost, void* out, long cap) {
  ultrahdr_metadata_struct md = make_md(max_boost, min_boost, 1.0f, 0.0f, 0.0f, min_boost, max_boost);
  const std::string s = generateXmpForPrimaryImage(secondary_length, md);
  return put(s.data(), s.size(), out, (size_t)cap);
}
long refc_xmp_secondary(float max_boost, float min_boost, float gamma, float off_sdr, float off_hdr, float cap_min, float cap_max, void* out,
                        long cap) {
  ultrahdr_metadata_struct md = make_md(max_boost, min_boost, gamma, off_sdr, off_hdr, cap_min, cap_max);
  const std::string s = generateXmpForSecondaryImage(md);
  return put(s.data(), s.size(), out, (size_t)cap);
}
/* md7: max, min, gamma, offsetSdr, offsetHdr, capMin, capMax; version copied to ver (<= 15 chars); returns 1 when the packet parses */
int refc_parse_xmp(const void* xmp, long n, float* md7, char* ver) {
  std::vector<uint8_t> buf((const uint8_t*)xmp, (const uint8_t*)xmp + n);
  ultrahdr_metadata_struct md;
  if (!getMetadataFromXMP(buf.data(), buf.size(), &md)) return 0;
  md7[0] = md.maxContentBoost; md7[1] = md.minContentBoost; md7[2] = md.gamma; md7[3] = md.offsetSdr; md7[4] = md.offsetHdr;
  md7[5] = md.hdrCapacityMin; md7[6] = md.hdrCapacityMax;
  strncpy(ver, md.version.c_str(), 15);
  ver[15] = 0;
  return 1;
}
long refc_mpf(int primary_size, int primary_offset, int secondary_size, int secondary_offset, void* out, long cap) {
  auto d = generateMpf(primary_size, primary_offset, secondary_size, secondary_offset);
  return put(d->getData(), (size_t)d->getLength(), out, (size_t)cap);
}
long refc_icc_write(int tf, int gamut, void* out, long cap) {
  auto d = IccHelper::writeIccProfile((ultrahdr_transfer_function)tf, (ultrahdr_color_gamut)gamut);
  if (!d) return -1;
  return put(d->getData(), (size_t)d->getLength(), out, (size_t)cap);
}
int refc_icc_read_gamut(const void* icc, long n) {
  std::vector<uint8_t> buf((const uint8_t*)icc, (const uint8_t*)icc + n);
  return (int)IccHelper::readIccColorGamut(buf.data(), buf.size());
}

}  // extern "C"

/* JpegDecoderHelper (lib/src/jpegdecoderhelper.cpp, the reference's own object code on the image's libjpeg):
 * decompressImage(DECODE_TO_YCBCR) -> bytes written (> cap: nothing copied), -1 when the reference's call fails; whg[3] = width,
 * height, 1 for a single-plane image (size == width * height) */
extern "C" long refc_jpeg_decode(const void* jpeg, int len, void* out, long cap, long* whg) {
  JpegDecoderHelper dec;
  if (!dec.decompressImage(jpeg, len, DECODE_TO_YCBCR)) return -1;
  whg[0] = (long)dec.getDecompressedImageWidth();
  whg[1] = (long)dec.getDecompressedImageHeight();
  const size_t n = dec.getDecompressedImageSize();
  whg[2] = n == (size_t)whg[0] * (size_t)whg[1] ? 1 : 0;
  return put(dec.getDecompressedImagePtr(), n, out, (size_t)cap);
}
/* getCompressedImageParameters: out5 = width, height, icc size, exif size, xmp size; 0 when the call fails */
extern "C" int refc_jpeg_params(const void* jpeg, int len, long* out5) {
  JpegDecoderHelper dec;
  if (!dec.getCompressedImageParameters(jpeg, len)) return 0;
  out5[0] = (long)dec.getDecompressedImageWidth(); out5[1] = (long)dec.getDecompressedImageHeight();
  out5[2] = (long)dec.getICCSize(); out5[3] = (long)dec.getEXIFSize(); out5[4] = (long)dec.getXMPSize();
  return 1;
}
