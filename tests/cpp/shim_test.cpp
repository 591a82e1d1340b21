// shim_test.cpp -- exercises ultrahdr::UltraHdrHip the way the reference's own harness does
// (tests/jpegr_test.cpp:2203-2304, JpegRBenchmark: generateGainMap then applyGainMap on the 1280x720
// fixture, HLG, P010 BT.2100 vs YUV420 BT.709, display boost = max content boost).
// usage: shim_test <raw_p010> <raw_yuv420> <out_dir>    -> writes map/apply/tonemap/convert outputs
#include <cfloat>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "ultrahdr_hip/ultrahdr_hip.h"

using namespace ultrahdr;

static std::vector<uint8_t> slurp(const char* p) {
  FILE* f = fopen(p, "rb");
  if (!f) { perror(p); exit(2); }
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  std::vector<uint8_t> b(n);
  if (fread(b.data(), 1, n, f) != (size_t)n) exit(2);
  fclose(f);
  return b;
}
static void dump(const std::string& p, const void* d, size_t n) {
  FILE* f = fopen(p.c_str(), "wb");
  fwrite(d, 1, n, f);
  fclose(f);
}
#define CHECK(x) do { if (!(x)) { fprintf(stderr, "CHECK failed: %s (line %d)\n", #x, __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  const size_t w = 1280, h = 720;
  std::vector<uint8_t> p010 = slurp(argv[1]), yuv = slurp(argv[2]);
  const std::string out = argv[3];
  UltraHdrHip uhdr(0);

  ultrahdr_uncompressed_struct yuv420{}, hdr{};
  yuv420.data = yuv.data(); yuv420.width = w; yuv420.height = h; yuv420.colorGamut = ULTRAHDR_COLORGAMUT_BT709;
  yuv420.luma_stride = w; yuv420.chroma_data = yuv.data() + w * h; yuv420.chroma_stride = w / 2;   // jpegr.cpp:265-278
  hdr.data = p010.data(); hdr.width = w; hdr.height = h; hdr.colorGamut = ULTRAHDR_COLORGAMUT_BT2100;
  hdr.luma_stride = w; hdr.chroma_data = p010.data() + w * h * 2; hdr.chroma_stride = w;

  // error behaviour first (ultrahdr.cpp:189-202)
  ultrahdr_metadata_struct metadata;
  ultrahdr_uncompressed_struct map{};
  CHECK(uhdr.generateGainMap(nullptr, &hdr, ULTRAHDR_TF_HLG, &metadata, &map) == ERROR_ULTRAHDR_BAD_PTR);
  ultrahdr_uncompressed_struct bad = hdr; bad.width = w + 2;
  CHECK(uhdr.generateGainMap(&yuv420, &bad, ULTRAHDR_TF_HLG, &metadata, &map) == ERROR_ULTRAHDR_RESOLUTION_MISMATCH);
  bad = hdr; bad.colorGamut = ULTRAHDR_COLORGAMUT_UNSPECIFIED;
  CHECK(uhdr.generateGainMap(&yuv420, &bad, ULTRAHDR_TF_HLG, &metadata, &map) == ERROR_ULTRAHDR_INVALID_COLORGAMUT);
  CHECK(uhdr.generateGainMap(&yuv420, &hdr, ULTRAHDR_TF_SRGB, &metadata, &map) == ERROR_ULTRAHDR_INVALID_TRANS_FUNC);

  CHECK(uhdr.generateGainMap(&yuv420, &hdr, ULTRAHDR_TF_HLG, &metadata, &map) == ULTRAHDR_NO_ERROR);
  std::unique_ptr<uint8_t[]> map_data(reinterpret_cast<uint8_t*>(map.data));   // caller owns the new[]'d map
  CHECK(map.width == w / 4 && map.height == h / 4 && map.luma_stride == w / 4);
  CHECK(map.pixelFormat == ULTRAHDR_PIX_FMT_MONOCHROME && map.colorGamut == ULTRAHDR_COLORGAMUT_UNSPECIFIED);
  CHECK(metadata.version == kGainMapVersion && metadata.minContentBoost == 1.0f && metadata.gamma == 1.0f);
  CHECK(metadata.maxContentBoost == 1000.0f / 203.0f && metadata.hdrCapacityMax == metadata.maxContentBoost);
  dump(out + "/map_hlg.bin", map.data, map.width * map.height);

  std::vector<uint8_t> rgba(w * h * 8);
  ultrahdr_uncompressed_struct dest{};
  dest.data = rgba.data();
  uhdr.setApplyMode(1);  // EXACT: the reference's bytes
  CHECK(uhdr.applyGainMap(&yuv420, &map, &metadata, ULTRAHDR_OUTPUT_HDR_HLG, metadata.maxContentBoost, &dest) == ULTRAHDR_NO_ERROR);
  CHECK(dest.width == w && dest.height == h && dest.colorGamut == ULTRAHDR_COLORGAMUT_BT709);
  dump(out + "/apply_hlg_exact.bin", rgba.data(), w * h * 4);
  CHECK(uhdr.applyGainMap(&yuv420, &map, &metadata, ULTRAHDR_OUTPUT_HDR_LINEAR, FLT_MAX, &dest) == ULTRAHDR_NO_ERROR);
  dump(out + "/apply_f16_exact.bin", rgba.data(), w * h * 8);
  uhdr.setApplyMode(0);
  CHECK(uhdr.applyGainMap(&yuv420, &map, &metadata, ULTRAHDR_OUTPUT_HDR_PQ, FLT_MAX, &dest) == ULTRAHDR_NO_ERROR);
  dump(out + "/apply_pq_fast.bin", rgba.data(), w * h * 4);
  ultrahdr_metadata_struct m2 = metadata; m2.version = "1.1";
  CHECK(uhdr.applyGainMap(&yuv420, &map, &m2, ULTRAHDR_OUTPUT_HDR_PQ, FLT_MAX, &dest) == ERROR_ULTRAHDR_BAD_METADATA);

  // toneMap into a 16-aligned-stride destination as API-0 does (jpegr.cpp:185-197)
  std::vector<uint8_t> sdr(w * h * 3 / 2, 0xAA);
  ultrahdr_uncompressed_struct tm{};
  tm.data = sdr.data(); tm.width = w; tm.height = h; tm.luma_stride = w; tm.chroma_data = sdr.data() + w * h; tm.chroma_stride = w / 2;
  tm.colorGamut = ULTRAHDR_COLORGAMUT_UNSPECIFIED;
  CHECK(uhdr.toneMap(&hdr, &tm) == ULTRAHDR_NO_ERROR && tm.colorGamut == ULTRAHDR_COLORGAMUT_BT2100);
  dump(out + "/tonemap.bin", sdr.data(), sdr.size());

  // convertYuv 709 -> 601 in place (jpegr.cpp:360)
  std::vector<uint8_t> cv = yuv;
  ultrahdr_uncompressed_struct ci = yuv420; ci.data = cv.data(); ci.chroma_data = cv.data() + w * h;
  CHECK(uhdr.convertYuv(&ci, ULTRAHDR_COLORGAMUT_BT709, ULTRAHDR_COLORGAMUT_P3) == ULTRAHDR_NO_ERROR);
  CHECK(uhdr.convertYuv(&ci, ULTRAHDR_COLORGAMUT_UNSPECIFIED, ULTRAHDR_COLORGAMUT_P3) == ERROR_ULTRAHDR_INVALID_COLORGAMUT);
  dump(out + "/convert_709_601.bin", cv.data(), cv.size());
  // editorhelper free functions on the 1280x720 SDR frame (tests/editorhelper_test.cpp shape: status + dims)
  std::vector<uint8_t> fx(w * h * 3 / 2 + 64);
  ultrahdr_uncompressed_struct fin = yuv420, fout{};
  fin.pixelFormat = ULTRAHDR_PIX_FMT_YUV420;
  fout.data = fx.data();
  CHECK(rotate(&fin, 90, &fout) == ULTRAHDR_NO_ERROR && fout.width == h && fout.height == w && fout.pixelFormat == ULTRAHDR_PIX_FMT_YUV420);
  dump(out + "/rotate90.bin", fx.data(), w * h * 3 / 2);
  CHECK(crop(&fin, 100, 739, 40, 519, &fout) == ULTRAHDR_NO_ERROR && fout.width == 640 && fout.height == 480);
  dump(out + "/crop.bin", fx.data(), 640 * 480 * 3 / 2);
  CHECK(mirror(&fin, ULTRAHDR_MIRROR_HORIZONTAL, &fout) == ULTRAHDR_NO_ERROR && fout.width == w);
  dump(out + "/mirror_h.bin", fx.data(), w * h * 3 / 2);
  CHECK(resize(&fin, 640, 360, &fout) == ULTRAHDR_NO_ERROR && fout.width == 640 && fout.height == 360);
  dump(out + "/resize.bin", fx.data(), 640 * 360 * 3 / 2);
  CHECK(rotate(&fin, 45, &fout) == ERROR_ULTRAHDR_INVALID_CROPPING_PARAMETERS);
  // addEffects with the reference's effect structs (editorhelper_test.cpp:553-600 shape, with a valid rotation)
  {
    ultrahdr_resize_effect rs; rs.new_width = (int)(w * 3 / 4); rs.new_height = (int)(h * 3 / 4);
    ultrahdr_mirror_effect mi; mi.mirror_dir = ULTRAHDR_MIRROR_VERTICAL;
    ultrahdr_rotate_effect ro; ro.clockwise_degree = 90;
    ultrahdr_crop_effect cr; cr.top = 10; cr.bottom = 99; cr.left = 20; cr.right = 149;
    std::vector<ultrahdr_effect*> effects = {&rs, &mi, &ro, &cr};
    ultrahdr_uncompressed_struct fo2{};
    fo2.data = fx.data();
    CHECK(addEffects(&fin, effects, &fo2) == ULTRAHDR_NO_ERROR && fo2.width == 130 && fo2.height == 90 && fo2.pixelFormat == ULTRAHDR_PIX_FMT_YUV420);
    CHECK(fo2.chroma_data == fx.data() + 130 * 90);
    dump(out + "/effects_chain.bin", fx.data(), 130 * 90 * 3 / 2);
  }
  // JPEG helpers: the gain map as a single-plane JPEG (jpegr.cpp:294-297, quality 85), the SDR frame at quality 95, and back
  JpegEncoderHelperHip enc_map, enc_sdr;
  CHECK(enc_map.compressImage(reinterpret_cast<uint8_t*>(map.data), nullptr, (int)map.width, (int)map.height, (int)map.luma_stride, 0, 85, nullptr, 0));
  CHECK(enc_map.getCompressedImageSize() > 0);
  dump(out + "/map_q85.jpg", enc_map.getCompressedImagePtr(), enc_map.getCompressedImageSize());
  const uint8_t icc[5] = {'i', 'c', 'c', '!', 0};
  CHECK(enc_sdr.compressImage(yuv.data(), yuv.data() + w * h, (int)w, (int)h, (int)w, (int)(w / 2), 95, icc, sizeof(icc)));
  dump(out + "/sdr_q95.jpg", enc_sdr.getCompressedImagePtr(), enc_sdr.getCompressedImageSize());
  JpegDecoderHelperHip dec;
  CHECK(dec.decompressImage(enc_sdr.getCompressedImagePtr(), (int)enc_sdr.getCompressedImageSize()));
  CHECK(dec.getDecompressedImageWidth() == w && dec.getDecompressedImageHeight() == h && !dec.isSingleChannel());
  CHECK(dec.getDecompressedImageSize() == w * h * 3 / 2);
  dump(out + "/sdr_q95_decoded.bin", dec.getDecompressedImagePtr(), dec.getDecompressedImageSize());
  CHECK(dec.decompressImage(enc_map.getCompressedImagePtr(), (int)enc_map.getCompressedImageSize()) && dec.isSingleChannel());
  dump(out + "/map_q85_decoded.bin", dec.getDecompressedImagePtr(), dec.getDecompressedImageSize());
  CHECK(!dec.decompressImage("not a jpeg", 10));
  CHECK(dec.decompressImage(enc_sdr.getCompressedImagePtr(), (int)enc_sdr.getCompressedImageSize(), DECODE_TO_RGBA) && dec.getDecompressedImageSize() == w * h * 4);
  dump(out + "/sdr_q95_rgba.bin", dec.getDecompressedImagePtr(), dec.getDecompressedImageSize());
  CHECK(!dec.decompressImage(enc_map.getCompressedImagePtr(), (int)enc_map.getCompressedImageSize(), DECODE_TO_RGBA));   // single plane: :258-262
  // ---- JpegRHip: the calls of the reference's own encode / decode tests (tests/jpegr_test.cpp: EncodeAPI0..4AndDecodeTest) -------------
  {
    JpegRHip codec;
    std::vector<uint8_t> file(w * h * 3), file2(w * h * 3);
    ultrahdr_compressed_struct jpgr{file.data(), 0, (int)file.size(), ULTRAHDR_COLORGAMUT_UNSPECIFIED};
    ultrahdr_uncompressed_struct raw_p010{}, raw_yuv{};                      // packed, defaults: strides 0 / chroma nullptr (jpegr.cpp:261-275)
    raw_p010.data = p010.data(); raw_p010.width = w; raw_p010.height = h; raw_p010.colorGamut = ULTRAHDR_COLORGAMUT_BT2100;
    raw_yuv.data = yuv.data(); raw_yuv.width = w; raw_yuv.height = h; raw_yuv.colorGamut = ULTRAHDR_COLORGAMUT_BT709;
    // API-0
    CHECK(codec.encodeJPEGR(&raw_p010, ULTRAHDR_TF_HLG, &jpgr, 101, nullptr) == ERROR_ULTRAHDR_INVALID_QUALITY_FACTOR);
    CHECK(codec.encodeJPEGR(static_cast<uhdr_uncompressed_ptr>(nullptr), ULTRAHDR_TF_HLG, &jpgr, 90, nullptr) == ERROR_ULTRAHDR_BAD_PTR);
    CHECK(codec.encodeJPEGR(&raw_p010, ULTRAHDR_TF_HLG, &jpgr, 90, nullptr) == ULTRAHDR_NO_ERROR && jpgr.length > 0);
    dump(out + "/api0.jpgr", jpgr.data, jpgr.length);
    // API-1, with EXIF
    uint8_t exif_bytes[16] = {'E', 'x', 'i', 'f', 0, 0, 'M', 'M', 0, 42, 0, 0, 0, 8, 0, 0};
    ultrahdr_exif_struct exif{exif_bytes, sizeof(exif_bytes)};
    CHECK(codec.encodeJPEGR(&raw_p010, &raw_yuv, ULTRAHDR_TF_HLG, &jpgr, 90, &exif) == ULTRAHDR_NO_ERROR);
    dump(out + "/api1.jpgr", jpgr.data, jpgr.length);
    ultrahdr_compressed_struct tiny{file2.data(), 0, 100, ULTRAHDR_COLORGAMUT_UNSPECIFIED};
    CHECK(codec.encodeJPEGR(&raw_p010, &raw_yuv, ULTRAHDR_TF_HLG, &tiny, 90, nullptr) == ERROR_ULTRAHDR_INSUFFICIENT_RESOURCE);
    // getJPEGRInfo on it
    jpeg_info_struct pinfo, ginfo;
    jpegr_info_struct info{0, 0, &pinfo, &ginfo};
    CHECK(codec.getJPEGRInfo(&jpgr, &info) == ULTRAHDR_NO_ERROR && info.width == w && info.height == h);
    CHECK(ginfo.width == w / 4 && ginfo.height == h / 4 && pinfo.exifData.size() == sizeof(exif_bytes) && !pinfo.iccData.empty() && !ginfo.xmpData.empty());
    CHECK(pinfo.imgData.size() + ginfo.imgData.size() == (size_t)jpgr.length);
    // decodeJPEGR: HDR rendition + exif + gain map + metadata
    std::vector<uint8_t> dec_out(w * h * 8), gm_plane(w * h / 16), exif_back(64);
    ultrahdr_uncompressed_struct decoded{}, gm_img{};
    decoded.data = dec_out.data(); gm_img.data = gm_plane.data();
    ultrahdr_exif_struct exif_out{exif_back.data(), exif_back.size()};
    ultrahdr_metadata_struct md_out;
    CHECK(codec.decodeJPEGR(&jpgr, &decoded, FLT_MAX, &exif_out, ULTRAHDR_OUTPUT_HDR_HLG, &gm_img, &md_out) == ULTRAHDR_NO_ERROR);
    CHECK(decoded.width == w && decoded.height == h && decoded.colorGamut == ULTRAHDR_COLORGAMUT_BT709 && gm_img.width == w / 4 && gm_img.height == h / 4);
    CHECK(exif_out.length == sizeof(exif_bytes) && memcmp(exif_back.data(), exif_bytes, sizeof(exif_bytes)) == 0);
    CHECK(md_out.version == kGainMapVersion && md_out.minContentBoost == 1.0f && md_out.hdrCapacityMin == 1.0f);
    dump(out + "/api1_decoded_hlg.bin", dec_out.data(), w * h * 4);
    dump(out + "/api1_decoded_map.bin", gm_plane.data(), gm_plane.size());
    ultrahdr_exif_struct small_exif{exif_back.data(), 4};
    CHECK(codec.decodeJPEGR(&jpgr, &decoded, FLT_MAX, &small_exif) == ERROR_ULTRAHDR_BUFFER_TOO_SMALL);
    CHECK(codec.decodeJPEGR(&jpgr, &decoded, 0.5f) == ERROR_ULTRAHDR_INVALID_DISPLAY_BOOST);
    CHECK(codec.decodeJPEGR(&jpgr, &decoded, FLT_MAX, nullptr, ULTRAHDR_OUTPUT_SDR) == ULTRAHDR_NO_ERROR && decoded.width == w && decoded.height == h);
    dump(out + "/api1_decoded_sdr.bin", dec_out.data(), w * h * 4);
    // API-2 / API-3 around the SDR JPEG made earlier (its ICC payload is not a profile -> gamut has to come from the struct... and is rejected)
    ultrahdr_compressed_struct sdr_jpg{enc_sdr.getCompressedImagePtr(), (int)enc_sdr.getCompressedImageSize(), (int)enc_sdr.getCompressedImageSize(),
                                       ULTRAHDR_COLORGAMUT_BT709};
    ultrahdr_compressed_struct jpgr2{file2.data(), 0, (int)file2.size(), ULTRAHDR_COLORGAMUT_UNSPECIFIED};
    JpegEncoderHelperHip enc_plain;                                       // the same frame without any ICC segment
    CHECK(enc_plain.compressImage(yuv.data(), yuv.data() + w * h, (int)w, (int)h, (int)w, (int)(w / 2), 95, nullptr, 0));
    ultrahdr_compressed_struct plain_jpg{enc_plain.getCompressedImagePtr(), (int)enc_plain.getCompressedImageSize(), (int)enc_plain.getCompressedImageSize(),
                                         ULTRAHDR_COLORGAMUT_BT709};
    CHECK(codec.encodeJPEGR(&raw_p010, &raw_yuv, &plain_jpg, ULTRAHDR_TF_HLG, &jpgr2) == ULTRAHDR_NO_ERROR);
    dump(out + "/api2.jpgr", jpgr2.data, jpgr2.length);
    CHECK(codec.encodeJPEGR(&raw_p010, &plain_jpg, ULTRAHDR_TF_HLG, &jpgr2) == ULTRAHDR_NO_ERROR);
    dump(out + "/api3.jpgr", jpgr2.data, jpgr2.length);
    plain_jpg.colorGamut = ULTRAHDR_COLORGAMUT_UNSPECIFIED;
    CHECK(codec.encodeJPEGR(&raw_p010, &plain_jpg, ULTRAHDR_TF_HLG, &jpgr2) == ERROR_ULTRAHDR_INVALID_COLORGAMUT);
    // API-4: the two compressed streams of the API-1 file + the decoded metadata give the file back minus its EXIF-less-ness
    ultrahdr_compressed_struct pj{pinfo.imgData.data(), (int)pinfo.imgData.size(), (int)pinfo.imgData.size(), ULTRAHDR_COLORGAMUT_BT709};
    ultrahdr_compressed_struct gj{ginfo.imgData.data(), (int)ginfo.imgData.size(), (int)ginfo.imgData.size(), ULTRAHDR_COLORGAMUT_UNSPECIFIED};
    CHECK(codec.encodeJPEGR(&pj, &gj, &md_out, &jpgr2) == ULTRAHDR_NO_ERROR);
    dump(out + "/api4.jpgr", jpgr2.data, jpgr2.length);
    md_out.version = "1.1";
    CHECK(codec.encodeJPEGR(&pj, &gj, &md_out, &jpgr2) == ERROR_ULTRAHDR_BAD_METADATA);
    // API-x: the planes + the gain map made at the top
    ultrahdr_compressed_struct jpgr3{file2.data(), 0, (int)file2.size(), ULTRAHDR_COLORGAMUT_UNSPECIFIED};
    CHECK(codec.encodeJPEGR(&raw_yuv, &map, &metadata, &jpgr3, 90, nullptr) == ULTRAHDR_NO_ERROR);
    dump(out + "/apix.jpgr", jpgr3.data, jpgr3.length);
  }
  printf("shim_test ok\n");
  return 0;
}
