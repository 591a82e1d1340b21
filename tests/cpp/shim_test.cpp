// shim_test.cpp -- exercises ultrahdr::UltraHdrHip the way the reference's own harness does
// (tests/jpegr_test.cpp:2203-2304, JpegRBenchmark: generateGainMap then applyGainMap on the 1280x720
// fixture, HLG, P010 BT.2100 vs YUV420 BT.709, display boost = max content boost).
// usage: shim_test <raw_p010> <raw_yuv420> <out_dir>    -> writes map/apply/tonemap/convert outputs
#include <cfloat>
#include <cstdio>
#include <cstdlib>
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
  printf("shim_test ok\n");
  return 0;
}
