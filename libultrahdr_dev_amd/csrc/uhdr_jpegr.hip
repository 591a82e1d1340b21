// uhdr_jpegr.hip -- the decode surface of JpegR on the device: JpegR::decodeJPEGR (lib/src/jpegr.cpp:655-822) for the HDR
// output formats.  A JPEG/R file is two concatenated JPEGs (primary SDR image, gain map); the gain map's APP1 carries the
// hdrgm:* XMP attributes.  Host work here is container bookkeeping only (marker walking, a dozen XMP attributes, three ICC
// colorant tags); both images are decompressed by the device decoder (uhdr_jpeg_dec.hip) into device memory and combined by
// the applyGainMap kernels without touching the host.
//   extractPrimaryImageAndGainMap  jpegr.cpp:823-876   (image ranges; the reference uses image_io's JpegScanner)
//   getMetadataFromXMP             jpegrutils.cpp:436-545 (+ the XMPXmlHandler getters :213-330)
//   IccHelper::readIccColorGamut   icc.cpp:615-685
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/uhdr_hip.h"
#include "uhdr_jpegr.h"

namespace uhdr {
namespace jpegr {

namespace {
unsigned rd16(const uint8_t* p) { return ((unsigned)p[0] << 8) | p[1]; }
uint32_t rd32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

// one image: SOI at pos, marker segments, entropy-coded data after every SOS, EOI.  Returns the index after EOI, 0 if malformed.
size_t walk_image(const uint8_t* d, size_t n, size_t pos) {
  if (pos + 4 > n || d[pos] != 0xFF || d[pos + 1] != 0xD8) return 0;
  pos += 2;
  for (;;) {
    while (pos + 1 < n && d[pos] == 0xFF && d[pos + 1] == 0xFF) pos++;   // fill bytes
    if (pos + 2 > n || d[pos] != 0xFF) return 0;
    const unsigned m = d[pos + 1];
    if (m == 0xD9) return pos + 2;
    if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) { pos += 2; continue; }   // stand-alone markers
    if (pos + 4 > n) return 0;
    const size_t len = rd16(d + pos + 2);
    if (len < 2 || pos + 2 + len > n) return 0;
    pos += 2 + len;
    if (m == 0xDA) {   // entropy-coded segment: up to the next marker that is neither a stuffed zero, a fill byte nor RSTn
      for (;;) {
        const void* f = pos + 1 < n ? memchr(d + pos, 0xFF, n - 1 - pos) : nullptr;
        if (f == nullptr) return 0;
        pos = (size_t)(static_cast<const uint8_t*>(f) - d);
        const unsigned k = d[pos + 1];
        if (k == 0x00 || (k >= 0xD0 && k <= 0xD7)) { pos += 2; continue; }
        if (k == 0xFF) { pos += 1; continue; }
        break;
      }
    }
  }
}
}  // namespace

int find_images(const uint8_t* d, size_t n, Range out[2]) {
  int count = 0;
  size_t pos = 0;
  while (count < 2) {
    const void* f = pos + 1 < n ? memchr(d + pos, 0xFF, n - 1 - pos) : nullptr;
    if (f == nullptr) break;
    pos = (size_t)(static_cast<const uint8_t*>(f) - d);
    if (d[pos + 1] != 0xD8) { pos++; continue; }
    const size_t end = walk_image(d, n, pos);
    if (end == 0) break;
    out[count].begin = pos;
    out[count].len = end - pos;
    ++count;
    pos = end;
  }
  return count;
}

bool find_app_segment(const uint8_t* jpg, size_t n, unsigned marker, const char* prefix, size_t prefix_len, const uint8_t** payload,
                      size_t* payload_len) {
  size_t pos = 2;
  while (pos + 4 <= n && jpg[pos] == 0xFF) {
    const unsigned m = jpg[pos + 1];
    if (m == 0xDA || m == 0xD9) break;
    const size_t len = rd16(jpg + pos + 2);
    if (len < 2 || pos + 2 + len > n) break;
    if (m == marker && len - 2 > prefix_len && memcmp(jpg + pos + 4, prefix, prefix_len) == 0) {   // "len > sizeof(prefix)", jpegdecoderhelper.cpp:222-240
      *payload = jpg + pos + 4;
      *payload_len = len - 2;
      return true;
    }
    pos += 2 + len;
  }
  return false;
}

namespace {
// value of attribute `name` (name="value" or name='value', blanks around '=' allowed) in an XMP packet
bool xmp_attribute(const std::string& xml, const char* name, std::string* value) {
  const size_t nl = strlen(name);
  size_t at = 0;
  for (;;) {
    at = xml.find(name, at);
    if (at == std::string::npos) return false;
    const bool starts_name = at == 0 || !(isalnum((unsigned char)xml[at - 1]) || xml[at - 1] == ':' || xml[at - 1] == '_' || xml[at - 1] == '-');
    size_t p = at + nl;
    while (p < xml.size() && isspace((unsigned char)xml[p])) p++;
    if (starts_name && p < xml.size() && xml[p] == '=') {
      p++;
      while (p < xml.size() && isspace((unsigned char)xml[p])) p++;
      if (p < xml.size() && (xml[p] == '"' || xml[p] == '\'')) {
        const char q = xml[p];
        const size_t e = xml.find(q, p + 1);
        if (e == std::string::npos) return false;
        *value = xml.substr(p + 1, e - p - 1);
        return true;
      }
    }
    at += nl;
  }
}
// `stringstream ss(str); float val; ss >> val` (jpegrutils.cpp:226-233): leading blanks skipped, a number must follow
bool parse_float(const std::string& s, float* v) {
  const char* b = s.c_str();
  char* e = nullptr;
  const float f = strtof(b, &e);
  if (e == b) return false;
  *v = f;
  return true;
}
}  // namespace

// getMetadataFromXMP: Version, GainMapMax and HDRCapacityMax are required; the others default (min 1, gamma 1, offsets 1/64,
// capacity min 1); GainMap* and HDRCapacity* are stored as log2; BaseRenditionIsHDR = "True" is refused.
bool metadata_from_xmp(const uint8_t* payload, size_t len, uhdr_hip_metadata_t* md) {
  static const char kNs[] = "http://ns.adobe.com/xap/1.0/";
  if (len < sizeof(kNs) + 2 || memcmp(payload, kNs, sizeof(kNs) - 1) != 0) return false;
  const std::string xml(reinterpret_cast<const char*>(payload) + sizeof(kNs), len - sizeof(kNs));
  std::string v;
  float f;
  memset(md, 0, sizeof(*md));
  if (!xmp_attribute(xml, "hdrgm:Version", &v)) return false;
  strncpy(md->version, v.c_str(), sizeof(md->version) - 1);
  if (!xmp_attribute(xml, "hdrgm:GainMapMax", &v) || !parse_float(v, &f)) return false;
  md->maxContentBoost = (float)exp2((double)f);
  if (!xmp_attribute(xml, "hdrgm:HDRCapacityMax", &v) || !parse_float(v, &f)) return false;
  md->hdrCapacityMax = (float)exp2((double)f);
  md->minContentBoost = 1.0f;
  if (xmp_attribute(xml, "hdrgm:GainMapMin", &v)) { if (!parse_float(v, &f)) return false; md->minContentBoost = (float)exp2((double)f); }
  md->gamma = 1.0f;
  if (xmp_attribute(xml, "hdrgm:Gamma", &v)) { if (!parse_float(v, &f)) return false; md->gamma = f; }
  md->offsetSdr = 1.0f / 64.0f;
  if (xmp_attribute(xml, "hdrgm:OffsetSDR", &v)) { if (!parse_float(v, &f)) return false; md->offsetSdr = f; }
  md->offsetHdr = 1.0f / 64.0f;
  if (xmp_attribute(xml, "hdrgm:OffsetHDR", &v)) { if (!parse_float(v, &f)) return false; md->offsetHdr = f; }
  md->hdrCapacityMin = 1.0f;
  if (xmp_attribute(xml, "hdrgm:HDRCapacityMin", &v)) { if (!parse_float(v, &f)) return false; md->hdrCapacityMin = (float)exp2((double)f); }
  if (xmp_attribute(xml, "hdrgm:BaseRenditionIsHDR", &v)) {
    if (v == "True") return false;        // "Base rendition of HDR is not supported" (:537-540)
    if (v != "False") return false;       // a present but unparsable field is an error (:531-534)
  }
  return true;
}

// readIccColorGamut: the rXYZ / gXYZ / bXYZ colorant tags compared with the three profiles the encoder writes
int gamut_from_icc(const uint8_t* payload, size_t len) {
  static const char kId[] = "ICC_PROFILE";
  const size_t kIdSize = 14, kHeader = 132;     // identifier + chunk count / index; ICC header + tag count
  if (payload == nullptr || len < kHeader + kIdSize || memcmp(payload, kId, sizeof(kId)) != 0) return UHDR_HIP_CG_UNSPECIFIED;
  const uint8_t* icc = payload + kIdSize;
  const size_t n = len - kIdSize;
  const uint32_t tags = rd32(icc + 128);
  uint32_t off[3] = {0, 0, 0}, sz[3] = {0, 0, 0};
  for (uint32_t t = 0; t < tags; ++t) {
    if (n < kHeader + (size_t)(t + 1) * 12) return UHDR_HIP_CG_UNSPECIFIED;
    const uint8_t* e = icc + kHeader + (size_t)t * 12;
    const int k = memcmp(e, "rXYZ", 4) == 0 ? 0 : memcmp(e, "gXYZ", 4) == 0 ? 1 : memcmp(e, "bXYZ", 4) == 0 ? 2 : -1;
    if (k >= 0 && off[k] == 0) { off[k] = rd32(e + 4); sz[k] = rd32(e + 8); }
  }
  for (int k = 0; k < 3; ++k)
    if (off[k] == 0 || sz[k] != 20 || (size_t)off[k] + 20 > n) return UHDR_HIP_CG_UNSPECIFIED;
  auto fixed = [](float x) { return (int32_t)floor((double)x * 65536.0 + 0.5); };   // float_round_to_fixed, icc.h:163-165
  auto ff = [](int v) { return (float)v * (1.0f / 65536.0f); };                       // FixedToFloat
  const float m[3][3][3] = {
      {{ff(0x6FA2), ff(0x6299), ff(0x24A0)}, {ff(0x38F5), ff(0xB785), ff(0x0F84)}, {ff(0x0390), ff(0x18DA), ff(0xB6CF)}},      // kSRGB, icc.h:115-123
      {{0.515102f, 0.291965f, 0.157153f}, {0.241182f, 0.692236f, 0.0665819f}, {-0.00104941f, 0.0418818f, 0.784378f}},            // kDisplayP3
      {{0.673459f, 0.165661f, 0.125100f}, {0.279033f, 0.675338f, 0.0456288f}, {-0.00193139f, 0.0299794f, 0.797162f}}};           // kRec2020
  const int gamut[3] = {UHDR_HIP_CG_BT709, UHDR_HIP_CG_P3, UHDR_HIP_CG_BT2100};
  for (int g = 0; g < 3; ++g) {
    bool same = true;
    for (int c = 0; c < 3 && same; ++c) {   // colorant c = column c of the matrix: (X, Y, Z) = m[0][c], m[1][c], m[2][c]
      const uint8_t* tag = icc + off[c];
      same = memcmp(tag, "XYZ ", 4) == 0 && rd32(tag + 4) == 0;
      for (int r = 0; r < 3 && same; ++r) same = (int32_t)rd32(tag + 8 + 4 * r) == fixed(m[g][r][c]);
    }
    if (same) return gamut[g];
  }
  return UHDR_HIP_CG_UNSPECIFIED;
}

}  // namespace jpegr
}  // namespace uhdr
