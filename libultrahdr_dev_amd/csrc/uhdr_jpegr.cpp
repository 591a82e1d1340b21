// uhdr_jpegr.cpp -- host-side bookkeeping of the JPEG/R container, both directions (plain C++: also built under AddressSanitizer by
// tests/cpp/fuzz_host_parsers.cpp).  A JPEG/R file is two concatenated JPEGs (primary SDR image, gain map); the gain map's APP1
// carries the hdrgm:* XMP attributes.  Nothing here touches pixels: marker walking, a dozen XMP attributes, three ICC colorant
// tags, and the writers of the same; the images themselves are (de)compressed and combined on the device (uhdr_capi.hip).
//   extractPrimaryImageAndGainMap  jpegr.cpp:823-876   (image ranges; the reference uses image_io's JpegScanner)
//   getMetadataFromXMP             jpegrutils.cpp:436-545 (+ the XMPXmlHandler getters :213-330)
//   IccHelper::readIccColorGamut   icc.cpp:615-685

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <string>
#include <utility>
#include <vector>

#include "../../include/uhdr_hip.h"
#include "uhdr_jpeg.h"
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
      pos = jpeg::skip_entropy_coded(d, pos, n);
      if (pos + 1 >= n) return 0;
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
// capacity min 1); GainMap* and HDRCapacity* are stored as log2; BaseRenditionIsHDR = "True" is refused.  The reference reads a
// float and calls exp2 on it (jpegrutils.cpp:225-229): the float overload, exp2f -- checked against its object code,
// tests/test_ref_container.py.
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
  md->maxContentBoost = exp2f(f);
  if (!xmp_attribute(xml, "hdrgm:HDRCapacityMax", &v) || !parse_float(v, &f)) return false;
  md->hdrCapacityMax = exp2f(f);
  md->minContentBoost = 1.0f;
  if (xmp_attribute(xml, "hdrgm:GainMapMin", &v)) { if (!parse_float(v, &f)) return false; md->minContentBoost = exp2f(f); }
  md->gamma = 1.0f;
  if (xmp_attribute(xml, "hdrgm:Gamma", &v)) { if (!parse_float(v, &f)) return false; md->gamma = f; }
  md->offsetSdr = 1.0f / 64.0f;
  if (xmp_attribute(xml, "hdrgm:OffsetSDR", &v)) { if (!parse_float(v, &f)) return false; md->offsetSdr = f; }
  md->offsetHdr = 1.0f / 64.0f;
  if (xmp_attribute(xml, "hdrgm:OffsetHDR", &v)) { if (!parse_float(v, &f)) return false; md->offsetHdr = f; }
  md->hdrCapacityMin = 1.0f;
  if (xmp_attribute(xml, "hdrgm:HDRCapacityMin", &v)) { if (!parse_float(v, &f)) return false; md->hdrCapacityMin = exp2f(f); }
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

// ====================================================================================================================
// assembly: generateXmpForPrimaryImage / generateXmpForSecondaryImage (jpegrutils.cpp:547-611), generateMpf
// (multipictureformat.cpp:30-92), IccHelper::writeIccProfile for the sRGB transfer (icc.cpp:410-600), JpegR::appendGainMap
// (jpegr.cpp:951-1130).  The text layout is image_io's XmlWriter's (two-space indentation, one attribute per line); numbers
// go through an ostream with default formatting, i.e. "%g".  Pinned: re-assembling the two JPEG streams of the reference's own
// tests/data/sample_jpegr.jpeg reproduces that file byte for byte (tests/test_jpegr_container.py).
// ====================================================================================================================
namespace {
std::string num(double v) {
  char b[64];
  snprintf(b, sizeof(b), "%g", v);
  return b;
}
const char kXmpHead[] =
    "<x:xmpmeta\n  xmlns:x=\"adobe:ns:meta/\"\n  x:xmptk=\"Adobe XMP Core 5.1.2\">\n  <rdf:RDF\n"
    "    xmlns:rdf=\"http://www.w3.org/1999/02/22-rdf-syntax-ns#\">\n    <rdf:Description\n";
void be16(std::vector<uint8_t>& b, unsigned v) { b.push_back((uint8_t)(v >> 8)); b.push_back((uint8_t)v); }
void be32(std::vector<uint8_t>& b, uint32_t v) { be16(b, v >> 16); be16(b, v & 0xFFFFu); }
}  // namespace

std::string xmp_primary(int secondary_image_length, const char* version) {
  std::string s = kXmpHead;
  s += "      xmlns:Container=\"http://ns.google.com/photos/1.0/container/\"\n";
  s += "      xmlns:Item=\"http://ns.google.com/photos/1.0/container/item/\"\n";
  s += "      xmlns:hdrgm=\"http://ns.adobe.com/hdr-gain-map/1.0/\"\n";
  s += std::string("      hdrgm:Version=\"") + version + "\">\n";
  s += "      <Container:Directory>\n        <rdf:Seq>\n          <rdf:li\n            rdf:parseType=\"Resource\">\n";
  s += "            <Container:Item\n              Item:Semantic=\"Primary\"\n              Item:Mime=\"image/jpeg\"/>\n          </rdf:li>\n";
  s += "          <rdf:li\n            rdf:parseType=\"Resource\">\n            <Container:Item\n              Item:Semantic=\"GainMap\"\n";
  s += "              Item:Mime=\"image/jpeg\"\n              Item:Length=\"" + std::to_string(secondary_image_length) + "\"/>\n          </rdf:li>\n";
  s += "        </rdf:Seq>\n      </Container:Directory>\n    </rdf:Description>\n  </rdf:RDF>\n</x:xmpmeta>\n";
  return s;
}

std::string xmp_secondary(const uhdr_hip_metadata_t& md) {
  std::string s = kXmpHead;
  s += "      xmlns:hdrgm=\"http://ns.adobe.com/hdr-gain-map/1.0/\"\n";
  s += std::string("      hdrgm:Version=\"") + md.version + "\"\n";
  s += "      hdrgm:GainMapMin=\"" + num((double)log2f(md.minContentBoost)) + "\"\n";      // log2(float): the float overload (jpegrutils.cpp:598)
  s += "      hdrgm:GainMapMax=\"" + num((double)log2f(md.maxContentBoost)) + "\"\n";
  s += "      hdrgm:Gamma=\"" + num((double)md.gamma) + "\"\n";
  s += "      hdrgm:OffsetSDR=\"" + num((double)md.offsetSdr) + "\"\n";
  s += "      hdrgm:OffsetHDR=\"" + num((double)md.offsetHdr) + "\"\n";
  s += "      hdrgm:HDRCapacityMin=\"" + num((double)log2f(md.hdrCapacityMin)) + "\"\n";
  s += "      hdrgm:HDRCapacityMax=\"" + num((double)log2f(md.hdrCapacityMax)) + "\"\n";
  s += "      hdrgm:BaseRenditionIsHDR=\"False\"/>\n  </rdf:RDF>\n</x:xmpmeta>\n";
  return s;
}

// 86 bytes: "MPF\0", big-endian TIFF header, index IFD with version / number of images / MP entries, two 16-byte entries
void mpf_segment(int primary_size, int primary_offset, int secondary_size, int secondary_offset, std::vector<uint8_t>& b) {
  b.clear();
  const uint8_t sig[8] = {'M', 'P', 'F', 0, 0x4D, 0x4D, 0x00, 0x2A};
  b.insert(b.end(), sig, sig + 8);
  be32(b, 8);                      // index IFD offset (endianness value + this field)
  be16(b, 3);                      // three tags
  be16(b, 0xB000); be16(b, 7); be32(b, 4); b.insert(b.end(), {'0', '1', '0', '0'});   // version
  be16(b, 0xB001); be16(b, 4); be32(b, 1); be32(b, 2);                                // number of images
  be16(b, 0xB002); be16(b, 7); be32(b, 32);                                           // MP entries: 2 x 16 bytes ...
  be32(b, (uint32_t)(b.size() - 4 + 4 + 4));                                          // ... at this offset from the endianness field
  be32(b, 0);                      // no attribute IFD
  be32(b, 0x030000); be32(b, (uint32_t)primary_size); be32(b, (uint32_t)primary_offset); be16(b, 0); be16(b, 0);
  be32(b, 0x000000); be32(b, (uint32_t)secondary_size); be32(b, (uint32_t)secondary_offset); be16(b, 0); be16(b, 0);
}

// writeIccProfile(ULTRAHDR_TF_SRGB, gamut): 'desc', the three colorants, 'wtpt', three parametric curves, 'cprt'; the payload
// starts with the "ICC_PROFILE\0" identifier, chunk count 1, chunk index 1 (what JpegEncoderHelper writes into APP2)
bool icc_profile_srgb_transfer(int gamut, std::vector<uint8_t>& out) {
  auto ff = [](int v) { return (float)v * 1.52587890625e-5f; };
  const float mats[3][3][3] = {
      {{ff(0x6FA2), ff(0x6299), ff(0x24A0)}, {ff(0x38F5), ff(0xB785), ff(0x0F84)}, {ff(0x0390), ff(0x18DA), ff(0xB6CF)}},
      {{0.515102f, 0.291965f, 0.157153f}, {0.241182f, 0.692236f, 0.0665819f}, {-0.00104941f, 0.0418818f, 0.784378f}},
      {{0.673459f, 0.165661f, 0.125100f}, {0.279033f, 0.675338f, 0.0456288f}, {-0.00193139f, 0.0299794f, 0.797162f}}};
  const char* names[3] = {"sRGB", "Display P3", "Rec2020"};
  if (gamut < UHDR_HIP_CG_BT709 || gamut > UHDR_HIP_CG_BT2100) return false;
  auto fixed = [](float x) {   // float_round_to_fixed (icc.h:157-165)
    float v = (float)floor((double)x * 65536.0 + 0.5);
    v = v < 2147483520.0f ? v : 2147483520.0f;
    v = v > -2147483520.0f ? v : -2147483520.0f;
    return (uint32_t)(int32_t)v;
  };
  auto text_tag = [](const std::string& t) {
    std::vector<uint8_t> b;
    b.insert(b.end(), {'m', 'l', 'u', 'c'});
    be32(b, 0); be32(b, 1); be32(b, 12); b.insert(b.end(), {'e', 'n', 'U', 'S'}); be32(b, (uint32_t)(2 * t.size())); be32(b, 28);
    for (char c : t) { b.push_back(0); b.push_back((uint8_t)c); }
    b.resize((((2 * t.size() + 28) + 2) >> 2) << 2, 0);
    return b;
  };
  auto xyz_tag = [&](float x, float y, float z) {
    std::vector<uint8_t> b;
    b.insert(b.end(), {'X', 'Y', 'Z', ' '});
    be32(b, 0); be32(b, fixed(x)); be32(b, fixed(y)); be32(b, fixed(z));
    return b;
  };
  auto para_tag = [&]() {   // kSRGB_TransFun (gainmapmath.h:67-68), kGABCDEF_ParaCurveType
    const float fn[7] = {2.4f, (float)(1 / 1.055), (float)(0.055 / 1.055), (float)(1 / 12.92), 0.04045f, 0.0f, 0.0f};
    std::vector<uint8_t> b;
    b.insert(b.end(), {'p', 'a', 'r', 'a'});
    be32(b, 0); be16(b, 4); be16(b, 0);
    for (float v : fn) be32(b, fixed(v));
    return b;
  };
  std::vector<std::pair<const char*, std::vector<uint8_t>>> tags;
  const auto& m = mats[gamut];
  tags.emplace_back("desc", text_tag(std::string(names[gamut]) + " Gamut with sRGB Transfer"));
  tags.emplace_back("rXYZ", xyz_tag(m[0][0], m[1][0], m[2][0]));
  tags.emplace_back("gXYZ", xyz_tag(m[0][1], m[1][1], m[2][1]));
  tags.emplace_back("bXYZ", xyz_tag(m[0][2], m[1][2], m[2][2]));
  tags.emplace_back("wtpt", xyz_tag(0.9642f, 1.0000f, 0.8249f));
  tags.emplace_back("rTRC", para_tag());
  tags.emplace_back("gTRC", para_tag());
  tags.emplace_back("bTRC", para_tag());
  tags.emplace_back("cprt", text_tag("Google Inc. 2022"));
  size_t data = 0;
  for (auto& t : tags) data += t.second.size();
  const size_t table = 12 * tags.size(), profile = 132 + table + data;
  out.clear();
  const uint8_t id[14] = {'I', 'C', 'C', '_', 'P', 'R', 'O', 'F', 'I', 'L', 'E', 0, 1, 1};
  out.insert(out.end(), id, id + 14);
  be32(out, (uint32_t)profile); be32(out, 0); be32(out, 0x04300000);
  out.insert(out.end(), {'m', 'n', 't', 'r', 'R', 'G', 'B', ' ', 'X', 'Y', 'Z', ' '});
  out.insert(out.end(), 12, 0);                                   // creation date
  out.insert(out.end(), {'a', 'c', 's', 'p'});
  out.insert(out.end(), 4 + 4 + 4 + 4 + 8, 0);                    // platform, flags, manufacturer, model, attributes
  be32(out, 1);                                                   // relative colorimetric
  be32(out, fixed(0.9642f)); be32(out, fixed(1.0000f)); be32(out, fixed(0.8249f));
  out.insert(out.end(), 4 + 16 + 28, 0);                          // creator, profile id, reserved
  be32(out, (uint32_t)tags.size());
  uint32_t off = (uint32_t)(132 + table);
  for (auto& t : tags) {
    out.insert(out.end(), t.first, t.first + 4);
    be32(out, off); be32(out, (uint32_t)t.second.size());
    off += (uint32_t)t.second.size();
  }
  for (auto& t : tags) out.insert(out.end(), t.second.begin(), t.second.end());
  return true;
}

// JpegDecoderHelper::extractEXIF (jpegdecoderhelper.cpp:146-188) as appendGainMap uses it: walk the header up to SOS; the position
// is accumulated over the APP0 / APP1 segments ONLY (the markers that call saves), so an EXIF segment that follows any other kind
// of segment gets the reference's (wrong) position -- kept, it decides which bytes copyJpegWithoutExif drops.
// returns false where jpeg_read_header would fail (no SOI, truncated segment, no SOF before SOS, EOI before SOS)
static bool extract_exif(const uint8_t* d, size_t n, long* exif_pos, const uint8_t** exif, size_t* exif_len) {
  *exif_pos = -1; *exif = nullptr; *exif_len = 0;
  if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) return false;
  size_t p = 2, pos = 2;
  bool sof = false;
  while (true) {
    while (p < n && d[p] != 0xFF) ++p;            // next_marker: skip garbage, then fill bytes
    while (p < n && d[p] == 0xFF) ++p;
    if (p >= n) return false;
    const unsigned m = d[p++];
    if (m == 0x00 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
    if (m == 0xD9) return false;                   // EOI before SOS: JERR_NO_IMAGE
    if (p + 2 > n) return false;
    const size_t len = ((size_t)d[p] << 8) | d[p + 1];
    if (len < 2 || p + len > n) return false;
    if (m == 0xDA) return sof;                     // SOS: header complete
    if (m == 0xC0 || m == 0xC1 || m == 0xC2) sof = true;
    if ((m == 0xE0 || m == 0xE1) && *exif_pos < 0) {
      const size_t dl = len - 2;
      pos += 4 + dl;
      static const uint8_t kExif[6] = {'E', 'x', 'i', 'f', 0, 0};
      if (m == 0xE1 && dl > sizeof(kExif) && memcmp(d + p + 2, kExif, sizeof(kExif)) == 0) {
        *exif = d + p + 2; *exif_len = dl; *exif_pos = (long)(pos - dl);
      }
    }
    p += len;
  }
}

// the packet scan of JpegDecoderHelper::decode (jpegdecoderhelper.cpp:221-249): among the APP1 / APP2 segments before SOS, the first
// XMP, the first EXIF and the first ICC packet, tested in that order per segment ("else if").  Offsets are relative to jpg.
void first_packets(const uint8_t* d, size_t n, size_t* xmp_off, size_t* xmp_len, size_t* exif_off, size_t* exif_len, size_t* icc_off, size_t* icc_len) {
  static const char kXmp[] = "http://ns.adobe.com/xap/1.0/";
  static const uint8_t kExif[6] = {'E', 'x', 'i', 'f', 0, 0};
  static const char kIcc[] = "ICC_PROFILE";
  *xmp_len = *exif_len = *icc_len = 0;
  *xmp_off = *exif_off = *icc_off = 0;
  size_t p = 2;
  while (p + 4 <= n && !(*xmp_len && *exif_len && *icc_len)) {
    while (p < n && d[p] != 0xFF) ++p;
    while (p < n && d[p] == 0xFF) ++p;
    if (p >= n) return;
    const unsigned m = d[p++];
    if (m == 0x00 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
    if (m == 0xD9 || m == 0xDA || p + 2 > n) return;
    const size_t len = ((size_t)d[p] << 8) | d[p + 1];
    if (len < 2 || p + len > n) return;
    const size_t dl = len - 2;
    const uint8_t* data = d + p + 2;
    if (m == 0xE1 || m == 0xE2) {
      if (!*xmp_len && dl > sizeof(kXmp) && memcmp(data, kXmp, sizeof(kXmp)) == 0) { *xmp_off = (size_t)(data - d); *xmp_len = dl; }
      else if (!*exif_len && dl > sizeof(kExif) && memcmp(data, kExif, sizeof(kExif)) == 0) { *exif_off = (size_t)(data - d); *exif_len = dl; }
      else if (!*icc_len && dl > sizeof(kIcc) && memcmp(data, kIcc, sizeof(kIcc)) == 0) { *icc_off = (size_t)(data - d); *icc_len = dl; }
    }
    p += len;
  }
}

bool first_xmp(const uint8_t* jpg, size_t n, const uint8_t** payload, size_t* payload_len) {
  size_t a, b, c, d, e, f;
  first_packets(jpg, n, &a, &b, &c, &d, &e, &f);
  *payload = jpg + a; *payload_len = b;
  return b != 0;
}
bool first_icc(const uint8_t* jpg, size_t n, const uint8_t** payload, size_t* payload_len) {
  size_t a, b, c, d, e, f;
  first_packets(jpg, n, &a, &b, &c, &d, &e, &f);
  *payload = jpg + e; *payload_len = f;
  return f != 0;
}

// image_width / image_height as jpeg_read_header leaves them: the first SOFn segment
bool dimensions(const uint8_t* d, size_t n, int* w, int* h) {
  size_t p = 2;
  while (p + 4 <= n) {
    while (p < n && d[p] != 0xFF) ++p;
    while (p < n && d[p] == 0xFF) ++p;
    if (p >= n) return false;
    const unsigned m = d[p++];
    if (m == 0x00 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
    if (m == 0xD9 || m == 0xDA || p + 2 > n) return false;
    const size_t len = ((size_t)d[p] << 8) | d[p + 1];
    if (len < 2 || p + len > n) return false;
    if (m >= 0xC0 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
      if (len < 7) return false;
      *h = (d[p + 3] << 8) | d[p + 4];
      *w = (d[p + 5] << 8) | d[p + 6];
      return true;
    }
    p += len;
  }
  return false;
}

bool has_valid_header(const uint8_t* jpg, size_t n) {
  long pos; const uint8_t* e; size_t el;
  return extract_exif(jpg, n, &pos, &e, &el);
}

// appendGainMap (jpegr.cpp:951-1130): 0 ok, else the reference's status.  The file is written straight into the caller's buffer
// (the two compressed streams are megabytes: one copy each, no intermediate container); *size is set whenever the size is known,
// ERROR_INSUFFICIENT_RESOURCE when cap is smaller (Write() running past maxLength, jpegr.cpp:46-61).
int append_gainmap_to(const uint8_t* primary_in, size_t n1, const uint8_t* gainmap, size_t n2, const uint8_t* exif, size_t exif_len,
                      const uint8_t* icc, size_t icc_len, const uhdr_hip_metadata_t& md, uint8_t* dst, size_t cap, size_t* size) {
  if (strncmp(md.version, "1.0", sizeof(md.version)) != 0) return UHDR_HIP_ERROR_BAD_METADATA;                  // :961-964
  if (md.maxContentBoost < md.minContentBoost) return UHDR_HIP_ERROR_BAD_METADATA;
  if (md.hdrCapacityMax < md.hdrCapacityMin || md.hdrCapacityMin < 1.0f) return UHDR_HIP_ERROR_BAD_METADATA;
  if (md.offsetSdr < 0.0f || md.offsetHdr < 0.0f) return UHDR_HIP_ERROR_BAD_METADATA;
  if (md.gamma <= 0.0f) return UHDR_HIP_ERROR_BAD_METADATA;
  if (n2 < 2) return UHDR_HIP_ERROR_BAD_PTR;
  static const char kNs[] = "http://ns.adobe.com/xap/1.0/";   // sizeof counts the terminator, as nameSpaceLength does
  const std::string xs = xmp_secondary(md);
  const int xs_len = 2 + (int)sizeof(kNs) + (int)xs.size();
  const int secondary_size = 2 + xs_len + (int)n2;
  const std::string xp = xmp_primary(secondary_size, md.version);
  const int xp_len = 2 + (int)sizeof(kNs) + (int)xp.size();

  // :1003-1033: an EXIF segment inside the primary JPEG moves in front of the XMP segment
  long exif_pos;
  const uint8_t* in_exif;
  size_t in_exif_len;
  if (!extract_exif(primary_in, n1, &exif_pos, &in_exif, &in_exif_len)) return UHDR_HIP_ERROR_DECODE_ERROR;
  // the primary image without its SOI, and without the EXIF segment if it has one (copyJpegWithoutExif :63-73): one or two pieces
  const uint8_t* piece[2] = {primary_in + 2, nullptr};
  size_t piece_len[2] = {n1 - 2, 0};
  if (exif_pos >= 0) {
    if (exif != nullptr) return UHDR_HIP_ERROR_MULTIPLE_EXIFS_RECEIVED;
    if ((size_t)exif_pos + in_exif_len > n1 || exif_pos < 4) return UHDR_HIP_ERROR_DECODE_ERROR;   // the reference would read out of bounds
    if (exif_pos - 4 < 2) return UHDR_HIP_ERROR_DECODE_ERROR;
    piece_len[0] = (size_t)exif_pos - 4 - 2;
    piece[1] = primary_in + exif_pos + in_exif_len; piece_len[1] = n1 - (size_t)exif_pos - in_exif_len;
    n1 = 2 + piece_len[0] + piece_len[1];
    exif = in_exif; exif_len = in_exif_len;
  }
  std::vector<uint8_t> head, mid;
  auto segment = [](std::vector<uint8_t>& o, unsigned marker, size_t payload) { o.push_back(0xFF); o.push_back((uint8_t)marker); be16(o, (unsigned)(payload + 2)); };
  head.reserve(exif_len + icc_len + xp.size() + 256);
  head.push_back(0xFF); head.push_back(0xD8);
  if (exif != nullptr) { segment(head, 0xE1, exif_len); head.insert(head.end(), exif, exif + exif_len); }
  segment(head, 0xE1, (size_t)xp_len - 2);
  head.insert(head.end(), kNs, kNs + sizeof(kNs));
  head.insert(head.end(), xp.begin(), xp.end());
  if (icc != nullptr && icc_len > 0) { segment(head, 0xE2, icc_len); head.insert(head.end(), icc, icc + icc_len); }
  {
    const int pos = (int)head.size(), length = 2 + 86;
    const int primary_size = pos + length + (int)n1;
    std::vector<uint8_t> mpf;
    mpf_segment(primary_size, 0, secondary_size, primary_size - pos - 8, mpf);
    segment(head, 0xE2, 86);
    head.insert(head.end(), mpf.begin(), mpf.end());
  }
  mid.push_back(0xFF); mid.push_back(0xD8);
  segment(mid, 0xE1, (size_t)xs_len - 2);
  mid.insert(mid.end(), kNs, kNs + sizeof(kNs));
  mid.insert(mid.end(), xs.begin(), xs.end());

  const size_t total = head.size() + piece_len[0] + piece_len[1] + mid.size() + (n2 - 2);
  *size = total;
  if (dst == nullptr || cap < total) return UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE;
  uint8_t* w = dst;
  memcpy(w, head.data(), head.size()); w += head.size();
  memcpy(w, piece[0], piece_len[0]); w += piece_len[0];
  if (piece_len[1]) { memcpy(w, piece[1], piece_len[1]); w += piece_len[1]; }
  memcpy(w, mid.data(), mid.size()); w += mid.size();
  memcpy(w, gainmap + 2, n2 - 2);
  return UHDR_HIP_NO_ERROR;
}

// the same into a byte vector (tests, fuzzing)
int append_gainmap(const uint8_t* primary, size_t n1, const uint8_t* gainmap, size_t n2, const uint8_t* exif, size_t exif_len,
                   const uint8_t* icc, size_t icc_len, const uhdr_hip_metadata_t& md, std::vector<uint8_t>& out) {
  size_t total = 0;
  out.clear();
  int rc = append_gainmap_to(primary, n1, gainmap, n2, exif, exif_len, icc, icc_len, md, nullptr, 0, &total);
  if (rc != UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE) return rc;
  out.resize(total);
  return append_gainmap_to(primary, n1, gainmap, n2, exif, exif_len, icc, icc_len, md, out.data(), out.size(), &total);
}

}  // namespace jpegr
}  // namespace uhdr
