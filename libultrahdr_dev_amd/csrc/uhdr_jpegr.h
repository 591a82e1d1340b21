// uhdr_jpegr.h -- JPEG/R container bookkeeping (uhdr_jpegr.cpp)
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <string>
#include <vector>

struct uhdr_hip_metadata;

namespace uhdr {
namespace jpegr {
struct Range { size_t begin, len; };
int find_images(const uint8_t* data, size_t n, Range out[2]);
// first APPn segment (before SOS) of `marker` whose payload starts with prefix; payload excludes the 2 length bytes
bool find_app_segment(const uint8_t* jpg, size_t n, unsigned marker, const char* prefix, size_t prefix_len, const uint8_t** payload,
                      size_t* payload_len);
bool metadata_from_xmp(const uint8_t* payload, size_t len, uhdr_hip_metadata* md);
int gamut_from_icc(const uint8_t* payload, size_t len);
std::string xmp_primary(int secondary_image_length, const char* version);
std::string xmp_secondary(const uhdr_hip_metadata& md);
void mpf_segment(int primary_size, int primary_offset, int secondary_size, int secondary_offset, std::vector<uint8_t>& out);
bool icc_profile_srgb_transfer(int gamut, std::vector<uint8_t>& out);
// exif / icc: payloads of the APP1 / APP2 segments to add (nullptr = none)
int append_gainmap(const uint8_t* primary, size_t n1, const uint8_t* gainmap, size_t n2, const uint8_t* exif, size_t exif_len,
                   const uint8_t* icc, size_t icc_len, const uhdr_hip_metadata& md, std::vector<uint8_t>& out);
// the XMP / ICC packet the reference's decoder would hand on: the first one of first_packets' scan (APP1 or APP2, fill bytes allowed)
bool first_xmp(const uint8_t* jpg, size_t n, const uint8_t** payload, size_t* payload_len);
bool first_icc(const uint8_t* jpg, size_t n, const uint8_t** payload, size_t* payload_len);
void first_packets(const uint8_t* jpg, size_t n, size_t* xmp_off, size_t* xmp_len, size_t* exif_off, size_t* exif_len, size_t* icc_off,
                   size_t* icc_len);
// ... written straight into dst[0, cap); *size receives the file size (ERROR_INSUFFICIENT_RESOURCE when cap is smaller)
int append_gainmap_to(const uint8_t* primary, size_t n1, const uint8_t* gainmap, size_t n2, const uint8_t* exif, size_t exif_len,
                      const uint8_t* icc, size_t icc_len, const uhdr_hip_metadata& md, uint8_t* dst, size_t cap, size_t* size);
bool dimensions(const uint8_t* jpg, size_t n, int* w, int* h);
bool has_valid_header(const uint8_t* jpg, size_t n);   // JpegDecoderHelper::getCompressedImageParameters succeeding
}  // namespace jpegr
}  // namespace uhdr
