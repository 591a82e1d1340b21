// uhdr_jpegr.h -- JPEG/R container bookkeeping (uhdr_jpegr.hip)
#pragma once
#include <stddef.h>
#include <stdint.h>

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
}  // namespace jpegr
}  // namespace uhdr
