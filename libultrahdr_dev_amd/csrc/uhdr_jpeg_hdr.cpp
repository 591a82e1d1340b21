// uhdr_jpeg_hdr.cpp -- the host-side JPEG header parser of the device decoder (baseline sequential, 8 bit, Huffman, 4:2:0 or
// single plane, with or without restart intervals).  Everything it returns sizes or indexes device buffers, so it sees the untrusted file first;
// plain C++ on purpose: tests/cpp/fuzz_host_parsers.cpp builds it with AddressSanitizer / UBSan on the CPU.
// Follows what jpeg_read_header + the checks of JpegDecoderHelper::decode accept (lib/src/jpegdecoderhelper.cpp:190-300).
#include <cstring>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#if defined(__x86_64__) && defined(__GNUC__)
#include <immintrin.h>
#define UHDR_JD_HAVE_AVX2_WALK 1
#endif

#include "uhdr_jpeg.h"

#include <new>

namespace uhdr {
namespace jpeg {

static unsigned rd16(const uint8_t* p) { return ((unsigned)p[0] << 8) | p[1]; }

// The walk over the entropy-coded segment (megabytes): every 0xFF is classified by the byte behind it -- 0x00: a stuffed zero (counted),
// 0xFF: a fill byte (skipped), anything else: a marker.  A quality-95 scan holds a 0xFF every ~250 bytes, and a branch per hit
// (mispredicted: the positions are random) costs more than the search, so 16 bytes at a time are classified without one and the
// zeros counted in byte lanes; only a real marker (RSTn, or the one that ends the segment) leaves the loop.  Returns the position of
// the next marker's 0xFF at or behind `e`, or of the first byte the loop did not look at, with `stuffed` advanced up to there.
#if defined(UHDR_JD_HAVE_AVX2_WALK)
// the same walk, 32 bytes at a time, where the processor has AVX2 (decided at run time)
__attribute__((target("avx2"))) static size_t skip_to_marker_avx2(const uint8_t* p, size_t e, size_t n, uint32_t* stuffed) {
  const __m256i ff = _mm256_set1_epi8((char)0xFF), zero = _mm256_setzero_si256();
  __m256i acc = zero;
  unsigned pending = 0;
  auto flush = [&]() __attribute__((target("avx2"))) {
    const __m256i sad = _mm256_sad_epu8(acc, zero);
    const __m128i lo = _mm256_castsi256_si128(sad), hi = _mm256_extracti128_si256(sad, 1);
    const __m128i sum = _mm_add_epi64(lo, hi);
    *stuffed += (uint32_t)_mm_cvtsi128_si32(sum) + (uint32_t)_mm_cvtsi128_si32(_mm_srli_si128(sum, 8));
    acc = zero;
    pending = 0;
  };
  while (e + 33 <= n) {
    const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(p + e));
    const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(p + e + 1));
    const __m256i is_ff = _mm256_cmpeq_epi8(a, ff);
    const __m256i z = _mm256_cmpeq_epi8(b, zero);
    const __m256i st = _mm256_and_si256(is_ff, z);
    const unsigned mk = (unsigned)_mm256_movemask_epi8(_mm256_andnot_si256(_mm256_or_si256(z, _mm256_cmpeq_epi8(b, ff)), is_ff));
    if (mk != 0u) {
      const unsigned m = (unsigned)__builtin_ctz(mk);
      flush();
      *stuffed += (uint32_t)__builtin_popcount((unsigned)_mm256_movemask_epi8(st) & (uint32_t)((1ull << m) - 1ull));
      return e + m;
    }
    acc = _mm256_sub_epi8(acc, st);   // a set lane is 0xFF = -1
    if (++pending == 255u) flush();
    e += 32;
  }
  flush();
  return e;
}
#endif

static size_t skip_to_marker(const uint8_t* p, size_t e, size_t n, uint32_t* stuffed) {
#if defined(UHDR_JD_HAVE_AVX2_WALK)
  static const bool avx2 = __builtin_cpu_supports("avx2");
  if (avx2) e = skip_to_marker_avx2(p, e, n, stuffed);
  if (e + 1 < n && p[e] == 0xFF && p[e + 1] != 0x00 && p[e + 1] != 0xFF) return e;   // at a marker
#endif
#if defined(__SSE2__)
  const __m128i ff = _mm_set1_epi8((char)0xFF), zero = _mm_setzero_si128();
  __m128i acc = zero;
  unsigned pending = 0;
  auto flush = [&]() {
    const __m128i sad = _mm_sad_epu8(acc, zero);
    *stuffed += (uint32_t)_mm_cvtsi128_si32(sad) + (uint32_t)_mm_cvtsi128_si32(_mm_srli_si128(sad, 8));
    acc = zero;
    pending = 0;
  };
  while (e + 17 <= n) {
    const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i*>(p + e));
    const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i*>(p + e + 1));
    const __m128i is_ff = _mm_cmpeq_epi8(a, ff);
    const __m128i z = _mm_cmpeq_epi8(b, zero);
    const __m128i st = _mm_and_si128(is_ff, z);
    const int mk = _mm_movemask_epi8(_mm_andnot_si128(_mm_or_si128(z, _mm_cmpeq_epi8(b, ff)), is_ff));
    if (mk != 0) {
      const unsigned m = (unsigned)__builtin_ctz((unsigned)mk);
      flush();
      *stuffed += (uint32_t)__builtin_popcount((unsigned)_mm_movemask_epi8(st) & ((1u << m) - 1u));
      return e + m;
    }
    acc = _mm_sub_epi8(acc, st);   // a set lane is 0xFF = -1
    if (++pending == 255u) flush();
    e += 16;
  }
  flush();
#endif
  (void)stuffed;
  return e;
}

// position of the 0xFF of the first marker at or behind `e` that is neither a stuffed zero, a fill byte nor an RSTn; n if there is none
size_t skip_entropy_coded(const uint8_t* p, size_t e, size_t n) {
  uint32_t stuffed = 0;
  for (;;) {
    e = skip_to_marker(p, e, n, &stuffed);
    while (e + 1 < n && p[e] != 0xFF) ++e;
    if (e + 1 >= n) return n;
    const uint8_t b = p[e + 1];
    if (b == 0x00 || (b & 0xF8) == 0xD0) { e += 2; continue; }
    if (b == 0xFF) { e += 1; continue; }
    return e;
  }
}

static int parse_header_impl(const uint8_t* jpg, size_t n, DecInfo* info);
// 0 ok, -1 malformed, -2 outside what is supported, -3 out of memory (the vectors of DecInfo; a progressive frame's coefficients):
// the callers are extern "C" entry points, nothing may throw through them
int parse_header(const uint8_t* jpg, size_t n, DecInfo* info) {
  try {
    return parse_header_impl(jpg, n, info);
  } catch (const std::bad_alloc&) {
    return -3;
  }
}
static int parse_header_impl(const uint8_t* jpg, size_t n, DecInfo* info) {
  if (jpg == nullptr || n < 4 || jpg[0] != 0xFF || jpg[1] != 0xD8) return -1;
  *info = DecInfo();
  static const uint8_t nat[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                  41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                  30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
  (void)nat;
  uint16_t quant[4][64];   // zigzag order, as stored in the file
  bool have_q[4] = {false, false, false, false};
  HuffSpec huff[2][4];
  memset(huff, 0, sizeof(huff));
  int nc = 0, hs[3] = {0, 0, 0}, vs[3] = {0, 0, 0}, tq[3] = {0, 0, 0}, cid[3] = {0, 0, 0};
  size_t pos = 2;
  for (;;) {
    while (pos + 1 < n && jpg[pos] == 0xFF && jpg[pos + 1] == 0xFF) pos++;
    if (pos + 4 > n || jpg[pos] != 0xFF) return -1;
    const unsigned m = jpg[pos + 1];
    const size_t len = rd16(jpg + pos + 2);
    const uint8_t* seg = jpg + pos + 4;
    if (len < 2 || pos + 2 + len > n) return -1;
    if (m == 0xDB) {
      for (size_t o = 0; o + 1 <= len - 2;) {
        const int pq = seg[o] >> 4, id = seg[o] & 15;
        const size_t sz = pq ? 128 : 64;
        if (id > 3 || o + 1 + sz > len - 2) return -1;
        for (int i = 0; i < 64; ++i) quant[id][i] = pq ? (uint16_t)rd16(seg + o + 1 + 2 * i) : seg[o + 1 + i];
        have_q[id] = true;
        o += 1 + sz;
      }
    } else if (m == 0xC4) {
      for (size_t o = 0; o + 17 <= len - 2;) {
        const int cls = seg[o] >> 4, id = seg[o] & 15;
        int cnt = 0;
        for (int i = 0; i < 16; ++i) cnt += seg[o + 1 + i];
        if (cls > 1 || id > 3 || cnt > 256 || o + 17 + (size_t)cnt > len - 2) return -1;
        HuffSpec& h = huff[cls][id];
        memset(&h, 0, sizeof(h));
        uint32_t code = 0, p = 0;
        for (int l = 1; l <= 16; ++l) {   // T.81 Annex C
          h.first_code[l] = (uint16_t)code;
          h.first_val[l] = (uint16_t)p;
          h.count[l] = seg[o + l];
          if (code + h.count[l] > (1u << l)) return -1;   // more codes than l bits hold: not a prefix code (libjpeg: JERR_BAD_HUFF_TABLE)
          code += h.count[l];
          p += h.count[l];
          code <<= 1;
        }
        memcpy(h.vals, seg + o + 17, (size_t)cnt);
        h.present = 1;
        o += 17 + (size_t)cnt;
      }
    } else if (m == 0xC0 || m == 0xC1) {
      if (len < 8 || seg[0] != 8) return -2;
      info->h = (int)rd16(seg + 1); info->w = (int)rd16(seg + 3); nc = seg[5];
      if (nc != 1 && nc != 3) return -2;
      if (len < (size_t)(8 + 3 * nc)) return -1;
      for (int c = 0; c < nc; ++c) { cid[c] = seg[6 + 3 * c]; hs[c] = seg[7 + 3 * c] >> 4; vs[c] = seg[7 + 3 * c] & 15; tq[c] = seg[8 + 3 * c]; }
    } else if (m == 0xC2) {
      return decode_progressive(jpg, n, info);   // every scan on the host, the device takes over at the coefficients (uhdr_jpeg_prog.cpp)
    } else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
      return -2;   // lossless, arithmetic, hierarchical: libjpeg reads some of these, this decoder does not
    } else if (m == 0xDD) {
      if (len < 4) return -1;
      info->restart_interval = rd16(seg);
    } else if (m == 0xDA) {
      if (nc == 0 || len < (size_t)(6 + 2 * nc) || seg[0] != nc) return -2;   // (the length first: seg[0] of a 2-byte segment at the end of the file is past the buffer)
      info->gray = nc == 1;
      if (!info->gray && !(hs[0] == 2 && vs[0] == 2 && hs[1] == 1 && vs[1] == 1 && hs[2] == 1 && vs[2] == 1)) return -1;   // the reference fails too, jpegdecoderhelper.cpp:283-289
      if (info->w <= 0 || info->h <= 0) return -1;
      for (int c = 0; c < nc; ++c) {
        if (seg[1 + 2 * c] != cid[c]) return -2;
        const int td = seg[2 + 2 * c] >> 4, ta = seg[2 + 2 * c] & 15;
        if (td > 3 || ta > 3 || !huff[0][td].present || !huff[1][ta].present || tq[c] > 3 || !have_q[tq[c]]) return -1;
        memcpy(info->quant[c], quant[tq[c]], sizeof(quant[0]));
        if (c <= 1) {   // luma tables in slots 0 (DC) / 1 (AC), chroma in 2 / 3; Cr must share Cb's tables
          info->tables.huff[2 * c] = huff[0][td];
          info->tables.huff[2 * c + 1] = huff[1][ta];
          info->td[c] = td; info->ta[c] = ta;
        } else if (td != info->td[1] || ta != info->ta[1]) {
          return -2;
        }
      }
      info->scan_offset = pos + 2 + len;
      // the entropy-coded segment ends at the first marker that is neither a stuffed zero, a fill byte nor (in a file with restart
      // intervals) an RSTn; the same walk notes where every restart interval begins once those bytes are gone
      size_t e = info->scan_offset;
      uint32_t stuffed = 0, markers = 0;
      if (info->restart_interval != 0) info->interval_start.push_back(0u);
      for (;;) {
        e = skip_to_marker(jpg, e, n, &stuffed);
        while (e + 1 < n && jpg[e] != 0xFF) ++e;
        if (e + 1 >= n) return -1;
        const uint8_t b = jpg[e + 1];
        if (b == 0x00) { ++stuffed; e += 2; continue; }
        if (b == 0xFF) { e += 1; continue; }
        if ((b & 0xF8) == 0xD0) {
          if (info->restart_interval == 0) return -2;   // RSTn without DRI cannot happen in a valid file
          ++markers;
          e += 2;
          info->interval_start.push_back((uint32_t)(e - info->scan_offset - stuffed - 2u * markers));
          continue;
        }
        break;
      }
      info->scan_bytes = e - info->scan_offset;
      info->raw_bytes = (uint32_t)(info->scan_bytes - stuffed - 2u * markers);
      if (info->restart_interval != 0) {   // every interval but the last holds restart_interval MCUs: their number is fixed by the image size
        const uint64_t mcus = info->gray ? (uint64_t)((info->w + 7) / 8) * (uint64_t)((info->h + 7) / 8)
                                         : (uint64_t)((info->w + 15) / 16) * (uint64_t)((info->h + 15) / 16);
        if (info->interval_start.size() != (mcus + info->restart_interval - 1) / info->restart_interval) return -1;
      }
      return 0;
    } else if (m == 0xD9) {
      return -1;
    }
    pos += 2 + len;
  }
}

}  // namespace jpeg
}  // namespace uhdr
