// uhdr_jpeg.hip -- baseline JPEG compression of the path's outputs on the GPU (encode side).
//
// Drop-in for the reference's JpegEncoderHelper::compressImage (lib/src/jpegencoderhelper.cpp:39-283, declared
// lib/include/ultrahdr/jpegencoderhelper.h:43-60): YUV 4:2:0 planar or a single 8-bit plane (the gain map,
// jpegr.cpp:294-297) -> the byte stream libjpeg writes in raw-data mode with default tables, quality scaling with
// force_baseline and the ISLOW DCT.  The whole encoder runs on the device; the host only builds the ~600-byte header:
//
//   k_jpeg_fdct_quant_count   one thread per 8x8 block (in entropy-coding order, dummy edge blocks included): level shift,
//                       13-bit fixed-point FDCT, quantisation by rounded division -> 64 int16 in zigzag order; and, from the same
//                       registers, DC difference + run/size symbols -> the block's number of bits
//   k_jpeg_emit         one thread per block: its bit offset (the totals of the workgroups in front + a block scan), then its
//                       codes written MSB-first there (whole words stored,
//                       the two boundary words OR-ed atomically); the last block pads the final byte with ones
//   k_jpeg_stuff_count  one thread per 64 bytes of the packed stream: number of 0xFF bytes (and per workgroup)
//   k_jpeg_stuff_copy   one thread per 64 bytes: copy to the output with 0x00 stuffed after every 0xFF, EOI at the end
//
// Byte-exact against libjpeg for every size, stride regime and quality tested (tests/test_gpu_jpeg.py against
// oracle/jpeg_oracle.c, which tests/test_jpeg_oracle.py pins to the libjpeg builds of the image).
#include <hip/hip_runtime.h>
#include "uhdr_wave_scan.h"

#include <cstring>
#include <vector>

#include "uhdr_jpeg.h"

namespace uhdr {
namespace jpeg {

// ---- ITU-T T.81 Annex K tables ----------------------------------------------------------------------------------
static const uint8_t kStdLumQuant[64] = {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,
                                         14, 13, 16, 24, 40,  57,  69,  56,  14, 17, 22, 29, 51,  87,  80,  62,
                                         18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
                                         49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
static const uint8_t kStdChrQuant[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99,
                                         99, 99, 47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                                         99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
static const uint8_t kNatural[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                     41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                     30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
static const uint8_t kBitsDcLum[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
static const uint8_t kBitsDcChr[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
static const uint8_t kValsDc[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static const uint8_t kBitsAcLum[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
static const uint8_t kValsAcLum[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81,
    0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18,
    0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48,
    0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75,
    0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99,
    0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5,
    0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
static const uint8_t kBitsAcChr[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
static const uint8_t kValsAcChr[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08,
    0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25,
    0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47,
    0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74,
    0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97,
    0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba,
    0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4,
    0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

// Huffman codes of the four default tables: [0] DC lum, [1] AC lum, [2] DC chroma, [3] AC chroma.  Packed (size << 16) | code.
__constant__ uint32_t c_huff[4][256];

static void derive_codes(const uint8_t bits[16], const uint8_t* vals, uint32_t out[256]) {  // T.81 Annex C
  memset(out, 0, sizeof(uint32_t) * 256);
  uint32_t code = 0;
  int p = 0;
  for (int l = 1; l <= 16; ++l) {
    for (int i = 0; i < bits[l - 1]; ++i, ++p) out[vals[p]] = ((uint32_t)l << 16) | code++;
    code <<= 1;
  }
}

hipError_t upload_tables() {
  uint32_t t[4][256];
  derive_codes(kBitsDcLum, kValsDc, t[0]);
  derive_codes(kBitsAcLum, kValsAcLum, t[1]);
  derive_codes(kBitsDcChr, kValsDc, t[2]);
  derive_codes(kBitsAcChr, kValsAcChr, t[3]);
  return hipMemcpyToSymbol(HIP_SYMBOL(c_huff), t, sizeof(t));
}

// jpeg_set_quality(quality, TRUE): jpeg_quality_scaling + jpeg_add_quant_table with force_baseline
void quant_table(int quality, bool chroma, uint16_t out_natural[64]) {
  if (quality <= 0) quality = 1;
  if (quality > 100) quality = 100;
  const int scale = quality < 50 ? 5000 / quality : 200 - quality * 2;
  const uint8_t* base = chroma ? kStdChrQuant : kStdLumQuant;
  for (int i = 0; i < 64; ++i) {
    long t = ((long)base[i] * scale + 50L) / 100L;
    if (t <= 0L) t = 1L;
    if (t > 255L) t = 255L;
    out_natural[i] = (uint16_t)t;
  }
}

void zigzag_table(const uint16_t natural[64], uint16_t zz[64]) {
  for (int i = 0; i < 64; ++i) zz[i] = natural[kNatural[i]];
}

// ---- header (everything up to and including SOS), marker layout of libjpeg's jcmarker.c --------------------------------
namespace {
void put16(std::vector<uint8_t>& b, unsigned v) { b.push_back((uint8_t)(v >> 8)); b.push_back((uint8_t)v); }
void put_dqt(std::vector<uint8_t>& b, int idx, const uint16_t q[64]) {
  put16(b, 0xFFDB); put16(b, 67); b.push_back((uint8_t)idx);
  for (int i = 0; i < 64; ++i) b.push_back((uint8_t)q[kNatural[i]]);
}
void put_dht(std::vector<uint8_t>& b, int cls_idx, const uint8_t bits[16], const uint8_t* vals) {
  int n = 0;
  for (int i = 0; i < 16; ++i) n += bits[i];
  put16(b, 0xFFC4); put16(b, (unsigned)(19 + n)); b.push_back((uint8_t)cls_idx);
  b.insert(b.end(), bits, bits + 16);
  b.insert(b.end(), vals, vals + n);
}
}  // namespace

void build_header(int w, int h, bool gray, int quality, const void* icc, size_t icc_n, std::vector<uint8_t>& b) {
  uint16_t ql[64], qc[64];
  quant_table(quality, false, ql);
  quant_table(quality, true, qc);
  b.clear();
  put16(b, 0xFFD8);
  put16(b, 0xFFE0); put16(b, 16);
  const uint8_t jfif[14] = {'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0};  // 1.01, no units, 1:1, no thumbnail
  b.insert(b.end(), jfif, jfif + 14);
  if (icc != nullptr && icc_n > 0) {  // jpeg_write_marker(JPEG_APP0 + 2) right after jpeg_start_compress (:98-100)
    put16(b, 0xFFE2); put16(b, (unsigned)(icc_n + 2));
    b.insert(b.end(), static_cast<const uint8_t*>(icc), static_cast<const uint8_t*>(icc) + icc_n);
  }
  put_dqt(b, 0, ql);
  if (!gray) put_dqt(b, 1, qc);
  const int nc = gray ? 1 : 3;
  put16(b, 0xFFC0); put16(b, (unsigned)(8 + 3 * nc)); b.push_back(8); put16(b, (unsigned)h); put16(b, (unsigned)w);
  b.push_back((uint8_t)nc);
  for (int c = 0; c < nc; ++c) {
    b.push_back((uint8_t)(c + 1));
    b.push_back((uint8_t)((c == 0 && !gray) ? 0x22 : 0x11));
    b.push_back((uint8_t)(c == 0 ? 0 : 1));
  }
  put_dht(b, 0x00, kBitsDcLum, kValsDc);
  put_dht(b, 0x10, kBitsAcLum, kValsAcLum);
  if (!gray) {
    put_dht(b, 0x01, kBitsDcChr, kValsDc);
    put_dht(b, 0x11, kBitsAcChr, kValsAcChr);
  }
  put16(b, 0xFFDA); put16(b, (unsigned)(6 + 2 * nc)); b.push_back((uint8_t)nc);
  for (int c = 0; c < nc; ++c) { b.push_back((uint8_t)(c + 1)); b.push_back((uint8_t)(c == 0 ? 0x00 : 0x11)); }
  b.push_back(0); b.push_back(63); b.push_back(0);
}

// ---- block geometry -----------------------------------------------------------------------------------------------
// Entropy-coding order (libjpeg jccoefct.c): one plane: blocks in raster order; 4:2:0: per MCU (16x16 pixels) Y00 Y01 Y10 Y11
// Cb Cr.  Blocks past the component's size in blocks are dummies: zero AC, DC of the block before them in the MCU.
struct BlockRef {
  int comp;        // 0 Y, 1 Cb, 2 Cr
  int br, bc;      // block row / column inside the component (of the source block for a dummy)
  bool dummy;
};
__device__ __forceinline__ BlockRef locate(const Job& j, uint32_t i) {
  BlockRef b;
  if (j.gray) {
    b.comp = 0; b.br = (int)(i / j.ybw); b.bc = (int)(i - (uint32_t)b.br * j.ybw); b.dummy = false;
    return b;
  }
  const uint32_t mcu = i / 6u, k = i - mcu * 6u;
  const int mr = (int)(mcu / j.mcus_x), mc = (int)(mcu - (uint32_t)mr * j.mcus_x);
  if (k >= 4u) { b.comp = (int)k - 3; b.br = mr; b.bc = mc; b.dummy = false; return b; }
  b.comp = 0;
  int kk = (int)k;
  b.br = 2 * mr + (kk >> 1); b.bc = 2 * mc + (kk & 1);
  b.dummy = !(b.br < (int)j.ybh && b.bc < (int)j.ybw);
  while (!(b.br < (int)j.ybh && b.bc < (int)j.ybw)) {  // a dummy inherits the DC of the block before it; block 0 is always real
    --kk;
    b.br = 2 * mr + (kk >> 1); b.bc = 2 * mc + (kk & 1);
  }
  return b;
}
// index of the previous block of the same component (DC prediction), or UINT32_MAX for the first one
__device__ __forceinline__ uint32_t dc_predecessor(const Job& j, uint32_t i) {
  if (j.gray) return i == 0u ? 0xFFFFFFFFu : i - 1u;
  const uint32_t mcu = i / 6u, k = i - mcu * 6u;
  if (k >= 4u) return mcu == 0u ? 0xFFFFFFFFu : i - 6u;
  if (k != 0u) return i - 1u;
  return mcu == 0u ? 0xFFFFFFFFu : i - 3u;  // Y11 of the previous MCU
}

// sample with the helper's padding rules (jpegencoderhelper.cpp:147-222, :239-278): rows past the height come from an
// all-zero row; columns past the width are zero when the row was copied into a zero-padded buffer (stride < aligned
// width) and the caller's bytes otherwise
__device__ __forceinline__ int sample(const Plane& p, int r, int c) {
  if (r >= p.h) return 0;
  if (c >= p.w && p.pad_cols) return 0;
  return p.p[(size_t)r * p.stride + c];
}

#define UJ_CONST_BITS 13
#define UJ_PASS1_BITS 2
__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }
// one 8-point pass of libjpeg's jfdctint.c ("islow"): in/out are 8 values with the given register stride
template <int PASS>
__device__ __forceinline__ void fdct8(int (&d)[64], int base, int stride) {
  const int d0 = d[base], d1 = d[base + stride], d2 = d[base + 2 * stride], d3 = d[base + 3 * stride];
  const int d4 = d[base + 4 * stride], d5 = d[base + 5 * stride], d6 = d[base + 6 * stride], d7 = d[base + 7 * stride];
  int tmp0 = d0 + d7, tmp7 = d0 - d7, tmp1 = d1 + d6, tmp6 = d1 - d6, tmp2 = d2 + d5, tmp5 = d2 - d5, tmp3 = d3 + d4, tmp4 = d3 - d4;
  const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
  constexpr int sh = PASS == 0 ? UJ_CONST_BITS - UJ_PASS1_BITS : UJ_CONST_BITS + UJ_PASS1_BITS;
  if (PASS == 0) {
    d[base] = (tmp10 + tmp11) << UJ_PASS1_BITS;
    d[base + 4 * stride] = (tmp10 - tmp11) << UJ_PASS1_BITS;
  } else {
    d[base] = descale(tmp10 + tmp11, UJ_PASS1_BITS);
    d[base + 4 * stride] = descale(tmp10 - tmp11, UJ_PASS1_BITS);
  }
  int z1 = (tmp12 + tmp13) * 4433;
  d[base + 2 * stride] = descale(z1 + tmp13 * 6270, sh);
  d[base + 6 * stride] = descale(z1 + tmp12 * (-15137), sh);
  z1 = tmp4 + tmp7;
  int z2 = tmp5 + tmp6, z3 = tmp4 + tmp6, z4 = tmp5 + tmp7;
  const int z5 = (z3 + z4) * 9633;
  tmp4 *= 2446; tmp5 *= 16819; tmp6 *= 25172; tmp7 *= 12299;
  z1 *= -7373; z2 *= -20995; z3 *= -16069; z4 *= -3196;
  z3 += z5; z4 += z5;
  d[base + 7 * stride] = descale(tmp4 + z1 + z3, sh);
  d[base + 5 * stride] = descale(tmp5 + z2 + z4, sh);
  d[base + 3 * stride] = descale(tmp6 + z2 + z3, sh);
  d[base + stride] = descale(tmp7 + z1 + z4, sh);
}
// jcdctmgr.c: the 8x-scaled coefficient divided by (quant << 3), rounded half away from zero.  The division is a 32-bit
// multiply-high by m = floor(2^32 / qval) + 1: exact for 0 <= t < 2^16 because t * qval < 2^27 (|coefficient| <= 2^14,
// qval <= 2040); an integer division costs ~35 instructions, and there are 64 per block.
__device__ __forceinline__ int quantise(int v, int q, uint32_t m) {
  const int qval = q << 3;
  int t = v < 0 ? -v : v;
  t += qval >> 1;
  t = (int)__umulhi((uint32_t)t, m);
  return v < 0 ? -t : t;
}

// the 64 level-shifted samples of block b (of its source block for a dummy), with the helper's padding rules
__device__ __forceinline__ void load_samples(const Plane& p, const BlockRef& b, int (&d)[64]) {
  const int r0 = b.br * 8, c0 = b.bc * 8;
  const bool inside = r0 + 8 <= p.h && (c0 + 8 <= p.w || !p.pad_cols);
  if (inside && p.aligned4) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const uint32_t* row = reinterpret_cast<const uint32_t*>(p.p + (size_t)(r0 + r) * p.stride + c0);
      const uint32_t a = row[0], bq = row[1];
      d[r * 8 + 0] = (int)(a & 0xff) - 128; d[r * 8 + 1] = (int)((a >> 8) & 0xff) - 128;
      d[r * 8 + 2] = (int)((a >> 16) & 0xff) - 128; d[r * 8 + 3] = (int)(a >> 24) - 128;
      d[r * 8 + 4] = (int)(bq & 0xff) - 128; d[r * 8 + 5] = (int)((bq >> 8) & 0xff) - 128;
      d[r * 8 + 6] = (int)((bq >> 16) & 0xff) - 128; d[r * 8 + 7] = (int)(bq >> 24) - 128;
    }
  } else {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int c = 0; c < 8; ++c) d[r * 8 + c] = sample(p, r0 + r, c0 + c) - 128;
  }
}

// ---- entropy coding ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int nbits_of(int a) { return a == 0 ? 0 : 32 - __builtin_clz((unsigned)a); }

// walks one block's symbols (jchuff.c encode_one_block); SINK::put(code, size) receives MSB-first bit strings.
// c: the block's 64 quantised coefficients in zigzag order (registers: every index is a compile-time constant)
template <typename SINK>
__device__ __forceinline__ void walk_coefs(const int16_t (&c)[64], int pred, int tbl_dc, int tbl_ac, SINK& s) {
  int temp = (int)c[0] - pred, temp2 = temp;
  if (temp < 0) { temp = -temp; temp2--; }
  int nb = nbits_of(temp);
  uint32_t h = c_huff[tbl_dc][nb];
  s.put(h & 0xffffu, (int)(h >> 16));
  if (nb) s.put((uint32_t)temp2 & ((1u << nb) - 1u), nb);
  int r = 0;
#pragma unroll
  for (int k = 1; k < 64; ++k) {
    temp = c[k];
    if (temp == 0) { r++; continue; }
    while (r > 15) { h = c_huff[tbl_ac][0xF0]; s.put(h & 0xffffu, (int)(h >> 16)); r -= 16; }
    temp2 = temp;
    if (temp < 0) { temp = -temp; temp2--; }
    nb = nbits_of(temp);
    h = c_huff[tbl_ac][(r << 4) + nb];
    s.put(((h & 0xffffu) << nb) | ((uint32_t)temp2 & ((1u << nb) - 1u)), (int)(h >> 16) + nb);   // <= 16 + 10 bits
    r = 0;
  }
  if (r > 0) { h = c_huff[tbl_ac][0]; s.put(h & 0xffffu, (int)(h >> 16)); }
}
template <typename SINK>
__device__ __forceinline__ void walk_block(const int16_t* __restrict__ zz, int pred, int tbl_dc, int tbl_ac, SINK& s) {
  int16_t c[64];
#pragma unroll
  for (int k = 0; k < 8; ++k) {   // 128 B per block
    const uint4 v = reinterpret_cast<const uint4*>(zz)[k];
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int m = 0; m < 4; ++m) { c[8 * k + 2 * m] = (int16_t)(w[m] & 0xffffu); c[8 * k + 2 * m + 1] = (int16_t)(w[m] >> 16); }
  }
  walk_coefs(c, pred, tbl_dc, tbl_ac, s);
}

struct CountSink {
  uint32_t bits = 0;
  __device__ __forceinline__ void put(uint32_t, int size) { bits += (uint32_t)size; }
};

__device__ __forceinline__ int pred_of(const Job& j, uint32_t i) {
  const uint32_t p = dc_predecessor(j, i);
  return p == 0xFFFFFFFFu ? 0 : (int)j.coef[(size_t)p * 64u];
}
__device__ __forceinline__ int comp_of(const Job& j, uint32_t i) { return j.gray ? 0 : ((i % 6u) < 4u ? 0 : 1); }

// One thread per 8x8 block: forward DCT, quantisation, the block's 64 coefficients to memory (zigzag order) -- and, while they are
// in registers, the number of bits its Huffman code takes (the prefix sum of those is where k_jpeg_emit writes).  The DC code needs
// the quantised DC of the component's previous block, which another thread computes: the islow DC is the plain sum of the 64
// level-shifted samples (pass 1 scales the row sums by 4, pass 2 descales by 4: exact), so it is had from that block's 64 bytes.
__device__ __forceinline__ uint32_t fdct_quant_count_block(const Job& j, const uint32_t i);
__global__ void __launch_bounds__(128) k_jpeg_fdct_quant_count(const Job j) {
  const uint32_t i = blockIdx.x * 128u + threadIdx.x;
  __shared__ uint32_t s_part[2];
  uint32_t my_bits = 0;
  if (i < j.nblk) my_bits = fdct_quant_count_block(j, i);
  // bits_blk[g]: the bits of workgroup g's 128 blocks.  k_jpeg_emit needs the bit offset of every block: inside a workgroup a block
  // scan, across workgroups the sum of the totals in front (a 4K frame has 1519), formed by every workgroup of the emit for itself
  const uint32_t total = block_sum<128>(my_bits, s_part);
  if (threadIdx.x == 0u) j.bits_blk[blockIdx.x] = total;
}

// the work of one thread of k_jpeg_fdct_quant_count: block i's coefficients to memory, its number of bits returned
__device__ __forceinline__ uint32_t fdct_quant_count_block(const Job& j, const uint32_t i) {
  const BlockRef b = locate(j, i);
  const Plane& p = j.plane[b.comp];
  const uint16_t* q = b.comp == 0 ? j.q_lum : j.q_chr;   // zigzag order
  const uint32_t* qm = b.comp == 0 ? j.m_lum : j.m_chr;
  int16_t* out = j.coef + (size_t)i * 64u;
  int d[64];
  load_samples(p, b, d);
  int16_t c[64];
  if (b.dummy) {  // zero AC, DC of the source block
    int s = 0;
#pragma unroll
    for (int k = 0; k < 64; ++k) s += d[k];
#pragma unroll
    for (int k = 1; k < 64; ++k) c[k] = 0;
    c[0] = (int16_t)quantise(s, q[0], qm[0]);
  } else {
#pragma unroll
    for (int r = 0; r < 8; ++r) fdct8<0>(d, r * 8, 1);
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) fdct8<1>(d, cc, 8);
    constexpr uint8_t nat[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                 41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
#pragma unroll
    for (int k = 0; k < 64; ++k) c[k] = (int16_t)quantise(d[nat[k]], q[k], qm[k]);
  }
  uint4* o = reinterpret_cast<uint4*>(out);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    uint32_t w[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) w[m] = (uint32_t)(uint16_t)c[8 * k + 2 * m] | ((uint32_t)(uint16_t)c[8 * k + 2 * m + 1] << 16);
    o[k] = make_uint4(w[0], w[1], w[2], w[3]);
  }
  int pred = 0;
  const uint32_t pi = dc_predecessor(j, i);
  if (pi != 0xFFFFFFFFu) {
    int e[64];
    load_samples(p, locate(j, pi), e);   // (the same component, so the same plane and quantiser)
    int s = 0;
#pragma unroll
    for (int k = 0; k < 64; ++k) s += e[k];
    pred = (int)(int16_t)quantise(s, q[0], qm[0]);
  }
  const int chroma = comp_of(j, i);
  CountSink cs;
  walk_coefs(c, pred, 2 * chroma, 2 * chroma + 1, cs);
  j.bits[i] = cs.bits;
  return cs.bits;
}

// MSB-first writer into a zero-initialised word buffer that other threads write next to: whole words are stored, the first
// and the last (partial) word are OR-ed atomically.  Words are kept big-endian in memory so that the buffer is the byte stream.
struct EmitSink {
  uint32_t* words;
  uint64_t acc;
  int nacc;       // valid low bits of acc (includes the leading alignment zeros at the start)
  uint32_t wi;    // next word index
  bool first;
  __device__ __forceinline__ void flush_word(uint32_t w, bool atomic) {
    const uint32_t be = __builtin_bswap32(w);
    if (atomic) atomicOr(&words[wi], be);
    else words[wi] = be;
    ++wi;
  }
  __device__ __forceinline__ void put(uint32_t code, int size) {
    acc = (acc << size) | (uint64_t)code;
    nacc += size;
    if (nacc >= 32) {
      flush_word((uint32_t)(acc >> (nacc - 32)), first);
      first = false;
      nacc -= 32;
    }
  }
  __device__ __forceinline__ void finish() {
    if (nacc > 0) flush_word((uint32_t)(acc << (32 - nacc)), true);
  }
};

__global__ void __launch_bounds__(128) k_jpeg_emit(const Job j) {
  __shared__ uint64_t s_part[2];
  __shared__ uint64_t s_base;
  const uint32_t i = blockIdx.x * 128u + threadIdx.x;
  uint64_t before = 0;
  for (uint32_t g = threadIdx.x; g < blockIdx.x; g += 128u) before += j.bits_blk[g];
  before = block_sum<128>(before, s_part);
  if (threadIdx.x == 0u) s_base = before;
  __syncthreads();
  const uint64_t my_bits = i < j.nblk ? j.bits[i] : 0u;
  const uint64_t in_front = block_exclusive_sum<128>(my_bits, s_part);
  if (i >= j.nblk) return;
  const uint64_t off = s_base + in_front;
  if (i == j.nblk - 1u) *j.total_bits = off + my_bits;   // read by the stuffing kernels behind this one
  EmitSink s;
  s.words = j.stream;
  s.acc = 0;
  s.nacc = (int)(off & 31u);   // bits of this word that belong to earlier blocks: zeros here, OR-ed there
  s.wi = (uint32_t)(off >> 5);
  s.first = true;
  const int chroma = comp_of(j, i);
  walk_block(j.coef + (size_t)i * 64u, pred_of(j, i), 2 * chroma, 2 * chroma + 1, s);
  if (i == j.nblk - 1u) {      // jchuff.c flush_bits: fill the last byte with ones
    const uint64_t end = off + my_bits;
    const int pad = (int)((8u - (uint32_t)(end & 7u)) & 7u);
    if (pad) s.put((1u << pad) - 1u, pad);
  }
  s.finish();
}

// ---- byte stuffing -------------------------------------------------------------------------------------------------
constexpr uint32_t kChunk = 64;
// ff_count[t]: 0xFF bytes in chunk t; ff_blk[g]: in the 256 chunks of workgroup g.  The copy needs the number of stuffed zeros in
// front of every chunk: inside a workgroup a block scan of 256 counts, across workgroups the sum of the totals in front, which
// every workgroup of the copy forms for itself (a 4K frame has ~100 of them) -- no device-wide scan between the two kernels.
__global__ void __launch_bounds__(256) k_jpeg_stuff_count(const Job j, const uint64_t* total_bits) {
  __shared__ uint32_t s_part[4];
  const uint32_t t = blockIdx.x * 256u + threadIdx.x;
  const uint64_t nbytes = (*total_bits + 7u) >> 3;
  const uint64_t b0 = (uint64_t)t * kChunk;
  if ((uint64_t)blockIdx.x * 256u * kChunk >= nbytes) return;   // uniform per workgroup: behind the end of the stream
  uint32_t n = 0;
  if (t < j.max_chunks && b0 < nbytes) {
    const uint8_t* p = reinterpret_cast<const uint8_t*>(j.stream) + b0;
    const uint32_t len = (uint32_t)(nbytes - b0 < kChunk ? nbytes - b0 : kChunk);
    for (uint32_t k = 0; k < len; ++k) n += p[k] == 0xFFu;
    j.ff_count[t] = n;
  }
  const uint32_t total = block_sum<256>(n, s_part);
  if (threadIdx.x == 0u) j.ff_blk[blockIdx.x] = total;
}
// One workgroup = 256 chunks = 16 KiB of the packed stream.  Every thread expands its chunk into LDS (0x00 after each
// 0xFF), then the workgroup writes the expanded run to the output as aligned dwords: the run's start in the output is
// arbitrary (header length + stuffed bytes before it), so a thread per chunk writing its own bytes would scatter byte
// stores over 256 cache lines per instruction (28 us per 4K frame measured; this form: see DESIGN.md).
__global__ void __launch_bounds__(256) k_jpeg_stuff_copy(const Job j, const uint64_t* total_bits, uint8_t* out, uint64_t out_cap,
                                                         uint64_t header_len, uint64_t* out_size) {
  __shared__ uint32_t s_part[4];
  __shared__ uint8_t s_buf[256 * kChunk * 2 + 8];
  __shared__ uint32_t s_len, s_base;
  const uint32_t first = blockIdx.x * 256u, t = first + threadIdx.x;
  const uint64_t nbytes = (*total_bits + 7u) >> 3;
  const uint64_t blk_b0 = (uint64_t)first * kChunk;
  if (blk_b0 >= nbytes) return;                       // uniform per workgroup
  uint32_t before = 0;
  for (uint32_t g = threadIdx.x; g < blockIdx.x; g += 256u) before += j.ff_blk[g];
  before = block_sum<256>(before, s_part);
  if (threadIdx.x == 0) { s_len = 0u; s_base = before; }
  __syncthreads();
  const uint32_t base_ff = s_base;
  const uint64_t b0 = (uint64_t)t * kChunk;
  const bool mine = t < j.max_chunks && b0 < nbytes;
  const uint32_t ff_before = block_exclusive_sum<256>(mine ? j.ff_count[t] : 0u, s_part);
  if (mine) {
    const uint32_t len = (uint32_t)(nbytes - b0 < kChunk ? nbytes - b0 : kChunk);
    uint32_t lo = (uint32_t)(b0 - blk_b0) + ff_before;
    const uint4* src = reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(j.stream) + b0);
#pragma unroll
    for (uint32_t q = 0; q < kChunk / 16u; ++q) {
      const uint4 v = src[q];                         // (reads up to 15 bytes past len inside the zeroed workspace)
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (uint32_t k = 0; k < 16u; ++k) {
        if (q * 16u + k < len) {
          const uint8_t by = (uint8_t)(w[k >> 2] >> (8u * (k & 3u)));
          s_buf[lo++] = by;
          if (by == 0xFFu) s_buf[lo++] = 0;
        }
      }
    }
    const bool last_of_stream = b0 + len == nbytes;
    if (last_of_stream) { s_buf[lo++] = 0xFF; s_buf[lo++] = 0xD9; }   // EOI
    if (last_of_stream || threadIdx.x == 255u) s_len = lo;
    if (last_of_stream) *out_size = header_len + blk_b0 + base_ff + lo;
  }
  __syncthreads();
  const uint32_t len = s_len;
  const uint64_t dst0 = header_len + blk_b0 + base_ff;              // offset of s_buf[0] in the output
  const uint32_t head = (uint32_t)((4u - ((reinterpret_cast<uintptr_t>(out) + dst0) & 3u)) & 3u);
  const uint32_t head_n = head < len ? head : len;
  if (threadIdx.x < head_n && dst0 + threadIdx.x < out_cap) out[dst0 + threadIdx.x] = s_buf[threadIdx.x];
  const uint32_t nwords = (len - head_n) >> 2;
  for (uint32_t k = threadIdx.x; k < nwords; k += 256u) {
    const uint32_t o = head_n + 4u * k;
    const uint32_t v = (uint32_t)s_buf[o] | ((uint32_t)s_buf[o + 1] << 8) | ((uint32_t)s_buf[o + 2] << 16) | ((uint32_t)s_buf[o + 3] << 24);
    if (dst0 + o + 4u <= out_cap) *reinterpret_cast<uint32_t*>(out + dst0 + o) = v;
    else for (uint32_t m = 0; m < 4u; ++m) if (dst0 + o + m < out_cap) out[dst0 + o + m] = s_buf[o + m];
  }
  const uint32_t tail0 = head_n + 4u * nwords;
  if (threadIdx.x < len - tail0 && dst0 + tail0 + threadIdx.x < out_cap) out[dst0 + tail0 + threadIdx.x] = s_buf[tail0 + threadIdx.x];
}

// ---- host side -----------------------------------------------------------------------------------------------------
size_t workspace_bytes(uint32_t nblk, Layout* l) {
  auto up = [](size_t v) { return (v + 255) / 256 * 256; };
  size_t o = 0;
  l->coef = o; o += up((size_t)nblk * 128);
  l->bits = o; o += up((size_t)(nblk + 1) * 4);
  l->bits_blk = o; o += up((size_t)(nblk / 128u + 2) * 4);
  // worst case per block: 20 bits of DC + 63 x 26 bits of AC = 1658 bits
  l->stream_bytes = up((size_t)nblk * 208 + 64);
  l->stream = o; o += l->stream_bytes;
  l->max_chunks = (uint32_t)(l->stream_bytes / kChunk);
  l->ff_count = o; o += up((size_t)l->max_chunks * 4);
  l->ff_blk = o; o += up((size_t)(l->max_chunks / 256u + 2) * 4);
  l->totals = o; o += 256;   // [0] total bits (uint64), [1] output size (uint64)
  l->scan_tmp_bytes = 0;
  l->scan_tmp = o;
  return o;
}

hipError_t encode_async(Job j, const Layout& l, uint8_t* ws, uint8_t* out, uint64_t out_cap, uint64_t header_len, hipStream_t s, uint64_t* out_size) {
  j.coef = reinterpret_cast<int16_t*>(ws + l.coef);
  j.bits = reinterpret_cast<uint32_t*>(ws + l.bits);
  j.bits_blk = reinterpret_cast<uint32_t*>(ws + l.bits_blk);
  j.stream = reinterpret_cast<uint32_t*>(ws + l.stream);
  j.ff_count = reinterpret_cast<uint32_t*>(ws + l.ff_count);
  j.ff_blk = reinterpret_cast<uint32_t*>(ws + l.ff_blk);
  j.max_chunks = l.max_chunks;
  uint64_t* totals = reinterpret_cast<uint64_t*>(ws + l.totals);
  hipError_t e;
  if ((e = hipMemsetAsync(j.stream, 0, l.stream_bytes, s)) != hipSuccess) return e;
  const dim3 gb((j.nblk + 127u) / 128u), bb(128);
  hipLaunchKernelGGL(k_jpeg_fdct_quant_count, gb, bb, 0, s, j);
  const uint64_t* total_bits = totals;   // written by the last thread of k_jpeg_emit
  j.total_bits = totals;
  hipLaunchKernelGGL(k_jpeg_emit, gb, bb, 0, s, j);
  const dim3 gc((j.max_chunks + 255u) / 256u), bc(256);
  hipLaunchKernelGGL(k_jpeg_stuff_count, gc, bc, 0, s, j, total_bits);
  hipLaunchKernelGGL(k_jpeg_stuff_copy, gc, bc, 0, s, j, total_bits, out, out_cap, header_len, out_size != nullptr ? out_size : totals + 1);
  return hipGetLastError();
}

}  // namespace jpeg
}  // namespace uhdr
