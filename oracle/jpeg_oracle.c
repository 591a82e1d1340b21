/*
 * jpeg_oracle.c -- TEST INFRASTRUCTURE ONLY (see uhdr_oracle.h for the rules: tests/, smoke() and the
 * cpu_baseline leg of bench.py are the only callers).
 *
 * CPU restatement of what the reference's JpegEncoderHelper::compressImage produces
 * (lib/src/jpegencoderhelper.cpp:39-283, declared lib/include/ultrahdr/jpegencoderhelper.h:43-60): a baseline
 * sequential JPEG of a YUV 4:2:0 planar image or of a single 8-bit plane, written by libjpeg in raw-data mode with
 *     jpeg_set_defaults; jpeg_set_quality(quality, TRUE); raw_data_in; dct_method = JDCT_ISLOW;
 *     sampling 2x2,1x1,1x1 (or 1x1 for one plane)                               (:119-136)
 * and fed 16 (8 for one plane) rows per call, rows past the image height from an all-zero row, columns past the
 * image width zero-filled when the caller's stride is smaller than the 16-aligned width and otherwise taken from
 * the caller's buffer (:138-232, :235-283).
 *
 * The algorithm itself lives in a third-party dependency that /root/reference does not vendor: libjpeg-turbo,
 * pinned to 3.0.1 by the reference's CMakeLists.txt:254-256.  What is restated here is the published baseline
 * process (ITU-T T.81: Annex A FDCT/quantisation, Annex F.1.2 Huffman coding, Annex B markers, Annex K tables)
 * with libjpeg's integer choices: the "islow" 13-bit fixed-point FDCT, quantisation by rounded division of the
 * 8x-scaled coefficients, quality scaling of the Annex K tables, one DQT/DHT segment per table, JFIF 1.01 APP0,
 * dummy blocks at the right / bottom edge with zero AC and the DC of the block before them.
 *
 * Parity status: PINNED against the libjpeg builds present in the image, driven with the reference's call
 * sequence (oracle/jpeg_libjpeg_harness.c over /opt/conda/lib/libjpeg.so.9 = IJG 9d; Pillow's bundled
 * libjpeg-turbo for MCU-aligned sizes) -- tests/test_jpeg_oracle.py.  Both produce identical bytes for this
 * configuration.  The reference's own jpegencoderhelper.cpp is not buildable here: it needs libjpeg-turbo's
 * headers (`boolean` is an int there, an enum in IJG 9), which the image lacks.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "uhdr_oracle.h"

/* ---- Annex K tables ------------------------------------------------------------------------------------ */
static const uint8_t kStdLumQuant[64] = {
    16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,  14, 13, 16, 24, 40,  57,  69,  56,
    14, 17, 22, 29, 51,  87,  80,  62,  18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
    49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
static const uint8_t kStdChrQuant[64] = {
    17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99,
    47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
static const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                    41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                    30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
static const uint8_t kDcLumBits[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
static const uint8_t kDcChrBits[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
static const uint8_t kDcVals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static const uint8_t kAcLumBits[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
static const uint8_t kAcLumVals[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71,
    0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72,
    0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37,
    0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59,
    0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83,
    0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
    0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
    0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
static const uint8_t kAcChrBits[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
static const uint8_t kAcChrVals[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22,
    0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1,
    0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36,
    0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58,
    0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a,
    0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a,
    0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba,
    0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
    0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

/* ---- quality -> quantisation tables (libjpeg jpeg_quality_scaling + jpeg_add_quant_table, force_baseline) - */
void orc_jpeg_quant_table(int quality, int chroma, uint16_t out[64]) {
  if (quality <= 0) quality = 1;
  if (quality > 100) quality = 100;
  const int scale = quality < 50 ? 5000 / quality : 200 - quality * 2;
  const uint8_t* base = chroma ? kStdChrQuant : kStdLumQuant;
  for (int i = 0; i < 64; ++i) {
    long t = ((long)base[i] * scale + 50L) / 100L;
    if (t <= 0L) t = 1L;
    if (t > 255L) t = 255L; /* force_baseline */
    out[i] = (uint16_t)t;
  }
}

/* ---- forward DCT, "islow" (13-bit constants, 2 extra bits between the passes), output = 8 x the DCT ------ */
#define CONST_BITS 13
#define PASS1_BITS 2
#define FIX_0_298631336 2446
#define FIX_0_390180644 3196
#define FIX_0_541196100 4433
#define FIX_0_765366865 6270
#define FIX_0_899976223 7373
#define FIX_1_175875602 9633
#define FIX_1_501321110 12299
#define FIX_1_847759065 15137
#define FIX_1_961570560 16069
#define FIX_2_053119869 16819
#define FIX_2_562915447 20995
#define FIX_3_072711026 25172
#define DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n)) /* arithmetic shift of a signed value, as libjpeg assumes */

static void fdct_1d(const int32_t in[8], int32_t out[8], int pass) {
  int32_t tmp0 = in[0] + in[7], tmp7 = in[0] - in[7];
  int32_t tmp1 = in[1] + in[6], tmp6 = in[1] - in[6];
  int32_t tmp2 = in[2] + in[5], tmp5 = in[2] - in[5];
  int32_t tmp3 = in[3] + in[4], tmp4 = in[3] - in[4];
  int32_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
  const int sh = pass == 0 ? CONST_BITS - PASS1_BITS : CONST_BITS + PASS1_BITS;
  if (pass == 0) {
    out[0] = (tmp10 + tmp11) << PASS1_BITS;
    out[4] = (tmp10 - tmp11) << PASS1_BITS;
  } else {
    out[0] = DESCALE(tmp10 + tmp11, PASS1_BITS);
    out[4] = DESCALE(tmp10 - tmp11, PASS1_BITS);
  }
  int32_t z1 = (tmp12 + tmp13) * FIX_0_541196100;
  out[2] = DESCALE(z1 + tmp13 * FIX_0_765366865, sh);
  out[6] = DESCALE(z1 + tmp12 * (-FIX_1_847759065), sh);
  z1 = tmp4 + tmp7;
  int32_t z2 = tmp5 + tmp6, z3 = tmp4 + tmp6, z4 = tmp5 + tmp7;
  int32_t z5 = (z3 + z4) * FIX_1_175875602;
  tmp4 *= FIX_0_298631336;
  tmp5 *= FIX_2_053119869;
  tmp6 *= FIX_3_072711026;
  tmp7 *= FIX_1_501321110;
  z1 *= -FIX_0_899976223;
  z2 *= -FIX_2_562915447;
  z3 *= -FIX_1_961570560;
  z4 *= -FIX_0_390180644;
  z3 += z5;
  z4 += z5;
  out[7] = DESCALE(tmp4 + z1 + z3, sh);
  out[5] = DESCALE(tmp5 + z2 + z4, sh);
  out[3] = DESCALE(tmp6 + z2 + z3, sh);
  out[1] = DESCALE(tmp7 + z1 + z4, sh);
}

/* samples: 64 bytes in raster order; coef: quantised coefficients in natural (raster) order */
void orc_jpeg_fdct_quant(const uint8_t samples[64], const uint16_t quant[64], int16_t coef[64]) {
  int32_t ws[64], col[8], res[8];
  for (int r = 0; r < 8; ++r) {
    int32_t row[8];
    for (int c = 0; c < 8; ++c) row[c] = (int32_t)samples[r * 8 + c] - 128;
    fdct_1d(row, &ws[r * 8], 0);
  }
  for (int c = 0; c < 8; ++c) {
    for (int r = 0; r < 8; ++r) col[r] = ws[r * 8 + c];
    fdct_1d(col, res, 1);
    for (int r = 0; r < 8; ++r) ws[r * 8 + c] = res[r];
  }
  for (int i = 0; i < 64; ++i) { /* jcdctmgr.c: divide the 8x-scaled coefficient by (quant << 3), rounding half away */
    int32_t qval = (int32_t)quant[i] << 3, temp = ws[i];
    if (temp < 0) {
      temp = -temp;
      temp += qval >> 1;
      temp = temp >= qval ? temp / qval : 0;
      temp = -temp;
    } else {
      temp += qval >> 1;
      temp = temp >= qval ? temp / qval : 0;
    }
    coef[i] = (int16_t)temp;
  }
}

/* ---- Huffman ------------------------------------------------------------------------------------------------ */
typedef struct { uint16_t code[256]; uint8_t size[256]; } huff_tbl;
static void derive(const uint8_t bits[16], const uint8_t* vals, huff_tbl* t) { /* T.81 Annex C */
  memset(t, 0, sizeof(*t));
  unsigned code = 0;
  int p = 0;
  for (int l = 1; l <= 16; ++l) {
    for (int i = 0; i < bits[l - 1]; ++i, ++p) {
      t->code[vals[p]] = (uint16_t)code++;
      t->size[vals[p]] = (uint8_t)l;
    }
    code <<= 1;
  }
}

typedef struct { uint8_t* out; long cap, n; uint32_t acc; int nacc; } bitw;
static void put_byte(bitw* w, uint8_t b) {
  if (w->n < w->cap) w->out[w->n] = b;
  w->n++;
}
static void emit_bits(bitw* w, unsigned code, int size) { /* MSB first, 0xFF followed by a stuffed 0x00 */
  if (size == 0) return;
  w->acc = (w->acc << size) | (code & ((1u << size) - 1u));
  w->nacc += size;
  while (w->nacc >= 8) {
    uint8_t b = (uint8_t)(w->acc >> (w->nacc - 8));
    put_byte(w, b);
    if (b == 0xFF) put_byte(w, 0);
    w->nacc -= 8;
  }
}
static int nbits_of(int v) {
  int n = 0;
  while (v) { n++; v >>= 1; }
  return n;
}
static void encode_block(bitw* w, const int16_t coef[64], int* last_dc, const huff_tbl* dc, const huff_tbl* ac) {
  int temp = coef[0] - *last_dc, temp2 = temp;
  *last_dc = coef[0];
  if (temp < 0) { temp = -temp; temp2--; }
  int nb = nbits_of(temp);
  emit_bits(w, dc->code[nb], dc->size[nb]);
  if (nb) emit_bits(w, (unsigned)temp2, nb);
  int r = 0;
  for (int k = 1; k < 64; ++k) {
    temp = coef[kZigzag[k]];
    if (temp == 0) { r++; continue; }
    while (r > 15) { emit_bits(w, ac->code[0xF0], ac->size[0xF0]); r -= 16; }
    temp2 = temp;
    if (temp < 0) { temp = -temp; temp2--; }
    nb = nbits_of(temp);
    int sym = (r << 4) + nb;
    emit_bits(w, ac->code[sym], ac->size[sym]);
    emit_bits(w, (unsigned)temp2, nb);
    r = 0;
  }
  if (r > 0) emit_bits(w, ac->code[0], ac->size[0]);
}

/* ---- markers ---------------------------------------------------------------------------------------------- */
static void put16(bitw* w, unsigned v) { put_byte(w, (uint8_t)(v >> 8)); put_byte(w, (uint8_t)v); }
static void put_dqt(bitw* w, int idx, const uint16_t q[64]) {
  put16(w, 0xFFDB); put16(w, 67); put_byte(w, (uint8_t)idx);
  for (int i = 0; i < 64; ++i) put_byte(w, (uint8_t)q[kZigzag[i]]);
}
static void put_dht(bitw* w, int cls_idx, const uint8_t bits[16], const uint8_t* vals) {
  int n = 0;
  for (int i = 0; i < 16; ++i) n += bits[i];
  put16(w, 0xFFC4); put16(w, (unsigned)(2 + 1 + 16 + n)); put_byte(w, (uint8_t)cls_idx);
  for (int i = 0; i < 16; ++i) put_byte(w, bits[i]);
  for (int i = 0; i < n; ++i) put_byte(w, vals[i]);
}
/* everything up to and including SOS; returns the header length (callers with a small cap still get the length) */
long orc_jpeg_header(int w, int h, int gray, int quality, const void* icc, unsigned icc_n, uint8_t* out, long cap) {
  bitw b = {out, cap, 0, 0, 0};
  uint16_t ql[64], qc[64];
  orc_jpeg_quant_table(quality, 0, ql);
  orc_jpeg_quant_table(quality, 1, qc);
  put16(&b, 0xFFD8);
  put16(&b, 0xFFE0); put16(&b, 16); /* JFIF 1.01, no units, 1:1, no thumbnail */
  put_byte(&b, 'J'); put_byte(&b, 'F'); put_byte(&b, 'I'); put_byte(&b, 'F'); put_byte(&b, 0);
  put_byte(&b, 1); put_byte(&b, 1); put_byte(&b, 0); put16(&b, 1); put16(&b, 1); put_byte(&b, 0); put_byte(&b, 0);
  if (icc != NULL && icc_n > 0) { /* jpeg_write_marker(JPEG_APP0 + 2, ...) right after start_compress (:98-100) */
    put16(&b, 0xFFE2); put16(&b, icc_n + 2);
    for (unsigned i = 0; i < icc_n; ++i) put_byte(&b, ((const uint8_t*)icc)[i]);
  }
  put_dqt(&b, 0, ql);
  if (!gray) put_dqt(&b, 1, qc);
  const int nc = gray ? 1 : 3;
  put16(&b, 0xFFC0); put16(&b, (unsigned)(8 + 3 * nc)); put_byte(&b, 8); put16(&b, (unsigned)h); put16(&b, (unsigned)w);
  put_byte(&b, (uint8_t)nc);
  for (int c = 0; c < nc; ++c) {
    put_byte(&b, (uint8_t)(c + 1));
    put_byte(&b, (uint8_t)((c == 0 && !gray) ? 0x22 : 0x11));
    put_byte(&b, (uint8_t)(c == 0 ? 0 : 1));
  }
  put_dht(&b, 0x00, kDcLumBits, kDcVals);
  put_dht(&b, 0x10, kAcLumBits, kAcLumVals);
  if (!gray) {
    put_dht(&b, 0x01, kDcChrBits, kDcVals);
    put_dht(&b, 0x11, kAcChrBits, kAcChrVals);
  }
  put16(&b, 0xFFDA); put16(&b, (unsigned)(6 + 2 * nc)); put_byte(&b, (uint8_t)nc);
  for (int c = 0; c < nc; ++c) { put_byte(&b, (uint8_t)(c + 1)); put_byte(&b, (uint8_t)(c == 0 ? 0x00 : 0x11)); }
  put_byte(&b, 0); put_byte(&b, 63); put_byte(&b, 0);
  return b.n;
}

/* ---- sample access with the helper's padding rules (jpegencoderhelper.cpp:147-222, :239-278) -------------- */
typedef struct { const uint8_t* p; int w, h, stride, pad_cols; } plane;
static uint8_t sample(const plane* pl, int r, int c) {
  if (r >= pl->h) return 0;                 /* the all-zero `empty` row */
  if (c >= pl->w && pl->pad_cols) return 0; /* stride < aligned width: row copied into a zero-padded buffer */
  return pl->p[(size_t)r * pl->stride + c];
}
static void fetch_block(const plane* pl, int brow, int bcol, uint8_t s[64]) {
  for (int r = 0; r < 8; ++r)
    for (int c = 0; c < 8; ++c) s[r * 8 + c] = sample(pl, brow * 8 + r, bcol * 8 + c);
}

/* quantised coefficients of every block in the order the entropy coder visits them (dummy edge blocks included);
 * coef must hold orc_jpeg_block_count() * 64 int16.  Returns the block count. */
long orc_jpeg_block_count(int w, int h, int gray) {
  if (gray) return (long)((w + 7) / 8) * ((h + 7) / 8);
  return (long)((w + 15) / 16) * ((h + 15) / 16) * 6;
}
long orc_jpeg_coefficients(const uint8_t* y, const uint8_t* uv, int w, int h, int ls, int cs, int quality, int16_t* coef) {
  const int gray = uv == NULL;
  uint16_t ql[64], qc[64];
  orc_jpeg_quant_table(quality, 0, ql);
  orc_jpeg_quant_table(quality, 1, qc);
  uint8_t s[64];
  long n = 0;
  const int aw = (w + 15) / 16 * 16;
  plane py = {y, w, h, ls, ls < aw};
  if (gray) {
    for (int br = 0; br < (h + 7) / 8; ++br)
      for (int bc = 0; bc < (w + 7) / 8; ++bc, ++n) {
        fetch_block(&py, br, bc, s);
        orc_jpeg_fdct_quant(s, ql, coef + n * 64);
      }
    return n;
  }
  const int cw = w / 2, ch = h / 2, acw = (cw + 7) / 8 * 8;
  plane pu = {uv, cw, ch, cs, cs < acw};
  plane pv = {uv + (size_t)cs * (size_t)h / 2, cw, ch, cs, cs < acw}; /* chroma_plane_size = chromaStride * height / 2 (:140) */
  const int ybw = (w + 7) / 8, ybh = (h + 7) / 8;              /* component sizes in blocks (libjpeg jcmaster.c) */
  const int cbw = ((w + 1) / 2 + 7) / 8, cbh = ((h + 1) / 2 + 7) / 8;
  for (int mr = 0; mr < (h + 15) / 16; ++mr)
    for (int mc = 0; mc < (w + 15) / 16; ++mc) {
      int16_t* mcu = coef + n * 64;
      int k = 0;
      for (int yi = 0; yi < 2; ++yi)       /* jccoefct.c compress_data: real blocks, then dummies with the previous DC */
        for (int xi = 0; xi < 2; ++xi, ++k) {
          const int br = mr * 2 + yi, bc = mc * 2 + xi;
          if (br < ybh && bc < ybw) {
            fetch_block(&py, br, bc, s);
            orc_jpeg_fdct_quant(s, ql, mcu + k * 64);
          } else {
            memset(mcu + k * 64, 0, 64 * sizeof(int16_t));
            mcu[k * 64] = mcu[(k - 1) * 64];
          }
        }
      for (int c = 0; c < 2; ++c, ++k) {
        if (mr < cbh && mc < cbw) {
          fetch_block(c == 0 ? &pu : &pv, mr, mc, s);
          orc_jpeg_fdct_quant(s, qc, mcu + k * 64);
        } else { /* cannot happen for even sizes; kept for the shape of the rule */
          memset(mcu + k * 64, 0, 64 * sizeof(int16_t));
          mcu[k * 64] = mcu[(k - 1) * 64];
        }
      }
      n += 6;
    }
  return n;
}

/* JpegEncoderHelper::compressImage (:39-52).  uv == NULL selects the single-plane form.  Returns the JPEG size
 * (also when it exceeds cap; then only cap bytes were written), or -1 for sizes libjpeg rejects. */
long orc_jpeg_encode(const uint8_t* y, const uint8_t* uv, int w, int h, int ls, int cs, int quality, const void* icc,
                     unsigned icc_n, uint8_t* out, long cap) {
  if (y == NULL || w <= 0 || h <= 0 || w > 65500 || h > 65500) return -1;
  const int gray = uv == NULL;
  const long nblk = orc_jpeg_block_count(w, h, gray);
  int16_t* coef = (int16_t*)malloc((size_t)nblk * 64 * sizeof(int16_t));
  if (!coef) return -1;
  orc_jpeg_coefficients(y, uv, w, h, ls, cs, quality, coef);
  bitw b = {out, cap, 0, 0, 0};
  b.n = orc_jpeg_header(w, h, gray, quality, icc, icc_n, out, cap);
  huff_tbl dcl, acl, dcc, acc;
  derive(kDcLumBits, kDcVals, &dcl);
  derive(kAcLumBits, kAcLumVals, &acl);
  derive(kDcChrBits, kDcVals, &dcc);
  derive(kAcChrBits, kAcChrVals, &acc);
  int last_dc[3] = {0, 0, 0};
  for (long i = 0; i < nblk; ++i) {
    const int k = gray ? 0 : (int)(i % 6);
    const int comp = gray ? 0 : (k < 4 ? 0 : k - 3);
    encode_block(&b, coef + i * 64, &last_dc[comp], comp == 0 ? &dcl : &dcc, comp == 0 ? &acl : &acc);
  }
  emit_bits(&b, 0x7F, 7); /* flush: pad the last byte with ones */
  b.nacc = 0;
  put16(&b, 0xFFD9);
  free(coef);
  return b.n;
}

/* =====================================================================================================================
 * Decoding: what JpegDecoderHelper::decompressImage(..., DECODE_TO_YCBCR) returns (lib/src/jpegdecoderhelper.cpp:188-327,
 * :352-516): libjpeg with raw_data_out and JDCT_ISLOW on a 4:2:0 YCbCr or a grayscale JPEG; the result buffer holds the
 * w x h luma plane followed (4:2:0) by the (w/2) x (h/2) Cb and Cr planes at w*h and w*h + w*h/4, i.e. the component
 * planes cropped to the image size.  Restated for the baseline sequential Huffman process (SOF0; SOF1 with 8-bit
 * samples is the same process): T.81 Annex B markers, F.2.2 decoding, A.3.3 dequantisation + libjpeg's "islow" IDCT.
 * Progressive and arithmetic-coded files (which libjpeg would also read) are outside this restatement: -2.
 * ===================================================================================================================== */
typedef struct { uint8_t look_len[65536]; uint8_t look_sym[65536]; int present; } dhuff;
static void dhuff_build(dhuff* t, const uint8_t bits[16], const uint8_t* vals) {
  memset(t->look_len, 0, sizeof(t->look_len));
  unsigned code = 0;
  int p = 0;
  for (int l = 1; l <= 16; ++l) {
    for (int i = 0; i < bits[l - 1]; ++i, ++p) {
      const unsigned lo = code << (16 - l), n = 1u << (16 - l);
      for (unsigned k = 0; k < n && lo + k < 65536u; ++k) { t->look_len[lo + k] = (uint8_t)l; t->look_sym[lo + k] = vals[p]; }
      code++;
    }
    code <<= 1;
  }
  t->present = 1;
}
typedef struct { const uint8_t* p; long n, pos; uint64_t acc; int nacc; int hit_marker; } bitr;
static void br_fill(bitr* b) { /* entropy-coded segment: FF00 -> FF, any other FFxx ends the segment (zeros are fed after it) */
  while (b->nacc <= 48) {
    unsigned byte = 0;
    if (!b->hit_marker && b->pos < b->n) {
      byte = b->p[b->pos];
      if (byte == 0xFF) {
        if (b->pos + 1 < b->n && b->p[b->pos + 1] == 0) b->pos += 2;
        else { b->hit_marker = 1; byte = 0; }
      } else b->pos++;
    }
    b->acc = (b->acc << 8) | byte;
    b->nacc += 8;
  }
}
static unsigned br_peek16(bitr* b) { br_fill(b); return (unsigned)((b->acc >> (b->nacc - 16)) & 0xFFFFu); }
static void br_skip(bitr* b, int n) { b->nacc -= n; }
static int br_get(bitr* b, int n) {
  if (n == 0) return 0;
  br_fill(b);
  int v = (int)((b->acc >> (b->nacc - n)) & ((1u << n) - 1u));
  b->nacc -= n;
  return v;
}
static int extend(int v, int n) { return n == 0 ? 0 : (v < (1 << (n - 1)) ? v - (1 << n) + 1 : v); } /* F.2.2.1 */

/* libjpeg jidctint.c ("islow"), coefficients in natural order already multiplied by the quantiser */
static void idct_1d(const int32_t in[8], int32_t out[8], int pass) {
  int32_t z2 = in[2], z3 = in[6];
  int32_t z1 = (z2 + z3) * FIX_0_541196100;
  int32_t tmp2 = z1 + z3 * (-FIX_1_847759065), tmp3 = z1 + z2 * FIX_0_765366865;
  z2 = in[0]; z3 = in[4];
  int32_t tmp0 = (z2 + z3) << CONST_BITS, tmp1 = (z2 - z3) << CONST_BITS;
  int32_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
  tmp0 = in[7]; tmp1 = in[5]; tmp2 = in[3]; tmp3 = in[1];
  z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
  int32_t z4 = tmp1 + tmp3, z5 = (z3 + z4) * FIX_1_175875602;
  tmp0 *= FIX_0_298631336; tmp1 *= FIX_2_053119869; tmp2 *= FIX_3_072711026; tmp3 *= FIX_1_501321110;
  z1 *= -FIX_0_899976223; z2 *= -FIX_2_562915447; z3 *= -FIX_1_961570560; z4 *= -FIX_0_390180644;
  z3 += z5; z4 += z5;
  tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
  const int sh = pass == 0 ? CONST_BITS - PASS1_BITS : CONST_BITS + PASS1_BITS + 3;
  out[0] = DESCALE(tmp10 + tmp3, sh); out[7] = DESCALE(tmp10 - tmp3, sh);
  out[1] = DESCALE(tmp11 + tmp2, sh); out[6] = DESCALE(tmp11 - tmp2, sh);
  out[2] = DESCALE(tmp12 + tmp1, sh); out[5] = DESCALE(tmp12 - tmp1, sh);
  out[3] = DESCALE(tmp13 + tmp0, sh); out[4] = DESCALE(tmp13 - tmp0, sh);
}
void orc_jpeg_idct(const int16_t coef_natural[64], const uint16_t quant_natural[64], uint8_t samples[64]) {
  int32_t ws[64], v[8], r[8];
  for (int c = 0; c < 8; ++c) { /* pass 1: columns */
    for (int k = 0; k < 8; ++k) v[k] = (int32_t)coef_natural[k * 8 + c] * (int32_t)quant_natural[k * 8 + c];
    idct_1d(v, r, 0);
    for (int k = 0; k < 8; ++k) ws[k * 8 + c] = r[k];
  }
  for (int row = 0; row < 8; ++row) { /* pass 2: rows, + 128, clamp (libjpeg's range_limit table) */
    idct_1d(&ws[row * 8], r, 1);
    for (int k = 0; k < 8; ++k) {
      int32_t x = r[k] + 128;
      samples[row * 8 + k] = (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x));
    }
  }
}

static unsigned rd16(const uint8_t* p) { return ((unsigned)p[0] << 8) | p[1]; }
/* returns bytes written (w*h*3/2 or w*h), -1 malformed, -2 unsupported process / sampling, -3 cap too small (sizes set) */
long orc_jpeg_decode(const uint8_t* jpg, long n, uint8_t* out, long cap, int* pw, int* ph, int* pgray) {
  if (jpg == NULL || n < 4 || jpg[0] != 0xFF || jpg[1] != 0xD8) return -1;
  static dhuff tbl[2][4]; /* [class][id]; not re-entrant: the oracle is called from one thread for decoding */
  uint16_t quant[4][64];
  int have_q[4] = {0, 0, 0, 0};
  for (int c = 0; c < 2; ++c) for (int i = 0; i < 4; ++i) tbl[c][i].present = 0;
  int w = 0, h = 0, nc = 0, hs[3] = {0}, vs[3] = {0}, tq[3] = {0}, cid[3] = {0}, restart = 0;
  long pos = 2;
  for (;;) {
    if (pos + 4 > n || jpg[pos] != 0xFF) return -1;
    while (pos < n && jpg[pos] == 0xFF && jpg[pos + 1] == 0xFF) pos++; /* fill bytes */
    const unsigned m = jpg[pos + 1];
    const long len = rd16(jpg + pos + 2);
    const uint8_t* seg = jpg + pos + 4;
    if (pos + 2 + len > n || len < 2) return -1;
    if (m == 0xDB) {
      for (long o = 0; o + 1 <= len - 2;) {
        const int pq = seg[o] >> 4, id = seg[o] & 15;
        if (id > 3 || (o + 1 + (pq ? 128 : 64)) > len - 2) return -1;
        for (int i = 0; i < 64; ++i) quant[id][kZigzag[i]] = pq ? (uint16_t)rd16(seg + o + 1 + 2 * i) : seg[o + 1 + i];
        have_q[id] = 1;
        o += 1 + (pq ? 128 : 64);
      }
    } else if (m == 0xC4) {
      for (long o = 0; o + 17 <= len - 2;) {
        const int cls = seg[o] >> 4, id = seg[o] & 15;
        int cnt = 0;
        for (int i = 0; i < 16; ++i) cnt += seg[o + 1 + i];
        if (cls > 1 || id > 3 || cnt > 256 || o + 17 + cnt > len - 2) return -1;
        dhuff_build(&tbl[cls][id], seg + o + 1, seg + o + 17);
        o += 17 + cnt;
      }
    } else if (m == 0xC0 || m == 0xC1) {
      if (seg[0] != 8) return -2;
      h = (int)rd16(seg + 1); w = (int)rd16(seg + 3); nc = seg[5];
      if (nc != 1 && nc != 3) return -2;
      for (int c = 0; c < nc; ++c) { cid[c] = seg[6 + 3 * c]; hs[c] = seg[7 + 3 * c] >> 4; vs[c] = seg[7 + 3 * c] & 15; tq[c] = seg[8 + 3 * c]; }
    } else if (m == 0xC2 || (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) {
      return -2;
    } else if (m == 0xDD) {
      restart = (int)rd16(seg);
    } else if (m == 0xDA) {
      if (nc == 0 || seg[0] != nc) return -2; /* one interleaved scan with all components */
      int td[3], ta[3];
      for (int c = 0; c < nc; ++c) {
        if (seg[1 + 2 * c] != cid[c]) return -2;
        td[c] = seg[2 + 2 * c] >> 4; ta[c] = seg[2 + 2 * c] & 15;
        if (td[c] > 3 || ta[c] > 3 || !tbl[0][td[c]].present || !tbl[1][ta[c]].present || tq[c] > 3 || !have_q[tq[c]]) return -1;
      }
      const int gray = nc == 1;
      if (gray ? 0 : !(hs[0] == 2 && vs[0] == 2 && hs[1] == 1 && vs[1] == 1 && hs[2] == 1 && vs[2] == 1)) return -2; /* :256-262 */
      if (w <= 0 || h <= 0) return -1;
      *pw = w; *ph = h; *pgray = gray;
      const long need = gray ? (long)w * h : (long)w * h + 2 * ((long)w * h / 4);
      if (need > cap) return -3;
      const int mcux = gray ? (w + 7) / 8 : (w + 15) / 16, mcuy = gray ? (h + 7) / 8 : (h + 15) / 16;
      const int bpm = gray ? 1 : 6;
      uint8_t* planes[3] = {out, out + (size_t)w * h, out + (size_t)w * h + (size_t)w * h / 4};
      const int pwid[3] = {w, w / 2, w / 2}, phgt[3] = {h, h / 2, h / 2};
      bitr b = {jpg, n, pos + 2 + len, 0, 0, 0};
      int pred[3] = {0, 0, 0}, rst_left = restart;
      for (int my = 0; my < mcuy; ++my)
        for (int mx = 0; mx < mcux; ++mx) {
          if (restart && rst_left == 0) { /* B.2.5 / F.2.2.4: byte-align, expect RSTn, reset predictors */
            b.nacc = 0; b.acc = 0;
            if (b.hit_marker) { b.pos += 2; b.hit_marker = 0; }
            else if (b.pos + 1 < b.n && b.p[b.pos] == 0xFF && (b.p[b.pos + 1] & 0xF8) == 0xD0) b.pos += 2;
            pred[0] = pred[1] = pred[2] = 0;
            rst_left = restart;
          }
          for (int k = 0; k < bpm; ++k) {
            const int c = gray ? 0 : (k < 4 ? 0 : k - 3);
            int16_t blk[64];
            memset(blk, 0, sizeof(blk));
            unsigned look = br_peek16(&b);
            int l = tbl[0][td[c]].look_len[look];
            if (l == 0) return -1;
            int s = tbl[0][td[c]].look_sym[look];
            br_skip(&b, l);
            pred[c] += extend(br_get(&b, s), s);
            blk[0] = (int16_t)pred[c];
            for (int z = 1; z < 64;) {
              look = br_peek16(&b);
              l = tbl[1][ta[c]].look_len[look];
              if (l == 0) return -1;
              const int rs = tbl[1][ta[c]].look_sym[look];
              br_skip(&b, l);
              const int r = rs >> 4, sz = rs & 15;
              if (sz == 0) { if (r == 15) { z += 16; continue; } break; }
              z += r;
              if (z > 63) return -1;
              blk[kZigzag[z]] = (int16_t)extend(br_get(&b, sz), sz);
              z++;
            }
            uint8_t smp[64];
            orc_jpeg_idct(blk, quant[tq[c]], smp);
            const int bx = gray ? mx : (c == 0 ? mx * 2 + (k & 1) : mx), by = gray ? my : (c == 0 ? my * 2 + (k >> 1) : my);
            for (int r2 = 0; r2 < 8; ++r2)
              for (int c2 = 0; c2 < 8; ++c2) {
                const int yy = by * 8 + r2, xx = bx * 8 + c2;
                if (yy < phgt[c] && xx < pwid[c]) planes[c][(size_t)yy * pwid[c] + xx] = smp[r2 * 8 + c2];
              }
          }
          if (restart) rst_left--;
        }
      return need;
    } else if (m == 0xD9) {
      return -1;
    }
    pos += 2 + len;
  }
}

/* ---------------------------------------------------------------------------------------------------------------------
 * libjpeg-turbo's path from decoded 4:2:0 planes to RGBA, as JpegDecoderHelper::decompressImage(..., DECODE_TO_RGBA)
 * configures it (lib/src/jpegdecoderhelper.cpp:251-281: JCS_EXT_RGBA, default do_fancy_upsampling, JDCT_ISLOW) and
 * decodeJPEGR(ULTRAHDR_OUTPUT_SDR) hands it out (lib/src/jpegr.cpp:768-786).  The arithmetic lives in the reference's pinned
 * dependency (libjpeg-turbo 3.0.1, CMakeLists.txt:254-256), not under /root/reference; restated from the published algorithm:
 *   jdsample.c h2v2_fancy_upsample: triangle filter, 9/16 3/16 3/16 1/16 of the four nearest chroma samples, rounding constants
 *     8 and 7 alternating by column, edge samples replicated (jdmainct.c makes the rows above the first / below the last
 *     duplicates of them); images whose chroma planes are at most 2 samples wide get h2v2_upsample (replication) instead
 *     (jdsample.c jinit_upsampler);
 *   jdcolor.c ycc_rgb_convert: R = Y + Cr_r[Cr], G = Y + ((Cb_g[Cb] + Cr_g[Cr]) >> 16), B = Y + Cb_b[Cb] with the 16-bit
 *     fixed-point tables of build_ycc_rgb_table, clamped to 0..255; alpha 0xFF.
 * Pinned against Pillow's bundled libjpeg-turbo on the JPEG corpus (tests/test_jpeg_oracle.py).  w, h even.
 * ------------------------------------------------------------------------------------------------------------------- */
int orc_ycc420_to_rgba(const uint8_t* yp, const uint8_t* cbp, const uint8_t* crp, int w, int h, uint8_t* rgba) {
  if (!yp || !cbp || !crp || !rgba || w <= 0 || h <= 0 || (w & 1) || (h & 1)) return -1;
  const int cw = w / 2, ch = h / 2;
  int cr_r[256], cb_b[256];
  long cr_g[256], cb_g[256];
#define ORC_FIX(x) ((long)((x) * 65536.0 + 0.5))
  for (int i = 0; i < 256; ++i) {
    const long x = i - 128;
    cr_r[i] = (int)((ORC_FIX(1.40200) * x + 32768) >> 16);
    cb_b[i] = (int)((ORC_FIX(1.77200) * x + 32768) >> 16);
    cr_g[i] = -ORC_FIX(0.71414) * x;
    cb_g[i] = -ORC_FIX(0.34414) * x + 32768;
  }
#undef ORC_FIX
  for (int r = 0; r < h; ++r) {
    const int i = r >> 1;
    int o = (r & 1) ? i + 1 : i - 1;
    if (o < 0) o = 0;
    if (o > ch - 1) o = ch - 1;
    for (int x = 0; x < w; ++x) {
      const int c = x >> 1;
      int n = (x & 1) ? c + 1 : c - 1;
      if (n < 0) n = 0;
      if (n > cw - 1) n = cw - 1;
      const int bias = (x & 1) ? 7 : 8;
      int cb, cr;
      if (cw > 2) {
        cb = (3 * (3 * cbp[i * cw + c] + cbp[o * cw + c]) + (3 * cbp[i * cw + n] + cbp[o * cw + n]) + bias) >> 4;
        cr = (3 * (3 * crp[i * cw + c] + crp[o * cw + c]) + (3 * crp[i * cw + n] + crp[o * cw + n]) + bias) >> 4;
      } else {   /* jinit_upsampler: the fancy filter needs downsampled_width > 2, narrower images get plain replication */
        cb = cbp[i * cw + c];
        cr = crp[i * cw + c];
      }
      const int y = yp[r * w + x];
      int R = y + cr_r[cr], G = y + (int)((cb_g[cb] + cr_g[cr]) >> 16), B = y + cb_b[cb];
      R = R < 0 ? 0 : (R > 255 ? 255 : R);
      G = G < 0 ? 0 : (G > 255 ? 255 : G);
      B = B < 0 ? 0 : (B > 255 ? 255 : B);
      uint8_t* px = rgba + ((size_t)r * w + x) * 4;
      px[0] = (uint8_t)R; px[1] = (uint8_t)G; px[2] = (uint8_t)B; px[3] = 0xFF;
    }
  }
  return 0;
}
