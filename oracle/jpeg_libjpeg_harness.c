/*
 * jpeg_libjpeg_harness.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Drives the libjpeg that ships in the image (/opt/conda: IJG libjpeg 9d) with the call sequence of the reference's
 * JpegEncoderHelper (lib/src/jpegencoderhelper.cpp:86-283), so that oracle/jpeg_oracle.c can be pinned against a real
 * libjpeg.  The reference's own helper does not compile against these headers (`return true;` from a function
 * returning libjpeg's `boolean`, an enum in IJG 9; it is written for libjpeg-turbo, where `boolean` is an int), and
 * no libjpeg-turbo headers exist in the image, so the sequence is restated in C here:
 *   jpeg_set_defaults, jpeg_set_quality(q, TRUE), raw_data_in, JDCT_ISLOW, sampling 2x2/1x1/1x1 or 1x1   (:119-136)
 *   optional APP2 marker right after jpeg_start_compress                                                (:97-100)
 *   16 luma + 8 chroma rows (or 8 rows of one plane) per jpeg_write_raw_data; rows past the height are one zero row;
 *   rows are copied into zero-padded buffers only when the stride is smaller than the 16-aligned width   (:138-283)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <jpeglib.h>

#define BATCH 16
#define ALIGNM(x, m) ((((x) + ((m)-1)) / (m)) * (m))

long lj_jpeg_encode(const unsigned char* y, const unsigned char* uv, int w, int h, int ls, int cs, int quality,
                    const void* icc, unsigned icc_n, unsigned char* out, long cap) {
  struct jpeg_compress_struct c;
  struct jpeg_error_mgr e;
  c.err = jpeg_std_error(&e);
  jpeg_create_compress(&c);
  unsigned char* mem = NULL;
  unsigned long memsz = 0;
  jpeg_mem_dest(&c, &mem, &memsz);
  const int gray = uv == NULL;
  c.image_width = w;
  c.image_height = h;
  c.input_components = gray ? 1 : 3;
  c.in_color_space = gray ? JCS_GRAYSCALE : JCS_YCbCr;
  jpeg_set_defaults(&c);
  jpeg_set_quality(&c, quality, TRUE);
  c.raw_data_in = TRUE;
  c.dct_method = JDCT_ISLOW;
  c.comp_info[0].h_samp_factor = gray ? 1 : 2;
  c.comp_info[0].v_samp_factor = gray ? 1 : 2;
  for (int i = 1; i < c.num_components; i++) {
    c.comp_info[i].h_samp_factor = 1;
    c.comp_info[i].v_samp_factor = 1;
  }
  jpeg_start_compress(&c, TRUE);
  if (icc != NULL && icc_n > 0) jpeg_write_marker(&c, JPEG_APP0 + 2, (const JOCTET*)icc, icc_n);

  const int aw = ALIGNM(w, BATCH), acw = ALIGNM(w / 2, BATCH / 2);
  const int pad_y = ls < aw, pad_c = cs < acw;
  unsigned char* empty = (unsigned char*)calloc(aw, 1);
  unsigned char* ybuf = (unsigned char*)calloc((size_t)aw * BATCH, 1);
  unsigned char* cbbuf = (unsigned char*)calloc((size_t)acw * BATCH / 2 + 1, 1);
  unsigned char* crbuf = (unsigned char*)calloc((size_t)acw * BATCH / 2 + 1, 1);
  JSAMPROW yr[BATCH], cbr[BATCH / 2], crr[BATCH / 2];
  JSAMPARRAY planes[3] = {yr, cbr, crr};
  const unsigned char* u = uv;
  const unsigned char* v = uv ? uv + (size_t)cs * h / 2 : NULL;
  while (c.next_scanline < c.image_height) {
    for (int i = 0; i < BATCH; i++) {
      size_t sl = c.next_scanline + i;
      if (sl < (size_t)h) {
        yr[i] = (JSAMPROW)(y + sl * ls);
        if (pad_y) {
          memcpy(ybuf + (size_t)i * aw, yr[i], w);
          yr[i] = ybuf + (size_t)i * aw;
        }
      } else {
        yr[i] = empty;
      }
    }
    if (!gray)
      for (int i = 0; i < BATCH / 2; i++) {
        size_t sl = c.next_scanline / 2 + i;
        if (sl < (size_t)h / 2) {
          cbr[i] = (JSAMPROW)(u + sl * cs);
          crr[i] = (JSAMPROW)(v + sl * cs);
          if (pad_c) {
            memcpy(cbbuf + (size_t)i * acw, cbr[i], w / 2);
            cbr[i] = cbbuf + (size_t)i * acw;
            memcpy(crbuf + (size_t)i * acw, crr[i], w / 2);
            crr[i] = crbuf + (size_t)i * acw;
          }
        } else {
          cbr[i] = crr[i] = empty;
        }
      }
    jpeg_write_raw_data(&c, planes, BATCH);
  }
  jpeg_finish_compress(&c);
  jpeg_destroy_compress(&c);
  const long n = (long)memsz;
  if (n <= cap) memcpy(out, mem, n);
  free(mem); free(empty); free(ybuf); free(cbbuf); free(crbuf);
  return n;
}

/* JpegDecoderHelper::decompressImage(..., DECODE_TO_YCBCR) (lib/src/jpegdecoderhelper.cpp:188-327 + decompressYUV :352-448,
 * decompressSingleChannel :450-516): raw_data_out, JDCT_ISLOW, 16 rows per jpeg_read_raw_data, planes cropped to w x h.
 * Returns the bytes written, or a negative value (-2: not 4:2:0 / grayscale, -3: cap too small, -1: libjpeg error). */
#include <setjmp.h>
struct lj_err { struct jpeg_error_mgr pub; jmp_buf jb; };
static void lj_error_exit(j_common_ptr c) { longjmp(((struct lj_err*)c->err)->jb, 1); }
static void lj_silent(j_common_ptr c) { (void)c; }
long lj_jpeg_decode(const unsigned char* jpg, long n, unsigned char* out, long cap, int* pw, int* ph, int* pgray) {
  struct jpeg_decompress_struct c;
  struct lj_err e;
  c.err = jpeg_std_error(&e.pub);
  e.pub.error_exit = lj_error_exit;
  e.pub.output_message = lj_silent;
  unsigned char* scratch = NULL;
  if (setjmp(e.jb)) { jpeg_destroy_decompress(&c); free(scratch); return -1; }
  jpeg_create_decompress(&c);
  jpeg_mem_src(&c, (unsigned char*)jpg, (unsigned long)n);
  if (jpeg_read_header(&c, TRUE) != JPEG_HEADER_OK) { jpeg_destroy_decompress(&c); return -1; }
  const int w = (int)c.image_width, h = (int)c.image_height;
  int gray;
  if (c.jpeg_color_space == JCS_YCbCr) {
    if (c.comp_info[0].h_samp_factor != 2 || c.comp_info[0].v_samp_factor != 2 || c.comp_info[1].h_samp_factor != 1 ||
        c.comp_info[1].v_samp_factor != 1 || c.comp_info[2].h_samp_factor != 1 || c.comp_info[2].v_samp_factor != 1) {
      jpeg_destroy_decompress(&c);
      return -2;
    }
    gray = 0;
  } else if (c.jpeg_color_space == JCS_GRAYSCALE) {
    gray = 1;
  } else {
    jpeg_destroy_decompress(&c);
    return -2;
  }
  *pw = w; *ph = h; *pgray = gray;
  const long need = gray ? (long)w * h : (long)w * h * 3 / 2;
  if (need > cap) { jpeg_destroy_decompress(&c); return -3; }
  c.out_color_space = c.jpeg_color_space;
  c.raw_data_out = TRUE;
  c.dct_method = JDCT_ISLOW;
  jpeg_start_decompress(&c);
  const int aw = ALIGNM(w, BATCH);
  /* always decode into the 16-aligned intermediate rows and copy the image part out (the reference does so whenever the
     width is not a multiple of 16 and writes in place otherwise: same bytes) */
  scratch = (unsigned char*)calloc((size_t)aw * BATCH * 3 / 2 + (size_t)aw, 1);
  unsigned char *yi = scratch, *ui = yi + (size_t)aw * BATCH, *vi = ui + (size_t)aw * BATCH / 4;
  JSAMPROW yr[BATCH], cbr[BATCH / 2], crr[BATCH / 2];
  JSAMPARRAY planes[3] = {yr, cbr, crr};
  for (int i = 0; i < BATCH; i++) yr[i] = yi + (size_t)i * aw;
  for (int i = 0; i < BATCH / 2; i++) { cbr[i] = ui + (size_t)i * (aw / 2); crr[i] = vi + (size_t)i * (aw / 2); }
  unsigned char *yp = out, *up = out + (size_t)w * h, *vp = up + (size_t)w * h / 4;
  while (c.output_scanline < c.image_height) {
    const size_t s0 = c.output_scanline;
    const int got = (int)jpeg_read_raw_data(&c, planes, BATCH);
    if (got <= 0) break;
    for (int i = 0; i < got; i++)
      if (s0 + i < (size_t)h) memcpy(yp + (s0 + i) * w, yr[i], w);
    if (!gray)
      for (int i = 0; i < BATCH / 2; i++)
        if (s0 / 2 + i < (size_t)h / 2) { memcpy(up + (s0 / 2 + i) * (w / 2), cbr[i], w / 2); memcpy(vp + (s0 / 2 + i) * (w / 2), crr[i], w / 2); }
  }
  jpeg_finish_decompress(&c);
  jpeg_destroy_decompress(&c);
  free(scratch);
  return need;
}

/* jpeg_read_coefficients: the quantised coefficients libjpeg holds after all scans of a (progressive or baseline) file, laid out
 * as the device decoder takes them -- blocks in MCU order (4:2:0: Y00 Y01 Y10 Y11 Cb Cr; padding blocks beyond a component's own
 * extent as zeros), 64 coefficients each in ZIGZAG order, DC as the value (not the difference).  Checker for the product's
 * host-side progressive entropy decoder (csrc/uhdr_jpeg_prog.cpp).  Returns the number of blocks, -1 on a libjpeg error,
 * -2 for a sampling other than 4:2:0 / grayscale, -3 when cap_blocks is too small. */
long lj_jpeg_coefficients(const unsigned char* jpg, long n, short* out, long cap_blocks, int* pw, int* ph, int* pgray) {
  static const unsigned char nat[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                        41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                        30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
  struct jpeg_decompress_struct c;
  struct lj_err e;
  c.err = jpeg_std_error(&e.pub);
  e.pub.error_exit = lj_error_exit;
  e.pub.output_message = lj_silent;
  if (setjmp(e.jb)) { jpeg_destroy_decompress(&c); return -1; }
  jpeg_create_decompress(&c);
  jpeg_mem_src(&c, (unsigned char*)jpg, (unsigned long)n);
  if (jpeg_read_header(&c, TRUE) != JPEG_HEADER_OK) { jpeg_destroy_decompress(&c); return -1; }
  const int w = (int)c.image_width, h = (int)c.image_height;
  int gray = c.jpeg_color_space == JCS_GRAYSCALE;
  if (!gray && (c.jpeg_color_space != JCS_YCbCr || c.comp_info[0].h_samp_factor != 2 || c.comp_info[0].v_samp_factor != 2 ||
                c.comp_info[1].h_samp_factor != 1 || c.comp_info[1].v_samp_factor != 1 || c.comp_info[2].h_samp_factor != 1 ||
                c.comp_info[2].v_samp_factor != 1)) { jpeg_destroy_decompress(&c); return -2; }
  *pw = w; *ph = h; *pgray = gray;
  const long mx = gray ? (w + 7) / 8 : (w + 15) / 16, my = gray ? (h + 7) / 8 : (h + 15) / 16;
  const long nblk = mx * my * (gray ? 1 : 6);
  if (nblk > cap_blocks) { jpeg_destroy_decompress(&c); return -3; }
  jvirt_barray_ptr* arrays = jpeg_read_coefficients(&c);
  memset(out, 0, (size_t)nblk * 128);
  for (int ci = 0; ci < c.num_components; ci++) {
    jpeg_component_info* comp = &c.comp_info[ci];
    for (JDIMENSION br = 0; br < comp->height_in_blocks; br++) {
      JBLOCKARRAY rows = (*c.mem->access_virt_barray)((j_common_ptr)&c, arrays[ci], br, 1, FALSE);
      for (JDIMENSION bc = 0; bc < comp->width_in_blocks; bc++) {
        long idx;
        if (gray) idx = (long)br * mx + bc;
        else if (ci == 0) idx = (((long)(br >> 1) * mx + (bc >> 1)) * 6 + ((br & 1) * 2 + (bc & 1)));
        else idx = ((long)br * mx + bc) * 6 + 3 + ci;
        if (idx >= nblk) continue;
        for (int k = 0; k < 64; k++) out[idx * 64 + k] = rows[0][bc][nat[k]];
      }
    }
  }
  jpeg_finish_decompress(&c);
  jpeg_destroy_decompress(&c);
  return nblk;
}
