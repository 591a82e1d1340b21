// uhdr_jpeg_prog.cpp -- progressive JPEG input (SOF2): the entropy decoding of every scan on the host, into the coefficient blocks
// the device decoder's dequantisation + IDCT kernel takes (uhdr_jpeg_dec.hip: k_jd_idct_multi).
//
// The reference decodes whatever libjpeg reads (lib/src/jpegdecoderhelper.cpp:190-320 has no progressive check), so a JPEG/R file
// whose primary image a camera or an editor wrote progressively must decode.  A progressive file is a sequence of scans over one
// coefficient array (T.81 Annex G): DC first / DC refinement (interleaved or not), AC first and AC refinement bands (one component
// each), every scan a Huffman-coded bit string of its own with end-of-band runs that span blocks and, in refinement scans, correction
// bits whose meaning depends on what earlier scans left in the array.  That dependence is what the baseline decoder's
// self-synchronisation (a subsequence decoded from a guessed state converges to the true one) does not have; the scans are walked
// here, once, by one host thread (plain C++: fuzzed under AddressSanitizer with the other host parsers), and the device does what it
// does for a baseline file from the coefficients on.  Decoding follows libjpeg's jdphuff.c decision by decision (corrupt data
// included: the same warnings-become-zeros tolerance is not reproduced -- a scan that violates the progression rules fails).
//
// Output (DecInfo): w, h, gray, quant (as latched when a component's first scan starts, like jdinput.c's latch_quant_tables),
// coef = nblk x 64 coefficients in zigzag order, blocks in the baseline decoder's order (4:2:0: Y00 Y01 Y10 Y11 Cb Cr per MCU),
// the DC term as the DIFFERENCE to the previous block of its component in that order (what a baseline scan would carry and the
// device's prefix sum undoes).  Complete files only: a file whose scans leave a coefficient short of full precision is what
// libjpeg smooths across blocks (jdcoefct.c: smoothing_ok); it is refused (-2).
#include <cstdint>
#include <cstring>
#include <vector>

#include "uhdr_jpeg.h"

namespace uhdr {
namespace jpeg {

namespace {

struct BitReader {
  const uint8_t* p;
  size_t pos, end;
  uint64_t acc = 0;     // bits left-aligned: the next bit is bit 63
  int nbits = 0;
  bool hit_marker = false;   // a marker other than a stuffed zero was met: zeros are fed from here on (jdhuff.c: insufficient data)
  BitReader(const uint8_t* d, size_t at, size_t n) : p(d), pos(at), end(n) {}
  void fill() {
    while (nbits <= 56) {
      uint8_t b = 0;
      if (!hit_marker && pos < end) {
        b = p[pos];
        if (b == 0xFF) {
          size_t q = pos + 1;
          while (q < end && p[q] == 0xFF) ++q;   // fill bytes
          if (q < end && p[q] == 0x00) { pos = q + 1; }
          else { hit_marker = true; b = 0; }
        } else {
          ++pos;
        }
      } else {
        hit_marker = true;
      }
      acc |= (uint64_t)b << (56 - nbits);
      nbits += 8;
    }
  }
  uint32_t peek(int n) { if (nbits < n) fill(); return (uint32_t)(acc >> (64 - n)); }
  void skip(int n) { acc <<= n; nbits -= n; }
  uint32_t get(int n) { if (n == 0) return 0; const uint32_t v = peek(n); skip(n); return v; }
  // restart: discard the rest of the byte, expect RSTn at the next marker position
  bool restart(unsigned expect) {
    acc = 0; nbits = 0;
    // position: the marker the reader stopped in front of (or scan forward to it)
    size_t q = pos;
    while (q + 1 < end && !(p[q] == 0xFF && p[q + 1] != 0x00 && p[q + 1] != 0xFF)) ++q;
    if (q + 1 >= end || p[q + 1] != (uint8_t)(0xD0 + expect)) return false;
    pos = q + 2;
    hit_marker = false;
    return true;
  }
};

struct HuffDec {
  const HuffSpec* h = nullptr;
  uint16_t look[512];   // 9-bit first level: (length << 8) | symbol, 0 = longer code
  int32_t maxcode[18];
  int32_t valoff[18];
  void build(const HuffSpec& s) {
    h = &s;
    memset(look, 0, sizeof(look));
    for (int l = 1; l <= 16; ++l) {
      maxcode[l] = s.count[l] ? (int32_t)s.first_code[l] + s.count[l] - 1 : -1;
      valoff[l] = (int32_t)s.first_val[l] - (int32_t)s.first_code[l];
      if (l <= 9)
        for (int i = 0; i < s.count[l]; ++i) {
          const uint32_t code = (uint32_t)(s.first_code[l] + i) << (9 - l);
          for (uint32_t f = 0; f < (1u << (9 - l)); ++f) look[code + f] = (uint16_t)((l << 8) | s.vals[s.first_val[l] + i]);
        }
    }
    maxcode[17] = 0x7FFFFFFF;
  }
  int decode(BitReader& br) const {
    const uint32_t v = br.peek(16);
    const uint16_t e = look[v >> 7];
    if (e) { br.skip(e >> 8); return e & 0xFF; }
    for (int l = 10; l <= 16; ++l) {
      const int32_t code = (int32_t)(v >> (16 - l));
      if (maxcode[l] >= 0 && code <= maxcode[l] && code >= (int32_t)h->first_code[l]) { br.skip(l); return h->vals[code + valoff[l]]; }
    }
    return -1;
  }
};

inline int extend(uint32_t v, int s) { return v < (1u << (s - 1)) ? (int)v - (1 << s) + 1 : (int)v; }
inline uint32_t rd16(const uint8_t* p) { return ((uint32_t)p[0] << 8) | p[1]; }

}  // namespace

// 0 ok, -1 malformed, -2 outside what is supported
int decode_progressive(const uint8_t* jpg, size_t n, DecInfo* info) {
  if (n < 4 || jpg[0] != 0xFF || jpg[1] != 0xD8) return -1;
  uint16_t quant[4][64];
  bool have_q[4] = {false, false, false, false};
  HuffSpec huff[2][4];
  memset(huff, 0, sizeof(huff));
  int nc = 0, hs[3] = {0, 0, 0}, vs[3] = {0, 0, 0}, tq[3] = {0, 0, 0}, cid[3] = {0, 0, 0};
  bool latched[3] = {false, false, false};
  uint32_t restart_interval = 0;
  int w = 0, h = 0;
  uint32_t mcus_x = 0, mcus_y = 0, nblk = 0;
  uint32_t bw[3] = {0, 0, 0}, bh[3] = {0, 0, 0};   // blocks a non-interleaved scan of the component covers
  // coef_bits[c][k]: -1 = not seen yet, else the Al the coefficient is known down to (jdphuff.c)
  int coef_bits[3][64];
  for (auto& cb : coef_bits) for (int& v : cb) v = -1;
  std::vector<int16_t>& coef = info->coef;
  bool frame = false;
  size_t pos = 2;
  for (;;) {
    while (pos + 1 < n && jpg[pos] == 0xFF && jpg[pos + 1] == 0xFF) pos++;
    if (pos + 2 > n || jpg[pos] != 0xFF) return -1;
    const unsigned m = jpg[pos + 1];
    if (m == 0xD9) break;   // EOI
    if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) { pos += 2; continue; }   // parameterless markers
    if (pos + 4 > n) return -1;
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
        HuffSpec& t = huff[cls][id];
        memset(&t, 0, sizeof(t));
        uint32_t code = 0, p = 0;
        for (int l = 1; l <= 16; ++l) {
          t.first_code[l] = (uint16_t)code; t.first_val[l] = (uint16_t)p; t.count[l] = seg[o + l];
          if (code + t.count[l] > (1u << l)) return -1;
          code += t.count[l]; p += t.count[l]; code <<= 1;
        }
        memcpy(t.vals, seg + o + 17, (size_t)cnt);
        t.present = 1;
        o += 17 + (size_t)cnt;
      }
    } else if (m == 0xC2) {
      if (frame || len < 8 || seg[0] != 8) return -2;
      h = (int)rd16(seg + 1); w = (int)rd16(seg + 3); nc = seg[5];
      if (nc != 1 && nc != 3) return -2;
      if (len < (size_t)(8 + 3 * nc) || w <= 0 || h <= 0) return -1;
      for (int c = 0; c < nc; ++c) { cid[c] = seg[6 + 3 * c]; hs[c] = seg[7 + 3 * c] >> 4; vs[c] = seg[7 + 3 * c] & 15; tq[c] = seg[8 + 3 * c]; if (tq[c] > 3) return -1; }
      if (nc == 3 && !(hs[0] == 2 && vs[0] == 2 && hs[1] == 1 && vs[1] == 1 && hs[2] == 1 && vs[2] == 1)) return -1;   // the reference fails too (:283-289)
      // (a single-component frame: sampling factors are irrelevant, every scan is non-interleaved -- T.81 A.2.2)
      info->w = w; info->h = h; info->gray = nc == 1;
      if (w > 65535 || h > 65535) return -1;
      if (w > 8192 || h > 8192) { info->progressive = true; info->scan_offset = 0; info->scan_bytes = 0; return 0; }   // the caller refuses the size (kMaxWidth)
      mcus_x = (uint32_t)(nc == 1 ? (w + 7) / 8 : (w + 15) / 16);
      mcus_y = (uint32_t)(nc == 1 ? (h + 7) / 8 : (h + 15) / 16);
      nblk = mcus_x * mcus_y * (nc == 1 ? 1u : 6u);
      bw[0] = (uint32_t)((w + 7) / 8); bh[0] = (uint32_t)((h + 7) / 8);
      bw[1] = bw[2] = (uint32_t)(((w + 1) / 2 + 7) / 8); bh[1] = bh[2] = (uint32_t)(((h + 1) / 2 + 7) / 8);
      frame = true;   // (the coefficient array -- up to 200 MB for an 8192 x 8192 frame -- is allocated at the first valid scan header)
    } else if (m == 0xC0 || m == 0xC1 || (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) {
      return -2;
    } else if (m == 0xDD) {
      if (len < 4) return -1;
      restart_interval = rd16(seg);
    } else if (m == 0xDA) {
      if (!frame) return -1;
      const int ns = len >= 3 ? seg[0] : 0;
      if (ns < 1 || ns > nc || len < (size_t)(6 + 2 * ns)) return -1;
      int sc[3], td[3], ta[3];
      for (int i = 0; i < ns; ++i) {
        int c = -1;
        for (int k = 0; k < nc; ++k) if (cid[k] == seg[1 + 2 * i]) c = k;
        if (c < 0) return -1;
        for (int k = 0; k < i; ++k) if (sc[k] == c) return -1;
        sc[i] = c; td[i] = seg[2 + 2 * i] >> 4; ta[i] = seg[2 + 2 * i] & 15;
        if (td[i] > 3 || ta[i] > 3) return -1;
      }
      if (coef.empty()) coef.assign((size_t)nblk * 64u, 0);
      const int Ss = seg[1 + 2 * ns], Se = seg[2 + 2 * ns], Ah = seg[3 + 2 * ns] >> 4, Al = seg[3 + 2 * ns] & 15;
      // jdphuff.c start_pass_phuff_decoder: the progression parameters must make sense
      const bool is_dc = Ss == 0;
      if (is_dc ? Se != 0 : (Se < Ss || Se > 63 || ns != 1)) return -1;
      if (Al > 13 || (Ah != 0 && Ah != Al + 1)) return -1;
      for (int i = 0; i < ns; ++i) {
        const int c = sc[i];
        if (!is_dc && coef_bits[c][0] < 0) return -1;   // AC without prior DC scan
        for (int k = Ss; k <= Se; ++k) {
          const int expected = coef_bits[c][k] < 0 ? 0 : coef_bits[c][k];
          if (Ah != expected) return -1;
          coef_bits[c][k] = Al;
        }
        if (!latched[c]) {   // latch_quant_tables (jdinput.c)
          if (!have_q[tq[c]]) return -1;
          memcpy(info->quant[c], quant[tq[c]], sizeof(quant[0]));
          latched[c] = true;
        }
      }
      HuffDec dc_t[3], ac_t;
      if (is_dc && Ah == 0) for (int i = 0; i < ns; ++i) { if (!huff[0][td[i]].present) return -1; dc_t[i].build(huff[0][td[i]]); }
      if (!is_dc) { if (!huff[1][ta[0]].present) return -1; ac_t.build(huff[1][ta[0]]); }
      BitReader br(jpg, pos + 2 + len, n);
      // the units of the scan: MCUs (interleaved: every block of the MCU, padding blocks included) or the blocks of one component
      const bool interleaved = ns > 1;
      const uint32_t units_x = interleaved ? mcus_x : bw[sc[0]], units_y = interleaved ? mcus_y : bh[sc[0]];
      const uint32_t total_units = units_x * units_y;
      int pred[3] = {0, 0, 0};
      uint32_t eobrun = 0, next_rst = 0, until_rst = restart_interval;
      const int p1 = 1 << Al, m1 = -(1 << Al);
      auto block_of = [&](int c, uint32_t br_, uint32_t bc_) -> int16_t* {   // component c's block (row, col) in the device decoder's order
        if (nc == 1) return coef.data() + ((size_t)br_ * mcus_x + bc_) * 64u;
        if (c == 0) return coef.data() + (((size_t)(br_ >> 1) * mcus_x + (bc_ >> 1)) * 6u + ((br_ & 1u) * 2u + (bc_ & 1u))) * 64u;
        return coef.data() + (((size_t)br_ * mcus_x + bc_) * 6u + 3u + (uint32_t)c) * 64u;
      };
      for (uint32_t u = 0; u < total_units; ++u) {
        if (restart_interval != 0 && until_rst == 0) {
          if (!br.restart(next_rst)) return -1;
          next_rst = (next_rst + 1) & 7;
          until_rst = restart_interval;
          pred[0] = pred[1] = pred[2] = 0;
          eobrun = 0;
        }
        if (restart_interval != 0) --until_rst;
        const uint32_t ur = u / units_x, uc = u - ur * units_x;
        if (is_dc) {
          for (int i = 0; i < ns; ++i) {
            const int c = sc[i];
            const uint32_t nb = interleaved ? (c == 0 ? 4u : 1u) : 1u;
            for (uint32_t k = 0; k < nb; ++k) {
              uint32_t rr = ur, cc = uc;
              if (interleaved && c == 0) { rr = 2 * ur + (k >> 1); cc = 2 * uc + (k & 1u); }
              int16_t* blk = block_of(c, rr, cc);
              if (Ah == 0) {
                const int s = dc_t[i].decode(br);
                if (s < 0 || s > 15) return -1;
                const int diff = s ? extend(br.get(s), s) : 0;
                // (unsigned: a hostile file can run the predictor or the shift out of int; what is stored is the low 16 bits either
                // way, as libjpeg's JCOEF assignment keeps them, and the device's running sum is taken modulo 2^16 as well)
                pred[c] = (int)((unsigned)pred[c] + (unsigned)diff);
                blk[0] = (int16_t)(uint16_t)((unsigned)pred[c] << Al);
              } else if (br.get(1)) {
                blk[0] = (int16_t)(blk[0] | p1);
              }
            }
          }
        } else {
          int16_t* blk = block_of(sc[0], ur, uc);
          if (Ah == 0) {   // decode_mcu_AC_first
            if (eobrun > 0) { --eobrun; continue; }
            for (int k = Ss; k <= Se; ++k) {
              const int rs = ac_t.decode(br);
              if (rs < 0) return -1;
              const int r = rs >> 4, s = rs & 15;
              if (s) {
                k += r;
                if (k > Se) return -1;
                blk[k] = (int16_t)(extend(br.get(s), s) * (1 << Al));
              } else if (r == 15) {
                k += 15;
              } else {
                eobrun = (1u << r);
                if (r) eobrun += br.get(r);
                --eobrun;
                break;
              }
            }
          } else {         // decode_mcu_AC_refine
            int k = Ss;
            if (eobrun == 0) {
              for (; k <= Se; ++k) {
                const int rs = ac_t.decode(br);
                if (rs < 0) return -1;
                int r = rs >> 4, s = rs & 15;
                if (s) {
                  if (s != 1) return -1;
                  s = br.get(1) ? p1 : m1;
                } else if (r != 15) {
                  eobrun = (1u << r);
                  if (r) eobrun += br.get(r);
                  break;   // force end-of-band
                }
                // advance over already-nonzero coefficients and r still-zero ones, appending correction bits to the nonzeroes
                do {
                  int16_t* t = blk + k;
                  if (*t != 0) {
                    if (br.get(1) && (*t & p1) == 0) *t = (int16_t)(*t >= 0 ? *t + p1 : *t + m1);
                  } else if (--r < 0) {
                    break;
                  }
                  ++k;
                } while (k <= Se);
                if (s) {
                  if (k > Se) return -1;
                  blk[k] = (int16_t)s;
                }
              }
            }
            if (eobrun > 0) {
              // the rest of the band: a correction bit for every coefficient that is already nonzero
              for (; k <= Se; ++k) {
                int16_t* t = blk + k;
                if (*t != 0 && br.get(1) && (*t & p1) == 0) *t = (int16_t)(*t >= 0 ? *t + p1 : *t + m1);
              }
              --eobrun;
            }
          }
        }
      }
      // the next marker behind the scan's data (the reader prefetches, so its position is not the place to look from): the walk
      // the baseline parser and the container scan use -- stuffed zeros, fill bytes and RSTn belong to the entropy-coded segment
      pos = skip_entropy_coded(jpg, pos + 2 + len, n);
      if (pos + 1 >= n) return -1;
      continue;
    }
    pos += 2 + len;
  }
  if (!frame) return -1;
  // complete files only: every coefficient of every component down to bit 0
  for (int c = 0; c < nc; ++c)
    for (int k = 0; k < 64; ++k)
      if (coef_bits[c][k] != 0) return -2;
  // DC values -> differences in the device decoder's block order (its prefix sum per component undoes this)
  {
    int prev[3] = {0, 0, 0};
    for (uint32_t b = 0; b < nblk; ++b) {
      const int c = nc == 1 ? 0 : ((b % 6u) < 4u ? 0 : (int)(b % 6u) - 3);
      int16_t* blk = coef.data() + (size_t)b * 64u;
      const int v = blk[0];
      blk[0] = (int16_t)(uint16_t)((unsigned)v - (unsigned)prev[c]);   // modulo 2^16: k_jd_idct reads the running sum as int16
      prev[c] = v;
    }
  }
  // the baseline parser's conventions for the fields the callers read
  info->progressive = true;
  info->restart_interval = 0;
  info->interval_start.clear();
  info->raw_bytes = 0;
  info->scan_offset = pos;   // the EOI marker: "the image ends behind its scans"
  info->scan_bytes = 0;
  return 0;
}

}  // namespace jpeg
}  // namespace uhdr
