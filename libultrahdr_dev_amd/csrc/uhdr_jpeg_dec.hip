// uhdr_jpeg_dec.hip -- baseline JPEG decompression on the GPU (decode side).
//
// Drop-in for the reference's JpegDecoderHelper::decompressImage(..., DECODE_TO_YCBCR)
// (lib/src/jpegdecoderhelper.cpp:188-327 + decompressYUV :352-448, decompressSingleChannel :450-516): a 4:2:0 YCbCr or a
// grayscale baseline JPEG -> the w x h luma plane followed by the (w/2) x (h/2) Cb and Cr planes, the bytes libjpeg
// returns with raw_data_out and JDCT_ISLOW.  This is the step in front of applyGainMap on the decode path (jpegr.cpp:796-801).
//
// A baseline scan without restart markers is ONE bit string: where a code starts is only known once everything before
// it has been decoded.  The decoder uses the self-synchronisation of Huffman codes (Klein & Wiseman 2003; Weissenberger &
// Schmidt 2018): a decoder started at a wrong bit position falls in step with the true decoder after a few symbols.
//   k_jd_prepare_multi     per 64-byte chunk, the bytes that stay once the 0x00 after every 0xFF is dropped; first-level lookup tables (the
//                    next 11 bits -> what the symbol implies) of the file's own Huffman tables; cleared flags and coefficients
//   k_jd_unstuff_copy_multi   compact -> raw bit string (offsets: the workgroups' totals summed, a block scan inside)
//   k_jd_sync_multi<0>     every thread decodes one 512-bit subsequence from a guessed state (block 0, coefficient 0) and
//                    records the state (bit, block-in-MCU, coefficient) it crosses the subsequence's end with
//   k_jd_sync_multi<1>     rounds: thread i re-decodes subsequence i from the end state of i-1 (only if that state changed);
//                    when no end state changes any more, every subsequence's start state is the true one (thread 0
//                    starts from the true state).  Bit and coefficient position fall in step within tens of bits, the
//                    block-in-MCU index (which decides luma vs. chroma tables) only after ~7 MCUs of a 4:2:0 file, so
//                    a 4K frame takes 5-8 rounds at quality 75 and 21-24 at quality 95; a round is one lane decoding
//                    512 bits (~17 us), whatever the number of subsequences: that product is the decoder's latency.
//                    A launch runs 4 rounds with the states in LDS and returns at once when the one before changed nothing.
//   (scan of the blocks completed per subsequence -> index of the block each subsequence starts in)
//   k_jd_write_multi       final decode from the true start states: DC differences and AC values into zeroed coefficient blocks
//   (one scan turns the DC differences into DC values, the three components side by side)
//   k_jd_idct_multi        one thread per block: DC value, dequantise, "islow" IDCT, +128, clamp, store cropped to the plane
// The decoding kernels read nothing from global memory inside their symbol loops: the workgroup's stretch of the bit string, the
// first-level tables and the canonical form of the longer codes are in LDS (a wave waits for all its memory operations at once).
// Restart intervals (DRI / RSTn) make the job easier, not harder: every interval is a byte-aligned bit string of its own whose
// start state is known, so its first subsequence plays the role of subsequence 0, nothing is carried across an interval boundary,
// and the block index and the DC predictors restart with it (segmented scans).  The host finds the markers while it looks for the
// end of the segment (parse_header) and the unstuffing pass drops them like the stuffed zeros.
// Every kernel has a second grid dimension over images (blockIdx.y; the per-image jobs sit in device memory): a JPEG/R file is two
// JPEGs and a server decodes many files at once, and since a decode is latency-bound the images of a call cost little more than one.
// Progressive files: their scans are entropy-decoded on the host (uhdr_jpeg_prog.cpp) and join the batch at the coefficient blocks
// (DC prefix sum + IDCT here).  Arithmetic-coded / lossless files return UHDR_HIP_ERROR_UNSUPPORTED_FEATURE; samplings other than
// 4:2:0 / grayscale fail as they do in the reference.
#include <hip/hip_runtime.h>
#include <cstring>

#include <rocprim/device/device_scan_by_key.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

#include "uhdr_jpeg.h"
#include "uhdr_wave_scan.h"

namespace uhdr {
namespace jpeg {

#ifndef UHDR_JD_SUBBITS
#define UHDR_JD_SUBBITS 512
#endif
constexpr uint32_t kSubBits = UHDR_JD_SUBBITS;   // bits per subsequence
constexpr uint32_t kUnstuffChunk = 64;  // bytes per thread in the unstuffing passes

// ---- unstuffing ----------------------------------------------------------------------------------------------------
// a byte of the entropy-coded segment that is not part of the bit string: the zero stuffed after a 0xFF, and (files with restart
// intervals) both bytes of an RSTn marker
__device__ __forceinline__ bool dropped_byte(const uint8_t* src, uint32_t n, uint32_t at, uint8_t v, uint8_t prev, int rst) {
  if (prev == 0xFF && (v == 0 || (rst && (v & 0xF8) == 0xD0))) return true;
  return rst && v == 0xFF && at + 1u < n && (src[at + 1u] & 0xF8) == 0xD0;
}
// kept[t]: bytes chunk t keeps; kept_blk[g]: bytes the 256 chunks of workgroup g keep.  The compaction needs the number of
// bytes kept in front of every chunk: inside a workgroup that is a block scan of 256 counts, across workgroups a sum of at most a
// few hundred totals, which every workgroup of the copy forms for itself -- no device-wide scan between the two kernels.
__device__ __forceinline__ void unstuff_count_body(const uint8_t* src, uint32_t n, uint32_t* kept, uint32_t* kept_blk, int rst) {
  __shared__ uint32_t s_part[4];
  const uint32_t t = blockIdx.x * 256u + threadIdx.x;
  const uint32_t b0 = t * kUnstuffChunk;
  if (blockIdx.x * 256u * kUnstuffChunk >= n) return;   // (uniform: a workgroup behind the end of this image's segment owns no total)
  uint32_t k = 0;
  if (b0 < n) {
    const uint32_t len = n - b0 < kUnstuffChunk ? n - b0 : kUnstuffChunk;
    uint8_t prev = b0 ? src[b0 - 1] : 0;
    for (uint32_t i = 0; i < len; ++i) {
      const uint8_t v = src[b0 + i];
      k += !dropped_byte(src, n, b0 + i, v, prev, rst);
      prev = v;
    }
    kept[t] = k;
  }
  const uint32_t total = block_sum<256>(k, s_part);
  if (threadIdx.x == 0u) kept_blk[blockIdx.x] = total;
}
// a workgroup compacts its 16 KiB into LDS and writes the run out as aligned dwords (its start in `dst` is arbitrary)
__device__ __forceinline__ void unstuff_copy_body(const uint8_t* src, uint32_t n, const uint32_t* kept, const uint32_t* kept_blk, uint8_t* dst, int rst) {
  __shared__ uint32_t s_part[4];
  __shared__ uint8_t s_buf[256 * kUnstuffChunk + 8];
  __shared__ uint32_t s_len, s_base;
  const uint32_t first = blockIdx.x * 256u, t = first + threadIdx.x;
  const uint32_t blk_b0 = first * kUnstuffChunk;
  if (blk_b0 >= n) return;
  uint32_t before = 0;
  for (uint32_t g = threadIdx.x; g < blockIdx.x; g += 256u) before += kept_blk[g];
  before = block_sum<256>(before, s_part);
  if (threadIdx.x == 0u) s_base = before;
  __syncthreads();
  const uint32_t base = s_base;
  const uint32_t b0 = t * kUnstuffChunk;
  if (threadIdx.x == 0) s_len = 0u;
  uint32_t lo = block_exclusive_sum<256>(b0 < n ? kept[t] : 0u, s_part);
  __syncthreads();
  if (b0 < n) {
    const uint32_t len = n - b0 < kUnstuffChunk ? n - b0 : kUnstuffChunk;
    uint8_t prev = b0 ? src[b0 - 1] : 0;
    const uint4* q = reinterpret_cast<const uint4*>(src + b0);     // src is 256-byte aligned, chunks are 64 bytes
#pragma unroll
    for (uint32_t k4 = 0; k4 < kUnstuffChunk / 16u; ++k4) {
      const uint4 v = q[k4];                                       // (may read up to 15 bytes past n inside the workspace)
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (uint32_t k = 0; k < 16u; ++k) {
        const uint8_t by = (uint8_t)(w[k >> 2] >> (8u * (k & 3u)));
        if (k4 * 16u + k < len) {
          if (!dropped_byte(src, n, b0 + k4 * 16u + k, by, prev, rst)) s_buf[lo++] = by;
          prev = by;
        }
      }
    }
    if (b0 + len == n || threadIdx.x == 255u) s_len = lo;
  }
  __syncthreads();
  const uint32_t len = s_len;
  uint8_t* d0 = dst + base;
  const uint32_t head = (uint32_t)((4u - (reinterpret_cast<uintptr_t>(d0) & 3u)) & 3u);
  const uint32_t head_n = head < len ? head : len;
  if (threadIdx.x < head_n) d0[threadIdx.x] = s_buf[threadIdx.x];
  const uint32_t nwords = (len - head_n) >> 2;
  for (uint32_t k = threadIdx.x; k < nwords; k += 256u) {
    const uint32_t o = head_n + 4u * k;
    *reinterpret_cast<uint32_t*>(d0 + o) = (uint32_t)s_buf[o] | ((uint32_t)s_buf[o + 1] << 8) | ((uint32_t)s_buf[o + 2] << 16) | ((uint32_t)s_buf[o + 3] << 24);
  }
  const uint32_t tail0 = head_n + 4u * nwords;
  if (threadIdx.x < len - tail0) d0[tail0 + threadIdx.x] = s_buf[tail0 + threadIdx.x];
}

#ifndef UHDR_JD_FASTBITS
#define UHDR_JD_FASTBITS 11
#endif
constexpr uint32_t kFastBitsEarly = UHDR_JD_FASTBITS;   // == kFastBits (first-level table width), needed before its definition
// ---- Huffman lookup tables -----------------------------------------------------------------------------------------
// First-level tables, indexed by the next kFastBits bits of the stream; a workgroup that decodes copies them into LDS.
// lut[x]: the 32-bit entry of the final pass (coef_entry), 0 when no code of at most kFastBits bits matches.
// adv[x]: what the position-only passes need from a symbol, in 16 bits: bits 8..13 = bits consumed (code + value bits, <= 27),
// bits 0..6 = advance of the coefficient index (DC: 1; AC: run + 1, ZRL 16, EOB 64); kLongEntry when no such code matches.  The
// upper 16 bits: the same for this symbol and the next one together, where both lie inside the pattern (0: no pair; their
// top bit is never set: at most 32 bits per pair).
// Longer codes (and "no code at all") are resolved from the canonical form of the table, also in LDS (LongCodes): a decoding loop
// must not load from memory -- a wave waits for all its outstanding vector memory operations at once, coefficient stores included.
constexpr uint32_t kLongEntry = 0x80000000u;   // first-level entry of the position-only tables: look among the longer codes
__device__ __forceinline__ uint32_t adv_entry(uint32_t tb, uint32_t len, uint32_t sym) {
  const uint32_t vb = sym & 15u, r = sym >> 4;
  const uint32_t dz = (tb & 1u) == 0u ? 1u : (vb != 0u ? r + 1u : (r == 15u ? 16u : 64u));
  return ((len + vb) << 8) | dz;
}
// What the final pass needs from a symbol: bits 0..4 code length, 5..8 value bits, 9..15 advance of the coefficient index, 16..19 run
// of zeros in front of the coefficient, bit 20 = the symbol carries a coefficient, bit 21 = not a code (one bit consumed).
constexpr uint32_t kNoCodeEntry = 1u | (1u << 21);
__device__ __forceinline__ uint32_t coef_entry(uint32_t tb, uint32_t len, uint32_t sym) {
  const uint32_t vb = sym & 15u, r = sym >> 4;
  if ((tb & 1u) == 0u) return len | (vb << 5) | (1u << 9) | (1u << 20);              // DC: the difference, position 0
  if (vb != 0u) return len | (vb << 5) | ((r + 1u) << 9) | (r << 16) | (1u << 20);   // AC: r zeros, then the value
  return len | ((r == 15u ? 16u : 64u) << 9);                                        // ZRL / EOB (runs of 1..14 zeros without a value: as EOB)
}
// the code of at most `avail` bits at the top of the kFastBits-bit pattern x: (length << 8) | symbol, or 0
__device__ __forceinline__ uint32_t short_code(const HuffSpec& h, uint32_t x, uint32_t avail) {
#pragma unroll 1
  for (uint32_t l = 1; l <= avail; ++l) {
    const uint32_t code = x >> (kFastBitsEarly - l);
    if (code >= h.first_code[l] && code - h.first_code[l] < h.count[l]) return (l << 8) | h.vals[h.first_val[l] + code - h.first_code[l]];
  }
  return 0u;
}
__device__ __forceinline__ void build_lut_body(const DecTables& t, uint32_t* lut, uint32_t* adv, uint32_t block) {
  const uint32_t g = block * 256u + threadIdx.x;   // 4 tables x 2^kFastBits
  if (g >= (4u << kFastBitsEarly)) return;
  const uint32_t tb = g >> kFastBitsEarly, x = g & ((1u << kFastBitsEarly) - 1u);
  const HuffSpec& h = t.huff[tb];
  const uint32_t e = h.present ? short_code(h, x, kFastBitsEarly) : 0u;
  lut[g] = e != 0u ? coef_entry(tb, e >> 8, e & 0xFFu) : 0u;
  uint32_t a = kLongEntry;
  if (e != 0u) {
    a = adv_entry(tb, e >> 8, e & 0xFFu);
    // A symbol that does not end the block is followed by an AC code of the same component -- the same table behind an AC symbol,
    // the AC table next to it behind a DC difference: when that code lies inside the pattern too, the upper half of the entry is the
    // pair (bits and index advance of both); decode_positions takes it when the first symbol neither fills the block nor crosses the
    // subsequence's end.  Two symbols per table lookup where the codes are short; a block of a flat area (DC difference, EOB) is one.
    const uint32_t bits1 = (a >> 8) & 31u, dz1 = a & 127u;
    const uint32_t tb2 = tb | 1u;
    if (dz1 < 64u && bits1 < kFastBitsEarly && t.huff[tb2].present) {
      const uint32_t e2 = short_code(t.huff[tb2], (x << bits1) & ((1u << kFastBitsEarly) - 1u), kFastBitsEarly - bits1);
      if (e2 != 0u) {
        const uint32_t a2 = adv_entry(tb2, e2 >> 8, e2 & 0xFFu);
        const uint32_t bits = bits1 + ((a2 >> 8) & 31u);
        // (a step may consume 32 bits at most: the window is refilled one word at a time)
        if (bits <= 32u) a |= ((bits << 8) | (dz1 + (a2 & 127u))) << 16;   // advance <= 16 + 64: 7 bits
      }
    }
  }
  adv[g] = a;
}

// ---- the sequential decoder one thread runs over a stretch of bits ------------------------------------------------------
struct DState { uint32_t p; uint32_t cz; };   // bit position; (block-in-MCU << 8) | coefficient index

// A symbol costs two dependent memory reads (stream bits, then the code table); from L2 that is ~1000 cycles per symbol, and a
// wave waits for the slowest of its lanes at every symbol (one s_waitcnt covers all outstanding loads).  So nothing in the
// symbol loop touches global memory: a workgroup first copies the stretch of the bit string its 256 subsequences cover into LDS
// (stage_bits), the bit window lives in registers (64 bits, refilled one word at a time from there), the first kFastBits bits of
// every code are looked up in LDS, and so are the longer codes (LongCodes).
constexpr uint32_t kFastBits = kFastBitsEarly;
// The codes of more than kFastBits bits, in canonical form (T.81 Annex C): a code of length l is the l-bit number in
// [first_code[l], first_code[l] + count[l]); with lim[l] = that upper bound shifted to 16 bits, the lengths' ranges follow one another,
// so the length of the code in front of a 16-bit peek is the first l with peek < lim[l].
constexpr uint32_t kLongLens = 16u - kFastBits;
struct LongCodes {
  uint32_t lim[4][kLongLens];   // [table][l - kFastBits - 1]
  int32_t off[4][kLongLens];    // first_val[l] - first_code[l]
  uint8_t vals[4][256];
};
// workgroup prologue (before a barrier)
__device__ __forceinline__ void load_long_codes(const DecTables& t, LongCodes& lc) {
  const uint32_t g = threadIdx.x;
  if (g < 4u * kLongLens) {
    const uint32_t tb = g / kLongLens, k = g - tb * kLongLens, l = kFastBits + 1u + k;
    const HuffSpec& h = t.huff[tb];
    lc.lim[tb][k] = h.present ? ((uint32_t)h.first_code[l] + (uint32_t)h.count[l]) << (16u - l) : 0u;
    lc.off[tb][k] = (int32_t)h.first_val[l] - (int32_t)h.first_code[l];
  }
  for (uint32_t q = g; q < 4u * 256u; q += blockDim.x) lc.vals[q >> 8][q & 255u] = t.huff[q >> 8].vals[q & 255u];
}
// (length << 8) | symbol of the code of more than kFastBits bits in front of `peek`, 0 if there is none
__device__ __forceinline__ uint32_t long_code(const LongCodes& lc, uint32_t tb, uint32_t peek) {
  uint32_t k = 0;
  while (k < kLongLens && peek >= lc.lim[tb][k]) ++k;
  if (k == kLongLens) return 0u;
  const uint32_t l = kFastBits + 1u + k;
  const uint32_t at = (uint32_t)(lc.off[tb][k] + (int32_t)(peek >> (16u - l))) & 255u;   // (in range for a table that is a prefix code; never outside vals)
  return (l << 8) | lc.vals[tb][at];
}
constexpr uint32_t kStageWords = 256u * (kSubBits / 32u) + 8u;   // 256 subsequences + the words a decoder reads past its end
struct Reader {
  const uint32_t* words;   // LDS copy of words [word0, word0 + kStageWords) of the stream
  uint32_t word0;
  uint64_t win;     // bits [base, base + 64) of the stream, MSB first
  uint32_t nextw;   // bits [base + 64, base + 96), still in memory byte order
  uint32_t base;    // multiple of 32
  __device__ __forceinline__ void init(const uint32_t* staged, uint32_t first_word, uint32_t p) {
    words = staged;
    word0 = first_word;
    const uint32_t i = (p >> 5) - word0;
    base = (p >> 5) << 5;
    win = ((uint64_t)__builtin_bswap32(staged[i]) << 32) | (uint64_t)__builtin_bswap32(staged[i + 1]);
    nextw = staged[i + 2];
  }
  __device__ __forceinline__ void advance_to(uint32_t p) {   // p - base < 64
    if (p - base >= 32u) {
      win = (win << 32) | (uint64_t)__builtin_bswap32(nextw);
      base += 32u;
      nextw = words[(base >> 5) + 2u - word0];
    }
  }
};

// One symbol of the final pass: the coefficient it carries (has: at zigzag position zpos of the current block, value val) and
// the new state.  Everything the symbol implies sits in one 32-bit table entry (coef_entry), so the walk has no branches but the
// rare long-code lookup and the window refill.  A bit pattern that is no code (only possible off-sync, or in a corrupt file)
// consumes one bit -- any deterministic rule will do for the synchronisation -- and is reported (returns false), as is a run
// that leaves the block.
// gray: the job's field by value -- read through the job it would be loaded again after every coefficient store (the compiler
// must assume the store hit it), and the wait for that load would sit out the store.
__device__ __forceinline__ bool step_coef(const bool gray, const LongCodes& lc, const uint32_t (*s_lut)[1u << kFastBits], Reader& rd, DState& s,
                                          bool& block_done, uint32_t& zpos, int& val, bool& has) {
  const uint32_t c = s.cz >> 8, z = s.cz & 0xFFu;
  const uint32_t tb = (gray ? 0u : (c >= 4u ? 2u : 0u)) + (z != 0u ? 1u : 0u);   // slots: 0 DC luma, 1 AC luma, 2 DC chroma, 3 AC chroma
  const uint32_t sh = s.p - rd.base;
  const uint64_t w = rd.win;
  const uint32_t peek = (uint32_t)(w >> (48u - sh)) & 0xFFFFu;
  uint32_t e = s_lut[tb][peek >> (16u - kFastBits)];
  if (e == 0u) {
    const uint32_t ls = long_code(lc, tb, peek);
    e = ls != 0u ? coef_entry(tb, ls >> 8, ls & 0xFFu) : kNoCodeEntry;
  }
  const uint32_t len = e & 31u, vbits = (e >> 5) & 15u, dz = (e >> 9) & 127u, run = (e >> 16) & 15u;
  zpos = z + run;
  has = ((e >> 20) & 1u) != 0u && zpos < 64u;
  const bool ok = ((e >> 21) & 1u) == 0u && (((e >> 20) & 1u) == 0u || zpos < 64u);
  const uint32_t raw = (uint32_t)(w >> (64u - sh - len - vbits)) & ((1u << vbits) - 1u);   // (sh + len + vbits <= 62)
  val = raw < ((1u << vbits) >> 1) ? (int)raw - (int)(1u << vbits) + 1 : (int)raw;         // T.81 F.2.2.1; no value bits: 0
  s.p += len + vbits;
  rd.advance_to(s.p);
  uint32_t nz = z + dz;
  block_done = nz >= 64u;
  nz = block_done ? 0u : nz;
  const uint32_t bpm = gray ? 1u : 6u;
  const uint32_t nc = block_done ? (c + 1u == bpm ? 0u : c + 1u) : c;
  s.cz = (nc << 8) | nz;
  return ok;
}

// workgroup prologue: one of the two first-level tables -> LDS (16 bytes per lane and turn; ends with a barrier)
template <typename T>
__device__ __forceinline__ void load_fast_table(const T* table, T (*s_tab)[1u << kFastBits]) {
  const uint4* src = reinterpret_cast<const uint4*>(table);
  uint4* dst = reinterpret_cast<uint4*>(&s_tab[0][0]);
  for (uint32_t g = threadIdx.x; g < (4u << kFastBits) * sizeof(T) / 16u; g += blockDim.x) dst[g] = src[g];
  __syncthreads();
}

// Where subsequence i lies.  Without restart intervals: bits [512 i, 512 (i + 1)) of one bit string, and only subsequence 0 knows
// its start state.  With them every interval is its own bit string (byte aligned, DC predictors reset, T.81 E.2.4): its first
// subsequence starts from the known state too and nothing is carried across an interval boundary.
__device__ __forceinline__ uint32_t sub_begin(const DecJob& j, uint32_t i) { return j.sub_start ? j.sub_start[i] : i * kSubBits; }
__device__ __forceinline__ uint32_t sub_end_bit(const DecJob& j, uint32_t i) {
  if (j.sub_end) return j.sub_end[i];
  return (i + 1u) * kSubBits < j.total_bits ? (i + 1u) * kSubBits : j.total_bits;
}
__device__ __forceinline__ bool sub_is_first(const DecJob& j, uint32_t i) { return i == 0u || (j.sub_key && j.sub_key[i] != j.sub_key[i - 1u]); }
__device__ __forceinline__ bool sub_is_last(const DecJob& j, uint32_t i) { return i + 1u == j.nsub || (j.sub_key && j.sub_key[i + 1u] != j.sub_key[i]); }

// workgroup prologue: words [first word of subsequence i0, last word a decoder of subsequence i1 - 1 can read] of the bit string -> LDS.
// A decoder starts inside its subsequence (or, in a round, less than one symbol past its start), stops with the first symbol that
// crosses its end (a symbol is at most 27 bits) and keeps three words in its window: 8 words of slack.  The buffer behind the
// bit string is zero-filled for 64 bytes (dec_workspace_bytes), so the slack of the last workgroup is there to read.
__device__ __forceinline__ uint32_t stage_bits(const DecJob& j, uint32_t i0, uint32_t* s_bits) {
  const uint32_t i1 = i0 + 256u < j.nsub ? i0 + 256u : j.nsub;
  const uint32_t w0 = sub_begin(j, i0) >> 5;
  const uint32_t w1 = ((sub_end_bit(j, i1 - 1u) + 27u) >> 5) + 3u;
  const uint32_t cnt = w1 - w0 + 1u < kStageWords ? w1 - w0 + 1u : kStageWords;
  if ((w0 & 3u) == 0u) {   // always, unless restart intervals put the subsequences at odd bytes (reads up to 3 words more: inside the slack)
    const uint4* src = reinterpret_cast<const uint4*>(j.raw + w0);
    uint4* dst = reinterpret_cast<uint4*>(s_bits);
    for (uint32_t k = threadIdx.x; k < (cnt + 3u) / 4u; k += 256u) dst[k] = src[k];
  } else {
    for (uint32_t k = threadIdx.x; k < cnt; k += 256u) s_bits[k] = j.raw[w0 + k];
  }
  return w0;
}

// One position-only decode of subsequence i from state s until its end is crossed; nb: blocks completed on the way.
// (A corrupt state cannot come out of the rounds -- positions only grow, by at most one symbol past an end -- but a read outside
// the staged words must be impossible, not unlikely: such a state decodes nothing and keeps its value.)
// A step takes one symbol -- or two, when the table entry holds a pair and the first symbol neither completes the block nor crosses
// the subsequence's end (the state a subsequence ends with is the one behind the first symbol that crosses it).  The loop is the
// decoder's latency (one lane, one wave per SIMD: every instruction is waited for), so its state is kept in the form the next
// step needs: bits left to the end instead of a position, the shift that brings the next 16 bits of the window down instead of
// the window's base, block-in-MCU and coefficient index apart, the LDS address of the next word.  No branches but the rare
// long-code lookup and the window refill.
__device__ __forceinline__ void decode_positions(const DecJob& j, const LongCodes& lc, const uint32_t (*s_adv)[1u << kFastBits], const uint32_t* s_bits,
                                                 uint32_t w0, uint32_t i, DState& s, uint32_t& nb) {
  const uint32_t end = sub_end_bit(j, i);
  const bool gray = j.gray != 0;
  nb = 0;
  if (!((s.p >> 5) >= w0 && (s.p >> 5) - w0 + 3u < kStageWords && end > s.p && end - s.p <= kSubBits + 32u)) return;
  const uint32_t* next_word = s_bits + ((s.p >> 5) - w0);
  uint64_t win = ((uint64_t)__builtin_bswap32(next_word[0]) << 32) | (uint64_t)__builtin_bswap32(next_word[1]);   // bits [base, base + 64), MSB first
  uint32_t nextw = next_word[2];                       // bits [base + 64, base + 96), still in memory byte order
  next_word += 3;
  int32_t down = 48 - (int32_t)(s.p & 31u);            // win >> down: the next 16 bits of the stream in bits 15..0; 17 <= down <= 48
  int32_t left = (int32_t)(end - s.p);                 // > 0 while the subsequence's end has not been crossed
  uint32_t c = s.cz >> 8, z = s.cz & 0xFFu;
  const uint32_t bpm = gray ? 1u : 6u, chroma_at = gray ? 0xFFu : 4u;
  const char* tables = reinterpret_cast<const char*>(&s_adv[0][0]);
  constexpr uint32_t kIndexMask = (1u << (kFastBits + 2u)) - 4u;   // byte offset of a 32-bit entry inside one table
  // The loop carries the table entry of the NEXT symbol: looked up at the end of a step, in the AC table of the same component
  // unless the step completed the block (then the DC table of the next component).  (Forcing the lookup to be issued before the
  // state update, under the guess "same AC table", measured no faster: 53.8 against 51.5 us per launch.)
  uint32_t tb = (c >= chroma_at ? 2u : 0u) + (z != 0u ? 1u : 0u);
  uint32_t x = (uint32_t)(win >> (uint32_t)down);
  uint32_t a = *reinterpret_cast<const uint32_t*>(tables + ((tb << (kFastBits + 2u)) | ((x >> (14u - kFastBits)) & kIndexMask)));
  do {
    if ((int32_t)a < 0) {                              // kLongEntry: no code of at most kFastBits bits
      const uint32_t e = long_code(lc, tb, x & 0xFFFFu);
      a = e != 0u ? adv_entry(tb, e >> 8, e & 0xFFu) : 0x0100u;   // no code: one bit consumed, index unchanged
    }
    const uint32_t pair = a >> 16;
    const bool both = pair != 0u && z + (a & 127u) < 64u && (int32_t)((a >> 8) & 63u) < left;
    a = both ? pair : a;
    const uint32_t bits = (a >> 8) & 63u;              // (at most 32: build_lut_body)
    left -= (int32_t)bits;
    down -= (int32_t)bits;
    if (down < 17) {                                   // the window's first word is used up
      win = (win << 32) | (uint64_t)__builtin_bswap32(nextw);
      down += 32;
      nextw = *next_word++;
    }
    x = (uint32_t)(win >> (uint32_t)down);
    const uint32_t index = (x >> (14u - kFastBits)) & kIndexMask;
    tb = (c >= chroma_at ? 3u : 1u);
    const uint32_t guess = *reinterpret_cast<const uint32_t*>(tables + ((tb << (kFastBits + 2u)) | index));
    z += a & 127u;
    const bool done = z >= 64u;
    z = done ? 0u : z;
    const uint32_t cn = c + 1u == bpm ? 0u : c + 1u;
    c = done ? cn : c;
    nb += done ? 1u : 0u;
    a = guess;
    if (done) {
      tb = c >= chroma_at ? 2u : 0u;
      a = *reinterpret_cast<const uint32_t*>(tables + ((tb << (kFastBits + 2u)) | index));
    }
  } while (left > 0);
  s.p = end - (uint32_t)left;
  s.cz = (c << 8) | z;
}

// The first pass: every subsequence from the guess (its own start, block 0, coefficient 0; the true state for the first one of the
// scan or of a restart interval).  Records the state each crosses its end with; every start state counts as new (dirty).
__device__ __forceinline__ void sync_first_body(const DecJob& j, const DecTables& tables, DState* next, uint8_t* dirty_out, uint32_t* nblocks) {
  __shared__ __attribute__((aligned(16))) uint32_t s_adv[4][1u << kFastBits];
  __shared__ __attribute__((aligned(16))) uint32_t s_bits[kStageWords];
  __shared__ LongCodes s_long;
  const uint32_t i0 = blockIdx.x * 256u, i = i0 + threadIdx.x;
  if (i0 >= j.nsub) return;
  const uint32_t w0 = stage_bits(j, i0, s_bits);
  load_long_codes(tables, s_long);
  load_fast_table(j.adv, s_adv);   // (ends with the barrier that also covers s_bits and s_long)
  if (i >= j.nsub) return;
  DState s;
  s.p = sub_begin(j, i); s.cz = 0u;
  uint32_t nb;
  decode_positions(j, s_long, s_adv, s_bits, w0, i, s, nb);
  next[i] = s;
  nblocks[i] = nb;            // blocks completed inside this subsequence, valid once its start state is the true one
  dirty_out[i + 1u] = 1;
}

// A launch of rounds.  In a round, every subsequence whose start state (the end state of its left neighbour) changed is decoded
// again from it; when nothing changes any more every start state is the true one.  A round is one lane's walk over 512 bits --
// latency, ~20 us, whatever the number of subsequences -- so the price of a launch around it (the gap between dependent kernels,
// the tables and the bits brought into LDS) is worth sharing: a workgroup runs kLocalRounds rounds on its 256 subsequences with
// the states in LDS.  Only its first lane depends on another workgroup (the end state of the subsequence in front, as of the
// previous launch), and only its last lane is waited for by one (dirty_out[i + 1]: "changed during this launch").
// dirty[i]: subsequence i has not been decoded from the current end state of i - 1 yet.
#ifndef UHDR_JD_LOCAL_ROUNDS
#define UHDR_JD_LOCAL_ROUNDS 4
#endif
constexpr uint32_t kLocalRounds = UHDR_JD_LOCAL_ROUNDS;
// launches the host enqueues before it looks at the ring (a quality-75 4K file needs 5-8 rounds, a quality-95 one 21-24)
// (counted in rounds of 512-bit subsequences; shorter subsequences need proportionally more)
constexpr uint32_t kRoundScale = 512u / kSubBits > 0u ? 512u / kSubBits : 1u;
// (A look at the ring leaves the device idle for ~35 us -- copy, host wake-up, next launch -- and a launch that returns at once
// costs ~7: so the first batch covers what a quality-95 file needs, and a file that needs a third of it pays four empty launches.)
constexpr uint32_t kFirstLaunches = 2u * ((24u * kRoundScale + 2u * kLocalRounds - 1u) / (2u * kLocalRounds));
constexpr uint32_t kMoreLaunches = 2u * ((16u * kRoundScale + 2u * kLocalRounds - 1u) / (2u * kLocalRounds));
__device__ __forceinline__ void sync_rounds_body(const DecJob& j, const DecTables& tables, const DState* prev, DState* next, const uint8_t* dirty_in,
                                                 uint8_t* dirty_out, uint32_t* nblocks, uint32_t* changed) {
  __shared__ __attribute__((aligned(16))) uint32_t s_adv[4][1u << kFastBits];
  __shared__ __attribute__((aligned(16))) uint32_t s_bits[kStageWords];
  __shared__ LongCodes s_long;
  __shared__ DState s_st[256];
  __shared__ uint8_t s_dirty[260];
  const uint32_t t = threadIdx.x, i0 = blockIdx.x * 256u, i = i0 + t;
  if (i0 >= j.nsub) return;
  const bool in = i < j.nsub;
  const bool live = in && !sub_is_first(j, i);          // a first subsequence starts from a state that is known
  const bool last = in && (t == 255u || i + 1u == j.nsub);
  DState mine; mine.p = 0u; mine.cz = 0u;
  if (in) mine = prev[i];
  const bool d0 = live && dirty_in[i] != 0;
  // a launch in which none of the workgroup's subsequences starts from a new state: nothing to decode, nothing to load
  if (!__syncthreads_or(d0 ? 1 : 0)) {
    if (in) {
      next[i] = mine;
      if (t != 0u) dirty_out[i] = 0;
      if (last) dirty_out[i + 1u] = 0;
    }
    return;
  }
  const uint32_t w0 = stage_bits(j, i0, s_bits);
  s_st[t] = mine;
  s_dirty[t] = d0 ? 1 : 0;
  DState start0; start0.p = 0u; start0.cz = 0u;
  if (t == 0u && live) start0 = prev[i - 1u];
  load_long_codes(tables, s_long);
  load_fast_table(j.adv, s_adv);   // (ends with the barrier that also covers s_bits, s_long, s_st and s_dirty)
  uint32_t nb = 0;
  bool decoded = false, changed_any = false;
  for (uint32_t r = 0; r < kLocalRounds; ++r) {
    const bool dd = live && s_dirty[t] != 0;
    DState s = start0;
    if (t != 0u) s = s_st[t - 1u];
    if (!__syncthreads_or(dd ? 1 : 0)) break;           // (the barrier also separates these reads from the writes below)
    bool ch = false;
    if (dd) {
      decode_positions(j, s_long, s_adv, s_bits, w0, i, s, nb);
      ch = s.p != mine.p || s.cz != mine.cz;
      mine = s;
      decoded = true;
      s_st[t] = s;
    }
    changed_any = changed_any || ch;
    if (t == 0u) s_dirty[0] = 0;                        // its start state cannot change during a launch
    s_dirty[t + 1u] = ch ? 1 : 0;                       // every entry is rewritten every round ([256]: nobody's)
    __syncthreads();
  }
  if (!in) return;
  next[i] = mine;
  if (decoded) nblocks[i] = nb;   // blocks completed inside this subsequence, valid once its start state is the true one
  if (t != 0u) dirty_out[i] = (live && s_dirty[t] != 0) ? 1 : 0;
  if (last) dirty_out[i + 1u] = changed_any ? 1 : 0;
  if (changed_any) *changed = 1u;
}

__device__ __forceinline__ void write_body(const DecJob& j, const DecTables& tables, const DState* st, const uint32_t* first_block, uint32_t* error) {
  __shared__ __attribute__((aligned(16))) uint32_t s_lut[4][1u << kFastBits];
  __shared__ __attribute__((aligned(16))) uint32_t s_bits[kStageWords];
  const uint32_t i0 = blockIdx.x * 256u, i = i0 + threadIdx.x;
  if (i0 >= j.nsub) return;
  __shared__ LongCodes s_long;
  const uint32_t w0 = stage_bits(j, i0, s_bits);
  load_long_codes(tables, s_long);
  load_fast_table(j.lut, s_lut);
  if (i >= j.nsub) return;
  DState s;
  if (sub_is_first(j, i)) { s.p = sub_begin(j, i); s.cz = 0u; } else s = st[i - 1u];
  const uint32_t end = sub_end_bit(j, i);
  // (as in decode_positions: a start outside the staged words is impossible with states that came out of the rounds; never read there)
  if ((s.p >> 5) < w0 || (s.p >> 5) - w0 + 3u >= kStageWords || (end > s.p && end - s.p > kSubBits + 32u)) { *error = 1u; return; }
  Reader rd;
  rd.init(s_bits, w0, s.p);
  // first_block: blocks completed before this subsequence, counted from the start of its restart interval (of the scan without them)
  uint32_t blk = first_block[i], blk_end = j.nblk;
  if (j.sub_key) {
    const uint32_t k = j.sub_key[i];
    blk += k * j.restart_blocks;
    blk_end = (k + 1u) * j.restart_blocks < j.nblk ? (k + 1u) * j.restart_blocks : j.nblk;
  }
  bool bd, has;
  uint32_t zp = 0;
  int v = 0;
  // The pointer comes out of a structure in memory, so the compiler would store through a FLAT instruction -- and those count as
  // LDS traffic too: the wait in front of the next symbol's table lookup would then sit out the scattered store of this one.
  // A global store is waited for by nothing in this loop (which is why the loop must not load from memory either).
  typedef __attribute__((address_space(1))) int16_t GlobalI16;
  GlobalI16* coef = (GlobalI16*)j.coef;
  const bool gray = j.gray != 0;
  bool bad = false;
  while (s.p < end && blk < blk_end) {   // (stops in front of the 1-bits that pad an interval to its byte boundary)
    const bool ok = step_coef(gray, s_long, s_lut, rd, s, bd, zp, v, has);
    bad = bad || !ok;
    if (has) coef[(size_t)blk * 64u + zp] = (int16_t)v;
    blk += bd;
  }
  if (bad || (sub_is_last(j, i) && (blk != blk_end || s.p > end))) *error = 1u;   // an interval (the scan) must end exactly after its last block
}

// ---- DC prediction: value = running sum of the differences of the same component ---------------------------------------
// one running sum per component, scanned together: a block contributes its difference to its own component's sum
struct Dc3 { int v[3]; };
struct Dc3Sum {
  __host__ __device__ Dc3 operator()(const Dc3& a, const Dc3& b) const { Dc3 r; r.v[0] = a.v[0] + b.v[0]; r.v[1] = a.v[1] + b.v[1]; r.v[2] = a.v[2] + b.v[2]; return r; }
};
struct BlkKey {   // restart interval a block belongs to: the DC predictors start from zero in each (T.81 F.1.1.5.1 / E.2.4)
  uint32_t per;
  __host__ __device__ uint32_t operator()(uint32_t b) const { return b / per; }
};
// ---- prefix sums over the concatenated per-image arrays of a batch: one segmented scan instead of one scan per image ----------
struct DecBatchJob;
struct SegOf {   // which image element g of a concatenation belongs to; off: n + 1 ascending element offsets (device memory)
  const uint32_t* off;
  int n;
  __host__ __device__ uint32_t operator()(uint32_t g) const {
    int lo = 0, hi = n;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (off[mid] <= g) lo = mid; else hi = mid; }
    return (uint32_t)lo;
  }
};

// ---- dequantisation + IDCT -----------------------------------------------------------------------------------------
__device__ __forceinline__ int dscale(int x, int n) { return (x + (1 << (n - 1))) >> n; }
template <int PASS>
__device__ __forceinline__ void idct8(int (&d)[64], int base, int stride) {   // libjpeg jidctint.c
  int z2 = d[base + 2 * stride], z3 = d[base + 6 * stride];
  int z1 = (z2 + z3) * 4433;
  int tmp2 = z1 + z3 * (-15137), tmp3 = z1 + z2 * 6270;
  z2 = d[base]; z3 = d[base + 4 * stride];
  int tmp0 = (z2 + z3) << 13, tmp1 = (z2 - z3) << 13;
  const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
  tmp0 = d[base + 7 * stride]; tmp1 = d[base + 5 * stride]; tmp2 = d[base + 3 * stride]; tmp3 = d[base + stride];
  z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
  int z4 = tmp1 + tmp3;
  const int z5 = (z3 + z4) * 9633;
  tmp0 *= 2446; tmp1 *= 16819; tmp2 *= 25172; tmp3 *= 12299;
  z1 *= -7373; z2 *= -20995; z3 *= -16069; z4 *= -3196;
  z3 += z5; z4 += z5;
  tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
  constexpr int sh = PASS == 0 ? 11 : 18;
  d[base] = dscale(tmp10 + tmp3, sh); d[base + 7 * stride] = dscale(tmp10 - tmp3, sh);
  d[base + stride] = dscale(tmp11 + tmp2, sh); d[base + 6 * stride] = dscale(tmp11 - tmp2, sh);
  d[base + 2 * stride] = dscale(tmp12 + tmp1, sh); d[base + 5 * stride] = dscale(tmp12 - tmp1, sh);
  d[base + 3 * stride] = dscale(tmp13 + tmp0, sh); d[base + 4 * stride] = dscale(tmp13 - tmp0, sh);
}

__device__ __forceinline__ void idct_body(const DecJob& j, const Dc3* dc) {
  const uint32_t b = blockIdx.x * 128u + threadIdx.x;
  if (b >= j.nblk) return;
  int comp, br, bc;
  if (j.gray) { comp = 0; br = (int)(b / j.mcus_x); bc = (int)(b - (uint32_t)br * j.mcus_x); }
  else {
    const uint32_t mcu = b / 6u, k = b - mcu * 6u;
    const int mr = (int)(mcu / j.mcus_x), mc = (int)(mcu - (uint32_t)mr * j.mcus_x);
    if (k < 4u) { comp = 0; br = 2 * mr + (int)(k >> 1); bc = 2 * mc + (int)(k & 1u); }
    else { comp = (int)k - 3; br = mr; bc = mc; }
  }
  const DecPlane& pl = j.plane[comp];
  if (br * 8 >= pl.h || bc * 8 >= pl.w) return;      // a dummy block of the encoder: nothing of it is inside the image
  const int dc_value = (int)(int16_t)dc[b].v[comp];  // the running sum of the component's differences (the coefficient array holds the difference)
  const uint16_t* q = j.quant[comp];                 // zigzag order
  constexpr uint8_t nat[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                               41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                               30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
  int d[64];
  const uint4* src = reinterpret_cast<const uint4*>(j.coef + (size_t)b * 64u);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const uint4 v = src[k];
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      d[nat[8 * k + 2 * m]] = (int)(int16_t)(w[m] & 0xffffu) * (int)q[8 * k + 2 * m];
      d[nat[8 * k + 2 * m + 1]] = (int)(int16_t)(w[m] >> 16) * (int)q[8 * k + 2 * m + 1];
    }
  }
  d[0] = dc_value * (int)q[0];
#pragma unroll
  for (int c = 0; c < 8; ++c) idct8<0>(d, c, 8);
#pragma unroll
  for (int r = 0; r < 8; ++r) idct8<1>(d, r * 8, 1);
  uint8_t* dst = pl.p + (size_t)(br * 8) * pl.stride + bc * 8;
  const bool whole = br * 8 + 8 <= pl.h && bc * 8 + 8 <= pl.w && pl.aligned8;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    uint32_t o[2] = {0u, 0u};
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      int x = d[r * 8 + c] + 128;
      x = x < 0 ? 0 : (x > 255 ? 255 : x);
      o[c >> 2] |= (uint32_t)x << (8 * (c & 3));
    }
    if (whole) *reinterpret_cast<uint2*>(dst + (size_t)r * pl.stride) = make_uint2(o[0], o[1]);
    else if (br * 8 + r < pl.h)
      for (int c = 0; c < 8; ++c)
        if (bc * 8 + c < pl.w) dst[(size_t)r * pl.stride + c] = (uint8_t)(o[c >> 2] >> (8 * (c & 3)));
  }
}

// ---- host side -----------------------------------------------------------------------------------------------------
// parse_header: csrc/uhdr_jpeg_hdr.cpp (plain C++, so that it can be fuzzed under AddressSanitizer on the CPU)

// ---- the kernels: one image per blockIdx.y of a launch, the jobs sit in device memory -- so that a batch of files costs the
// launches of one image (decode_device_batch); the bodies above take blockIdx.x / threadIdx.x as "their" image's grid --------------
struct DecBatchJob {
  DecJob j;
  const uint8_t* src; uint32_t src_bytes; int rst;     // unstuffing: stuffed segment -> j.raw
  uint32_t* kept; uint32_t* kept_blk; uint8_t* raw_out;
  DecTables tables; uint32_t* lut_out; uint32_t* adv_out;
  DState* st[2]; uint8_t* dirty[2]; uint32_t* nblocks; uint32_t* flags;
  const uint32_t* first_block; const Dc3* dc;
  uint8_t* zero[3]; uint32_t zero_words[3];            // ranges k_jd_prepare_multi clears (16-byte multiples)
};
struct SubKeyBatch {   // (image, restart interval) of subsequence g of the concatenated nblocks arrays
  const DecBatchJob* jobs;
  SegOf seg;
  __host__ __device__ uint64_t operator()(uint32_t g) const {
    const uint32_t img = seg(g);
    const uint32_t* sk = jobs[img].j.sub_key;
    return ((uint64_t)img << 32) | (sk ? sk[g - seg.off[img]] : 0u);
  }
};
struct BlkKeyBatch {   // (image, restart interval) of block g of the concatenated DC arrays
  const DecBatchJob* jobs;
  SegOf seg;
  __host__ __device__ uint64_t operator()(uint32_t g) const {
    const uint32_t img = seg(g), rb = jobs[img].j.restart_blocks;
    return ((uint64_t)img << 32) | (rb ? (g - seg.off[img]) / rb : 0u);
  }
};
struct DcPickBatch {   // the DC difference of block g of the concatenation, in its component's slot
  const DecBatchJob* jobs;
  SegOf seg;
  __host__ __device__ Dc3 operator()(uint32_t g) const {
    const uint32_t img = seg(g), b = g - seg.off[img];
    const DecJob& j = jobs[img].j;
    const int c = j.gray ? 0 : ((b % 6u) < 4u ? 0 : (int)(b % 6u) - 3);
    Dc3 r; r.v[0] = r.v[1] = r.v[2] = 0;
    r.v[c] = (int)j.coef[(size_t)b * 64u];
    return r;
  }
};
typedef rocprim::counting_iterator<uint32_t> CountIt;
typedef rocprim::transform_iterator<CountIt, SubKeyBatch, uint64_t> SubKeyIt;
typedef rocprim::transform_iterator<CountIt, BlkKeyBatch, uint64_t> BlkKeyIt;
typedef rocprim::transform_iterator<CountIt, DcPickBatch, Dc3> DcPickIt;

// What depends on nothing but the uploaded jobs, in one launch (a launch and the gap behind it cost more than any of the three):
// workgroups [0, gu) count the bytes each 64-byte chunk of the stuffed segment keeps, the next kLutBlocks build the first-level
// tables, the last kZeroBlocks clear the flags, the buffer of the bit string and the coefficients.
constexpr uint32_t kLutBlocks = (4u << kFastBits) / 256u, kZeroBlocks = 512u;
__global__ void __launch_bounds__(256) k_jd_prepare_multi(const DecBatchJob* jobs, uint32_t gu) {
  const DecBatchJob& b = jobs[blockIdx.y];
  if (blockIdx.x < gu) {
    if (b.src_bytes == 0u) return;   // an image that failed on the host: it owns no slice of the batch arrays
    unstuff_count_body(b.src, b.src_bytes, b.kept, b.kept_blk, b.rst);
  } else if (blockIdx.x < gu + kLutBlocks) {
    build_lut_body(b.tables, b.lut_out, b.adv_out, blockIdx.x - gu);
  } else {
    const uint32_t first = (blockIdx.x - gu - kLutBlocks) * 256u + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      uint4* p = reinterpret_cast<uint4*>(b.zero[k]);
      const uint32_t n16 = b.zero_words[k];
      for (uint32_t i = first; i < n16; i += kZeroBlocks * 256u) p[i] = make_uint4(0u, 0u, 0u, 0u);
    }
  }
}
__global__ void __launch_bounds__(256) k_jd_unstuff_copy_multi(const DecBatchJob* jobs) { const DecBatchJob& b = jobs[blockIdx.y]; unstuff_copy_body(b.src, b.src_bytes, b.kept, b.kept_blk, b.raw_out, b.rst); }
// parity: which of the two state / dirty buffers is read (the other one is written).  round: number of the launch, 1, 2, ... --
// words 64..127 of an image's flags are a ring of "launch r changed an end state".  A launch that follows one without a change has
// nothing to do (both state buffers are equal by then) and returns at once, image by image: the host enqueues launches without
// knowing how many a file needs, and the surplus costs launches, not decodes.  Every launch clears the ring slot of the next one.
constexpr uint32_t kFlagWords = 128;   // per image: [1] = corrupt data, [64, 128) = the ring
template <int ROUND>
__global__ void __launch_bounds__(256) k_jd_sync_multi(const DecBatchJob* jobs, int parity, uint32_t round) {
  const DecBatchJob& b = jobs[blockIdx.y];
  uint32_t* ring = b.flags + 64;
  if (ROUND == 0) { sync_first_body(b.j, b.tables, b.st[0], b.dirty[0], b.nblocks); return; }
  if (blockIdx.x == 0u && threadIdx.x == 0u) ring[(round + 1u) & 63u] = 0u;
  if (round >= 2u && ring[(round - 1u) & 63u] == 0u) return;
  sync_rounds_body(b.j, b.tables, b.st[parity], b.st[parity ^ 1], b.dirty[parity], b.dirty[parity ^ 1], b.nblocks, ring + (round & 63u));
}
__global__ void __launch_bounds__(256) k_jd_write_multi(const DecBatchJob* jobs) { const DecBatchJob& b = jobs[blockIdx.y]; write_body(b.j, b.tables, b.st[0], b.first_block, b.flags + 1); }
__global__ void __launch_bounds__(128) k_jd_idct_multi(const DecBatchJob* jobs) { const DecBatchJob& b = jobs[blockIdx.y]; idct_body(b.j, b.dc); }

size_t dec_workspace_bytes(const DecInfo& info, DecLayout* l) {
  auto up = [](size_t v) { return (v + 255) / 256 * 256; };
  const uint32_t w = (uint32_t)info.w, h = (uint32_t)info.h;
  l->mcus_x = info.gray ? (w + 7) / 8 : (w + 15) / 16;
  const uint32_t mcus_y = info.gray ? (h + 7) / 8 : (h + 15) / 16;
  l->nblk = l->mcus_x * mcus_y * (info.gray ? 1u : 6u);
  const size_t nbytes = info.scan_bytes;
  l->nchunks = (uint32_t)((nbytes + kUnstuffChunk - 1) / kUnstuffChunk);
  const uint32_t nint = (uint32_t)info.interval_start.size();   // 0 without restart intervals; each interval may end in a short subsequence
  l->nsub_max = (uint32_t)((nbytes * 8 + kSubBits - 1) / kSubBits) + 1u + nint;
  // per image: the stuffed and the unstuffed segment, the Huffman tables, the two state / dirty buffers of the rounds, the
  // coefficients; the arrays the prefix sums run over belong to the batch (batch_layout)
  size_t o = 0;
  l->src = o; o += up(nbytes + 16);
  l->raw = o; o += up(nbytes + 64);
  l->lut = o; o += up((size_t)(4u << kFastBits) * 4);   // the first-level tables
  l->adv = o; o += up((size_t)(4u << kFastBits) * 4);
  l->st_a = o; o += up((size_t)l->nsub_max * sizeof(DState));
  l->st_b = o; o += up((size_t)l->nsub_max * sizeof(DState));
  l->dirty_a = o; o += up((size_t)l->nsub_max + 2);
  l->dirty_b = o; o += up((size_t)l->nsub_max + 2);
  l->coef = o; o += up((size_t)l->nblk * 128);
  l->sub_start = l->sub_end = l->sub_key = 0;
  if (nint) {
    l->sub_start = o; o += up((size_t)l->nsub_max * 4);
    l->sub_end = o; o += up((size_t)l->nsub_max * 4);
    l->sub_key = o; o += up((size_t)(l->nsub_max + 1) * 4);
  }
  return o;
}

// where the batch-level arrays sit inside the scratch buffer
struct BatchLayout {
  size_t jobs, flags, offs, kept, kept_blk, nblocks, first_block, dc, tmp, tmp_bytes, total;
  uint32_t n_kept, n_sub, n_blk;
};
static BatchLayout batch_layout(int n, const DecLayout l[]) {
  auto up = [](size_t v) { return (v + 255) / 256 * 256; };
  BatchLayout B;
  B.n_kept = B.n_sub = B.n_blk = 0;
  for (int k = 0; k < n; ++k) { B.n_kept += l[k].nchunks + 1u; B.n_sub += l[k].nsub_max + 1u; B.n_blk += l[k].nblk; }
  size_t o = 0;
  B.jobs = o; o += (size_t)n * sizeof(DecBatchJob);        // the jobs and, right behind them, the three offset tables: one upload
  B.offs = o; o += up((size_t)3 * (n + 1) * 4 + 256);
  o = up(o);
  B.flags = o; o += up((size_t)n * kFlagWords * 4);
  B.kept = o; o += up((size_t)B.n_kept * 4);
  B.kept_blk = o; o += up(((size_t)B.n_kept / 256u + (size_t)n + 1u) * 4);
  B.nblocks = o; o += up((size_t)B.n_sub * 4);
  B.first_block = o; o += up((size_t)B.n_sub * 4);
  B.dc = o; o += up((size_t)B.n_blk * sizeof(Dc3) + 16);
  size_t t2 = 0, t3 = 0;
  const SegOf seg{nullptr, n};
  CountIt cnt(0u);
  (void)rocprim::exclusive_scan_by_key(nullptr, t2, SubKeyIt(cnt, SubKeyBatch{nullptr, seg}), (const uint32_t*)nullptr, (uint32_t*)nullptr, 0u, (size_t)B.n_sub,
                                        rocprim::plus<uint32_t>(), rocprim::equal_to<uint64_t>());
  (void)rocprim::inclusive_scan_by_key(nullptr, t3, BlkKeyIt(cnt, BlkKeyBatch{nullptr, seg}), DcPickIt(cnt, DcPickBatch{nullptr, seg}), (Dc3*)nullptr,
                                        (size_t)(B.n_blk ? B.n_blk : 1u), Dc3Sum(), rocprim::equal_to<uint64_t>());
  B.tmp_bytes = up(std::max(t2, t3) + 256);
  B.tmp = o; o += B.tmp_bytes;
  B.total = o;
  return B;
}
size_t dec_batch_scratch_bytes(int n, const DecLayout l[]) { return batch_layout(n, l).total; }

int decode_device_batch(int n, const DecInfo* const info[], const DecLayout l[], uint8_t* const ws[], DecPlane (*planes[])[3], hipStream_t s,
                        uint8_t* batch_ws, hipError_t* herr, int* image_rc) {
#define JD_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { *herr = _e; (void)hipStreamSynchronize(s); return 1; } } while (0)
  if (n < 1) return -1;
#ifdef UHDR_JD_TIMING
  auto TT = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) { const auto t = std::chrono::steady_clock::now(); fprintf(stderr, "[jd]   %-28s %.0f us\n", what, std::chrono::duration<double, std::micro>(t - TT).count()); TT = t; };
#define JD_LAP(x) lap(x)
#else
#define JD_LAP(x) do {} while (0)
#endif
  // jobs and offset tables are assembled in page-locked memory that lives across calls (per host thread): their upload is then a
  // real asynchronous copy and needs no synchronisation of its own (every call ends with the stream idle, so reuse is safe)
  static thread_local void* t_pinned = nullptr;
  static thread_local size_t t_pinned_cap = 0;
  const size_t pinned_need = (size_t)n * sizeof(DecBatchJob) + (size_t)3 * (n + 1) * 4 + 64;
  if (pinned_need > t_pinned_cap) {
    if (t_pinned) (void)hipHostFree(t_pinned);
    t_pinned = nullptr; t_pinned_cap = 0;
    JD_TRY(hipHostMalloc(&t_pinned, pinned_need * 2, hipHostMallocDefault));
    t_pinned_cap = pinned_need * 2;
  }
  DecBatchJob* jobs = static_cast<DecBatchJob*>(t_pinned);
  std::vector<int> bad((size_t)n, 0);
  std::vector<std::vector<uint32_t>> keep;   // restart-interval tables: alive until the uploads have happened
  const BatchLayout B = batch_layout(n, l);
  DecBatchJob* djobs = reinterpret_cast<DecBatchJob*>(batch_ws + B.jobs);
  uint32_t* dflags = reinterpret_cast<uint32_t*>(batch_ws + B.flags);   // kFlagWords per image
  uint32_t* doffs = reinterpret_cast<uint32_t*>(batch_ws + B.offs);      // three offset tables of n + 1 entries: kept, subsequences, blocks
  uint32_t* offs = reinterpret_cast<uint32_t*>(jobs + n);
  const size_t noffs = (size_t)3 * (n + 1);
  uint32_t* koff = offs; uint32_t* soff = koff + (n + 1); uint32_t* boff = soff + (n + 1);
  koff[0] = soff[0] = boff[0] = 0u;
  uint32_t gu = 1, gsync = 1, gidct = 1;
  for (int k = 0; k < n; ++k) {
    DecBatchJob& b = jobs[k];
    memset(&b, 0, sizeof(b));
    const DecInfo& in = *info[k];
    const DecLayout& L = l[k];
    uint8_t* w = ws[k];
    b.src = w + L.src; b.src_bytes = (uint32_t)in.scan_bytes; b.rst = in.restart_interval != 0 ? 1 : 0;
    // the arrays the prefix sums run over are slices of batch-level concatenations (one segmented scan for all images)
    b.kept = reinterpret_cast<uint32_t*>(batch_ws + B.kept) + koff[k]; b.kept_blk = reinterpret_cast<uint32_t*>(batch_ws + B.kept_blk) + koff[k] / 256u + (uint32_t)k;
    b.raw_out = w + L.raw;
    b.tables = in.tables; b.lut_out = reinterpret_cast<uint32_t*>(w + L.lut); b.adv_out = reinterpret_cast<uint32_t*>(w + L.adv);
    b.st[0] = reinterpret_cast<DState*>(w + L.st_a); b.st[1] = reinterpret_cast<DState*>(w + L.st_b);
    b.dirty[0] = w + L.dirty_a; b.dirty[1] = w + L.dirty_b;
    b.nblocks = reinterpret_cast<uint32_t*>(batch_ws + B.nblocks) + soff[k];
    b.flags = dflags + kFlagWords * (uint32_t)k;
    b.first_block = reinterpret_cast<const uint32_t*>(batch_ws + B.first_block) + soff[k];
    b.dc = reinterpret_cast<const Dc3*>(batch_ws + B.dc) + boff[k];
    DecJob& j = b.j;
    j.raw = reinterpret_cast<const uint32_t*>(w + L.raw);
    j.lut = b.lut_out; j.adv = b.adv_out;
    j.total_bits = in.raw_bytes * 8u;
    j.nsub = (j.total_bits + kSubBits - 1u) / kSubBits;
    j.gray = in.gray; j.nblk = L.nblk; j.mcus_x = L.mcus_x;
    j.dc_tbl[0] = 0; j.ac_tbl[0] = 1; j.dc_tbl[1] = 2; j.ac_tbl[1] = 3;
    j.coef = reinterpret_cast<int16_t*>(w + L.coef);
    for (int c = 0; c < 3; ++c) { j.plane[c] = (*planes[k])[c]; memcpy(j.quant[c], in.quant[c], sizeof(j.quant[c])); }
    if (b.rst) {
      const size_t nint = in.interval_start.size();
      std::vector<uint32_t> sb_, se_, sk_;
      for (size_t q = 0; q < nint && !bad[k]; ++q) {
        const uint32_t b0 = in.interval_start[q] * 8u, b1 = (q + 1 < nint ? in.interval_start[q + 1] : in.raw_bytes) * 8u;
        if (b1 <= b0) { bad[k] = 1; break; }
        for (uint32_t x = b0; x < b1; x += kSubBits) { sb_.push_back(x); se_.push_back(x + kSubBits < b1 ? x + kSubBits : b1); sk_.push_back((uint32_t)q); }
      }
      j.nsub = (uint32_t)sb_.size();
      if (!bad[k] && j.nsub != 0u && j.nsub <= L.nsub_max) {
        sk_.push_back(0xFFFFFFFFu);
        JD_TRY(hipMemcpyAsync(w + L.sub_start, sb_.data(), sb_.size() * 4, hipMemcpyHostToDevice, s));
        JD_TRY(hipMemcpyAsync(w + L.sub_end, se_.data(), se_.size() * 4, hipMemcpyHostToDevice, s));
        JD_TRY(hipMemcpyAsync(w + L.sub_key, sk_.data(), sk_.size() * 4, hipMemcpyHostToDevice, s));
        keep.push_back(std::move(sb_)); keep.push_back(std::move(se_)); keep.push_back(std::move(sk_));
        j.sub_start = reinterpret_cast<const uint32_t*>(w + L.sub_start);
        j.sub_end = reinterpret_cast<const uint32_t*>(w + L.sub_end);
        j.sub_key = reinterpret_cast<const uint32_t*>(w + L.sub_key);
        j.restart_blocks = in.restart_interval * (in.gray ? 1u : 6u);
      }
    }
    // a progressive file arrives with its coefficients decoded (uhdr_jpeg_prog.cpp): no segment, no subsequences -- every
    // entropy-decoding kernel's bounds check skips the image; its coefficients are uploaded behind the clearing kernel below and
    // the DC prefix sum and the IDCT treat it like any other image
    const bool prog = in.progressive;
    if (prog) {
      j.nsub = 0u; j.total_bits = 0u; b.src_bytes = 0u;
      if (in.coef.size() != (size_t)L.nblk * 64u) bad[k] = 1;
    } else if (j.nsub == 0u || j.nsub > L.nsub_max) bad[k] = 1;
    if (bad[k]) { j.nsub = 0u; j.nblk = 0u; b.src_bytes = 0u; }   // every kernel's bounds check then skips this image
    b.zero[0] = reinterpret_cast<uint8_t*>(b.flags); b.zero_words[0] = kFlagWords / 4u;
    b.zero[1] = w + L.raw; b.zero_words[1] = bad[k] ? 0u : (uint32_t)((((size_t)in.scan_bytes + 64 + 255) / 256 * 256) / 16);
    b.zero[2] = reinterpret_cast<uint8_t*>(j.coef); b.zero_words[2] = prog ? 0u : (uint32_t)(((size_t)j.nblk * 128) / 16);
    koff[k + 1] = koff[k] + (bad[k] ? 0u : L.nchunks + 1u);
    soff[k + 1] = soff[k] + j.nsub;
    boff[k + 1] = boff[k] + j.nblk;
    gu = std::max(gu, (L.nchunks + 255u) / 256u);
    gsync = std::max(gsync, (j.nsub + 255u) / 256u);
    gidct = std::max(gidct, (j.nblk + 127u) / 128u);
  }
  JD_TRY(hipMemcpyAsync(djobs, jobs, (size_t)n * sizeof(DecBatchJob) + noffs * 4, hipMemcpyHostToDevice, s));   // jobs + offset tables (adjacent on both sides)
  if (!keep.empty()) JD_TRY(hipStreamSynchronize(s));   // restart tables are in pageable host memory
  JD_LAP("jobs assembled + uploaded");
  const SegOf sseg{doffs + (n + 1), n}, bseg{doffs + 2 * (n + 1), n};
  const CountIt cnt0(0u);
  uint8_t* stmp = batch_ws + B.tmp;
  keep.clear();
  const dim3 b256(256);
  const unsigned ny = (unsigned)n;
  hipLaunchKernelGGL(k_jd_prepare_multi, dim3(gu + kLutBlocks + kZeroBlocks, ny), b256, 0, s, (const DecBatchJob*)djobs, gu);
  for (int k = 0; k < n; ++k)   // progressive images: the host's coefficients (pageable memory: the copy returns when it has been staged)
    if (info[k]->progressive && !bad[k])
      JD_TRY(hipMemcpyAsync(jobs[k].j.coef, info[k]->coef.data(), info[k]->coef.size() * sizeof(int16_t), hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_jd_unstuff_copy_multi, dim3(gu, ny), b256, 0, s, (const DecBatchJob*)djobs);
  hipLaunchKernelGGL(k_jd_sync_multi<0>, dim3(gsync, ny), b256, 0, s, (const DecBatchJob*)djobs, 0, 0u);
  JD_LAP("enqueued through sync<0>");
  std::vector<uint32_t> flags((size_t)n * kFlagWords, 0u);
  uint32_t max_nsub = 0;
  for (int k = 0; k < n; ++k) max_nsub = std::max(max_nsub, jobs[k].j.nsub);
  uint32_t round = 1;
  for (uint32_t batch = kFirstLaunches;; batch = kMoreLaunches) {   // (even numbers)
    if (round > max_nsub + 34u) { (void)hipStreamSynchronize(s); return -1; }   // cannot happen: every launch fixes at least one more subsequence
    for (uint32_t r = 0; r < batch; r += 2, round += 2) {   // odd rounds read buffer 0, even ones buffer 1: the result is in buffer 0
      hipLaunchKernelGGL(k_jd_sync_multi<1>, dim3(gsync, ny), b256, 0, s, (const DecBatchJob*)djobs, 0, round);
      hipLaunchKernelGGL(k_jd_sync_multi<1>, dim3(gsync, ny), b256, 0, s, (const DecBatchJob*)djobs, 1, round + 1u);
    }
    JD_TRY(hipMemcpyAsync(flags.data(), dflags, flags.size() * 4, hipMemcpyDeviceToHost, s));
    JD_TRY(hipStreamSynchronize(s));
    JD_LAP("round batch synced");
#ifdef UHDR_JD_TIMING
    for (int k = 0; k < n; ++k) {   // which launches of rounds still changed an end state
      uint32_t last = 0;
      for (uint32_t r = 1; r < round; ++r) if (flags[(size_t)k * kFlagWords + 64u + (r & 63u)] != 0u) last = r;
      fprintf(stderr, "[jd]   image %d: %u subsequences, last launch of rounds that changed a state: %u of %u\n", k, jobs[k].j.nsub, last, round - 1u);
    }
#endif
    bool any = false;
    for (int k = 0; k < n; ++k) any = any || flags[(size_t)k * kFlagWords + 64u + ((round - 1u) & 63u)] != 0u;
    if (!any) break;
  }
  if (soff[n]) {   // blocks completed before each subsequence, counted from the start of its image (of its restart interval)
    size_t tmp = B.tmp_bytes;
    JD_TRY(rocprim::exclusive_scan_by_key(stmp, tmp, SubKeyIt(cnt0, SubKeyBatch{djobs, sseg}), reinterpret_cast<const uint32_t*>(batch_ws + B.nblocks),
                                          reinterpret_cast<uint32_t*>(batch_ws + B.first_block), 0u, (size_t)soff[n], rocprim::plus<uint32_t>(),
                                          rocprim::equal_to<uint64_t>(), s));
  }
  hipLaunchKernelGGL(k_jd_write_multi, dim3(gsync, ny), b256, 0, s, (const DecBatchJob*)djobs);
  if (boff[n]) {   // DC differences -> DC values: per component, image and restart interval
    size_t tmp = B.tmp_bytes;
    JD_TRY(rocprim::inclusive_scan_by_key(stmp, tmp, BlkKeyIt(cnt0, BlkKeyBatch{djobs, bseg}), DcPickIt(cnt0, DcPickBatch{djobs, bseg}),
                                          reinterpret_cast<Dc3*>(batch_ws + B.dc), (size_t)boff[n], Dc3Sum(), rocprim::equal_to<uint64_t>(), s));
  }
  hipLaunchKernelGGL(k_jd_idct_multi, dim3(gidct, ny), dim3(128), 0, s, (const DecBatchJob*)djobs);
  JD_LAP("tail enqueued");
  JD_TRY(hipMemcpyAsync(flags.data(), dflags, flags.size() * 4, hipMemcpyDeviceToHost, s));
  JD_TRY(hipStreamSynchronize(s));
  JD_LAP("final sync");
  JD_TRY(hipGetLastError());
  int out = 0;
  for (int k = 0; k < n; ++k) {
    if (bad[k] || flags[kFlagWords * (size_t)k + 1u]) { bad[k] = 1; out = -1; }
    if (image_rc) image_rc[k] = bad[k] ? -1 : 0;
  }
  return out;
#undef JD_TRY
}

}  // namespace jpeg
}  // namespace uhdr
