// uhdr_jpeg.h -- internal interface of the GPU baseline-JPEG encoder (uhdr_jpeg.hip)
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include <vector>

namespace uhdr {
namespace jpeg {

struct Plane {
  const uint8_t* p;
  int w, h, stride;
  int pad_cols;   // columns >= w read as zero (the reference copies the row into a zero-padded buffer) instead of from p
  int aligned4;   // p and stride are multiples of 4: whole-dword row loads
};
struct Job {
  Plane plane[3];          // Y, Cb, Cr (one plane: only [0])
  uint16_t q_lum[64], q_chr[64];   // quantisation tables in zigzag order
  uint32_t m_lum[64], m_chr[64];   // floor(2^32 / (q << 3)) + 1: n / (q << 3) == mulhi(n, m) for n < 2^16 (exact: n * (q << 3) < 2^32)
  uint32_t nblk;           // blocks in entropy-coding order, dummy edge blocks included
  uint32_t ybw, ybh;       // luma size in blocks
  uint32_t mcus_x;         // MCUs per row (4:2:0)
  int gray;
  // workspace (device)
  int16_t* coef;           // nblk x 64, zigzag order
  uint32_t* bits;          // nblk (+1 zero)
  uint32_t* bits_blk;      // per workgroup of 128 blocks: their bits
  uint64_t* total_bits;    // the stream's length in bits (written by k_jpeg_emit)
  uint32_t* stream;        // packed entropy-coded bits before byte stuffing, big-endian words
  uint32_t* ff_count;      // per 64-byte chunk of the stream
  uint32_t* ff_blk;
  uint32_t max_chunks;
};
struct Layout {
  size_t coef, bits, bits_blk, stream, stream_bytes, ff_count, ff_blk, totals, scan_tmp, scan_tmp_bytes;
  uint32_t max_chunks;
};

hipError_t upload_tables();
void quant_table(int quality, bool chroma, uint16_t out_natural[64]);
void zigzag_table(const uint16_t natural[64], uint16_t zz[64]);
void build_header(int w, int h, bool gray, int quality, const void* icc, size_t icc_n, std::vector<uint8_t>& out);
size_t workspace_bytes(uint32_t nblk, Layout* l);
// enqueues the whole encoder; the JPEG (without the header, which the caller places at out[0, header_len)) lands at
// out + header_len, its total size (header included) in the uint64 at ws + l.totals + 8
// out_size: where the kernel that appends EOI reports the file's size (device-accessible; nullptr: word [1] of the workspace's totals)
hipError_t encode_async(Job j, const Layout& l, uint8_t* ws, uint8_t* out, uint64_t out_cap, uint64_t header_len, hipStream_t s,
                        uint64_t* out_size = nullptr);

// ---- decoder (uhdr_jpeg_dec.hip) -------------------------------------------------------------------------------------
struct HuffSpec {            // one DHT table in canonical form (T.81 Annex C)
  uint16_t first_code[17];   // first code of each length (index = length)
  uint16_t first_val[17];    // index into vals of that code
  uint16_t count[17];
  uint8_t vals[256];
  int present;
};
struct DecTables { HuffSpec huff[4]; };   // [0] DC luma, [1] AC luma, [2] DC chroma, [3] AC chroma
struct DecPlane {
  uint8_t* p;
  int w, h, stride;
  int aligned8;
};
struct DecInfo {
  int w = 0, h = 0, gray = 0;
  uint16_t quant[3][64] = {};     // per component, zigzag order (as stored in the file)
  DecTables tables = {};
  int td[2] = {0, 0}, ta[2] = {0, 0};
  size_t scan_offset = 0, scan_bytes = 0;   // the entropy-coded segment inside the file
  // restart intervals (DRI / RSTn, T.81 B.2.4.4 + E.2.4): MCUs per interval (0 = none) and, from a host scan of the segment, where
  // each interval starts in the UNSTUFFED stream (stuffed zeros and the RSTn markers themselves removed), and that stream's length
  uint32_t restart_interval = 0;
  std::vector<uint32_t> interval_start;
  uint32_t raw_bytes = 0;
  // progressive files (SOF2): every scan is entropy-decoded on the host (uhdr_jpeg_prog.cpp); coef = nblk x 64 coefficients, zigzag
  // order, blocks in the device decoder's order, DC as the difference to the component's previous block; scan_bytes = 0 and
  // scan_offset = the position of EOI.  The device runs its DC prefix sum and the IDCT.
  bool progressive = false;
  std::vector<int16_t> coef;
};
struct DecLayout {
  size_t src, raw, lut, adv, st_a, st_b, dirty_a, dirty_b, coef;
  size_t sub_start, sub_end, sub_key;   // restart-interval files only
  uint32_t nchunks, nsub_max, nblk, mcus_x;
};
struct DecJob {
  const uint32_t* raw;       // unstuffed entropy-coded bits, big-endian words, zero padded
  const uint32_t* lut;       // 4 first-level tables of the final pass (coef_entry), see build_lut_body
  const uint32_t* adv;       // 4 first-level tables of position-only entries (one symbol | two symbols)
  uint32_t total_bits, nsub, nblk, mcus_x;
  // restart-interval files: subsequence i covers bits [sub_start[i], sub_end[i]) of restart interval sub_key[i] (it never spans
  // two intervals); restart_blocks = blocks per interval.  NULL / 0: one interval, subsequence i = bits [512 i, 512 (i + 1))
  const uint32_t* sub_start;
  const uint32_t* sub_end;
  const uint32_t* sub_key;
  uint32_t restart_blocks;
  int gray;
  uint32_t dc_tbl[2], ac_tbl[2];
  int16_t* coef;             // nblk x 64, zigzag order
  DecPlane plane[3];
  uint16_t quant[3][64];
};
// 0 ok, -1 malformed, -2 outside what this decoder (or the reference: sampling) supports
int parse_header(const uint8_t* jpg, size_t n, DecInfo* info);
// the same for a progressive file, all scans decoded (parse_header calls it when it meets SOF2)
int decode_progressive(const uint8_t* jpg, size_t n, DecInfo* info);
// the walk over an entropy-coded segment on its own (the container scan uses it): position of the 0xFF of the first marker at or
// behind `e` that is neither a stuffed zero, a fill byte nor an RSTn; n if there is none
size_t skip_entropy_coded(const uint8_t* p, size_t e, size_t n);
size_t dec_workspace_bytes(const DecInfo& info, DecLayout* l);
// n images on one stream with one launch per decoder step for all of them (blockIdx.y = image); batch_ws: device scratch of
// dec_batch_scratch_bytes(n, layouts)
size_t dec_batch_scratch_bytes(int n, const DecLayout l[]);
int decode_device_batch(int n, const DecInfo* const info[], const DecLayout l[], uint8_t* const ws[], DecPlane (*planes[])[3], hipStream_t s,
                        uint8_t* batch_ws, hipError_t* herr, int* image_rc);

}  // namespace jpeg
}  // namespace uhdr
