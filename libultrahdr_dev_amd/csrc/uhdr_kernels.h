// uhdr_kernels.h -- internal host<->kernel interface (POD kernel arguments + launchers).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

namespace uhdr {

// Per image (321 KiB since round 4, 64 KiB before): header, then EITHER kStatLists lists of kStatCap entries (0.2 % of a 4K map's pixels are in doubt: ~1000 entries per
// image, ~16 per list).  Several lists per image because a wave appends with ONE returning atomic on its list's count, and atomics
// on one address are served one after the other at the memory side (~1 us each: a single count per image cost the kernel 30 %).
// kStatEst: in launches of few images every wave of an image starts at once, finds nothing published and publishes its estimates' extremes:
// thousands of atomics on two words, served one after the other (one 4K image: 80 us for a 10 us kernel).  Such launches
// (GenConsts::stat_spread: at most 16 images of at least 512 waves) publish to and prune by one pair of words per LIST instead;
// the others keep the single pair, which prunes better.
// Round 4: launches whose images have at most kStatSlotWaves waves each (a 4K frame at four spans per block: 1016) need no
// atomic at all -- every wave owns kStatSlotWords words behind the header, word 0 its count (written by every wave of every launch,
// so nothing has to be cleared), the others its entries; a wave with more entries than fit sets word 6 of the header and the image
// is swept.  The returning atomic cost the streaming kernel 17 us per 64 x 4K launch (0.435 ms against 0.418 with it compiled out,
// profiles/r02_generate_ab.txt): a wave waits a trip to the memory side for it before it can exit.  Other launches keep the lists.
// A wave's count word (kStatSlotCnt + wave): plain entries | saved entries << 8.  Its slots: [1, kStatSlotPlain]: plain entries (pair index << 3 | flags: the resolve
// kernel samples the pair again); then kStatSlotSaved entries of 16 words (64 B): (pair index << 3 | flags) and the pair AS SAMPLED --
// r, g, b, hr, hg, hb of both pixels, what the transfer functions start from -- written by the streaming kernel the moment its
// filter finds a pixel in doubt (the twelve floats are still in registers then), so that the resolve kernel reads one contiguous
// line per such pair instead of fourteen scattered ones (that round of reads was 14 of its 24 us).
constexpr uint32_t kStatSlotWaves = 1024, kStatSlotPlain = 15, kStatSlotSaved = 4, kStatSlotWords = 16 + 16 * kStatSlotSaved;
// (slot mode keeps the waves' count words side by side in front of the slots, kStatSlotCnt: the resolve kernel reads them as 4 KiB)
constexpr uint32_t kStatLists = 64, kStatCap = 252, kStatEst = 8 + kStatLists, kStatSlotCnt = kStatEst + 2 * kStatLists,
                   kStatHdr = kStatSlotCnt + kStatSlotWaves,
                   kStatWords = kStatHdr + (kStatLists * kStatCap > kStatSlotWaves * kStatSlotWords ? kStatLists * kStatCap : kStatSlotWaves * kStatSlotWords);
static_assert(kStatHdr % 4u == 0u && kStatSlotWords % 4u == 0u && kStatWords % 4u == 0u, "saved entries are read and written as 16-byte pieces");
constexpr int kMaxChunk = 64;  // images per launch (descriptors travel in the 4 KiB kernarg segment: 64 x 56 B + consts)

// ---- LUT mode (gainmapmath.cpp:21-64 static tables; opt-in) --------------------
// one device buffer per device, filled once at uhdr_hip_init() by k_build_luts: table[i] = f((float)i / (float)(N - 1))
constexpr uint32_t kLutSrgbInvN = 1024, kLutHlgInvN = 4096, kLutPqInvN = 4096, kLutHlgN = 65536, kLutPqN = 65536;
constexpr uint32_t kLutSrgbInv = 0, kLutHlgInv = kLutSrgbInv + kLutSrgbInvN, kLutPqInv = kLutHlgInv + kLutHlgInvN,
                   kLutHlg = kLutPqInv + kLutPqInvN, kLutPq = kLutHlg + kLutHlgN, kLutTotal = kLutPq + kLutPqN;
// FAST apply only (not the reference's tables): transfer functions as line segments (c0, c1), 8 bytes per cell (k_apply_s4).
// A channel is  code = OETF(EOTF_sRGB(c) * F) * 1023  with F = 2^E the gain factor.  Both OETFs are smooth, nearly linear
// functions of a POWER of their argument -- HLG of u = sqrt(x) (exactly linear below its junction), PQ of u = x^m1 (a rational
// function) -- so the path is  u = T(c) * 2^(g E),  T = EOTF^g  (g = 1/2 for HLG, m1 for PQ, 1 for the linear formats):
//   stage 1  T(c):     cell = the float's own exponent and top 4 mantissa bits (16 cells per octave: the sRGB toe makes T
//                      singular at 0, cells that follow the exponent hold a power law to a constant relative error, 6e-5 here);
//                      `(bits >> 16) & 0x0FF8` -- one SDWA v_and -- is the byte offset of the entry as it stands: five exponent
//                      bits (inputs are 0 or lie in [2^-31, 1]), 4 KiB; the pixels use the top few octaves: 1-2 entries per bank.
//                      For g = 1 (linear output formats, calls whose values may pass 1.0) 16 cells per octave are too coarse:
//                      that table takes its cell from the HALF-PRECISION bit pattern of c (bits 14..3: 128 cells per octave).
//   stage 2  code(u):  129 UNIFORM cells over 2 + 2u in [2, 4] -- the cell is byte 2 of the float itself -- each entry
//                      replicated into 32 lane slots in LDS (cell * 256 + (lane & 31) * 8): no bank conflicts whatever the data.
//                      The entry evaluates to -(2 + n 2^-22), n the integer code (the kernel rounds toward zero), whose bits are
//                      0xC0000000 | n: red as it stands (alpha included), green and blue after a shift that drops the upper bits.
// Appended to the LUT buffer by uhdr_hip_init; byte sizes are multiples of 16.
constexpr uint32_t kTabS1Bytes = 0x3C00 + 16;                    // g = 1, half-precision cells up to 1.0
constexpr uint32_t kTabS1PowBytes = 512u * 8u;                   // g < 1: cell = (exponent & 31) << 4 | top 4 mantissa bits (inputs are 0 or in [2^-31, 1])
constexpr uint32_t kTabS2Cells = 129, kTabS2Floats = 260;        // 129 x (c0, c1), padded
constexpr uint32_t kTabS2LdsBytes = kTabS2Cells * 256u;          // replicated
constexpr uint32_t kTabS1Lin = kLutTotal, kTabS1Hlg = kTabS1Lin + kTabS1Bytes / 4, kTabS1Pq = kTabS1Hlg + kTabS1PowBytes / 4,
                   kTabS2Hlg = kTabS1Pq + kTabS1PowBytes / 4, kTabS2Pq = kTabS2Hlg + kTabS2Floats, kTabEnd = kTabS2Pq + kTabS2Floats;
// LUT-mode apply, scale 4: what becomes of an OETF table entry is its 10-bit code, (uint32_t)(table[i] * 1023.0f) & 0x3ff
// (gainmapmath.cpp:722-727) -- the 65536 codes of a table are 128 KiB of uint16 and fit into a workgroup's LDS where the 256 KiB of
// floats do not.  Built once per device behind the float tables (k_build_lut_codes).
constexpr uint32_t kCodeHlg = kTabEnd, kCodePq = kCodeHlg + kLutHlgN / 2, kLutBufferFloats = kCodePq + kLutPqN / 2;
constexpr uint32_t kGainLutN = 1024;  // kGainFactorNumEntries, gainmapmath.h:149-150

// ---- generate ----------------------------------------------------------------------------------
struct GenConsts {
  float sdr_cr, sdr_gcb, sdr_gcr, sdr_cb;  // SDR YUV->RGB (gamut of the SDR image, or 601)
  float hdr_cr, hdr_gcb, hdr_gcr, hdr_cb;  // HDR YUV->RGB (gamut of the P010 image)
  float lum_r, lum_g, lum_b;               // luminance of the SDR gamut (used for both images)
  float gm[9];                             // HDR->SDR gamut matrix
  int gm_identity;
  float hdr_white_nits;
  float bias4096;                          // == 4096.0f (kept opaque to the optimizer, see gen_pair)
  float min_boost, max_boost, log2_min, log2_max;
  // encode_gain_guarded: 255/(double)(log2_max - log2_min) and the bytes of gain == min / gain == max
  double enc_scale;
  uint32_t enc_byte_min, enc_byte_max;
  uint32_t width, height, map_w, map_h;
  uint32_t* stat_keys;  // 2 words per image of the launch (min key, max key), or nullptr
  uint32_t stat_stride; // words between the key pairs of consecutive images (2 in a caller's min/max array)
  // filtered kernel (always): per image kStatWords words -- [0] ~key of the smallest ESTIMATE seen, [1] key of the largest,
  // [3] slices of k_generate_resolve that have finished, [4] [5] the exact keys (stat_keys points here, stat_stride = kStatWords),
  // [8 + l] entries in list l, [kStatEst + 2 l], [kStatEst + 2 l + 1] words 0 and 1 once more per list (launches of few images),
  // [kStatHdr + l * kStatCap ..] list l: pairs with a pixel in doubt or a statistics candidate (a block
  // appends to list blk % kStatLists; a count beyond kStatCap means: sweep the image).  k_generate_resolve
  // redoes them on the exact path, writes the image's (min, max) to stat_out (when not null) and clears the header for the next
  // launch: no memset and no finalize kernel around the launch.
  uint32_t* stat_ws;
  float* stat_out;
  const float* lut;     // device LUT buffer (LUT mode only)
  // f32 pre-filter of the exact path (see gen_pair): code-value scale, half-width of the "too close to an integer"
  // band, and the gains below / above which the clamp certainly applies
  float flt_scale, flt_delta, flt_lo, flt_hi;
  float flt_gain_rel;   // 2 x kRel: how far (relatively) a filter estimate of the gain may sit from the exact one
  uint32_t stat_spread; // the estimates' extremes are published per list (kStatEst): launches of few, large images
  uint32_t stat_slots;  // != 0: the waves of an image (this many, <= kStatSlotWaves) append to slots of their own, no atomics (above)
};
struct EvalConsts {
  float min_boost, max_boost, log2_min, log2_max;
  double enc_scale;
  uint32_t enc_byte_min, enc_byte_max;
  const float* lut;
  double log2_min_d, log2_max_d;
};
struct GenImage {  // 56 bytes; the V plane is u + c_stride * (height / 2) (gainmapmath.cpp:568)
  const uint8_t* y;
  const uint8_t* u;
  const uint16_t* hy;
  const uint16_t* huv;
  uint8_t* map;
  uint32_t y_stride, c_stride, hy_stride, huv_stride;
};
struct GenBatch {
  GenImage img[kMaxChunk];
};

// ---- apply -------------------------------------------------------------------------------------
// constants of the FAST scale-4 kernel (see k_apply_s4): wD[oy][pair][k-2] = (w_k(ox=2*pair), w_k(ox=2*pair+1)) * A / 255
// E = B + A255 * m1 + sum_{k=2..4} (m_k - m1) * wD[.][.][k-2][.]   (m: the four map bytes as floats; the weights of a cell sum to 1)
struct AppFast {
  float A, B, A255;
  float wD[4][2][3][2];
};
struct AppConsts {
  uint32_t width, height, map_w, map_h, scale;
  uint32_t cells_per_thread;      // k_apply_s4: map cells a thread walks (set by launch_apply from the size of the launch)
  uint32_t step_x, step_y;        // k_apply_s4: a block's 512 cells as (rows, columns) of the map: 512 = step_y * map_w + step_x
  float display_boost, inv_display_boost, max_boost, inv_max_boost;
  double log2_min_d, log2_max_d;  // log2((double)minContentBoost), log2((double)maxContentBoost)
  const float* idw;               // device: 4 tables (std, NR, NB, C) of scale*scale*4 floats
  const float* lut;               // device LUT buffer (LUT mode only)
  const float* tab;               // device: the LUT buffer (line-segment tables at kTabS1* / kTabS2*), FAST scale-4 kernel
  float lut_boost_factor;         // GainLUT(metadata, displayBoost): displayBoost > 0 ? displayBoost / max : 1 (gainmapmath.h:162)
  uint32_t lut_plain_div;         // LUT mode: x / display_boost may run as the IEEE expansion WITHOUT its operand scaling (lut_cell_pk):
                                  // the operands of this call cannot reach the ranges in which v_div_scale_f32 rescales them
  uint32_t* ex_ws;                // EXACT mode behind the f32 pre-filter: the lists of pixels in doubt (layout below), or nullptr
  uint32_t ex_cap;                // entries per list
  AppFast fast;
};
// EXACT apply's workspace, one per stream: kMaxChunk headers of kExHdrWords words (word 3: slices of k_apply_resolve that have
// finished; words 8 ..: the lengths of the image's kExLists lists), then kMaxChunk x kExLists lists of ex_cap pixel indices.  The
// headers sit in front so that their place does not depend on the image size: every launch leaves them cleared.
constexpr uint32_t kExLists = 1024;
constexpr uint32_t kExHdrWords = 8 + kExLists;
inline size_t ex_ws_bytes(uint32_t cap) { return ((size_t)kMaxChunk * kExHdrWords + (size_t)kMaxChunk * kExLists * cap) * 4u; }
// capacity for a sixteenth of the image's pixels (the filter leaves ~1 % in doubt), spread over the lists
inline uint32_t ex_list_cap(uint64_t pixels) { return (uint32_t)(pixels / 16u / kExLists) + 64u; }
struct AppImage {
  const uint8_t* y;
  const uint8_t* u;
  const uint8_t* v;
  const uint8_t* map;
  void* dst;
  uint32_t y_stride, c_stride;
};
struct AppBatch {
  AppImage img[kMaxChunk];
};

// ---- toneMap / convertYuv ----------------------------------------------------------------------
struct ToneImage {
  const uint16_t* sy;
  const uint16_t* suv;
  uint8_t* dy;
  uint8_t* du;
  uint8_t* dv;
  uint32_t sy_stride, suv_stride, dy_stride, dc_stride;
  uint32_t width, height;
};
constexpr int kToneChunk = 32;  // images per toneMap / convertYuv launch (grid.z = image; 32 x 112 B of descriptors)
struct ToneBatch {
  ToneImage img[kToneChunk];
};
struct CvtImage {
  uint8_t* y;   // where the converted planes go ...
  uint8_t* u;
  uint8_t* v;
  uint32_t y_stride, c_stride, width, height;
  float m[9];
  const uint8_t* sy;   // ... and where the samples come from: the same planes (JpegR::convertYuv works in place), or the caller's
  const uint8_t* su;   // image when the conversion is also the private copy encodeJPEGR API-1 takes first (jpegr.cpp:297-358)
  const uint8_t* sv;
  uint32_t sy_stride, sc_stride;
};
struct CvtBatch {   // m, width, height are taken from img[0]: one launch converts equally sized images between the same encodings
  CvtImage img[kToneChunk];
};

// ---- editorhelper effects (crop / mirror / rotate / resize): byte gathers over planes ---------------
enum FxOp : int { FX_COPY = 0, FX_FLIP_V, FX_FLIP_H, FX_ROT90, FX_ROT180, FX_ROT270, FX_RESIZE };
struct FxJob {            // dst[i][j] = src[f(i, j)] for i < rows, j < cols
  const uint8_t* src;
  uint8_t* dst;
  uint32_t rows, cols, dst_stride, src_stride;
  uint32_t in_w, in_h;    // extent of the source plane the index maps refer to
  uint32_t row_num, row_den, col_num, col_den;  // FX_RESIZE: src row = i*row_num/row_den, src col = j*col_num/col_den
  int op;
};
struct FxJobs {
  FxJob job[3];
  int n;
};

static_assert(sizeof(GenConsts) + sizeof(GenBatch) <= 4096, "generate kernel arguments exceed the kernarg segment");
static_assert(sizeof(AppConsts) + sizeof(AppBatch) <= 4096, "apply kernel arguments exceed the kernarg segment");

// launchers (enqueue only; return hipError_t of the launch)
hipError_t launch_generate(const GenConsts& c, const GenBatch& b, int n, int hdr_tf, bool aligned, bool lut,
                           bool filter, hipStream_t s);
// a launch of n images this size is too small to be worth k_generate_resolve's latency (a single 4K image: 13 us against 24)
bool generate_is_small(const GenConsts& c, int n);
bool generate_resolve_pays(const GenConsts& c, int n);   // a launch with statistics: the filtered kernel + k_generate_resolve beat the exact kernel
uint32_t generate_slot_waves(const GenConsts& c, int n);  // waves per image of the filtered + deferred launch of n such images, or 0 when they exceed kStatSlotWaves
hipError_t launch_stats_init(uint32_t* keys, int n, hipStream_t s);
// after every filtered launch: the pixels it left in doubt and its statistics candidates on the exact path
hipError_t launch_stats_resolve(const GenConsts& c, const GenBatch& b, int n, int hdr_tf, bool aligned, hipStream_t s);
constexpr size_t kStatWsBytes = sizeof(uint32_t) * kStatWords * (size_t)kMaxChunk;
hipError_t launch_stats_finalize(uint32_t* keys, int n, hipStream_t s);
// mode: 0 FAST, 1 EXACT, 2 LUT
hipError_t launch_apply(const AppConsts& c, const AppBatch& b, int n, int fmt, int mode,
                        bool fast_s4, hipStream_t s);
hipError_t launch_build_luts(float* lut /* kLutTotal floats */, hipStream_t s);
hipError_t launch_build_lut_codes(float* lut /* the whole buffer: reads the two OETF tables, writes kCodeHlg / kCodePq */, hipStream_t s);
// GainLUT table (kGainLutN floats, device) for (log2 min, log2 max, boost factor)
hipError_t launch_build_gain_lut(float* table, double log2_min, double log2_max, float boost_factor, hipStream_t s);
hipError_t launch_tonemap(const ToneBatch& b, int n, bool aligned, hipStream_t s);      // n <= kToneChunk images of equal width / height
hipError_t launch_convert_yuv(const CvtBatch& b, int n, bool aligned, hipStream_t s);
static_assert(sizeof(CvtBatch) <= 4096 && sizeof(ToneBatch) <= 4096, "toneMap / convertYuv kernel arguments exceed the kernarg segment");
// decoded 4:2:0 planes -> RGBA8888 with libjpeg-turbo's arithmetic (k_ycc420_rgba); w, h even, strides in bytes
hipError_t launch_ycc420_to_rgba(const uint8_t* y, const uint8_t* cb, const uint8_t* cr, uint32_t w, uint32_t h, uint32_t y_stride,
                                 uint32_t c_stride, uint8_t* rgba, hipStream_t s);
hipError_t upload_idw4(const float* tables /* 4*64 floats */);
hipError_t launch_effect(const FxJobs& j, hipStream_t s);
hipError_t launch_eval_transfer(int fn, const float* in, float* out, size_t n, const EvalConsts& ec, hipStream_t s);
hipError_t launch_synth_lcg(uint16_t* p010, uint8_t* yuv, uint32_t n_luma, uint32_t n, uint32_t seed, hipStream_t s);

}  // namespace uhdr
