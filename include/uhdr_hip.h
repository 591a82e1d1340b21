/*
 * uhdr_hip.h -- C-ABI of the MI355X (gfx950) gain-map pixel path.
 *
 * This is the drop-in boundary for the per-pixel hot path of libultrahdr_dev.  The reference has
 * no FFI layer: the path sits behind three protected C++ members of ultrahdr::UltraHdr plus one
 * private member of JpegR.  Each entry point below names the reference interface it replaces
 * (paths relative to the reference tree); INTEGRATION.md shows the binding a maintainer adds.
 *
 *   uhdr_hip_generate_gainmap   <- UltraHdr::generateGainMap   lib/include/ultrahdr/ultrahdr.h:348-350
 *                                                              lib/src/ultrahdr.cpp:185-358
 *   uhdr_hip_apply_gainmap      <- UltraHdr::applyGainMap      lib/include/ultrahdr/ultrahdr.h:370-372
 *                                                              lib/src/ultrahdr.cpp:360-515
 *   uhdr_hip_tonemap            <- UltraHdr::toneMap           lib/include/ultrahdr/ultrahdr.h:381
 *                                                              lib/src/ultrahdr.cpp:517-558
 *   uhdr_hip_convert_yuv        <- JpegR::convertYuv           lib/include/ultrahdr/jpegr.h:329-330
 *                                                              lib/src/jpegr.cpp:1132-1206
 *
 * Conventions
 *  - plain C, POD only; no torch / STL types.  Enum VALUES are the reference's
 *    (lib/include/ultrahdr/ultrahdr.h:36-120).  Return value is the reference's status_t value
 *    for the same inputs (first failing check wins, same order); HIP failures map to
 *    UHDR_HIP_UNKNOWN_ERROR (-1); "no usable GPU / library not initialised" is
 *    UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE (-20009).  There is NO CPU fallback.
 *  - images use the reference descriptor (ultrahdr_uncompressed_struct, ultrahdr.h:152-181):
 *    strides in PIXELS (P010 chroma stride counts uint16 elements of the interleaved plane),
 *    chroma_data must be non-NULL (callers normalise exactly as jpegr.cpp:265-278 does).
 *    YUV420: U plane at chroma_data, V plane at chroma_data + chroma_stride*(height/2).
 *  - mem_space says where every data pointer of the call lives:
 *      UHDR_HIP_MEM_DEVICE  device pointers; the call only enqueues kernels on `stream`
 *                           (asynchronous; graph-capturable; nothing is allocated or copied -- with three
 *                           exceptions, each the FIRST use of a stream's workspace: a generate launch of more
 *                           than a few images allocates the stream's 21 MiB statistics workspace, an EXACT apply
 *                           its lists of pixels in doubt (first call, and again for larger images), and an apply
 *                           with a map scale factor the device holds no weight table for uploads one and waits.
 *                           uhdr_hip_stream_reserve() does all of that ahead of time -- call it before capturing
 *                           a graph --, uhdr_hip_stream_release() gives a stream's workspace back);
 *      UHDR_HIP_MEM_HOST    host pointers; the library stages through its own device workspace
 *                           (H2D, kernels, D2H) and returns after the result is back in host memory.
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *  - the library never allocates result memory on behalf of the caller: the gain map buffer
 *    ((width/4)*(height/4) bytes) is caller-provided in dest->data.  (The C++ shim in
 *    include/ultrahdr_hip/ultrahdr_hip.h reproduces the reference's new[] contract on top of this.)
 */
#ifndef UHDR_HIP_H
#define UHDR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UHDR_HIP_ABI_VERSION 3

/* ultrahdr_color_gamut, ultrahdr.h:36-42 */
#define UHDR_HIP_CG_UNSPECIFIED (-1)
#define UHDR_HIP_CG_BT709 0
#define UHDR_HIP_CG_P3 1
#define UHDR_HIP_CG_BT2100 2
/* ultrahdr_transfer_function, ultrahdr.h:46-53 */
#define UHDR_HIP_TF_LINEAR 0
#define UHDR_HIP_TF_HLG 1
#define UHDR_HIP_TF_PQ 2
#define UHDR_HIP_TF_SRGB 3
/* ultrahdr_output_format, ultrahdr.h:56-64 */
#define UHDR_HIP_OUTPUT_SDR 0
#define UHDR_HIP_OUTPUT_HDR_LINEAR 1            /* RGBA F16, 8 B/pixel */
#define UHDR_HIP_OUTPUT_HDR_PQ 2                /* RGBA1010102, 4 B/pixel */
#define UHDR_HIP_OUTPUT_HDR_HLG 3               /* RGBA1010102, 4 B/pixel */
#define UHDR_HIP_OUTPUT_HDR_LINEAR_RGB_10BIT 4  /* three planar uint16 planes, 6 B/pixel */
/* ultrahdr_pixel_format, ultrahdr.h:67-75 */
#define UHDR_HIP_PIX_FMT_UNSPECIFIED (-1)
#define UHDR_HIP_PIX_FMT_P010 0
#define UHDR_HIP_PIX_FMT_YUV420 1
#define UHDR_HIP_PIX_FMT_MONOCHROME 2
/* status_t, ultrahdr.h:91-120 */
#define UHDR_HIP_NO_ERROR 0
#define UHDR_HIP_UNKNOWN_ERROR (-1)
#define UHDR_HIP_ERROR_BAD_PTR (-10001)
#define UHDR_HIP_ERROR_INVALID_COLORGAMUT (-10003)
#define UHDR_HIP_ERROR_INVALID_TRANS_FUNC (-10005)
#define UHDR_HIP_ERROR_RESOLUTION_MISMATCH (-10006)
#define UHDR_HIP_ERROR_BAD_METADATA (-10010)
#define UHDR_HIP_ERROR_INVALID_CROPPING_PARAMETERS (-10011)
#define UHDR_HIP_ERROR_UNSUPPORTED_FEATURE (-30000)
#define UHDR_HIP_ERROR_UNSUPPORTED_MAP_SCALE_FACTOR (-20008)
#define UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE (-20009)

/* where the planes of a call live.  DEVICE: the call only enqueues kernels on `stream` (asynchronous, graph-capturable).  HOST
 * (what a caller of the reference binds): the call copies the planes in, runs, copies the result out and returns when it is there.
 * Threads: every entry point may be called from several host threads.  Device-memory calls on different streams overlap; the
 * host-memory forms of generate / apply / tonemap / convert_yuv lease a staging set per call, so callers on their own streams
 * overlap their copies and kernels as well (one thread moves 0.54, eight 0.94 4K pairs per ms over the host link); the codec entry
 * points (jpeg_*, jpegr_*, the effects and table calls through host memory) lease a codec context per call -- staging buffers,
 * decoder workspaces, a side stream -- so callers on their own streams overlap too (a JPEG decode is latency-bound: two threads
 * decode nearly twice as many files per second as one). */
#define UHDR_HIP_MEM_HOST 0
#define UHDR_HIP_MEM_DEVICE 1

/* arithmetic mode of uhdr_hip_apply_gainmap*:
 *   FAST  : transfer functions from line-segment tables in LDS (special-function unit where a call can exceed 1.0); every
 *           10-bit channel within 1 LSB and every F16 channel within 1 half-ULP of the reference CPU path;
 *   EXACT : the same bytes as the reference CPU path: its float/double promotion pattern replayed with correctly
 *           rounded pow/exp/log/exp2, on the pixels an f32 estimate with measured error bounds cannot settle (~1 %).
 *           The first EXACT call on a stream (and the first with larger images) allocates that stream's list workspace;
 *           after it the call only enqueues two kernels and can be captured into a graph like the others.
 * generate/tonemap/convert_yuv have a single, bit-exact mode. */
#define UHDR_HIP_APPLY_FAST 0
#define UHDR_HIP_APPLY_EXACT 1
/* LUT mode (opt-in): the loops as a build that sees jpegr.cpp:33-38's
 * USE_*_LUT = 1 compiles them (upstream libultrahdr's configuration; dead code in this fork, whose
 * ultrahdr.cpp never sees those macros).  apply: srgbInvOetfLUT + applyGainLUT(GainLUT(metadata,
 * display_boost)) + hlgOetfLUT / pqOetfLUT (ultrahdr.cpp:433,446,470,481); generate: srgbInvOetfLUT +
 * hlgInvOetfLUT / pqInvOetfLUT (:230,238,319).  Tables are built on the device by the exact functions
 * (gainmapmath.cpp:21-64); with them the LUT pipelines are pure float/integer work and BIT-EXACT against
 * the reference's LUT functions. */
#define UHDR_HIP_APPLY_LUT 2
/* EXACT without its f32 pre-filter: every pixel takes the double-precision path (what EXACT always does for HDR_PQ).  Same
 * bytes as UHDR_HIP_APPLY_EXACT; kept as a verification switch for tests. */
#define UHDR_HIP_APPLY_EXACT_UNFILTERED 3
#define UHDR_HIP_GENERATE_EXACT 0
#define UHDR_HIP_GENERATE_LUT 1
/* EXACT without the f32 pre-filter: every pixel takes the double-precision path.  Same bytes and statistics as
 * UHDR_HIP_GENERATE_EXACT by construction (the filter only decides which waves can skip that path); kept as a
 * verification switch for tests. */
#define UHDR_HIP_GENERATE_UNFILTERED 2

/* POD mirror of ultrahdr_uncompressed_struct (ultrahdr.h:152-181) */
typedef struct uhdr_hip_image {
  void* data;           /* luma plane (or gain map / packed output) */
  size_t width;         /* luma width in pixels */
  size_t height;        /* luma height in pixels */
  int32_t colorGamut;   /* UHDR_HIP_CG_* */
  void* chroma_data;    /* U plane (YUV420) / interleaved UV plane (P010); must be non-NULL */
  size_t luma_stride;   /* pixels */
  size_t chroma_stride; /* pixels (P010: uint16 elements of the interleaved plane) */
  int32_t pixelFormat;  /* UHDR_HIP_PIX_FMT_* */
} uhdr_hip_image_t;

/* POD mirror of ultrahdr_metadata_struct (ultrahdr.h:129-147); version is the NUL-terminated
 * string the reference keeps in a std::string ("1.0", ultrahdr.h:210) */
typedef struct uhdr_hip_metadata {
  char version[8];
  float maxContentBoost;
  float minContentBoost;
  float gamma;
  float offsetSdr;
  float offsetHdr;
  float hdrCapacityMin;
  float hdrCapacityMax;
} uhdr_hip_metadata_t;

/* ---- library / device management -------------------------------------------------------- */

/* ABI version of the loaded library (== UHDR_HIP_ABI_VERSION of the header it was built from) */
int uhdr_hip_abi_version(void);
/* number of visible HIP devices (0 when there is no GPU; never fails) */
int uhdr_hip_device_count(void);
/* bind the calling process to `device` (hipSetDevice), upload the constant tables, create the
 * staging workspace.  Must be called once per device before any compute call. */
int uhdr_hip_init(int device);
/* release the workspace of every initialised device */
int uhdr_hip_shutdown(void);
/* Allocate, now, what device-memory calls on `stream` would otherwise allocate at their first use (see mem_space above): the
 * statistics workspace of generate, and -- when exact_images > 0 -- the lists EXACT apply needs for launches of up to
 * exact_images images of width x height pixels (at most 64 per launch; 0 for width / height with exact_images == 0), and the
 * sampleMap weight table of `map_scale_factor` (0: none; 4 is always resident).  After this the calls it covers only enqueue. */
int uhdr_hip_stream_reserve(void* stream, int exact_images, size_t width, size_t height, int map_scale_factor);
/* Free what the library holds for `stream` (waits for the stream first): its statistics workspace, its EXACT-apply lists and the
 * smaller lists those outgrew.  A service that creates and destroys streams calls this before hipStreamDestroy: workspaces are
 * keyed by the stream handle, and a recycled handle would inherit the old one.  NOT returned: what belongs to the device rather
 * than to a stream (staging sets of host-memory calls, codec contexts, weight tables) -- uhdr_hip_shutdown() frees those. */
int uhdr_hip_stream_release(void* stream);
/* last HIP error text seen by the library on this thread ("" if none) */
const char* uhdr_hip_last_error(void);

/* ---- where resident images lie in device memory (no reference counterpart: the reference's images live in malloc'ed host memory) ----
 * Any device pointer works with every call.  But WHICH physical memory a batch of images occupies decides what the card's HBM gives
 * the streaming kernels: a batch in one physically contiguous stretch (what hipMalloc returns on a device whose memory is mostly free)
 * runs apply 6-7 % slower than the same batch in pieces taken from all over a region three or more times its size, whatever the
 * virtual layout (profiles/r04_placement.txt).  A pool takes `bytes` of device memory as chunks of `chunk_bytes` through
 * the HIP virtual-memory calls; an allocation is one contiguous range of virtual addresses backed by chunks spaced evenly over the
 * pool's free ones -- so the images of several allocations interleave physically.  Intended use: at start-up, one pool over a wide
 * stretch of the card (tens of GiB and more: a pool no larger than what it holds is fast or not by where it happens to lie), one
 * allocation per arena of what stays resident (frames, maps, renditions), then uhdr_hip_mem_pool_trim() for the rest.
 *   bytes        rounded up to whole chunks; fails with ERROR_INSUFFICIENT_RESOURCE when the device does not have them
 *   chunk_bytes  0 = 16 MiB; a multiple of 2 MiB
 * ERROR_BAD_PTR for a NULL argument, ERROR_UNSUPPORTED_FEATURE for a size of 0, a chunk size that is no multiple of 2 MiB, a device
 * that does not exist or a pointer the pool did not hand out.
 * alloc: ERROR_INSUFFICIENT_RESOURCE when fewer free chunks are left than the size needs; free returns the chunks to the pool (the
 * pointer must be one alloc returned, with no work outstanding on it); trim gives the chunks no allocation uses back to the device;
 * destroy unmaps and releases everything (allocations included).  Calls on one pool are serialised by the library. */
typedef struct uhdr_hip_mem_pool uhdr_hip_mem_pool_t;
int uhdr_hip_mem_pool_create(int device, size_t bytes, size_t chunk_bytes, uhdr_hip_mem_pool_t** pool);
int uhdr_hip_mem_pool_alloc(uhdr_hip_mem_pool_t* pool, size_t bytes, void** ptr);
int uhdr_hip_mem_pool_free(uhdr_hip_mem_pool_t* pool, void* ptr);
int uhdr_hip_mem_pool_trim(uhdr_hip_mem_pool_t* pool);
int uhdr_hip_mem_pool_destroy(uhdr_hip_mem_pool_t* pool);
/* chunks the pool holds / of those, chunks no allocation uses (either pointer may be NULL) */
int uhdr_hip_mem_pool_stats(uhdr_hip_mem_pool_t* pool, size_t* chunks, size_t* free_chunks);

/* ---- single image ------------------------------------------------------------------------ */

/* UltraHdr::generateGainMap (ultrahdr.cpp:185-358).  Writes the (width/4)x(height/4) u8 map into
 * dest->data (caller-allocated, stride == map width), fills dest->{width,height,colorGamut,
 * luma_stride,chroma_data,chroma_stride,pixelFormat} and *metadata (the reference's constants,
 * ultrahdr.cpp:250-257). */
int uhdr_hip_generate_gainmap(const uhdr_hip_image_t* yuv420_image, const uhdr_hip_image_t* p010_image,
                              int hdr_tf, uhdr_hip_metadata_t* metadata, uhdr_hip_image_t* dest,
                              int sdr_is_601, int mem_space, void* stream);

/* the same with an arithmetic mode: UHDR_HIP_GENERATE_EXACT (what uhdr_hip_generate_gainmap runs) or _LUT */
int uhdr_hip_generate_gainmap_ex(const uhdr_hip_image_t* yuv420_image, const uhdr_hip_image_t* p010_image,
                                 int hdr_tf, uhdr_hip_metadata_t* metadata, uhdr_hip_image_t* dest,
                                 int sdr_is_601, int generate_mode, int mem_space, void* stream);

/* UltraHdr::applyGainMap (ultrahdr.cpp:360-515).  dest->data is caller-allocated:
 * width*height*{8|4|6} bytes for HDR_LINEAR | HDR_PQ,HDR_HLG | HDR_LINEAR_RGB_10BIT; any other
 * output_format writes nothing and returns NO_ERROR, like the reference (ultrahdr.cpp:491-493). */
int uhdr_hip_apply_gainmap(const uhdr_hip_image_t* yuv420_image, const uhdr_hip_image_t* gainmap_image,
                           const uhdr_hip_metadata_t* metadata, int output_format,
                           float max_display_boost, uhdr_hip_image_t* dest, int apply_mode,
                           int mem_space, void* stream);

/* UltraHdr::toneMap (ultrahdr.cpp:517-558): P010 -> YUV420 by bit shift, stride padding zeroed. */
int uhdr_hip_tonemap(const uhdr_hip_image_t* src, uhdr_hip_image_t* dest, int mem_space, void* stream);

/* JpegR::convertYuv (jpegr.cpp:1132-1206): in-place YUV420 matrix re-encode between
 * BT.709 / BT.601(P3) / BT.2100 encodings. */
int uhdr_hip_convert_yuv(uhdr_hip_image_t* image, int src_encoding, int dest_encoding, int mem_space,
                         void* stream);

/* The same two over n images in DEVICE memory (every op in single and batched form; the reference runs them once
 * per image, ultrahdr.cpp:517-558 / jpegr.cpp:1132-1206).  Images of equal size share one launch (grid.z = image, up to 32), so an
 * API-0 / API-1 encode of a batch does not pay a launch pair per image.  Checks are the single form's, all images before any launch. */
int uhdr_hip_tonemap_batch(int n, const uhdr_hip_image_t* srcs, uhdr_hip_image_t* dests, void* stream);
int uhdr_hip_convert_yuv_batch(int n, uhdr_hip_image_t* images, int src_encoding, int dest_encoding, void* stream);

/* ---- editing effects (SURVEY.md 8(f) "next", rank 3) -------------------------------------------------
 * crop / mirror / rotate / resize of lib/src/editorhelper.cpp:26-360 (lib/include/ultrahdr/editorhelper.h:49-63)
 * on YUV420 or MONOCHROME images, byte-identical to the reference including its layout rules: the output is
 * tightly packed (luma then U then V at out->data) except mirror and rotate-180, whose output strides follow
 * the INPUT luma stride; crop's chroma copy runs over the full output height (:72).  out->data is
 * caller-allocated; the call fills the other fields of *out.  in->chroma_data == NULL means "right after luma",
 * strides of 0 mean "width" / "luma stride / 2", as in the reference. */
int uhdr_hip_crop(const uhdr_hip_image_t* in_img, int left, int right, int top, int bottom, uhdr_hip_image_t* out_img,
                  int mem_space, void* stream);
/* mirror_dir: 0 = ULTRAHDR_MIRROR_VERTICAL, 1 = ULTRAHDR_MIRROR_HORIZONTAL */
int uhdr_hip_mirror(const uhdr_hip_image_t* in_img, int mirror_dir, uhdr_hip_image_t* out_img, int mem_space, void* stream);
int uhdr_hip_rotate(const uhdr_hip_image_t* in_img, int clockwise_degree, uhdr_hip_image_t* out_img, int mem_space,
                    void* stream);
int uhdr_hip_resize(const uhdr_hip_image_t* in_img, int out_width, int out_height, uhdr_hip_image_t* out_img,
                    int mem_space, void* stream);

/* addEffects (lib/src/editorhelper.cpp:362-446; the way ultrahdr.cpp applies a configuration's effects to the SDR image and to
 * the gain map, :886-1429): the effects in order, every intermediate image tightly packed in device memory, the last one
 * copied to out->data (caller-allocated: the largest intermediate extent must fit) with out's fields set as the reference
 * leaves them (chroma_data = data + luma_stride * height for YUV420).  n == 0 copies width*height(*3/2) bytes and the
 * descriptor.  Where the reference is undefined this call reports instead: the status of a failing effect (the reference
 * ignores it and copies uninitialised fields), and ERROR_UNSUPPORTED_FEATURE for a mirror / rotate-180 of an image whose
 * strides exceed its width (the reference overruns its temporary buffer). */
typedef struct uhdr_hip_effect {
  int32_t type;          /* 0 crop, 1 mirror, 2 rotate, 3 resize */
  int32_t a, b, c, d;    /* crop: left, right, top, bottom; mirror: direction; rotate: clockwise degrees; resize: width, height */
} uhdr_hip_effect_t;
int uhdr_hip_add_effects(const uhdr_hip_image_t* in_img, const uhdr_hip_effect_t* effects, int n, uhdr_hip_image_t* out_img,
                         int mem_space, void* stream);

/* ---- JPEG compression of the path's outputs (encode side) -------------------------------
 * JpegEncoderHelper::compressImage (lib/src/jpegencoderhelper.cpp:39-283; lib/include/ultrahdr/jpegencoderhelper.h:43-60):
 * baseline JPEG of a YUV420 image (image->data = Y, image->chroma_data = U, V at chroma_stride * height / 2) or, when
 * image->pixelFormat == UHDR_HIP_PIX_FMT_MONOCHROME, of the single plane image->data -- the bytes libjpeg writes in
 * raw-data mode with default tables, jpeg_set_quality(quality, TRUE) and the ISLOW DCT, including the reference's padding
 * rules (rows past the height are zero; columns past the width are zero when the stride is smaller than the 16-aligned
 * width, the caller's bytes otherwise).  icc (HOST memory, may be NULL) becomes an APP2 segment after the JFIF header.
 * FDCT, quantisation, Huffman coding and byte stuffing all run on the device.  out (capacity out_capacity) lives in the
 * memory space given by mem_space like the image planes; *out_size (HOST) receives the JPEG size.  The call waits for
 * the stream.  Returns ERROR_INSUFFICIENT_RESOURCE with *out_size set when out_capacity is too small. */
int uhdr_hip_jpeg_encode(const uhdr_hip_image_t* image, int quality, const void* icc, size_t icc_size, void* out,
                         size_t out_capacity, size_t* out_size, int mem_space, void* stream);

/* Diagnostics, host only (no GPU): the quantised coefficients of a progressive (SOF2) file after all of its scans -- what the
 * host-side entropy decoder hands to the device: blocks in MCU order (4:2:0: Y00 Y01 Y10 Y11 Cb Cr), zigzag order inside a block.
 * *blocks receives the block count (also when coef is NULL / too small: INSUFFICIENT_RESOURCE).  Baseline files, whose entropy
 * decoding runs on the device: UNSUPPORTED_FEATURE. */
int uhdr_hip_jpeg_progressive_coefficients(const void* jpeg, size_t jpeg_size, int16_t* coef, size_t capacity_blocks, size_t* blocks,
                                           int* width, int* height, int* gray);

/* JpegDecoderHelper::decompressImage(image, length, DECODE_TO_YCBCR) (lib/src/jpegdecoderhelper.cpp:188-516;
 * lib/include/ultrahdr/jpegdecoderhelper.h:54-56): a baseline 4:2:0 YCbCr or grayscale JPEG (HOST memory) -> the bytes
 * libjpeg returns with raw_data_out and JDCT_ISLOW, laid out as the reference's result buffer: width x height luma, then
 * (4:2:0) the (width/2) x (height/2) Cb and Cr planes at width*height and width*height + width*height/4.  out lives in
 * the memory space given by mem_space.  *desc is filled (data = out, chroma_data, strides, pixelFormat YUV420 or
 * MONOCHROME) when the header is readable, also on ERROR_INSUFFICIENT_RESOURCE (out_capacity too small: width*height*3/2
 * resp. width*height bytes are needed).  Huffman decoding (self-synchronising parallel decoder), dequantisation and IDCT
 * run on the device; restart intervals (DRI / RSTn) are read.  Progressive files (SOF2) are read too: their scans are entropy-decoded
 * on the host, dequantisation and IDCT run on the device (complete files; the planes are libjpeg's).  ERROR_UNSUPPORTED_FEATURE:
 * arithmetic-coded / lossless files (libjpeg reads some of them, this decoder does not); UNKNOWN_ERROR: malformed file, or a sampling other than 4:2:0 / single plane, where the reference's call
 * returns false as well (:283-289) -- and a host allocation that failed while parsing (a progressive file's coefficient array).
 * ERROR_INSUFFICIENT_RESOURCE with out == NULL is therefore always the size probe's answer: the header parsed and *desc holds the
 * size; on every status returned before that point *desc is zeroed.  The call waits for the stream. */
int uhdr_hip_jpeg_decode(const void* jpeg, size_t jpeg_size, void* out, size_t out_capacity, uhdr_hip_image_t* desc,
                         int mem_space, void* stream);

/* JpegDecoderHelper::decompressImage(..., DECODE_TO_RGBA) (lib/src/jpegdecoderhelper.cpp:251-281): a YCbCr 4:2:0 baseline JPEG ->
 * width*height RGBA8888 pixels (alpha 0xFF) with libjpeg-turbo's arithmetic (fancy upsampling, fixed-point colour conversion; see
 * uhdr_hip_jpegr_decode's UHDR_HIP_OUTPUT_SDR).  Same calling convention and status values as uhdr_hip_jpeg_decode; a single-plane
 * JPEG is UNKNOWN_ERROR (the reference's call returns false). */
int uhdr_hip_jpeg_decode_rgba(const void* jpeg, size_t jpeg_size, void* out, size_t out_capacity, uhdr_hip_image_t* desc, int mem_space,
                              void* stream);

/* JpegR::decodeJPEGR (lib/src/jpegr.cpp:655-822) for the HDR output formats: a JPEG/R file (HOST memory: primary JPEG + gain
 * map JPEG, the gain map's APP1 carrying the hdrgm:* XMP attributes) -> the HDR rendition applyGainMap produces.  Container
 * scan (extractPrimaryImageAndGainMap, :823-876), XMP metadata (getMetadataFromXMP, jpegrutils.cpp:436-545) and the ICC gamut
 * of the primary image (IccHelper::readIccColorGamut, icc.cpp:615-685) are host bookkeeping; both JPEGs are decompressed and
 * combined on the device.  dest_data (memory space mem_space) receives width*height*{8|4|6} bytes; *dest gets width, height
 * and colorGamut; *metadata (optional) the parsed metadata.  Status values are the reference's: BAD_PTR,
 * INVALID_DISPLAY_BOOST (max_display_boost < 1), INVALID_OUTPUT_FORMAT, NO_IMAGES_FOUND, GAIN_MAP_IMAGE_NOT_FOUND, DECODE_ERROR,
 * METADATA_ERROR, then applyGainMap's own; plus ERROR_INSUFFICIENT_RESOURCE when dest_capacity is too small (*dest is filled)
 * and ERROR_UNSUPPORTED_FEATURE for arithmetic-coded JPEGs (progressive ones are read).  UHDR_HIP_OUTPUT_SDR (:768-786) returns the primary image alone as
 * RGBA8888 (4 bytes per pixel, alpha 0xFF) with the arithmetic libjpeg-turbo applies for DECODE_TO_RGBA (fancy 4:2:0 upsampling and
 * its fixed-point YCbCr -> RGB tables; jpegdecoderhelper.cpp:251-281); the gain map is then not decompressed and its XMP packet is
 * only read when `metadata` is not NULL, as in the reference. */
#define UHDR_HIP_ERROR_INVALID_DISPLAY_BOOST (-10008)
#define UHDR_HIP_ERROR_INVALID_OUTPUT_FORMAT (-10009)
#define UHDR_HIP_ERROR_DECODE_ERROR (-20002)
#define UHDR_HIP_ERROR_GAIN_MAP_IMAGE_NOT_FOUND (-20003)
#define UHDR_HIP_ERROR_METADATA_ERROR (-20005)
#define UHDR_HIP_ERROR_NO_IMAGES_FOUND (-20006)
int uhdr_hip_jpegr_decode(const void* jpegr, size_t jpegr_size, int output_format, float max_display_boost, void* dest_data,
                          size_t dest_capacity, uhdr_hip_image_t* dest, uhdr_hip_metadata_t* metadata, int apply_mode,
                          int mem_space, void* stream);

/* The same for n files in one call (no reference counterpart: the reference decodes one file per call).  A JPEG decode on the
 * device is latency-bound, so the 2 n JPEGs of the call share every kernel launch (one grid row per image) and their
 * synchronisation rounds run side by side: 16 4K files take 4.5x the time of one.  Arrays are indexed by file; dest_data[i] (memory space mem_space) needs dest_capacity[i] bytes; status (optional)
 * receives each file's status, the return value is the first one that is not NO_ERROR; a file that fails does not disturb the
 * others.  dest_data[i] == NULL asks for file i's size only (ERROR_INSUFFICIENT_RESOURCE, dests[i] filled). */
int uhdr_hip_jpegr_decode_batch(int n, const void* const* jpegr, const size_t* jpegr_size, int output_format, float max_display_boost,
                                void* const* dest_data, const size_t* dest_capacity, uhdr_hip_image_t* dests,
                                uhdr_hip_metadata_t* metadata, int* status, int apply_mode, int mem_space, void* stream);

/* JpegR::appendGainMap (lib/src/jpegr.cpp:951-1130): primary JPEG + gain-map JPEG + metadata -> JPEG/R container (XMP packets of
 * jpegrutils.cpp:547-611, MPF segment of multipictureformat.cpp:30-92).  exif / icc: payloads of an APP1 / APP2 segment to add, or
 * NULL; an EXIF segment found inside primary_jpeg moves in front of the XMP segment (and ERROR_MULTIPLE_EXIFS_RECEIVED if exif is
 * given as well), with JpegDecoderHelper::extractEXIF's position arithmetic (jpegdecoderhelper.cpp:146-188).  Pure host code, no
 * device needed.  *out_size receives the size (also when out_capacity is too small: ERROR_INSUFFICIENT_RESOURCE). */
#define UHDR_HIP_ERROR_MULTIPLE_EXIFS_RECEIVED (-20007)
int uhdr_hip_jpegr_append_gainmap(const void* primary_jpeg, size_t primary_size, const void* gainmap_jpeg, size_t gainmap_size,
                                  const void* exif, size_t exif_size, const void* icc, size_t icc_size,
                                  const uhdr_hip_metadata_t* metadata, void* out, size_t out_capacity, size_t* out_size);
/* IccHelper::writeIccProfile (lib/src/icc.cpp:410-600) for transfer_function == UHDR_HIP_TF_SRGB (the profile of the SDR base
 * image): "ICC_PROFILE\0" + chunk bytes + profile, as it goes into the primary JPEG's APP2.  Host code. */
int uhdr_hip_icc_profile(int transfer_function, int color_gamut, void* out, size_t out_capacity, size_t* out_size);

/* JpegR::encodeJPEGR, every overload (lib/include/ultrahdr/jpegr.h:81-185,263-265; lib/src/jpegr.cpp:186-631).  toneMap,
 * generateGainMap, the BT.601 re-encode of the SDR image (16-aligned zero-padded copy + convertYuv unless it is P3 already), the
 * JPEG compressions (gain map at quality 85, SDR image at `quality` with the ICC profile) and API-3's JPEG decode run on the
 * device; the container is assembled on the host.  Raw images live in mem_space; compressed inputs, exif and `out` are HOST
 * memory.  exif: payload of the APP1 segment ("Exif\0\0"...), or NULL.  sdr_jpeg_gamut: the colorGamut field of the reference's
 * compressed struct (used when the JPEG carries no ICC profile).  Status values and their order are the reference's
 * (areInputArgumentsValid :75-183, then each overload's own checks); ERROR_INSUFFICIENT_RESOURCE (with *out_size set) when
 * out_capacity is too small, where the reference's Write() fails the same way (:46-61).
 *   api0: P010                       -> toneMap, then as api1                          (:186-247)
 *   api1: P010 + YUV420                                                                (:249-381)
 *   api2: P010 + YUV420 + SDR JPEG   -> gain map from the planes, container around the given JPEG   (:384-437)
 *   api3: P010 + SDR JPEG            -> JPEG decoded (BT.601), gain map, container     (:439-500)
 *   api4: SDR JPEG + gain-map JPEG + metadata -> container; host only                  (:502-560)
 *   apix: YUV420 + gain-map plane + metadata  -> both compressed, container            (:562-631) */
#define UHDR_HIP_ERROR_UNSUPPORTED_WIDTH_HEIGHT (-10002)
#define UHDR_HIP_ERROR_INVALID_STRIDE (-10004)
#define UHDR_HIP_ERROR_INVALID_QUALITY_FACTOR (-10007)
#define UHDR_HIP_ERROR_ENCODE_ERROR (-20001)
int uhdr_hip_jpegr_encode_api0(const uhdr_hip_image_t* p010_image, int hdr_tf, int quality, const void* exif, size_t exif_size,
                               void* out, size_t out_capacity, size_t* out_size, int mem_space, void* stream);
int uhdr_hip_jpegr_encode_api1(const uhdr_hip_image_t* p010_image, const uhdr_hip_image_t* yuv420_image, int hdr_tf, int quality,
                               const void* exif, size_t exif_size, void* out, size_t out_capacity, size_t* out_size, int mem_space,
                               void* stream);
int uhdr_hip_jpegr_encode_api2(const uhdr_hip_image_t* p010_image, const uhdr_hip_image_t* yuv420_image, const void* sdr_jpeg,
                               size_t sdr_jpeg_size, int sdr_jpeg_gamut, int hdr_tf, void* out, size_t out_capacity,
                               size_t* out_size, int mem_space, void* stream);
int uhdr_hip_jpegr_encode_api3(const uhdr_hip_image_t* p010_image, const void* sdr_jpeg, size_t sdr_jpeg_size, int sdr_jpeg_gamut,
                               int hdr_tf, void* out, size_t out_capacity, size_t* out_size, int mem_space, void* stream);
int uhdr_hip_jpegr_encode_api4(const void* sdr_jpeg, size_t sdr_jpeg_size, int sdr_jpeg_gamut, const void* gainmap_jpeg,
                               size_t gainmap_jpeg_size, const uhdr_hip_metadata_t* metadata, void* out, size_t out_capacity,
                               size_t* out_size);
int uhdr_hip_jpegr_encode_apix(const uhdr_hip_image_t* yuv420_image, const uhdr_hip_image_t* gainmap_image,
                               const uhdr_hip_metadata_t* metadata, int quality, const void* exif, size_t exif_size, void* out,
                               size_t out_capacity, size_t* out_size, int mem_space, void* stream);

/* JpegR::getJPEGRInfo (lib/src/jpegr.cpp:633-653; parseJpegInfo :878-915): the two images of a JPEG/R file and what
 * JpegDecoderHelper::getCompressedImageParameters reports for each -- size and the first ICC / EXIF / XMP packets
 * (jpegdecoderhelper.cpp:221-249; the reference copies them into vectors, here they are [offset, size) ranges into the file, size 0
 * = absent; the XMP range is the reference's buffer minus its extra terminating zero).  gainmap may be NULL.  Host code.
 * NO_IMAGES_FOUND / GAIN_MAP_IMAGE_NOT_FOUND from the split, DECODE_ERROR for an unreadable header or one over 8192x8192. */
typedef struct uhdr_hip_jpeg_info {
  size_t offset, size;            /* the JPEG itself (imgData) */
  size_t width, height;
  size_t icc_offset, icc_size;    /* iccData: "ICC_PROFILE\0" + chunk bytes + profile */
  size_t exif_offset, exif_size;  /* exifData: "Exif\0\0" + TIFF */
  size_t xmp_offset, xmp_size;    /* xmpData: namespace + '\0' + packet */
} uhdr_hip_jpeg_info_t;
int uhdr_hip_jpegr_info(const void* jpegr, size_t jpegr_size, uhdr_hip_jpeg_info_t* primary, uhdr_hip_jpeg_info_t* gainmap);

/* getMetadataFromXMP (lib/src/jpegrutils.cpp:436-545) on the gain-map image of a JPEG/R file, as uhdr_dec_probe does after
 * getJPEGRInfo (lib/src/ultrahdr_api.cpp:1038-1108).  Host code.  METADATA_ERROR when the packet is missing or malformed. */
int uhdr_hip_jpegr_metadata(const void* jpegr, size_t jpegr_size, uhdr_hip_metadata_t* metadata);

/* ---- batches (device memory only, asynchronous on `stream`) ------------------------------ */
/* The reference processes one image per call; a batch is n independent calls with identical
 * (hdr_tf, sdr_is_601 | metadata, output_format, max_display_boost).  Images of equal size share
 * one kernel launch (grid.y = image).  Descriptor arrays live in HOST memory and are consumed
 * before the call returns; the data pointers inside them are DEVICE pointers. */

/* content_minmax (optional, DEVICE pointer to 2*n floats): per image i, [2i] = min and [2i+1] =
 * max over map pixels of the UNCLAMPED gain (gainmapmath.cpp:531-534 before the clamp).  This
 * statistic has no reference counterpart (the reference writes constants, ultrahdr.cpp:250-257);
 * the emitted metadata stays the reference's constants. */
int uhdr_hip_generate_gainmap_batch(int n, const uhdr_hip_image_t* yuv420_images,
                                    const uhdr_hip_image_t* p010_images, int hdr_tf,
                                    uhdr_hip_metadata_t* metadata, uhdr_hip_image_t* dests,
                                    int sdr_is_601, float* content_minmax, void* stream);

int uhdr_hip_generate_gainmap_batch_ex(int n, const uhdr_hip_image_t* yuv420_images,
                                       const uhdr_hip_image_t* p010_images, int hdr_tf,
                                       uhdr_hip_metadata_t* metadata, uhdr_hip_image_t* dests,
                                       int sdr_is_601, int generate_mode, float* content_minmax, void* stream);

int uhdr_hip_apply_gainmap_batch(int n, const uhdr_hip_image_t* yuv420_images,
                                 const uhdr_hip_image_t* gainmap_images,
                                 const uhdr_hip_metadata_t* metadata, int output_format,
                                 float max_display_boost, uhdr_hip_image_t* dests, int apply_mode,
                                 void* stream);

/* ---- introspection for tests -------------------------------------------------------------- */
/* copies the 4 Shepard IDW weight tables (standard, no-right, no-bottom, corner; each
 * scale*scale*4 floats; gainmapmath.h:184-228) the apply kernels use for `scale` into out[] */
int uhdr_hip_idw_tables(int scale, float* out);

/* copies static LUT `which` (0 kSrgbInvOETF[1024], 1 kHlgInvOETF[4096], 2 kPqInvOETF[4096], 4 kHlgOETF[65536],
 * 5 kPqOETF[65536]; gainmapmath.cpp:21-64) from the device into out[capacity] (HOST); *count = its length */
int uhdr_hip_lut_table(int which, float* out, size_t capacity, size_t* count);
/* GainLUT::mGainTable (gainmapmath.h:151-182) as LUT-mode apply builds it: 1024 floats into out (HOST);
 * with_display_boost == 0 is GainLUT(metadata), otherwise GainLUT(metadata, display_boost) */
int uhdr_hip_gain_lut(const uhdr_hip_metadata_t* metadata, int with_display_boost, float display_boost, float* out);

/* evaluates one scalar device function over n floats (DEVICE pointers), out[i] = f(in[i]):
 *   fn 0/1/2  sRGB / HLG / PQ inverse OETF as generate computes them (lean f64 + rounding test + exact fallback)
 *   fn 3      encodeGain byte (as float) of gain in[i] for (min_boost, max_boost), generate's version
 *   fn 10..13 the same four through the exact (ocml f64) path;  14/15 HLG / PQ OETF exact
 *   fn 20/24/25 apply-FAST sRGB EOTF / HLG OETF / PQ OETF;  21/22 the f32 HLG / PQ inverse OETF of generate's pre-filter; 23 v_log_f32
 *   fn 40/41/42/44/45 srgbInvOetfLUT / hlgInvOetfLUT / pqInvOetfLUT / hlgOetfLUT / pqOetfLUT; 46 GainLUT(min, max,
 *              displayBoost = max).getGainFactor(in[i])
 *   fn 30/31   gain-map byte -> float through the constant division / the IEEE division (in[i] = byte as float)
 *   fn 100/101 1.0 where the lean path of fn 0/1 passed its rounding test, else 0.0
 * Used by the exhaustive transfer-function tests. */
int uhdr_hip_eval_transfer(int fn, const float* in, float* out, size_t n, float min_boost, float max_boost,
                           void* stream);

/* Bench / test support: the deterministic synthetic frame pair of SURVEY.md 8(d) (an LCG; tests/ and bench.py compare it with the
 * oracle's serial loop), written into device memory: p010 = width*height*3/2 uint16 (luma, then interleaved UV, values in the
 * legal 10-bit ranges << 6), yuv = width*height*3/2 bytes (Y, U, V).  width and height even; enqueued on `stream`. */
int uhdr_hip_synth_lcg_frame(size_t width, size_t height, unsigned int seed, void* p010, void* yuv, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* UHDR_HIP_H */
