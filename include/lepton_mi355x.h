/*
 * lepton_mi355x.h -- C ABI of liblepton_mi355x.so, the MI355X-native drop-in for Lepton's
 * arithmetic-coding hot path (the code behind BaseEncoder::encode_chunk / BaseDecoder::decode_chunk,
 * src/lepton/base_coders.hh:26-65 of dropbox/lepton).  Plain pointers and sizes only.
 *
 * Layers, bottom up:
 *   1. lep_gpu_*        the hot path itself: batches of (image x thread-segment) work items coded by
 *                        HIP kernels on gfx950.  Replaces VP8ComponentEncoder::vp8_full_encoder's
 *                        per-segment loop (src/lepton/vp8_encoder.cc:239-445, 460-519) and
 *                        VP8ComponentDecoder::decode_chunk / LeptonCodec::decode_row
 *                        (src/lepton/vp8_decoder.cc:387-490, src/lepton/lepton_codec.cc:7-47,119-309).
 *   2. lep_jpeg_* / lep_file_*   host-side callers either side of the hot path (JPEG <-> coefficient
 *                        frames, .lep container); what src/lepton/jpgcoder.cc + recoder.cc do.
 *   3. lep_compress / lep_decompress   whole-file convenience = `lepton in.jpg out.lep` and back.
 *
 * Return values are the reference's process exit codes (src/vp8/util/memory.hh:13-40); 0 = SUCCESS.
 * There is NO CPU fallback: every lep_gpu_* entry point fails with LEP_GPU_ERROR when no gfx950
 * device or kernel image is available.
 */
#ifndef LEPTON_MI355X_H
#define LEPTON_MI355X_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum {
    LEP_SUCCESS = 0, LEP_ASSERTION_FAILURE = 1, LEP_CODING_ERROR = 2, LEP_SHORT_READ = 3,
    LEP_UNSUPPORTED_4_COLORS = 4, LEP_COEFFICIENT_OUT_OF_RANGE = 6, LEP_STREAM_INCONSISTENT = 7,
    LEP_PROGRESSIVE_UNSUPPORTED = 8, LEP_SAMPLING_BEYOND_TWO_UNSUPPORTED = 10,
    LEP_THREADING_PARTIAL_MCU = 12, LEP_VERSION_UNSUPPORTED = 13, LEP_ONLY_GARBAGE_NO_JPEG = 14, LEP_OS_ERROR = 33,
    LEP_SAMPLING_BEYOND_FOUR_UNSUPPORTED = 11,
    LEP_OOM = 37,                      /* a .lep whose header claims sizes beyond the reference's 576 MiB arena (its allocator's exit code) */
    LEP_TOO_MUCH_MEMORY_NEEDED = 38,   /* coefficient frame beyond the reference's default budget of 4,423,680 blocks (566 MB) */
    LEP_ROUNDTRIP_FAILURE = 41, LEP_UNSUPPORTED_JPEG = 42, LEP_UNSUPPORTED_JPEG_WITH_ZERO_IDCT_0 = 43,
    LEP_BUFFER_TOO_SMALL = 100,
    LEP_GPU_PATH_DECLINED = 101,       /* lep_file_recode_finish: the scan encoder stopped at the cut of a truncated file before the byte bound was
                                          reached (or does not take the file): call lep_file_recode, the host re-coder, instead */
    LEP_GPU_ERROR = 120
};

#define LEP_MAX_COMPONENTS 3   /* default reference build: ColorChannel::NumBlockTypes == 3 */
#define LEP_MAX_SEGMENTS 16    /* MuxReader::MAX_STREAM_ID, src/io/MuxReader.hh:201 */

/* One image's coefficient frame: what UncompressedComponents exposes to encode_chunk
 * (src/lepton/uncompressed_components.hh:24-302; consumed at src/lepton/vp8_encoder.cc:521-548). */
typedef struct lep_image_desc {
    int32_t ncomp;                                   /* get_num_components() */
    int32_t mcu_rows;                                /* get_mcu_count_vertical() */
    int32_t width_blocks[LEP_MAX_COMPONENTS];        /* block_width(c) */
    int32_t height_blocks[LEP_MAX_COMPONENTS];       /* full_component_nosync(c).original_height() */
    int32_t coded_blocks[LEP_MAX_COMPONENTS];        /* component_size_in_blocks(c) (truncated files) */
    int32_t coded_height[LEP_MAX_COMPONENTS];        /* get_max_coded_heights()[c] */
    uint16_t qtable_zigzag[LEP_MAX_COMPONENTS][64];  /* get_quantization_tables(c), zig-zag order */
    /* per component: width*height AlignedBlocks = 64 x int16 in "aligned" order
     * (src/vp8/util/aligned_block.hh:32-44).  HOST pointers for lep_gpu_*_host, DEVICE pointers
     * for lep_gpu_*_device. */
    int16_t *blocks[LEP_MAX_COMPONENTS];
} lep_image_desc;

/* One thread segment of one image = one independent arithmetic-coded stream = one wavefront.
 * luma_y_start/end come from ThreadHandoff (src/lepton/thread_handoff.hh:8-39). */
typedef struct lep_segment {
    int32_t image;          /* index into the images[] array of the call */
    int32_t luma_y_start;
    int32_t luma_y_end;
    int32_t is_last;        /* last segment of its image runs to the end (vp8_encoder.cc:280) */
} lep_segment;

typedef struct lep_bytes { uint8_t *data; size_t len; size_t cap; } lep_bytes;

/* ---- layer 1: the GPU hot path ------------------------------------------------------------- */
typedef struct lep_gpu lep_gpu;   /* owns a HIP stream, per-segment models and staging buffers */

int lep_gpu_create(int device, lep_gpu **out);
void lep_gpu_destroy(lep_gpu *g);
const char *lep_gpu_last_error(lep_gpu *g);
int lep_gpu_device(lep_gpu *g);   /* the HIP device this object was created on */
size_t lep_gpu_debug_huffenc(lep_gpu *g, void *out, size_t cap);   /* diagnosis: the lane-per-unit scan encoder's work area after its last launch */
int lep_gpu_pci_bus_id(lep_gpu *g, char *out, int cap);   /* its PCI address ("0000:c1:00.0", cap >= 16): which physical GPU a rank drives */

/* Encode nseg segments of nimg images.  Host variant: blocks[] are host pointers, copied to HBM,
 * coded, streams copied back into out[i] (out[i].data with capacity out[i].cap; len is set).
 * status[i] receives the per-segment exit code. */
int lep_gpu_encode_host(lep_gpu *g, const lep_image_desc *images, int nimg, const lep_segment *segs, int nseg,
                        lep_bytes *out, int32_t *status);
/* Decode: in[i] are the de-multiplexed streams; coefficient rows of each segment are written into
 * images[].blocks (host). */
int lep_gpu_decode_host(lep_gpu *g, const lep_image_desc *images, int nimg, const lep_segment *segs, int nseg,
                        const lep_bytes *in, int32_t *status);

/* Device-resident variants (inputs already in HBM; nothing crosses PCIe inside the call):
 * images[].blocks are device pointers; streams live in one device arena: segment i uses
 * [stream_offsets[i], stream_offsets[i+1]) of d_streams; d_stream_len[i] (device, uint32) is
 * written by encode and read by decode.  hip_stream may be NULL (library stream).
 * The calls enqueue work and return; lep_gpu_sync waits.  kernel_ms (may be NULL) receives the
 * HIP-event time of the kernel(s) when sync'd via lep_gpu_last_kernel_ms. */
int lep_gpu_encode_device(lep_gpu *g, const lep_image_desc *images, int nimg, const lep_segment *segs, int nseg,
                          uint8_t *d_streams, const uint64_t *stream_offsets, uint32_t *d_stream_len,
                          int32_t *d_status, void *hip_stream);
int lep_gpu_decode_device(lep_gpu *g, const lep_image_desc *images, int nimg, const lep_segment *segs, int nseg,
                          const uint8_t *d_streams, const uint64_t *stream_offsets, const uint32_t *d_stream_len,
                          int32_t *d_status, void *hip_stream);
/* Workspace set (0 or 1) the next lep_gpu_*_device launches use.  Two launches that overlap in time (different hip streams) must
 * use different sets; a set may be reused once the launch that used it has finished (stream order does that when the same
 * stream always goes with the same set).  Default 0. */
int lep_gpu_use_arena(lep_gpu *g, int k);
/* on = 1: the decode launches that follow will share the chip with a neighbour launch on another stream (both workspace sets in use):
 * they take the register budget that lets two launches' wavefronts sit on one SIMD.  0 = back to choosing by launch size. */
int lep_gpu_expect_company(lep_gpu *g, int on);
/* A caller that passed streams of its own (`hip_stream` arguments) calls this BEFORE destroying them: the object's descriptor-upload
 * ring holds events recorded on those streams; this waits for the copies and drops the events. */
int lep_gpu_settle_uploads(lep_gpu *g);
int lep_gpu_sync(lep_gpu *g);
double lep_gpu_last_kernel_ms(lep_gpu *g);          /* HIP-event duration of the most recent encode / decode / scan launch (ms; -1: none timed) */
/* The split-phase encoder (lep_enc5.h) is several kernels: stage times of the most recent encode launch that used it, in
 * order count + plan, emit, fold, gather, write (HIP events on the launch stream); returns how many were written (0: the
 * launch was a single-kernel one). */
int lep_gpu_last_stage_ms(lep_gpu *g, double *ms, int cap);
const char *lep_gpu_last_kernel_name(lep_gpu *g);   /* which kernel generation / register-budget variant that launch used */
/* JPEG Huffman re-encode of decoded coefficient frames on the GPU (replaces recode_one_mcu_row / encode_block_seq,
 * src/lepton/recoder.cc:316-412, 245-314, for whole, untruncated sequential scans): one wavefront per thread segment writes
 * that segment's scan bytes (FF00-stuffed, RST markers included) to d_out + segs[i].out_off, at most segs[i].out_cap of
 * them; d_out_len[i] receives the count.  images[].blocks are device pointers.  lep_file_recode_plan fills both structs. */
typedef struct lep_huff_image {
    int32_t ncomp, mcuh, mcuv, mcuc;
    int32_t rsti, padbit;
    uint32_t rst_limit;
    int32_t interleaved;                 /* 1: MCUs of hs x vs blocks per component.  A one-component file (never interleaved: MCU = one block, the scan steps over the
                                          * frame's padding blocks) is planned as mcuh x mcuv = its nch x ncv blocks, hs = vs = 1, block rows bch apart, segments in block rows */
    int32_t hs[4], vs[4], bch[4];
    int32_t dc_tbl[4], ac_tbl[4];
    int32_t scan_cmp[4];
    int32_t trunc_bc[4];                 /* a file cut inside its scan (EEE section): blocks of each component in front of the cut; 0 = whole */
    const int16_t *blocks[4];
    uint32_t code[4][256];               /* [0..1] DC, [2..3] AC tables: code length << 16 | code */
} lep_huff_image;
typedef struct lep_huff_segment {
    int32_t image, mcu_row0, mcu_row1;
    uint32_t overhang;                   /* overhang_byte | num_overhang_bits << 8 (ThreadHandoff) */
    int16_t last_dc[4];
    uint64_t out_off;
    uint32_t out_cap;
    uint32_t pad;
} lep_huff_segment;
/* What a segment's writer ends in -- partial byte, its bit count, last DC per component -- i.e. what the NEXT hand-off must
 * have recorded: the reference asserts all three at every segment end (recode_physical_thread, src/lepton/recoder.cc:625-640);
 * lep_file_recode_finish holds them against the file's hand-offs.  attempted = bytes before clipping to out_cap. */
typedef struct lep_huff_end {
    uint32_t attempted;
    uint8_t overhang_byte, num_overhang_bits;
    int16_t last_dc[4];
    uint16_t pad;
} lep_huff_end;
/* d_ends: nseg records in device memory, or NULL.  Segments (of MCU-interleaved scans and of one-component files, with or without restart intervals) are written with one
 * lane per run of at most eight MCUs (lep_huff_simt.h: count, prefix sums, code, stuff; a run ends where its restart interval does and
 * carries the pad bits and the marker), the others with one wavefront per segment (lep_huff.h);
 * same bytes, same end states (LEP_HUFFENC_SIMT=0: the wavefront form for all).  `pad` of a segment is the library's: pass it as 0 (lep_file_recode_plan does).
 * The call copies its arrays before it returns and does not wait for the stream. */
int lep_gpu_huffman_encode_device(lep_gpu *g, const lep_huff_image *images, int nimg, const lep_huff_segment *segs, int nseg,
                                  uint8_t *d_out, uint32_t *d_out_len, lep_huff_end *d_ends, void *hip_stream);
/* The same for PROGRESSIVE files (BASELINE.json configs[4]; replaces the scan loop of recode_jpeg, src/lepton/jpgcoder.cc:3309-3716,
 * with encode_dc_prg_*, encode_ac_prg_fs / _sa, encode_eobrun, encode_crbits :4991-5400): every scan of a progressive file is a
 * function of the finished frame alone, so its bytes (FF00-stuffed, restart markers included) are written to d_out +
 * scans[i].out_off by one LANE per 32 blocks (lep_huffprog_simt.h: count / place / assign / code / stuff; files with restart
 * intervals: one wavefront per scan, lep_huffprog.h); d_out_len[i] = byte count, bit 31 set = the scan outgrew its slot or its
 * scratch (let the host re-coder do that file).  d_corr: scratch for correction bits held back behind end-of-band runs
 * (scans[i].corr_off / corr_cap dwords).  lep_file_recode_plan_progressive fills both structs. */
typedef struct lep_huffprog_image {
    int32_t ncomp, mcuh, mcuv, mcuc;
    int32_t rsti, padbit;
    int32_t hs[4], vs[4], bch[4], bcv[4], nch[4], ncv[4], mbs[4];
    const int16_t *blocks[4];
} lep_huffprog_image;
typedef struct lep_huffprog_scan {
    int32_t image;
    int32_t cmpc, cmp[4];                /* components of the scan, in scan order */
    int32_t from, to, sah, sal;          /* spectral band, successive approximation high / low */
    int32_t max_eobrun;
    int32_t tbl[4];                      /* DC scans: table slot (0 / 1) of each scan component */
    int32_t rsti;                        /* this scan's restart interval (a DRI segment may stand in front of any scan: phone cameras write one per scan);
                                          * -1 = the image's (lep_huffprog_image.rsti).  Sits where the struct had four bytes of padding. */
    uint64_t out_off;
    uint32_t out_cap;
    uint32_t corr_off, corr_cap;         /* dwords */
    uint32_t file_bound;                 /* bytes ALL scans of this scan's image produce together at most (they are parts of one file: its
                                          * size); 0 = unknown, the sum of their out_cap.  Sizes the bit buffers of the lane-per-unit kernels. */
    uint32_t code[4][256];               /* length << 16 | code.  DC scans: [0..1] = DC tables 0 / 1; AC scans: [0] = the component's AC table; scans of
                                          * sequential frames (from 0 / to 63): [0..1] = DC tables 0 / 1, [2..3] = AC tables 0 / 1, tbl[i] = DC table | AC table << 8
                                          * of scan component i */
} lep_huffprog_scan;
int lep_gpu_huffman_progressive_encode_device(lep_gpu *g, const lep_huffprog_image *images, int nimg, const lep_huffprog_scan *scans,
                                              int nscan, uint8_t *d_out, uint32_t *d_corr, uint32_t *d_out_len, void *hip_stream);
/* JPEG Huffman scan decode on the GPU (replaces decode_jpeg / decode_block_seq, src/lepton/jpgcoder.cc:2799-3302,
 * 4893-4966, for single-scan sequential files -- an interleaved scan of all components, or a one-component file, which is planned as
 * mcuh x mcuv = its nch x ncv blocks with hs = vs = 1 and block rows bch apart): one wavefront per image decodes the un-stuffed scan into
 * the zero-filled device frame images[i].blocks and writes images[i].mcuv + 1 records (bit position + last DC per MCU row,
 * final record: pad-bit pattern and status) at d_rows + images[i].rows_off.  lep_jpeg_open_gpu fills the struct. */
#define LEP_HUFFDEC_EARLY_EOF 1          /* lep_huffdec_image.flags */
#define LEP_HUFFDEC_RST_TABLE 2          /* ... a scan with restart intervals whose markers all stand where they should: their positions (uint32
                                            offsets into the un-stuffed scan, (mcuc - 1) / rsti of them) follow the scan bytes at
                                            scan + LEP_HUFFDEC_SCAN_ROOM(scan_len); every interval is then decoded by a lane of its own */
#define LEP_HUFFDEC_SCAN_ROOM(scan_len) ((((size_t)(scan_len)) + 64 + 15) & ~(size_t)15)   /* scan bytes + zero padding, 16-byte multiple */
#define LEP_HUFFDEC_ROW_TRUNCATED 0x40000000   /* final lep_huffdec_row.aux: the scan stopped in mid-image; .bitpos = blocks decoded (scan order) */
typedef struct lep_huffdec_image {
    const uint8_t *scan;                 /* device: un-stuffed scan bytes, 16-byte aligned, followed by >= 32 zero bytes */
    uint32_t scan_len;
    int32_t ncomp, mcuh, mcuv, mcuc, rsti;
    int32_t flags;                       /* LEP_HUFFDEC_EARLY_EOF: the file ends inside its scan (no EOI): the scan may stop in mid-image */
    int32_t reserved0;
    int32_t hs[4], vs[4], bch[4], dc_tbl[4], ac_tbl[4], scan_cmp[4];
    int16_t *blocks[4];                  /* device: zero-filled coefficient frame */
    uint64_t rows_off;
    uint16_t lut[4][512];
    int32_t maxcode[4][8], valoff[4][8];   /* codes of 9..16 bits: largest code per length (-1 none), symbol index - code */
    uint8_t longsym[4][256];               /* their symbols in canonical order */
} lep_huffdec_image;
typedef struct lep_huffdec_row {
    uint32_t bitpos;
    int16_t last_dc[4];
    int32_t aux;
} lep_huffdec_row;
int lep_gpu_huffman_decode_device(lep_gpu *g, const lep_huffdec_image *images, int nimg, lep_huffdec_row *d_rows, void *hip_stream);
/* The same for PROGRESSIVE files (replaces the progressive branches of decode_jpeg's scan loop, src/lepton/jpgcoder.cc:2975-3260,
 * with decode_dc_prg_*, decode_ac_prg_fs / _sa, decode_eobrun_sa, skip_eobrun :4968-5335, :5462-5500): one wavefront per
 * (image, scan) whose 64 lanes decode the codes that would start at the next 64 bits while the scalar unit hops from code to
 * code (lep_huffprogdec_win.h; scans with restart intervals: uniform vector code, lep_huffprogdec.h).  A refinement scan must see what the earlier scans of its band wrote, so every descriptor carries a
 * dependency `level`.  Up to 16384 scans go out as ONE launch, ordered by level, in which a scan waits -- MCU row by MCU row,
 * on a progress word the scans in front of it publish -- for the scans of its file (same frame pointers) whose component and
 * band meet its own: a file then takes as long as its longest scan instead of the sum over its levels.  Larger calls (or
 * LEP_HUFFPROG_PIPELINE=0) launch level after level on the stream.  Each scan writes its coefficients into the
 * zero-filled frame t.blocks, the first scan of a file also one record per MCU row at d_rows + t.rows_off, and every scan a
 * final record {bits consumed, last DC, pad bits | status << 8} at d_rows + result_off; a non-zero status anywhere sends the
 * whole file to the host parser.  lep_jpeg_open_gpu_progressive fills the descriptors. */
typedef struct lep_huffprogdec_scan {
    lep_huffdec_image t;                 /* scan = this scan's bytes; lut[0..1] DC tables 0 / 1, lut[2] the scan's AC table */
    int32_t cmpc, cmp[4];
    int32_t from, to, sah, sal;
    int32_t bcv[4], nch[4], ncv[4], mbs[4];
    int32_t tbl[4];
    int32_t max_eobrun;
    int32_t want_rows;
    int32_t level;
    int32_t pad;
    uint64_t result_off;
} lep_huffprogdec_scan;
int lep_gpu_huffman_progressive_decode_device(lep_gpu *g, const lep_huffprogdec_scan *scans, int nscan, lep_huffdec_row *d_rows, void *hip_stream);
/* The same result with one LANE per subsequence (lep_huffdec_simt.h; the batch compressor's default): a scan is cut into thousands of subsequences, every lane decodes one with a
 * bit reader of its own -- a guess from its first bit, settle passes from where the lane in front ended until no end state moves,
 * a prefix sum, a write pass into the ZERO-FILLED frame.  Same records, same frame, same status semantics as the single-wave
 * kernel (bit-exact against it on MI355X and in the lane-loop emulation); scans with restart intervals are taken when the image
 * carries LEP_HUFFDEC_RST_TABLE (lane = restart interval) and refused otherwise (LEP_ASSERTION_FAILURE: the single-wave kernel's).
 * LEP_HUFFDEC_SIMT_BITS forces the subsequence length (tests). */
int lep_gpu_huffman_decode_simt_device(lep_gpu *g, const lep_huffdec_image *images, int nimg, lep_huffdec_row *d_rows, void *hip_stream);
/* plain device memory helpers so non-torch callers need no HIP binding */
/* Device self-test of kernel arithmetic that has no CPU twin (exhaustive: the float-reciprocal Branch probability of the
 * kernels against integer division, src/vp8/model/branch.hh:82-125).  0 = exact everywhere. */
int lep_gpu_selftest(lep_gpu *g);
/* Gives back the device memory the object caches between launches (per-segment models, neighbour rings, the split-phase
 * encoder's scratch: ~140 MB per 4K image of the largest launch so far) -- to the object's own POOL: the workspaces are virtual
 * address ranges into which 512 MB chunks are mapped, a trimmed workspace is unmapped and its chunks wait for the next workspace that
 * grows (of whatever kind).  Cheap in both directions (96 GB: 19 ms to unmap, 14 ms to map again) -- what is NOT cheap is taking
 * memory from the driver, which clears it: ~40 ms per GB (hipMalloc of 96 GB: 3 - 4 s); round 3 measured that as "memory serves
 * the kernels slower after a trim".  Waits for the device. */
int lep_gpu_trim(lep_gpu *g);
/* lep_gpu_trim, and the pool handed to the driver: for a process that wants the device's memory for something else.  (The
 * library does this itself before it reports an allocation failure of its staging.) */
int lep_gpu_release_memory(lep_gpu *g);
/* Profiling builds (-DLEP_PROF) only: per-phase shader-clock totals [64 segments][32 slots] of the last decoder launch. */
int lep_gpu_debug_prof(lep_gpu *g, uint64_t *out);
int lep_gpu_malloc(lep_gpu *g, size_t bytes, void **dptr);
int lep_gpu_free(lep_gpu *g, void *dptr);
int lep_gpu_memcpy_h2d(lep_gpu *g, void *dst, const void *src, size_t bytes);
int lep_gpu_memcpy_d2h(lep_gpu *g, void *dst, const void *src, size_t bytes);
int lep_gpu_memcpy_d2d(lep_gpu *g, void *dst, const void *src, size_t bytes);
int lep_gpu_memset(lep_gpu *g, void *dst, int value, size_t bytes);

/* ---- layer 2: host-side callers of the hot path --------------------------------------------- */
typedef struct lep_jpeg lep_jpeg;   /* a parsed JPEG: header bytes, coefficient frame, row hand-offs */
typedef struct lep_file lep_file;   /* a parsed .lep: header sections, hand-offs, de-muxed streams */

/* JPEG -> coefficient frame (read_jpeg + decode_jpeg, src/lepton/jpgcoder.cc:2269-2466, 2799-3302) */
int lep_jpeg_open(const uint8_t *jpg, size_t len, int allow_progressive, lep_jpeg **out);
/* the same, with the coefficient frame decoded straight into caller-provided memory (frame_cap bytes, planes back to back,
 * e.g. pinned staging memory): no allocation, page faults or later copy; falls back to owned storage when it is too small */
int lep_jpeg_open_into(const uint8_t *jpg, size_t len, int allow_progressive, void *frame_mem, size_t frame_cap, lep_jpeg **out);
/* frame size from the SOF marker alone (no scan decode) */
int lep_jpeg_peek_frame_bytes(const uint8_t *jpg, size_t len, size_t *bytes);
/* JPEG parse with the scan decode left to lep_gpu_huffman_decode_device: splits the file and reads the tables only.
 * *eligible = 0: the file needs the host decoder (call lep_jpeg_open / lep_jpeg_open_into instead).  Otherwise *image is
 * filled except for scan / blocks / rows_off (device addresses, the caller's), lep_jpeg_scan_bytes gives the bytes to
 * upload, and lep_jpeg_finish_gpu turns the kernel's row records into hand-offs (non-zero: irregular scan, use the host). */
int lep_jpeg_open_gpu(const uint8_t *jpg, size_t len, lep_jpeg **out, lep_huffdec_image *image, int *eligible);
/* the parse behind lep_compress_slice: len already bounded by -trunc; lep_jpeg_plan / lep_jpeg_write_lep then produce the 'Y' file */
int lep_jpeg_open_slice(const uint8_t *jpg, size_t len, size_t start_byte, lep_jpeg **out);
/* `lepton -embedding=<offset>` (jpgcoder.cc:1135-1137, 2275-2282; test_suite/test_embedded.sh): a JPEG that sits `offset` bytes
 * into a larger blob; the .lep ('PGE' section + ordinary garbage) restores the whole blob */
int lep_jpeg_open_embedded(const uint8_t *blob, size_t len, size_t offset, lep_jpeg **out);
int lep_compress_embedded(lep_gpu *g, const uint8_t *blob, size_t len, size_t offset, lep_bytes *out);
/* progressive files, and sequential frames coded in several scans (their scans: from 0 / to 63, t = the frame with all four tables,
 * a record per MCU row of the scan's own geometry; lep_gpu_huffman_progressive_decode_device hands them to the sequential scan decoders):
 * after lep_jpeg_open_gpu answered *eligible = 0.  Fills up to `cap` scan descriptors (t.scan = byte offset of
 * the scan inside lep_jpeg_scan_bytes, t.rows_off / result_off relative to the file's first record -- the caller adds its arena
 * offsets and sets t.blocks); *rows_needed = records the file needs in the row arena.  _finish turns the kernels' records into
 * the hand-offs and bookkeeping of the .lep header (non-zero: irregular, use the host parser). */
int lep_jpeg_open_gpu_progressive(lep_jpeg *j, lep_huffprogdec_scan *scans, int cap, int *nscan, int *rows_needed, int *eligible);
int lep_jpeg_finish_gpu_progressive(lep_jpeg *j, const lep_huffprogdec_scan *scans, int nscan, const lep_huffdec_row *rows);
/* The Huffman half of the round-trip check (validation.cc:97-218) for a file whose scans lep_jpeg_open_gpu_progressive took:
 * the plan that writes every scan of the parsed file again on the GPU (lep_gpu_huffman_progressive_encode_device; the caller
 * sets image->blocks, out_off, corr_off) and where each scan's own bytes lie in the file (first byte, length -- up to the
 * marker that ends the scan), to be compared with what the kernel wrote.  *eligible = 0: check on the host instead. */
int lep_jpeg_plan_progressive_check(lep_jpeg *j, size_t jpeg_len, lep_huffprog_image *image, lep_huffprog_scan *scans,
                                    uint32_t *file_first, uint32_t *file_len, int cap, int *nscan, int *eligible);
int lep_jpeg_scan_bytes(const lep_jpeg *j, const uint8_t **data, size_t *len);
/* The restart markers of a file lep_jpeg_open_gpu flagged LEP_HUFFDEC_RST_TABLE: the offset in the un-stuffed scan bytes at which
 * each stood.  The caller puts them, as uint32, at scan + LEP_HUFFDEC_SCAN_ROOM(scan_len) on the device. */
int lep_jpeg_scan_restarts(const lep_jpeg *j, const uint32_t **pos, size_t *count);
int lep_jpeg_finish_gpu(lep_jpeg *j, const lep_huffdec_row *rows);
void lep_jpeg_close(lep_jpeg *j);
int lep_jpeg_describe(const lep_jpeg *j, lep_image_desc *desc);          /* host pointers into j */
int lep_jpeg_is_progressive(const lep_jpeg *j);   /* 1: not a single interleaved sequential scan (needs the progressive re-coder) */
/* segment choice of write_ujpg (src/lepton/jpgcoder.cc:3856-3934); returns count, fills segs */
int lep_jpeg_plan(const lep_jpeg *j, int max_threads, lep_segment *segs, int image_index);
/* -maxencodethreads / -minencodethreads / -evensplit (jpgcoder.cc:1064-1095): they change how many thread segments a file gets
 * and where they are cut, i.e. the .lep bytes; 0 / 0 / 0 leaves the reference's defaults (8, 1, by compressed size) */
int lep_jpeg_set_encode_options(lep_jpeg *j, int max_threads, int min_threads, int even_split);
/* `lepton -brotliheader` (src/lepton/jpgcoder.cc:1116-1119, 4038): container format version 2 -- the header is a brotli stream
 * (BrotliCodec::Compress, src/io/BrotliCompression.cc:45-98, with the vendored brotli 1.0.0 encoder: compiled into the library from
 * the reference's dependency tree when build() finds it) and the packets end with FF FE FF.  1 = the default (zlib).  Returns
 * LEP_VERSION_UNSUPPORTED for a version this build cannot write; lep_container_can_write_version asks without a file. */
int lep_jpeg_set_container_version(lep_jpeg *j, int version);
int lep_container_can_write_version(int version);
/* whole .lep file from the per-segment streams (header + mux + trailer) */
int lep_jpeg_write_lep(const lep_jpeg *j, int max_threads, const lep_bytes *streams, int nstreams, lep_bytes *out);
/* The Huffman half of the reference's default round-trip check (src/lepton/validation.cc:97-218; `lepton` without
 * -skipverify exits with ROUNDTRIP_FAILURE when the file it would restore differs from its input): parses `lepdata` (what
 * lep_jpeg_write_lep just produced), re-codes j's own coefficient frame with the .lep's header, and compares the result
 * with `want` (the whole input; for a slice the bytes [start_byte, trunc); for an embedded JPEG the whole blob).  Host
 * only.  The arithmetic-coder half (streams -> the same coefficients) is lep_batch_options.verify's on-GPU comparison.
 * Returns 0, LEP_ROUNDTRIP_FAILURE, or the exit code of a .lep that does not parse.  Files the reference itself cannot
 * restore -- e.g. a one-component frame with sampling factors 2x2, restart markers and more than one MCU row -- end here. */
int lep_jpeg_check_restores(const lep_jpeg *j, const uint8_t *lepdata, size_t lep_len, const uint8_t *want, size_t want_len);

/* .lep -> streams + frame geometry (read_ujpg, src/lepton/jpgcoder.cc:4117-4362) */
int lep_file_open(const uint8_t *lepdata, size_t len, lep_file **out);
/* Format versions >= 2 (brotli header, `lepton -brotliheader`) mark the end of their packets, so several files may follow
 * each other in one stream and restore the concatenation of their JPEGs (jpgcoder.cc:1868-1897): lep_file_consumed = bytes
 * the parsed file occupies; lep_chained_file_follows = another file's magic stands behind them; lep_file_open_next parses
 * it, handing over what `prev` left in the shared header reader (the "CNT" sections `lepton -lepcat` writes,
 * concat.cc:84-95).  lep_decompress walks such streams itself. */
size_t lep_file_consumed(const lep_file *f);
int lep_chained_file_follows(const uint8_t *lepdata, size_t len, size_t consumed);
int lep_file_open_next(const uint8_t *lepdata, size_t len, const lep_file *prev, lep_file **out);
void lep_file_close(lep_file *f);
int lep_file_describe(lep_file *f, lep_image_desc *desc);                /* allocates zeroed host frame */
int lep_file_describe_into(lep_file *f, void *frame_mem, size_t frame_cap, lep_image_desc *desc);   /* frame in caller memory; (NULL, (size_t)-1) = geometry only */
int lep_file_segments(const lep_file *f, lep_segment *segs, lep_bytes *streams, int image_index);
uint32_t lep_file_jpeg_size(const lep_file *f);
size_t lep_file_frame_bytes(const lep_file *f);   /* bytes of the coefficient frame lep_file_describe will expose */
/* coefficient frame -> original JPEG bytes (recode_baseline_jpeg, src/lepton/recoder.cc:694-889) */
int lep_file_recode(lep_file *f, lep_bytes *out);
/* The same with the Huffman coding done by lep_gpu_huffman_encode_device: _plan walks the header and, if the file is
 * eligible (*gpu_ok = 1), fills the image / per-segment parameters (blocks[], image index and out_off are the caller's to
 * set; out_cap is the segment's byte bound); _finish glues header, the segments' scan bytes and the trailer together. */
int lep_file_recode_plan(lep_file *f, lep_huff_image *image, lep_huff_segment *segs, int *nseg, int *gpu_ok);
/* ends: the kernel's per-segment end states (NULL: only the byte counts are held against the hand-offs) */
int lep_file_recode_finish(lep_file *f, const lep_bytes *seg_bytes, const lep_huff_end *ends, int nseg, lep_bytes *out);
/* progressive files: _plan fills the image and up to `cap` scan descriptors (out_cap / corr_cap = what each scan may need;
 * image index, out_off, corr_off and blocks[] are the caller's to set); *gpu_ok = 0: the file keeps the host re-coder
 * (truncated, withheld restart markers ...).  SEQUENTIAL frames coded in several scans are planned here too: their scans have
 * from 0 / to 63 (no progressive scan has), all four tables in code[] and the components' choice of them in tbl[], and
 * lep_gpu_huffman_progressive_encode_device hands them to the sequential scan encoders.  _finish glues header pieces, scans and trailer. */
/* Compression with verification, baseline files: the plan that writes the parsed file's scan again on the GPU from its coefficient frame
 * (lep_gpu_huffman_encode_device; images[].blocks and the segments' out_off are the caller's to set) and, per thread segment, the
 * bytes of the file it must reproduce -- the Huffman half of the reference's round-trip check (src/lepton/validation.cc:97-218),
 * executed, not argued.  *eligible = 0: the file keeps the host check (lep_jpeg_check_restores).
 * NOT a read-only call: the parsed file is lent to the plan for its duration (the tables in front of the scan are read again from the
 * header bytes, a warning level may be set) -- the handle must not be used from another thread while it runs. */
int lep_jpeg_plan_scan_check(lep_jpeg *j, size_t jpeg_len, lep_huff_image *image, lep_huff_segment *segs, uint32_t *file_first, uint32_t *file_len, int cap,
                             int *nseg, int *eligible);
int lep_jpeg_scan_file_range(const lep_jpeg *j, uint32_t *first, uint32_t *len);
int lep_file_recode_plan_progressive(lep_file *f, lep_huffprog_image *image, lep_huffprog_scan *scans, int cap, int *nscan, int *gpu_ok);
int lep_file_recode_finish_progressive(lep_file *f, const lep_bytes *scan_bytes, int nscan, lep_bytes *out);


/* .lep framing pieces, exposed for callers that assemble containers themselves */
typedef struct lep_handoff {           /* ThreadHandoff, src/lepton/thread_handoff.hh:8-39 */
    uint16_t luma_y_start, luma_y_end;
    uint32_t segment_size;
    uint8_t overhang_byte, num_overhang_bits;
    int16_t last_dc[4];
} lep_handoff;
/* 'H', n, then n 16-byte records (ThreadHandoff::serialize / deserialize, thread_handoff.cc:4-76) */
int lep_handoffs_serialize(const lep_handoff *h, int n, uint8_t *out, size_t out_cap);
int lep_handoffs_parse(const uint8_t *data, size_t len, lep_handoff *out, int out_cap);
/* lep_jpeg_plan's choice as hand-off records (what write_ujpg serialises into the header, jpgcoder.cc:3873-3934):
 * segment_size = the JPEG scan bytes each thread segment covers.  Returns the count, or -LEP_BUFFER_TOO_SMALL. */
int lep_jpeg_plan_handoffs(const lep_jpeg *j, int max_threads, lep_handoff *out, int cap);
/* MuxWriter policy (src/io/MuxReader.hh:336-522) fed in the encoder's slice order (vp8_encoder.cc:575-594) */
int lep_mux(const lep_bytes *streams, int nstreams, int version, lep_bytes *out);
/* MuxReader (src/io/MuxReader.hh:230-283): packets until the data runs out; streams[16] are malloc'd */
int lep_demux(const uint8_t *data, size_t len, lep_bytes *streams16);

/* ---- layer 3: whole files ------------------------------------------------------------------- */
int lep_compress(lep_gpu *g, const uint8_t *jpg, size_t len, lep_bytes *out);
/* `lepton -startbyte=<start_byte> -trunc=<trunc>` (jpgcoder.cc:1132-1140, 3801-3843; test_suite/test_2nd_block.sh): the .lep
 * restores bytes [start_byte, trunc) of the JPEG (trunc 0 = to the end) -- how a JPEG stored as fixed-size blocks is
 * compressed block by block.  Format flag 'Y': header segments reduced to the ones the scan needs, hand-off rows in front of
 * start_byte dropped, the bytes up to the first remaining MCU row kept verbatim.  lep_decompress reads such files like any
 * other.  Progressive files cannot be sliced (PROGRESSIVE_UNSUPPORTED), as in the reference. */
int lep_compress_slice(lep_gpu *g, const uint8_t *jpg, size_t len, size_t start_byte, size_t trunc, lep_bytes *out);
int lep_decompress(lep_gpu *g, const uint8_t *lepdata, size_t len, lep_bytes *out);

/* Whole batches of files as a pipeline (what `lepton -socket` workers / src/lepton/socket_serve.cc:312-390 would hand to
 * the GPU): JPEG parsing + Huffman scan decode on a host thread pool, coefficient frames over PCIe on a copy stream while
 * the previous chunk's coder kernels run, streams back, .lep containers written on the host pool -- and the mirror image
 * for decompression.  verify != 0 restores the reference's default round-trip check (src/lepton/validation.cc:97-218) on
 * the GPU: what was just encoded is decoded into a scratch frame and compared with the input frame before the .lep is
 * released; a mismatch gives that file status LEP_ROUNDTRIP_FAILURE.
 * status[i] = exit code of file i (0 = ok, outs[i] is malloc'd: release with lep_free); the return value is non-zero only
 * for failures of the machinery itself (LEP_GPU_ERROR). */
typedef struct lep_batch_options {
    int32_t host_threads;        /* 0 = the CPUs this process may use (affinity mask capped by the cgroup CPU quota) */
    int32_t verify;              /* compress: on-GPU round-trip verification */
    size_t chunk_frame_bytes;    /* cap on coefficient-frame bytes per pipeline chunk; 0 = 32 GiB (1024 4K frames) */
    int32_t host_huffman;        /* 1 = JPEG Huffman decode / re-encode on the host pool (frames cross PCIe) instead of on the GPU */
    int32_t chunk_images;        /* images per pipeline chunk; 0 = automatic: at most 1024 images and 8192 thread segments (the decoder
                                    wavefronts the chip holds at once), chunks of a call balanced.  The decode kernel takes as
                                    long for 100 segments as for 8192, so chunks must be this big */
    int32_t overlap_launches;    /* compress: consecutive chunks' coder kernels on two streams / two workspace sets (experimental; also
                                    LEP_BATCH_OVERLAP=1) */
} lep_batch_options;
typedef struct lep_batch_stats {
    double wall_s;               /* whole call */
    double pipeline_s;           /* first upload .. last container (excludes stream / buffer creation and the first parse) */
    double parse_s, stage_s, write_s;   /* host pool: parse, copies into pinned staging, container / Huffman re-code (summed) */
    double h2d_bytes, d2h_bytes; /* PCIe traffic */
    double alloc_s;              /* inside pipeline_s: (re)allocation of the pinned / device staging buffers (kept between calls) */
    double redone_files;         /* compress: files whose streams outgrew the space reserved from their JPEG size and went through lep_compress */
    double gpu_huffman_files;    /* files whose Huffman scans the GPU decoded (compress) / wrote (decompress); the rest took the host coder */
    double gpu_verified_scans;   /* compress with verify: thread segments (baseline) and scans (progressive) written again on the GPU from the
                                    device frame and compared with the file's own bytes -- the Huffman half of the round-trip check */
} lep_batch_stats;
int lep_compress_batch(lep_gpu *g, const lep_bytes *jpgs, int n, lep_bytes *outs, int32_t *status,
                       const lep_batch_options *opt, lep_batch_stats *stats);
int lep_decompress_batch(lep_gpu *g, const lep_bytes *leps, int n, lep_bytes *outs, int32_t *status,
                         const lep_batch_options *opt, lep_batch_stats *stats);
/* the chunking lep_compress_batch will apply: file_bytes[i] / frame_bytes[i] (lep_jpeg_peek_frame_bytes; 0 = not a usable file)
 * -> chunk k holds files [chunk_first[k], chunk_first[k + 1]); returns the number of chunks (cap = entries of chunk_first) */
int lep_batch_plan(const size_t *file_bytes, const size_t *frame_bytes, int n, const lep_batch_options *opt, int *chunk_first, int cap);
/* progressive files on the GPU scan decoder: scans of its one-launch form that gave up waiting for a scan in front of them since
 * the process started (their files went to the host parser; expected 0 -- the wait cannot deadlock, the count is the safety net's) */
uint64_t lep_jpeg_gpu_scan_wait_timeouts(void);
void lep_batch_release(void);   /* frees the staging buffers the two calls above keep between invocations (not re-entrant) */
/* what the batch calls keep between invocations: pinned host bytes / device bytes of the large staging arenas (either pointer may be NULL) */
void lep_batch_footprint(size_t *pinned_bytes, size_t *device_bytes);
/* test hook: overwrite the pinned staging buffers kept between batch calls with `value` (stale-staging regression tests) */
void lep_batch_debug_poison(int value);

/* ---- serving surface ------------------------------------------------------------------------------
 * The `lepton -socket[=name] / -listen[=port]` protocol (src/lepton/socket_serve.cc:312-390, jpgcoder.cc:1162-1186):
 * a client connects, sends one whole file, half-closes (shutdown(SHUT_WR)) and reads the converted file until EOF --
 * JPEG in -> .lep out, .lep in -> JPEG out, chosen by the first two bytes (jpgcoder.cc:2178-2235).  The zlib socket
 * (<name>.z0 / -zliblisten) returns a decoded JPEG as a zlib stream of stored blocks (src/io/Zlib0.cc:36-120); so does a
 * .lep whose magic is 0xce 0xb6 (jpgcoder.cc:552).  A failed request gets no bytes, only the close; the exit code the
 * reference's forked child would have died with is logged and counted.  -timebound: time_bound_ms after a request's
 * first byte the connection is closed, answered or not (the reference's SIGALRM, jpgcoder.cc:1740-1752).
 * Where the reference forks one process per connection (accept_new_connection, socket_serve.cc:86-116) and bounds them
 * with -maxchildren, this server gathers the requests that arrive within batch_window_us (or until max_batch are
 * waiting) and hands them to lep_compress_batch / lep_decompress_batch as ONE GPU batch; max_connections plays
 * -maxchildren's part (further clients wait in the listen backlog).  One server per GPU: run one process per device. */
typedef int (*lep_serve_process_fn)(void *user, int kind /* 0: JPEG -> .lep, 1: .lep -> JPEG */, const lep_bytes *in, int n,
                                    lep_bytes *outs /* malloc'd by the callee */, int32_t *status);
typedef struct lep_serve_options {
    const char *uds_path;        /* -socket=<name>; the zlib socket is <name>.z0, the lock file <name>.lock; NULL = no UDS */
    const char *zlib_uds_path;   /* NULL = <uds_path>.z0 (the reference's random names are /tmp/<id>.uport and /tmp/<id>.z0) */
    int32_t tcp_port;            /* -listen=<port>; 0 = no TCP listener */
    int32_t zlib_tcp_port;       /* -zliblisten=<port>; 0 = none */
    int32_t listen_backlog;      /* -listenbacklog (default 16) */
    int32_t max_connections;     /* -maxchildren: connections in flight before accept() pauses; 0 = unbounded */
    uint32_t max_file_bytes;     /* larger uploads are dropped; 0 = 256 MiB */
    uint32_t time_bound_ms;      /* -timebound; 0 = none */
    int32_t max_batch;           /* files per GPU batch; 0 = 1024 */
    int32_t batch_window_us;     /* how long the batcher waits for company once a request is complete; 0 = 2000 */
    lep_gpu *gpu;                /* used by the default processor */
    lep_batch_options batch;     /* passed through to lep_*_batch (verify = the reference's default round-trip check) */
    lep_serve_process_fn process;   /* NULL = the GPU batch pipeline; tests substitute the CPU oracle here */
    void *process_user;
} lep_serve_options;
typedef struct lep_serve_stats {
    uint64_t accepted, answered, failed, timed_out, rejected;   /* connections */
    uint64_t batches, largest_batch;
    uint64_t bytes_in, bytes_out;
    int32_t last_failure_code;   /* exit code of the most recent failed request */
} lep_serve_stats;
typedef struct lep_server lep_server;
/* binds and starts the IO + batcher threads; LEP_OS_ERROR if a socket cannot be bound (or the lock is held: another
 * server owns <name>, socket_serve.cc:331-356) */
int lep_serve_start(const lep_serve_options *opt, lep_server **out);
void lep_serve_get_stats(lep_server *s, lep_serve_stats *out);
void lep_serve_stop(lep_server *s);   /* closes listeners and connections, joins, removes the socket files it owns */
/* zlib stream of stored blocks exactly as Zlib0Writer emits it (src/io/Zlib0.cc) */
int lep_zlib0_wrap(const uint8_t *data, size_t len, lep_bytes *out);

void lep_free(void *p);   /* frees lep_bytes.data returned by this library */
const char *lep_version(void);

#ifdef __cplusplus
}
#endif
#endif
