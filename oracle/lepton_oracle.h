/*
 * lepton_oracle.h -- CPU restatement of Lepton's per-block context-modelled binary arithmetic
 * coder.  TEST INFRASTRUCTURE ONLY: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may call into this.  The product (lepton_amd/) never links or loads it.
 *
 * Parity is PINNED: see tests/test_oracle_golden.py (golden .lep files of the reference's own
 * test-suite decode to the reference's md5s; streams equal the ones the real reference binary
 * built by oracle/Makefile.ref emits).
 */
#ifndef LEPTON_ORACLE_H
#define LEPTON_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* One image's coefficient frame, as the reference hands it to BaseEncoder::encode_chunk
 * (src/lepton/base_coders.hh:26-41, src/lepton/uncompressed_components.hh:24-302). */
typedef struct lor_image {
    int ncomp;                    /* 1..3 */
    int16_t *blocks[4];           /* per component: width*height blocks of 64 int16 in "aligned" order
                                     (src/vp8/util/aligned_block.hh:32-44,98-161) */
    int width_blocks[4];          /* bch */
    int height_blocks[4];         /* bcv (theoretical height) */
    int coded_blocks[4];          /* trunc_bc  (component_size_in_blocks) */
    int coded_height[4];          /* trunc_bcv (max_coded_heights) */
    int mcu_rows;                 /* mcuv */
    uint16_t qtable_zigzag[4][64];/* per component, zig-zag order as in the DQT segment */
} lor_image;

/* exit codes mirror src/vp8/util/memory.hh:13-40 */
enum { LOR_OK = 0, LOR_ASSERTION_FAILURE = 1, LOR_CODING_ERROR = 2,
       LOR_COEFFICIENT_OUT_OF_RANGE = 6, LOR_STREAM_INCONSISTENT = 7,
       LOR_UNSUPPORTED_ZERO_IDCT_0 = 43, LOR_BUFFER_TOO_SMALL = 100 };

/* Encode one thread segment (rows with luma_y in [luma_y_start, luma_y_end); the last segment runs
 * to the end of the image) into one raw bool-coder stream, including the start marker bin, the 32
 * stop bins and the 0xC0 tail rule.  Returns an exit code. */
int lor_encode_segment(const lor_image *img, int luma_y_start, int luma_y_end, int is_last_segment,
                       uint8_t *out, size_t out_cap, size_t *out_len, uint64_t *bins_coded);

/* Inverse: fills img->blocks for the rows of the segment. */
int lor_decode_segment(lor_image *img, int luma_y_start, int luma_y_end, int is_last_segment,
                       const uint8_t *in, size_t in_len, uint64_t *bins_coded);

size_t lor_model_bytes(void);

/* test knob: see lepton_oracle.c (streams with impossible edge non-zero counts); 0 = off */
extern int lor_test_edge_count_bias;

#ifdef __cplusplus
}
#endif
#endif
