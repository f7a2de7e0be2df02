/* abi_over_oracle.c -- TEST INFRASTRUCTURE.  The four layer-1 entry points integration/mi355x_coders.cc calls
 * (lep_gpu_create / _destroy / _encode_host / _decode_host) implemented over the CPU oracle, so that the adapter -- linked
 * into the real reference by oracle/Makefile.ref (target adapter_check) -- can be EXECUTED in a container without a GPU:
 * the mux writing, the size trailer, decode_row / decode_chunk and the hand-off plumbing run for real; only the arithmetic
 * coding itself is the oracle's instead of the kernels'.  The product library is not involved and never links this. */
#include <stdlib.h>
#include <string.h>

#include "../../include/lepton_mi355x.h"
#include "../../oracle/lepton_oracle.h"

struct lep_gpu { int unused; };

int lep_gpu_create(int device, lep_gpu **out) { (void)device; *out = (lep_gpu *)calloc(1, sizeof(lep_gpu)); return *out ? 0 : LEP_OS_ERROR; }
void lep_gpu_destroy(lep_gpu *g) { free(g); }

static void to_lor(const lep_image_desc *d, lor_image *im) {
    memset(im, 0, sizeof *im);
    im->ncomp = d->ncomp; im->mcu_rows = d->mcu_rows;
    for (int c = 0; c < d->ncomp; ++c) {
        im->blocks[c] = d->blocks[c]; im->width_blocks[c] = d->width_blocks[c]; im->height_blocks[c] = d->height_blocks[c];
        im->coded_blocks[c] = d->coded_blocks[c]; im->coded_height[c] = d->coded_height[c];
        memcpy(im->qtable_zigzag[c], d->qtable_zigzag[c], 128);
    }
}

int lep_gpu_encode_host(lep_gpu *g, const lep_image_desc *images, int nimg, const lep_segment *segs, int nseg, lep_bytes *out, int32_t *status) {
    (void)g; (void)nimg;
    int worst = 0;
    for (int s = 0; s < nseg; ++s) {
        lor_image im;
        to_lor(&images[segs[s].image], &im);
        size_t n = 0;
        const int rc = lor_encode_segment(&im, segs[s].luma_y_start, segs[s].luma_y_end, segs[s].is_last, out[s].data, out[s].cap, &n, NULL);
        out[s].len = rc ? 0 : n;
        status[s] = rc;
        if (rc && !worst) worst = rc;
    }
    return worst;
}

int lep_gpu_decode_host(lep_gpu *g, const lep_image_desc *images, int nimg, const lep_segment *segs, int nseg, const lep_bytes *in, int32_t *status) {
    (void)g; (void)nimg;
    int worst = 0;
    for (int s = 0; s < nseg; ++s) {
        lor_image im;
        to_lor(&images[segs[s].image], &im);
        const int rc = lor_decode_segment(&im, segs[s].luma_y_start, segs[s].luma_y_end, segs[s].is_last, in[s].data, in[s].len, NULL);
        status[s] = rc;
        if (rc && !worst) worst = rc;
    }
    return worst;
}
