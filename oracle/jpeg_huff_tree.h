/* jpeg_huff_tree.h -- TEST INFRASTRUCTURE ONLY (nothing under lepton_amd/ may include it).
 *
 * The reference's Huffman decoding tree, restated: build_huffcodes (src/lepton/jpgcoder.cc:5507-5606) and the
 * bit-by-bit walk of next_huffcode (src/lepton/jpgcoder.cc:5407-5425).  The product decodes over a sorted list of
 * code words (lepton_amd/csrc/jpeg_scan.cc build_huff_table, jpeg_bits.h next_huffcode); tests/emu/huff_table_check.cc
 * holds the two against each other on tables that follow T.81 Annex C and on tables that do not.
 */
#ifndef LEP_ORACLE_JPEG_HUFF_TREE_H
#define LEP_ORACLE_JPEG_HUFF_TREE_H
#include <stdint.h>
#include <string.h>

typedef struct {
    uint16_t clen[256], cval[256];
    uint16_t l[256], r[256];        /* links: 0 = none, 1..255 = inner node, >= 256 = leaf of symbol (link - 256) */
    int max_eobrun;
} oracle_huff_tree;

/* jpgcoder.cc:5507-5606.  jpeg != 0: the file type is JPEG (a table that runs out of nodes is refused). */
static int oracle_huff_tree_build(const uint8_t* clen, size_t clen_avail, const uint8_t* cval, size_t cval_avail,
                                  int jpeg, oracle_huff_tree* t) {
    int i, j, k = 0, code = 0, node, nextfree = 1;
    memset(t, 0, sizeof *t);
    for (i = 0; i < 16; i++) {                                         /* :5529-5543 */
        int n = (size_t)i < clen_avail ? clen[i] : 0;
        for (j = 0; j < n; j++) {
            size_t at = (size_t)(k & 0xff);
            uint8_t v = at < cval_avail ? cval[at] : 0;
            t->clen[v] = (uint16_t)(1 + i);
            t->cval[v] = (uint16_t)code;
            k++;
            code++;
        }
        code <<= 1;
    }
    for (i = 14; i >= 0; i--)                                          /* :5546-5552 */
        if (t->clen[(i << 4) & 255] > 0) { t->max_eobrun = (2 << i) - 1; break; }
    for (i = 0; i < 256; i++) {                                        /* :5560-5601 */
        node = 0;
        for (j = t->clen[i] - 1; j > 0; j--) {
            if (node <= 0xff) {
                uint16_t* side = ((t->cval[i] >> j) & 1) ? t->r : t->l;
                if (side[node] == 0) side[node] = (uint16_t)nextfree++;
                node = side[node];
            } else if (jpeg) {
                return 0;
            }
        }
        if (node <= 0xff) {
            if (t->clen[i] > 0) {
                if (t->cval[i] & 1) t->r[node] = (uint16_t)(i + 256);
                else t->l[node] = (uint16_t)(i + 256);
            }
        } else if (jpeg) {
            return 0;
        }
    }
    return 1;
}

/* jpgcoder.cc:5407-5425 as a walk over caller-supplied bits: next_bit(ctx) returns the next bit of the scan (0 past its end).
 * Returns the symbol, or a negative value for bits that are no code; *used = bits taken either way. */
static int oracle_huff_tree_walk(const oracle_huff_tree* t, int (*next_bit)(void*), void* ctx, int* used) {
    int node = 0;
    *used = 0;
    while (node < 256) {
        node = next_bit(ctx) ? t->r[node] : t->l[node];
        ++*used;
        if (node == 0) break;
    }
    return node - 256;
}
#endif
