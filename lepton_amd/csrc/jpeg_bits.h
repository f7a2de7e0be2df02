// jpeg_bits.h -- MSB-first bit reader / writer over un-stuffed JPEG scan bytes, with the exact
// end-of-data and position semantics of the reference's abitreader / abitwriter
// (src/lepton/bitops.hh:66-362), because hand-off records (byte position, overhang bits) and the
// handling of truncated files depend on them.
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>
#include "jpeg_model.h"

namespace lep {

struct BitReader {
    const uint8_t* data;
    int size;
    int next_byte = 0;     // first byte not yet loaded into `window`
    int avail = 0;         // unread bits in `window` (right-aligned)
    uint64_t window = 0;
    bool eof = false;

    BitReader(const uint8_t* d, int n) : data(d), size(n) {}

    static inline uint64_t low_bits(uint64_t v, int n) { return n == 0 ? 0 : (v & (~0ULL >> (64 - n))); }

    unsigned read(int nbits) {
        if (eof || !nbits) return 0;
        unsigned out;
        if (nbits > 32 && avail > 0) {
            // Only a corrupt DHT gets here (a DC category used as a bit count).  How many bits such a read consumes in the
            // reference depends on where its 8-byte buffer stands (at most what is left in it + one refill, bitops.hh:262-300),
            // and top_up() keeps this window fuller than that: fall back to the reference's phase first -- its buffer is
            // refilled every 64 consumed bits counted from the start of the data -- then read as it does.
            const long consumed = (long)next_byte * 8 - avail;
            const int cb = (int)((64 - consumed % 64) % 64);   // what the reference's buffer still holds
            if (cb <= avail && ((avail - cb) & 7) == 0) {        // (near the end of the data there may be nothing to give back)
                const int drop = (avail - cb) >> 3;
                next_byte -= drop;
                window = drop >= 8 ? 0 : window >> (drop * 8);
                avail = cb;
            }
        }
        if (nbits >= avail) {
            int took = avail;
            // nbits can be anything up to 255 on a corrupt DHT (a DC category is used as a bit count unchecked, like the
            // reference does, bitops.hh:262-270): shift counts are reduced the way x86 reduces them instead of being undefined
            out = (unsigned)((low_bits(window, avail) << ((nbits - took) & 63)) & ((1u << (nbits & 31)) - 1));
            int want = nbits - took;
            window = took >= 64 ? 0 : window >> took;
            avail = 0;
            if (next_byte == size) { eof = true; return out; }
            int nb = size - next_byte < 8 ? size - next_byte : 8;
            uint64_t w = 0;
            for (int i = 0; i < nb; ++i) w = (w << 8) | data[next_byte + i];
            window = w;
            next_byte += nb;
            avail = nb * 8;
            if (want) {
                if (want <= avail) { out |= (unsigned)(low_bits(window, avail) >> (avail - want)); avail -= want; }
                else { out |= (unsigned)window; window = 0; avail = 0; }
            }
        } else {
            out = (unsigned)(low_bits(window, avail) >> (avail - nbits));
            avail -= nbits;
        }
        return out;
    }
    // Fast path for the Huffman decoders: keep the window topped up and look at the next bits without consuming them.
    // getpos / overhang / the end-of-data rule depend only on (next_byte, avail), not on how the window was filled.
    inline void top_up() {
        while (avail <= 56 && next_byte < size) { window = (window << 8) | data[next_byte++]; avail += 8; }
    }
    inline unsigned peek(int nbits) const { return (unsigned)(low_bits(window, avail) >> (avail - nbits)); }   // needs avail >= nbits
    inline void skip(int nbits) { avail -= nbits; }

    // pad-bit pattern of the current partial byte (consumes it)
    uint8_t unpad(uint8_t fillbit) {
        if ((avail & 7) == 0 || eof) return fillbit;
        int last = (int)read(1);
        fillbit = (uint8_t)last;
        int off = 1;
        while (avail & 7) { last = (int)read(1); fillbit |= (uint8_t)(last << off); ++off; }
        while (off < 7) { fillbit |= (uint8_t)(last << off); ++off; }
        return fillbit;
    }
    // 1 + index of the byte holding the next unread bit
    int getpos() const { return next_byte - 7 + ((64 - avail) >> 3); }
    // bits already consumed from the current byte, and those bits (left-aligned)
    void overhang(uint8_t* nbits, uint8_t* byte) const {
        int rem = (64 - avail) & 7;
        uint8_t cur = 0;
        if (rem) {
            // byte containing the next unread bit: it is the one whose low (8-rem) bits are still in window
            int bit_index_from_low = avail - (8 - rem);   // shift that brings that byte to the bottom
            cur = (uint8_t)(window >> bit_index_from_low);
        }
        *nbits = (uint8_t)rem;
        *byte = (uint8_t)(cur & (uint8_t)(((1 << rem) - 1) << (8 - rem)));
    }
};

// One Huffman symbol (>= 0) or a negative value for bits that are no code of the table.  What the callers can observe besides
// the symbol is how far the reader has moved and whether it has met the end of the data (a failed code AT the end is a truncated
// file, before it a refused one): a code is consumed bit by bit as far as some word of the table still begins with the bits
// read, the first bit that leaves every word included -- jpgcoder.cc:5407-5425 walks its tree that far.
inline int next_huffcode(BitReader& br, const HuffTable& t) {
    if (!br.eof) {
        br.top_up();
        unsigned e = br.avail >= 10 ? t.lut[br.peek(10)] : 0;
        if (!e && br.avail >= 16) {               // a longer code, away from the end of the data
            const int k = t.word_at(br.peek(16));
            if (k >= 0) e = ((unsigned)t.wlen[k] << 8) | t.wsym[k];
        }
        if (e) {
            br.skip((int)(e >> 8));
            if (br.avail == 0 && br.next_byte == br.size) br.eof = true;   // read() sets eof with the read that takes the last bit of the data
            return (int)(e & 255);
        }
    }
    unsigned bits = 0;
    for (int n = 1, k = 0; n <= 16; ++n) {        // k: first word not below the patterns that begin with `bits`
        bits = (bits << 1) | br.read(1);
        const unsigned lo = bits << (16 - n), span = 1u << (16 - n);
        while (k < t.nwords && t.wfirst[k] < lo) ++k;
        if (k == t.nwords || t.wfirst[k] - lo >= span) break;
        if (t.wlen[k] == n) return (int)t.wsym[k];
    }
    return -256;
}

// the value a magnitude category and its extra bits stand for (T.81 F.2.2.1): the top half of the category's range as read, the
// bottom half shifted down to the negative numbers.  A category past 16 only comes out of a corrupt DHT: the shift counts are
// reduced as x86 reduces them, which is what the reference's expression does there.
inline int extend(int category, int bits) {
    if (category == 0) return bits;
    const int half = (int)(1u << ((category - 1) & 31));
    return bits >= half ? bits : (int)((unsigned)bits + 1u - (1u << (category & 31)));
}

// MSB-first writer producing un-stuffed bytes; optional hard bound on produced bytes.
struct BitWriter {
    std::vector<uint8_t> bytes;
    uint64_t acc = 0;   // pending bits, left-aligned
    int nacc = 0;       // number of pending bits (< 8 after drain)
    uint8_t fillbit = 1;
    // bits the reference's 64-bit buffer would hold at this point (64 - cbit2): it is emptied whenever it fills, at every pad
    // and -- down to the last partial byte -- at the end of every MCU row (partial_bytewise_flush).  The baseline re-coder
    // hands bytes to its bounded output only at those row ends, at pads, and behind a block that leaves the buffer exactly
    // empty (no_remainder(), recoder.cc:366-369), so WHEN a byte bound is noticed depends on it (RowCoder::mcu_row).
    int phase = 0;
    bool buffer_empty() const { return phase == 0; }
    void row_flush() { phase &= 7; }

    void put(unsigned val, int nbits) {
        if (excess) { put_after_oversized_seed(val, nbits); return; }
        phase += nbits;
        if (phase >= 64) phase -= 64;
        while (nbits > 0) {
            int take = nbits > 32 ? 32 : nbits;
            uint64_t v = (val >> (nbits - take)) & (take == 32 ? 0xffffffffull : ((1ull << take) - 1));
            acc |= v << (64 - nacc - take);
            nacc += take;
            nbits -= take;
            while (nacc >= 8) { bytes.push_back((uint8_t)(acc >> 56)); acc <<= 8; nacc -= 8; }
        }
    }
    void pad(uint8_t pattern) {
        int off = 1;
        while (nacc & 7) { put((pattern & off) ? 1 : 0, 1); off <<= 1; }
        phase = 0;
    }
    // state a thread segment starts in (abitwriter::reset_from_overhang_byte_and_num_bits, bitops.hh:203-214: buf = byte << 56,
    // cbit2 = 64 - nbits).  A well-formed hand-off has nbits < 8; a damaged one can say anything up to 255.  Up to 64 the
    // reference's 64-bit buffer simply holds the byte and nbits - 8 zero bits.  Beyond 64 cbit2 is negative and the first write
    // (abitwriter::write, bitops.hh:120-163) flushes the eight buffer bytes, widens its value by the excess, and -- if that makes
    // it wider than the buffer -- drops it and leaves the buffer "full" of zeros, which the next write flushes as eight more
    // bytes.  Bits of the byte below the nbits it claims stay in the buffer and are OR-ed with what is written next, there as here.
    int excess = 0;   // > 0: seeded with 64 + excess bits, first put() pending
    void seed(uint8_t overhang_byte, int nbits) {
        bytes.clear(); acc = (uint64_t)overhang_byte << 56; excess = 0;
        if (nbits > 64) { excess = nbits - 64; nacc = 64; }
        else nacc = nbits;
        phase = nacc;
        while (nacc >= 8) { bytes.push_back((uint8_t)(acc >> 56)); acc <<= 8; nacc -= 8; }
    }
    void put_after_oversized_seed(unsigned val, int nbits) {
        const int wide = nbits + excess;
        excess = 0;
        if (wide > 64) { for (int i = 0; i < 8; ++i) bytes.push_back(0); phase = 0; return; }   // the value is lost, 64 zero bits take its place
        acc = wide ? ((uint64_t)val & (~0ull >> (64 - wide))) << (64 - wide) : 0;
        nacc = wide;
        phase = wide & 63;
        while (nacc >= 8) { bytes.push_back((uint8_t)(acc >> 56)); acc <<= 8; nacc -= 8; }
    }
    uint8_t overhang_byte() const { return (uint8_t)(acc >> 56); }
    int overhang_bits() const { return nacc; }
};

}  // namespace lep
