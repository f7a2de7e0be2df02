// mi355x_coders.cc -- the reference-side adapter of INTEGRATION.md as a real translation unit: BaseEncoder / BaseDecoder
// (src/lepton/base_coders.hh:26-65 of dropbox/lepton) implemented on top of the C ABI of liblepton_mi355x.so.
// This file belongs in the REFERENCE's tree (src/lepton/), next to vp8_encoder.cc / vp8_decoder.cc; it is kept here so that
// tests/test_integration_adapter.py can type-check it against the reference's own headers wherever a reference checkout is
// present (g++ -fsyntax-only, the reference's include paths and default defines).  The three factories of jpgcoder.cc:440-471
// then become:
//     g_encoder.reset(make_mi355x_encoder(g_threaded, g_threaded));      (jpgcoder.cc:1710)
//     g_decoder = make_mi355x_decoder(g_threaded, g_threaded);           (jpgcoder.cc:1727)
// (oracle/mi355x_factories.sed is exactly that patch; oracle/Makefile.ref applies it to a copy of jpgcoder.cc and links
// oracle/_ref/lepton-mi355x, which tests/test_integration_adapter.py runs) and nothing else in jpgcoder.cc / recoder.cc changes.  The process must be able to reach /dev/kfd, i.e. run -unjailed or
// create the lep_gpu before installStrictSyscallFilter (jpgcoder.cc:1765) with a filter that admits the HIP runtime's calls.
#include <pthread.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "lepton_mi355x.h"

#include "../io/MuxReader.hh"
#include "../io/ioutil.hh"
#include "base_coders.hh"
#include "thread_handoff.hh"
#include "uncompressed_components.hh"

extern unsigned char ujgversion;   // jpgcoder.cc:544

namespace {

// what encode_chunk consumes of UncompressedComponents (vp8_encoder.cc:521-548): geometry, truncation bounds, zig-zag
// quantisation tables, the dense AlignedBlock arrays
// with_frame = false: geometry only.  The baseline decode path never allocates the full components of `in` (the reference
// decodes through two-row framebuffers there, uncompressed_components.hh:155-167), so neither their dimensions nor their
// blocks may be asked of the BlockBasedImage objects -- raster(0) of an unallocated image is an OOM exit
// (block_based_image.hh:216-224); the header's own block counts are always there.
void fill_desc(const UncompressedComponents *in, lep_image_desc *d, bool with_frame) {
    memset(d, 0, sizeof *d);
    d->ncomp = in->get_num_components();
    d->mcu_rows = in->get_mcu_count_vertical();
    Sirikata::Array1d<uint32_t, (uint32_t)ColorChannel::NumBlockTypes> coded_h = in->get_max_coded_heights();
    for (int c = 0; c < d->ncomp && c < LEP_MAX_COMPONENTS; ++c) {
        d->width_blocks[c] = (int32_t)in->block_width(c);
        // the frame's theoretical height (bcv); block_height() is the TRUNCATED height.  Without a full component the caller
        // takes it from the row framebuffers it was handed (initialize_baseline_decoder)
        d->height_blocks[c] = with_frame ? (int32_t)in->full_component_nosync(c).original_height() : 0;
        d->coded_blocks[c] = (int32_t)in->component_size_in_blocks(c);
        d->coded_height[c] = (int32_t)coded_h[c];
        memcpy(d->qtable_zigzag[c], in->get_quantization_tables((BlockType)c), 64 * sizeof(uint16_t));
        d->blocks[c] = with_frame ? const_cast<int16_t *>(in->block_nosync((BlockType)c, 0).raw_data()) : NULL;
    }
}

void die_on(int rc) {   // same contract as the reference: data errors end the process with the ExitCode (memory.hh:13-40)
    if (rc) custom_exit((ExitCode)rc);
}

}  // namespace

// ---- replaces VP8ComponentEncoder (vp8_encoder.cc:75-82, 460-632) ------------------------------------------------------------
class MI355XEncoder : public BaseEncoder {
    lep_gpu *gpu_;

public:
    MI355XEncoder() : gpu_(NULL) { die_on(lep_gpu_create(0, &gpu_)); }
    ~MI355XEncoder() { lep_gpu_destroy(gpu_); }
    void registerWorkers(GenericWorker *, unsigned int) {}   // the segments are wavefronts, not CPU threads
    size_t get_decode_model_memory_usage() const { return 0; }
    size_t get_decode_model_worker_memory_usage() const { return 0; }

    CodingReturnValue encode_chunk(const UncompressedComponents *in, IOUtil::FileWriter *out, const ThreadHandoff *splits,
                                   unsigned int n) {
        lep_image_desc d;
        fill_desc(in, &d, true);
        lep_segment seg[LEP_MAX_SEGMENTS];
        lep_bytes st[LEP_MAX_SEGMENTS];
        int32_t status[LEP_MAX_SEGMENTS];
        std::vector<std::vector<uint8_t> > buf(n);
        size_t blocks = 0;
        for (int c = 0; c < d.ncomp; ++c) blocks += (size_t)d.width_blocks[c] * d.height_blocks[c];
        always_assert(n <= LEP_MAX_SEGMENTS);
        for (unsigned int i = 0; i < n; ++i) {
            seg[i].image = 0;
            seg[i].luma_y_start = splits[i].luma_y_start;
            seg[i].luma_y_end = splits[i].luma_y_end;
            seg[i].is_last = i + 1 == n;
            buf[i].resize(blocks * 40 / n + 65536);   // the same head room the batch pipeline reserves
            st[i].data = buf[i].data(); st[i].len = 0; st[i].cap = buf[i].size();
        }
        die_on(lep_gpu_encode_host(gpu_, &d, 1, seg, (int)n, st, status));
        for (unsigned int i = 0; i < n; ++i) die_on(status[i]);
        // from here on: the reference's own tail (vp8_encoder.cc:575-611) -- round-robin mux slices, then the size trailer
        Sirikata::MuxWriter mux(out, Sirikata::JpegAllocator<uint8_t>(), ujgversion);
        size_t off[Sirikata::MuxReader::MAX_STREAM_ID] = {0};
        for (bool any = true; any;) {
            any = false;
            for (unsigned int i = 0; i < n; ++i) {
                if (st[i].len <= off[i]) continue;
                any = true;
                const size_t slice = off[i] == 0 ? 256 : (off[i] == 256 ? 4096 : 65536);
                off[i] += mux.Write((uint8_t)i, st[i].data + off[i], (unsigned int)std::min(slice, st[i].len - off[i])).first;
            }
        }
        mux.Close();
        uint32_t total = (uint32_t)out->getsize() + 4;
        uint8_t le[4] = {uint8_t(total), uint8_t(total >> 8), uint8_t(total >> 16), uint8_t(total >> 24)};
        out->Write(le, 4);
        return CODING_DONE;
    }
};

// ---- replaces VP8ComponentDecoder (vp8_decoder.cc:18-24, 173-179, 316-490; lepton_codec.cc:7-47) ------------------------------
// The reference's baseline decoder keeps two block rows per component per thread and decodes them on demand
// (block_based_image.hh:60-66, recoder.cc:516); here every segment of the image is decoded by one kernel launch into a full
// frame owned by the adapter, and decode_row hands rows out of it.
class MI355XDecoder : public BaseDecoder {
    lep_gpu *gpu_;
    Sirikata::DecoderReader *in_;
    Sirikata::MuxReader mux_;
    std::vector<ThreadHandoff> handoff_;
    GenericWorker *workers_;
    unsigned int num_workers_;
    // baseline path: the adapter's own frame, decoded once by whichever of the re-coder's threads asks first (recoder.cc
    // runs recode_physical_thread on up to NUM_THREADS workers, each calling decode_row for its own segments)
    pthread_mutex_t once_;
    bool decoded_;
    lep_image_desc desc_;
    std::vector<int16_t> frame_[LEP_MAX_COMPONENTS];

    void collect_streams(std::vector<std::vector<uint8_t> > *streams) {   // stream id == segment index (vp8_encoder.cc:575-594)
        streams->assign(Sirikata::MuxReader::MAX_STREAM_ID, std::vector<uint8_t>());
        Sirikata::MuxReader::ResizableByteBuffer pkt;
        for (;;) {
            pkt.resize(0);
            std::pair<uint8_t, Sirikata::JpegError> r = mux_.nextDataPacket(pkt);
            if (r.second != Sirikata::JpegError::nil()) break;
            std::vector<uint8_t> &s = (*streams)[r.first];
            s.insert(s.end(), pkt.data(), pkt.data() + pkt.size());
        }
    }
    void decode_into(lep_image_desc *d) {
        std::vector<std::vector<uint8_t> > streams;
        collect_streams(&streams);
        const size_t n = handoff_.size();
        always_assert(n >= 1 && n <= LEP_MAX_SEGMENTS);
        lep_segment seg[LEP_MAX_SEGMENTS];
        lep_bytes st[LEP_MAX_SEGMENTS];
        int32_t status[LEP_MAX_SEGMENTS];
        for (size_t i = 0; i < n; ++i) {
            seg[i].image = 0;
            seg[i].luma_y_start = handoff_[i].luma_y_start;
            seg[i].luma_y_end = handoff_[i].luma_y_end;
            seg[i].is_last = i + 1 == n;
            st[i].data = streams[i].empty() ? NULL : &streams[i][0];
            st[i].len = st[i].cap = streams[i].size();
        }
        die_on(lep_gpu_decode_host(gpu_, d, 1, seg, (int)n, st, status));
        for (size_t i = 0; i < n; ++i) die_on(status[i]);
    }

public:
    MI355XDecoder() : gpu_(NULL), in_(NULL), mux_(Sirikata::JpegAllocator<uint8_t>()), workers_(NULL), num_workers_(0), decoded_(false) {
        pthread_mutex_init(&once_, NULL);
        memset(&desc_, 0, sizeof desc_);
        die_on(lep_gpu_create(0, &gpu_));
    }
    ~MI355XDecoder() { lep_gpu_destroy(gpu_); pthread_mutex_destroy(&once_); }

    void initialize(Sirikata::DecoderReader *input, const std::vector<ThreadHandoff> &thread_transition_info) {
        in_ = input;
        mux_.init(input);
        handoff_ = thread_transition_info;
        decoded_ = false;
    }
    // frame-pull path (progressive files, uncompressed_components.hh:106-108): the whole frame in one call
    CodingReturnValue decode_chunk(UncompressedComponents *dst) {
        lep_image_desc d;
        fill_desc(dst, &d, true);
        if (!handoff_.empty()) handoff_.back().luma_y_end = (uint16_t)dst->block_height(0);   // vp8_decoder.cc:366-368
        decode_into(&d);
        // what the reference's decode_chunk signals when it returns CODING_DONE (vp8_decoder.cc:483-487): the callers spin in
        // wait_for_worker_on_dpos / _bpos / _bit on these counters, calling decode_chunk again until they move
        for (int c = 0; c < d.ncomp; ++c) dst->worker_mark_cmp_finished((BlockType)c);
        dst->worker_update_coefficient_position_progress(64);   // every coefficient of every block is there
        dst->worker_update_bit_progress(16);
        return CODING_DONE;
    }
    void registerWorkers(GenericWorker *workers, unsigned int num_workers) { workers_ = workers; num_workers_ = num_workers; }
    GenericWorker *getWorker(unsigned int i) { return workers_ + i; }
    unsigned int getNumWorkers() const { return num_workers_; }

    // row-pull path (baseline files, recoder.cc:694-889): geometry now, the frame on the first decode_row
    std::vector<ThreadHandoff> initialize_baseline_decoder(
        const UncompressedComponents *const colldata,
        Sirikata::Array1d<BlockBasedImagePerChannel<true>, MAX_NUM_THREADS> &framebuffer) {
        fill_desc(colldata, &desc_, false);
        for (int c = 0; c < desc_.ncomp; ++c) {
            desc_.height_blocks[c] = (int32_t)framebuffer[0][c]->original_height();   // allocate_channel_framebuffer: init(bch, bcv, ...)
            frame_[c].assign((size_t)desc_.width_blocks[c] * desc_.height_blocks[c] * 64, 0);
            desc_.blocks[c] = frame_[c].empty() ? NULL : &frame_[c][0];
        }
        if (!handoff_.empty()) handoff_.back().luma_y_end = (uint16_t)colldata->block_height(0);
        decoded_ = false;
        return handoff_;
    }
    void decode_row(int, BlockBasedImagePerChannel<true> &image_data,
                    Sirikata::Array1d<uint32_t, (uint32_t)ColorChannel::NumBlockTypes> component_size_in_blocks, int component,
                    int curr_y) {
        pthread_mutex_lock(&once_);
        if (!decoded_) { decode_into(&desc_); decoded_ = true; }
        pthread_mutex_unlock(&once_);
        const uint32_t w = (uint32_t)desc_.width_blocks[component];
        const int16_t *row = &frame_[component][(size_t)curr_y * w * 64];
        for (uint32_t x = 0; x < w && (uint32_t)curr_y * w + x < component_size_in_blocks[component]; ++x)
            memcpy(image_data[component]->at((uint32_t)curr_y, x).raw_data(), row + (size_t)x * 64, 64 * sizeof(int16_t));
    }
    size_t get_model_memory_usage() const { return 0; }          // the models live in HBM
    size_t get_model_worker_memory_usage() const { return 0; }
    void flush() {}
    void map_logical_thread_to_physical_thread(int, int) {}
    void clear_thread_state(int, int, BlockBasedImagePerChannel<true> &) {}
    void reset_all_comm_buffers() {}
};

// The factories, with the signatures and the worker-thread registration of the ones they replace (makeEncoder / makeDecoder /
// makeBoth, jpgcoder.cc:440-471).  The decoder needs the reference's worker threads: the baseline re-coder runs its Huffman
// half on them (recoder.cc:740-790 reaches them through g_decoder->getWorker); the encoder's segments are wavefronts.
GenericWorker *get_worker_threads(unsigned int num_workers);   // jpgcoder.cc:429
BaseEncoder *make_mi355x_encoder(bool /*threaded*/, bool /*start_workers*/) { return new MI355XEncoder; }
BaseDecoder *make_mi355x_decoder(bool /*threaded*/, bool start_workers) {
    MI355XDecoder *d = new MI355XDecoder;
    if (start_workers) d->registerWorkers(get_worker_threads(NUM_THREADS), NUM_THREADS);
    return d;
}
