# The reference-side change of INTEGRATION.md as a patch (applied by Makefile.ref to a copy of src/lepton/jpgcoder.cc under
# oracle/_ref/): the coder factories return the MI355X adapter (integration/mi355x_coders.cc) instead of VP8ComponentEncoder /
# VP8ComponentDecoder.  Three call sites -- jpgcoder.cc:1710 (encoder), :1727 (decoder), :1213-1217 (the -preload pair).
s|^BaseDecoder\* g_decoder = NULL;|BaseDecoder* g_decoder = NULL; BaseEncoder *make_mi355x_encoder(bool, bool); BaseDecoder *make_mi355x_decoder(bool, bool);|
s|g_encoder.reset(makeEncoder<VPXBoolReader>(g_threaded, g_threaded));|g_encoder.reset(make_mi355x_encoder(g_threaded, g_threaded));|
s|g_decoder = makeDecoder(g_threaded, g_threaded, ujgversion == 3);|g_decoder = make_mi355x_decoder(g_threaded, g_threaded);|
s|if (g_do_preload \&\& g_skip_validation) {|if (false \&\& g_do_preload \&\& g_skip_validation) {|
