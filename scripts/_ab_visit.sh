set -u
export TMPDIR=/tmp
python - <<'PY'
# a corpus of restart-interval files through the batch compressor: lane-per-interval kernels against the single-wave kernel
import io, os, sys, time
sys.path.insert(0, os.getcwd())
from PIL import Image
from lepton_amd import corpus
from lepton_amd.codec import GpuCodec
base = [corpus.synth_jpeg(3840, 2160, 500 + i) for i in range(8)]
def with_rst(j, **kw):
    im = Image.open(io.BytesIO(j)); buf = io.BytesIO(); im.save(buf, format="JPEG", quality=90, subsampling="4:2:0", **kw); return buf.getvalue()
for label, kw in (("one restart interval per MCU row", dict(restart_marker_rows=1)), ("a restart interval of 8 MCUs", dict(restart_marker_blocks=8)), ("no restart intervals", dict())):
    rst = [with_rst(j, **kw) for j in base]
    jpgs = [rst[i % 8] for i in range(896)]
    mb = sum(map(len, jpgs)) / 1e6
    for env in ("1", "0"):
        os.environ["LEP_HUFFDEC_SIMT"] = env
        c = GpuCodec(0)
        c.compress_batch(jpgs)
        t0 = time.time(); out, st, stats = c.compress_batch(jpgs); dt = time.time() - t0
        assert not any(st)
        t0 = time.time(); back, st2, ds = c.decompress_batch(out); dt2 = time.time() - t0
        assert back == jpgs
        print("LEP_HUFFDEC_SIMT=%s: 896 x 4K, %s: compress %.0f MB/s (wall %.3f s, parse_s %.3f, gpu_huffman_files %d), decompress %.0f MB/s" % (env, label, mb / dt, dt, stats["parse_s"], stats["gpu_huffman_files"], mb / dt2), flush=True)
        c.close()
PY
