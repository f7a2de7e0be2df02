set -u
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_reference_corpus.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "restart or batch or rst or huffman or cut_inside" 2>&1 | tail -4
python - <<'PY'
# a corpus of restart-interval files through the batch compressor: lane-per-interval kernels against the single-wave kernel
import io, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from PIL import Image
from lepton_amd import corpus
from lepton_amd.codec import GpuCodec
base = [corpus.synth_jpeg(3840, 2160, 500 + i) for i in range(8)]
def with_rst(j, rows):
    im = Image.open(io.BytesIO(j)); buf = io.BytesIO(); im.save(buf, format="JPEG", quality=90, subsampling="4:2:0", restart_marker_rows=rows); return buf.getvalue()
rst = [with_rst(j, 1) for j in base]
jpgs = [rst[i % 8] for i in range(512)]
mb = sum(map(len, jpgs)) / 1e6
for env in ("1", "0"):
    os.environ["LEP_HUFFDEC_SIMT"] = env
    c = GpuCodec(0)
    c.compress_batch(jpgs[:64])
    t0 = time.time(); out, st, stats = c.compress_batch(jpgs); dt = time.time() - t0
    assert not any(st)
    print("LEP_HUFFDEC_SIMT=%s: 512 x 4K with one restart interval per MCU row: compress %.0f MB/s (wall %.3f s, parse_s %.3f, gpu_huffman_files %d)" % (env, mb / dt, dt, stats["parse_s"], stats["gpu_huffman_files"]))
    c.close()
PY
