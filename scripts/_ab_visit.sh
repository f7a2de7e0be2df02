set -u
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_reference_corpus.py -m gpu -q -x -p no:cacheprovider -k "batch or streams_of_the_reference or register_budget or mixed" 2>&1 | tail -3
V=""
for v in ew0 ew1 eww; do V="$V LEP_LIB_PATH=$PWD/lepton_amd/liblepton_$v.so"; done
bash scripts/gpu_ab.sh r5h -b resident -- "" $V
LEP_DEC_CLASSES=0 python bench.py --no-cpu-baseline --no-end-to-end > gpurun_out/r5h_bench_classes0.json 2> gpurun_out/r5h_bench_classes0.err
python bench.py --no-cpu-baseline --no-end-to-end > gpurun_out/r5h_bench_classes1.json 2> gpurun_out/r5h_bench_classes1.err
python - <<'PY'
import json
for k in ("0","1"):
    d=json.loads(open('gpurun_out/r5h_bench_classes%s.json'%k).read().strip().splitlines()[-1])
    print('classes',k,'mixed',{a:d['mixed'][a] for a in ('compress_MBps','decompress_MBps','value')},'skewed resident',d['value_skewed'].get('value'),d['value_skewed'].get('decode_kernel_ms'),'extra.skewed',{a:d['extra']['skewed'].get(a) for a in ('compress_MBps','decompress_MBps','value')}, 'c1080p', d['extra']['c1080p'].get('value'))
PY
