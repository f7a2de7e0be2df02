set -u
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_reference_corpus.py -m gpu -q -x -p no:cacheprovider -k "batch or pipeline or decompress" 2>&1 | tail -3
for v in x 0; do
  if [ $v = x ]; then unset LEP_BATCH_DEC_OVERLAP; else export LEP_BATCH_DEC_OVERLAP=$v; fi
  echo "LEP_BATCH_DEC_OVERLAP=$v"
  LEP_BATCH_TRACE=1 python scripts/trace_decode_overlap.py 2>&1 | awk '/MARK/{f=1} f{print}' | grep -v "^MARK\|\[batch\]   " | head -12
done
