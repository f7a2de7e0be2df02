set -u
export TMPDIR=/tmp
export LEP_BATCH_DEC_OVERLAP=0
for v in "HSA_ENABLE_SDMA=1" "HSA_ENABLE_SDMA=0" "GPU_MAX_HW_QUEUES=16" "HIP_USE_SDMA=1"; do
  echo "== $v"
  env $v LEP_BATCH_TRACE=1 python scripts/trace_decode_overlap.py 2>&1 | awk '/MARK/{f=1} f{print}' | grep "first=1024 begins\|first=1024 done\|decompress"
done
