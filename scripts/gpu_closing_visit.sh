#!/bin/bash
# round 3 closing visit: the whole GPU suite, the default bench line (with extras) under the kernel trace, then the PMC passes
# over the shipped kernels (separate --pmc runs, kernel trace only) -> profiles-ready files under gpurun_out/<tag>/
set -u
TAG=${1:-r10z}; export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
t0=$(date +%s)
if [ "${SKIP_TESTS:-0}" != 1 ]; then
  timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$? ($(( $(date +%s)-t0 )) s)"; tail -3 $OUT/pytest_gpu.log
fi
pmc_passes() {
B="python bench.py --steps 1 --warmup 0 --no-extras --no-end-to-end --no-cpu-baseline --mixed-images 0"
pass() {  # name, images, counters...
  local name=$1 images=$2; shift 2
  timeout 200 rocprofv3 --pmc "$@" --kernel-trace -d $OUT/pmc_$name -o pmc --output-format csv -- $B --images $images > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err; echo "$name rc=$? $(( $(date +%s)-t0 )) s"
}
pass sq 1024 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM
pass in 1024 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES
pass mem 1024 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum
pass mem2 1024 TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_READ_sum
pass mem256 256 TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum   # (four TCC counters at most: a fifth is refused and rocprofv3 then sits until the timeout)
python scripts/make_pmc_traffic.py $OUT
rm -rf $OUT/pmc_sq $OUT/pmc_in $OUT/pmc_mem $OUT/pmc_mem2 $OUT/pmc_mem256
}
if [ "${PMC_FIRST:-0}" = 1 ]; then   # the bench line then quotes the table of its own build
  pmc_passes; cp $OUT/pmc_traffic.json profiles/pmc_traffic.json
fi
if [ "${SKIP_BENCH:-0}" != 1 ]; then
  timeout 1500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$? ($(( $(date +%s)-t0 )) s)"
  ARGS="--steps 3 --warmup 1 --no-extras --no-end-to-end --no-cpu-baseline --mixed-images 0"
  timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof -o b --output-format csv -- python bench.py $ARGS > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err; echo "rocprof rc=$? ($(( $(date +%s)-t0 )) s)"
  for f in $(find $OUT/prof -name "*kernel_stats.csv"); do cp $f $OUT/rocprofv3_kernel_stats.csv; done
  rm -rf $OUT/prof
fi
if [ "${SERVE:-0}" = 1 ]; then   # the daemon: what one caller waits, then a small concurrent load (every answer checked)
  timeout 600 python scripts/bench_serve.py --requests 256 --clients 64 --procs 4 --unique 8 > $OUT/serve_4k.json 2> $OUT/serve_4k.err; echo "serve rc=$? ($(( $(date +%s)-t0 )) s)"; head -c 1200 $OUT/serve_4k.json; echo
fi
if [ "${RESTART:-0}" = 1 ]; then   # files with restart intervals through the batch pipeline, scan kernels A/B
  timeout 600 python scripts/bench_restart_corpora.py > $OUT/restart_interval_corpora_ab.txt 2> $OUT/restart_interval_corpora.err; echo "restart corpora rc=$? ($(( $(date +%s)-t0 )) s)"; cat $OUT/restart_interval_corpora_ab.txt
fi
if [ "${SKIP_PMC:-0}" = 1 ] || [ "${PMC_FIRST:-0}" = 1 ]; then echo "total $(( $(date +%s)-t0 )) s"; exit 0; fi
pmc_passes
echo "total $(( $(date +%s)-t0 )) s"
