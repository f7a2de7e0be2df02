#!/bin/bash
# Per-level times of the progressive scan decoder on 256 x 4K progressive files (level by level: three launches, each as long as its
# longest scan), for the product and for experiment builds: scripts/prog_levels.sh <tag> [<variant lib name> ...]
set -u
export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
for V in "" "$@"; do
  n=${V:-product}
  ( [ -n "$V" ] && export LEP_LIB_PATH=$PWD/lepton_amd/liblepton_$V.so
    LEP_HUFFPROG_PIPELINE=0 rocprofv3 --kernel-trace -d $OUT/prof_$n -- python scripts/bench_batch.py --images 256 --unique 8 --width 3840 --height 2160 --progressive > $OUT/bench_$n.json 2> $OUT/err_$n.txt )
  echo "== $n"; python scripts/trace_kernels.py $OUT/prof_$n/*/*.db --timeline progdec | grep "progdec"
done
