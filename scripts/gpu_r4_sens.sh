#!/bin/bash
# what the decode kernel is short of: launch time against n extra independent instructions per block of one kind
# (experiment builds: scripts/build_variant.sh padv128 -DLEP_DEC4_PAD_VALU=128, pads128 -DLEP_DEC4_PAD_SALU=128, padl16 -DLEP_DEC4_PAD_LDS=16, ...)
set -u
TAG=${1:-r5w}; shift; export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
t0=$(date +%s)
B="python bench.py --steps 3 --warmup 1 --unique 16 --no-extras --no-end-to-end --no-cpu-baseline --mixed-images 0"
for lib in "$@"; do
  LEP_LIB_PATH=$PWD/lepton_amd/liblepton_$lib.so timeout 300 $B > $OUT/b_$lib.json 2>> $OUT/err.txt; python -c "
import json;d=json.load(open('$OUT/b_$lib.json'));r=d['roofline'];print('$lib', 'decode ms', r['decode_kernel_ms'], 'encode ms', r['encode_kernel_ms'])" | tee -a $OUT/sensitivity.txt
done
echo "total $(( $(date +%s)-t0 )) s"
