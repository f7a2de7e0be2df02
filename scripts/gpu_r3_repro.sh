#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r3repro; mkdir -p $OUT
A="--mixed-images 0 --no-cpu-baseline --steps 1"
timeout 900 /opt/rocm/bin/rocgdb -batch -ex "handle SIGUSR1 nostop noprint" -ex run -ex "bt 40" -ex "info sharedlibrary lepton" --args python bench.py $A > $OUT/b.gdb 2>&1
grep "rank 0" $OUT/b.gdb | cut -c1-200
grep -A50 "received signal" $OUT/b.gdb | cut -c1-260 | head -80
