#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r3repro; mkdir -p $OUT
A="--mixed-images 0 --no-end-to-end --no-cpu-baseline --steps 1"
timeout 600 python bench.py $A > $OUT/b.json 2> $OUT/b.err; rc=$?; echo "bench rc=$rc"; grep "rank 0" $OUT/b.err | cut -c1-300
if [ $rc -ne 0 ]; then
  timeout 900 /opt/rocm/bin/rocgdb -batch -ex "handle SIGUSR1 nostop noprint" -ex run -ex "bt 40" --args python bench.py $A > $OUT/b.gdb 2>&1
  grep -A45 "received signal" $OUT/b.gdb | cut -c1-240 | head -70
fi
