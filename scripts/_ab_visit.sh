set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r11d
t0=$(date +%s)
for k in 8mcu none; do
  timeout 300 python scripts/bench_restart_corpora.py --only $k --simt 1,0 --verbose --repeats 3 2>&1 | grep -v "^W2026" | tee -a gpurun_out/r11d/restart_verbose.txt
done
echo "total $(( $(date +%s)-t0 )) s"
