set -u
export TMPDIR=/tmp
t0=$(date +%s)
V=""
for v in lat6 lat14 lat10; do V="$V LEP_LIB_PATH=$PWD/lepton_amd/liblepton_$v.so"; done
bash scripts/gpu_ab.sh r11e -b latency-decode -- "" $V
cat gpurun_out/r11e/latency-decode_*.json
bash scripts/gpu_ab.sh r11f -b resident -- "" "LEP_LIB_PATH=$PWD/lepton_amd/liblepton_pairor.so"
echo "total $(( $(date +%s)-t0 )) s"
