set -u
export TMPDIR=/tmp
V=""
for v in ww3 ww4; do V="$V LEP_LIB_PATH=$PWD/lepton_amd/liblepton_$v.so"; done
bash scripts/gpu_ab.sh r5x -k "4k_roundtrip" -b resident -- "" $V
LEP_LIB_PATH=$PWD/lepton_amd/liblepton_ww3.so timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "4k_roundtrip or streams_equal" 2>&1 | tail -2
