set -u
export TMPDIR=/tmp
bash scripts/gpu_ab.sh r5m -b progressive -- "" "LEP_LIB_PATH=$PWD/lepton_amd/liblepton_pprio.so" "" "LEP_LIB_PATH=$PWD/lepton_amd/liblepton_pprio.so"
