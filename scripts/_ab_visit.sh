set -u
export TMPDIR=/tmp
V=""
for v in c3 sc0 sc10 sc6; do V="$V LEP_LIB_PATH=$PWD/lepton_amd/liblepton_$v.so"; done
bash scripts/gpu_ab.sh r5w -b resident -- "" $V ""
