set -u
export TMPDIR=/tmp
V=""
for v in r1 r2 r3 r4 r5 r6 r7 r8; do V="$V LEP_LIB_PATH=$PWD/lepton_amd/liblepton_$v.so"; done
bash scripts/gpu_ab.sh r5e -b resident -- "" $V ""
