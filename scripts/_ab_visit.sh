set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r12d
t0=$(date +%s)
timeout 400 python scripts/diag_scan_encode.py ALL > gpurun_out/r12d/diag_su.txt 2>&1; echo "diag su rc=$?"; grep -c " ok " gpurun_out/r12d/diag_su.txt; grep -A4 "DIFFERS" gpurun_out/r12d/diag_su.txt | head -80; tail -2 gpurun_out/r12d/diag_su.txt
LEP_LIB_PATH=$PWD/lepton_amd/liblepton_nosu.so timeout 300 python scripts/diag_scan_encode.py ALL > gpurun_out/r12d/diag_nosu.txt 2>&1; echo "diag nosu rc=$?"; grep -A4 "DIFFERS" gpurun_out/r12d/diag_nosu.txt | head -40; tail -2 gpurun_out/r12d/diag_nosu.txt
LEP_HUFFENC_SIMT=0 timeout 300 python scripts/diag_scan_encode.py ALL > gpurun_out/r12d/diag_su_wave.txt 2>&1; echo "diag su wave-kernel rc=$?"; grep -A4 "DIFFERS" gpurun_out/r12d/diag_su_wave.txt | head -40; tail -2 gpurun_out/r12d/diag_su_wave.txt
echo "total $(( $(date +%s)-t0 )) s"
