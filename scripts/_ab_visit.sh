set -u
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_reference_corpus.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "cut_inside or batch_pipeline_on_the_reference or decompress_pipeline or truncat or hostile" 2>&1 | tail -4
bash scripts/gpu_ab.sh r5g -b latency-decode -- "" "LEP_LIB_PATH=$PWD/lepton_amd/liblepton_lat.so LEP_DEC_LATENCY_MASK=6" "LEP_LIB_PATH=$PWD/lepton_amd/liblepton_lat.so LEP_DEC_LATENCY_MASK=14" "LEP_LIB_PATH=$PWD/lepton_amd/liblepton_lat.so LEP_DEC_LATENCY_MASK=6 LEP_DEC_LATENCY_MAX=4096" "LEP_DEC_WAVES=8"
python bench.py > gpurun_out/r5g_bench.json 2> gpurun_out/r5g_bench.err; tail -c 300 gpurun_out/r5g_bench.err
