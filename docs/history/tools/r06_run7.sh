set -x
mkdir -p gpurun_out/r06d
export ACT_LIB_PATH=$PWD/anonymous-credit-tokens_amd/libact_fused.so
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_sodium.py tests/test_gpu_tiny.py -x -q -m gpu > gpurun_out/r06d/parity_fused.log 2>&1; tail -3 gpurun_out/r06d/parity_fused.log
unset ACT_LIB_PATH
STEPS=3 timeout 900 bash tools/ab_bench.sh pair2 fused > gpurun_out/r06d/ab4.txt 2>&1; cat gpurun_out/r06d/ab4.txt
