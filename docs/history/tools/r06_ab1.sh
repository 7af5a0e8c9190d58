set -x
mkdir -p gpurun_out/r06a
export ACT_LIB_PATH=$PWD/anonymous-credit-tokens_amd/libact_oneasm_pf.so
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_sodium.py -x -q -m gpu > gpurun_out/r06a/parity_oneasm_pf.log 2>&1; tail -3 gpurun_out/r06a/parity_oneasm_pf.log
unset ACT_LIB_PATH
STEPS=3 timeout 1500 bash tools/ab_bench.sh base oneasm pf oneasm_pf > gpurun_out/r06a/ab1.txt 2>&1
cat gpurun_out/r06a/ab1.txt
