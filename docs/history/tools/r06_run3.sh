set -x
mkdir -p gpurun_out/r06c
( time timeout 2400 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_wire.py tests/test_gpu_hygiene.py tests/test_gpu_table_sharing.py tests/test_cbor.py tests/test_gpu_cbor_verify.py tests/test_gpu_tiny.py tests/test_gpu_small_batch.py tests/test_gpu_node.py tests/test_gpu_redeem.py -m gpu -q ) > gpurun_out/r06c/gpu_tests_subset.log 2>&1; tail -25 gpurun_out/r06c/gpu_tests_subset.log
STEPS=3 timeout 900 bash tools/ab_bench.sh base3 pair2 > gpurun_out/r06c/ab3.txt 2>&1; cat gpurun_out/r06c/ab3.txt
timeout 900 python3 tools/strong_share_probe.py > gpurun_out/r06c/strong_share.txt 2>&1; head -8 gpurun_out/r06c/strong_share.txt
timeout 600 python3 tools/soak_wire.py 150 20 > gpurun_out/r06c/soak_wire.log 2>&1; tail -3 gpurun_out/r06c/soak_wire.log
