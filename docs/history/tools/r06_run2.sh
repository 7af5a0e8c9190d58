set -x
mkdir -p gpurun_out/r06b
( time timeout 2400 python3 -m pytest tests -m gpu -x -q ) > gpurun_out/r06b/gpu_tests.log 2>&1; tail -15 gpurun_out/r06b/gpu_tests.log
STEPS=3 timeout 900 bash tools/ab_bench.sh base2 pair > gpurun_out/r06b/ab2.txt 2>&1; cat gpurun_out/r06b/ab2.txt
( time python3 bench.py > gpurun_out/r06b/bench_default.json 2> gpurun_out/r06b/bench_default.err ) 2>&1 | tail -4
python3 -c "
import json; d=json.load(open('gpurun_out/r06b/bench_default.json')); print(round(d['value']), d['ms_per_step'], d['roofline']['frac'], d['config'])"
