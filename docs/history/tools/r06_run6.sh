set -x
( time timeout 2400 python3 -m pytest tests -m gpu -q ) > gpurun_out/r06_y_gpu_tests.log 2>&1; tail -4 gpurun_out/r06_y_gpu_tests.log
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_y_bench_20steps.json 2> gpurun_out/r06_y_bench_20steps.err ) 2>&1 | tail -4
python3 -c "
import json; d=json.load(open('gpurun_out/r06_y_bench_20steps.json')); print(round(d['value']), d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'])"
( time python3 bench.py --gpus 2 --force-device 0 --dist-backend gloo --steps 2 --warmup 1 --no-extras --no-cpu-baseline > gpurun_out/r06_two_ranks_one_gpu.json 2> gpurun_out/r06_two_ranks_one_gpu.err ) 2>&1 | tail -4
python3 -c "
import json; d=json.load(open('gpurun_out/r06_two_ranks_one_gpu.json')); print(d['n_gpus'], round(d['value']), d['scaling'], [r.get('verifies_per_s') for r in d.get('per_rank',[])], d['config'].get('fixed_base_window_bits_g_h1_h2_h3'))"
tail -5 gpurun_out/r06_two_ranks_one_gpu.err
