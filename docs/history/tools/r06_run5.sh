set -x
root=$PWD
export ACT_LIB_PATH=$root/anonymous-credit-tokens_amd/libact_roctx.so
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --marker-trace --output-format csv -d $root/gpurun_out/r06_x_roctx -- python3 $root/bench.py --steps 1 --warmup 1 --batch-log2 18 --no-extras --no-cpu-baseline --no-node-multi > $root/gpurun_out/r06_x_roctx_bench.json 2> $root/gpurun_out/r06_x_roctx.err
cd $root
unset ACT_LIB_PATH
python3 tools/roctx_summarize.py gpurun_out/r06_x_roctx 140 > gpurun_out/r06_x_roctx_timeline.txt 2>&1
rm -rf gpurun_out/r06_x_roctx
head -30 gpurun_out/r06_x_roctx_timeline.txt
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_x_bench_20steps.json 2> gpurun_out/r06_x_bench_20steps.err ) 2>&1 | tail -4
python3 -c "
import json; d=json.load(open('gpurun_out/r06_x_bench_20steps.json')); print(round(d['value']), d['ms_per_step'], d['roofline']['frac'], d['roofline']['frac_vs_nominal_2p4GHz'], d['roofline']['traffic'])"
( time python3 -c "import __graft_entry__ as g; g.smoke()" ) 2>&1 | tail -5
