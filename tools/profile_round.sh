#!/bin/bash
# One GPU-box call that produces the evidence bench.py's roofline cites, for tag $1 (e.g. r02_a):
#   gpurun_out/<tag>_bench.json                  the default bench line (two chunks in flight, extras, cpu baseline)
#   gpurun_out/<tag>_bench_depth1.json           the same bench with one chunk in flight, run under rocprofv3 --kernel-trace --stats
#   gpurun_out/<tag>_kernel_stats_depth1.csv     rocprofv3's per-kernel summary of that run: with one chunk in flight its average
#                                                k_spend_bits duration is an un-overlapped launch and must agree with roofline.avg_launch_ms
#   gpurun_out/<tag>_pmc_valu.json, _pmc_hbm_traffic.json   separate --pmc passes (tools/pmc_profile.sh)
tag=${1:-r04_a}
root=${GRAFT_REPO_ROOT:-$PWD}
cd $root
python3 bench.py --steps 3 --warmup 1 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/${tag}_prof -- python3 $root/bench.py --steps 2 --warmup 1 --pipeline-depth 1 --no-extras --no-cpu-baseline --no-node-multi > $root/gpurun_out/${tag}_bench_depth1.json 2> $root/gpurun_out/${tag}_prof.err
cd $root
f=$(find gpurun_out/${tag}_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f gpurun_out/${tag}_kernel_stats_depth1.csv
# the pipeline as named ranges (tracing): the -DACT_ROCTX build (tools/build_variant.sh roctx -DACT_ROCTX, built in
# the authoring container) under --marker-trace, one chunk sequence of a 2^18 batch, summarised as a timeline
if [ -f anonymous-credit-tokens_amd/libact_roctx.so ]; then
  export ACT_LIB_PATH=$root/anonymous-credit-tokens_amd/libact_roctx.so
  cd /tmp
  rocprofv3 --kernel-trace --marker-trace --output-format csv -d $root/gpurun_out/${tag}_roctx -- python3 $root/bench.py --steps 1 --warmup 1 --batch-log2 18 --no-extras --no-cpu-baseline --no-node-multi > $root/gpurun_out/${tag}_roctx_bench.json 2> $root/gpurun_out/${tag}_roctx.err
  cd $root
  unset ACT_LIB_PATH
  python3 tools/roctx_summarize.py gpurun_out/${tag}_roctx 160 > gpurun_out/${tag}_roctx_timeline.txt 2>&1
  rm -rf gpurun_out/${tag}_roctx
fi
bash tools/pmc_profile.sh $tag > gpurun_out/${tag}_pmc.log 2>&1
# the suite (incl. the full-size BASELINE configurations, whose rates land in <tag>_configs_from_tests.json) and the other configs
ACT_WRITE_RATES=${tag}_configs_from_tests.json python3 -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" > gpurun_out/${tag}_gpu_tests.log
python3 tools/bench_configs.py > gpurun_out/${tag}_other_configs_1gpu.json 2> /dev/null
rm -rf gpurun_out/${tag}_prof gpurun_out/${tag}_pmc
head -4 gpurun_out/${tag}_kernel_stats_depth1.csv
python3 - <<P
import json
for n in ("bench", "bench_depth1"):
    d = json.load(open("gpurun_out/${tag}_%s.json" % n))
    print(n, round(d["value"]), "avg_launch_ms", round(d["roofline"]["avg_launch_ms"], 2), "frac", round(d["roofline"]["frac"], 4))
print(open("gpurun_out/${tag}_pmc.log").read()[-600:])
P
