#!/bin/bash
# Same-box A/B of ONE measurement knob (act_tuning_set through its ACT_* variable): bench.py's timed region, alternating, ROUNDS times,
# then the shares of a strong-scaling split.  usage: tools/ab_knob.sh ACT_HARD_STAGGER [rounds] [steps] [value_a value_b]   -> gpurun_out/ab_<knob>.txt
knob=$1; rounds=${2:-2}; steps=${3:-8}; va=${4:-0}; vb=${5:-1}
out=gpurun_out/ab_${knob}.txt; mkdir -p gpurun_out; : > $out
for r in $(seq 1 $rounds); do
  for v in $va $vb; do
    line=$(env $knob=$v timeout 300 python3 bench.py --steps $steps --warmup 2 --no-extras --no-cpu-baseline --no-node-multi 2>/dev/null | tail -1)
    echo "$knob=$v round $r: $(python3 -c "import json,sys; d=json.loads(sys.argv[1]); print(round(d['value']), 'verifies/s, avg launch', round(d['roofline']['avg_launch_ms'],2), 'ms, frac', round(d['roofline']['frac'],4))" "$line")" | tee -a $out
  done
done
for v in $va $vb; do
  echo "$knob=$v: shares of one 2^20 batch" | tee -a $out
  env $knob=$v timeout 600 python3 tools/strong_share_probe.py --one 2>/dev/null | tee -a $out
done
