#!/bin/bash
# Collects the PMC evidence bench.py's roofline object cites, for k_spend_bits on one chunk of NB (default 65536) L=128 proofs
# (tools/pmc_run.py).  Separate rocprofv3 --pmc passes (with --kernel-trace only), as MI355X_MICROARCH.md prescribes.
# usage (on the GPU box): tools/pmc_profile.sh <tag>     -> gpurun_out/<tag>_pmc_valu.json, gpurun_out/<tag>_pmc_hbm_traffic.json
tag=${1:-r01_e}
export NB=${NB:-65536}
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum" "SQC_ICACHE_REQ SQC_ICACHE_MISSES" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  REPS=1 timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $root/gpurun_out/${tag}_pmc/p$i -- python3 $root/tools/pmc_run.py > /dev/null 2>&1
done
cd $root
python3 tools/pmc_summarize.py $tag
