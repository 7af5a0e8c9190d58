#!/bin/bash
# PMC A/B: instruction counts, wave cycles and instruction-cache behaviour of k_spend_bits for each libact_<name>.so
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  export ACT_LIB_PATH=$GRAFT_REPO_ROOT/anonymous-credit-tokens_amd/libact_$v.so
  for set in "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_LDS" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"; do
    tag=$(echo $set | cut -c1-12 | tr ' ' '_')
    NB=16384 REPS=1 timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$v/$tag -- python3 $GRAFT_REPO_ROOT/tools/pmc_run.py > /dev/null 2>&1
  done
done
cd $GRAFT_REPO_ROOT
python3 - "$@" <<'P'
import csv, glob, sys, collections
for v in sys.argv[1:]:
    tot = collections.defaultdict(float); n = collections.Counter()
    for f in glob.glob("gpurun_out/pmc_%s/*/**/*counter_collection.csv" % v, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_spend_bits" in r["Kernel_Name"]:
                tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print(v, {k: (tot[k], n[k]) for k in sorted(tot)})
P
