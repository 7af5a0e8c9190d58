#!/bin/bash
# Where do k_spend_bits's issue cycles go?  SQ activity counters for the kernel (tools/pmc_run.py) next to the same counters on
# tools/ubench_mix's blocks of pure multiply-accumulates and of multiply-accumulates + simple instructions (the calibration).
# usage (GPU box): bash tools/pmc_stall.sh <tag>   -> gpurun_out/<tag>_pmc_stall.txt
tag=${1:-r03_g}
export NB=${NB:-65536}
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_CYCLES SQ_BUSY_CU_CYCLES" \
           "SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU" \
           "SQ_WAVE_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU2 SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_INT64"; do
  i=$((i+1))
  REPS=1 timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $root/gpurun_out/${tag}_stall/k$i -- python3 $root/tools/pmc_run.py > /dev/null 2>&1
  timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $root/gpurun_out/${tag}_stall/u$i -- $root/tools/ubench_mix > /dev/null 2>&1
done
cd $root
python3 - $tag <<'P'
import csv, glob, sys, collections
tag = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("gpurun_out/%s_stall/**/*counter_collection.csv" % tag, recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_spend_bits" in k: key = "k_spend_bits"
        elif "k_mix" in k: key = k[k.index("k_mix"):][:12] + " grid=%s" % r["Grid_Size"]
        else: continue
        tot[key][r["Counter_Name"]] += float(r["Counter_Value"]); n[key][r["Counter_Name"]] += 1
with open("gpurun_out/%s_pmc_stall.txt" % tag, "w") as o:
    for key in sorted(tot):
        c = {m: tot[key][m] / n[key][m] for m in tot[key]}          # per launch
        o.write(key + "\n")
        for m in sorted(c): o.write("   %-24s %16.0f\n" % (m, c[m]))
        if "SQ_INSTS_VALU" in c and "SQ_WAVE_CYCLES" in c:
            o.write("   -> quad-cycles of wave residency per VALU instruction %.3f; ACTIVE_INST_VALU per VALU instruction %.3f; ACTIVE_INST_ANY / WAVE_CYCLES %.3f; IFETCH_LEVEL / IFETCH %.1f\n" % (
                c["SQ_WAVE_CYCLES"] / c["SQ_INSTS_VALU"], c.get("SQ_ACTIVE_INST_VALU", 0) / c["SQ_INSTS_VALU"], c.get("SQ_ACTIVE_INST_ANY", 0) / c["SQ_WAVE_CYCLES"], c.get("SQ_IFETCH_LEVEL", 0) / max(1.0, c.get("SQ_IFETCH", 0))))
print(open("gpurun_out/%s_pmc_stall.txt" % tag).read())
P
rm -rf gpurun_out/${tag}_stall
