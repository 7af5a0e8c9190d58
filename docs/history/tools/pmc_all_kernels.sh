#!/bin/bash
# VALU instructions of EVERY kernel of one verify chunk (NB proofs, L = 128; tools/pmc_run.py), one rocprofv3 --pmc pass:
# the evidence behind "the path is work-bound" (DESIGN.md section 6): sum over kernels of instructions x cycles per instruction
# against the chunk's time.   usage (GPU box): docs/history/tools/pmc_all_kernels.sh <tag>  -> gpurun_out/<tag>_pmc_all_kernels.json
tag=${1:-r02_f}
export NB=${NB:-65536}
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
REPS=1 timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VMEM --kernel-trace --output-format csv -d $root/gpurun_out/${tag}_pmc_all -- python3 $root/tools/pmc_run.py > /dev/null 2>&1
cd $root
python3 - $tag <<'P'
import csv, glob, json, sys, collections
tag = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob("gpurun_out/%s_pmc_all/**/*counter_collection.csv" % tag, recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("act::", "")
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES": calls[k] += 1
out = {"proofs": int(__import__("os").environ.get("NB", "65536")), "note": "all launches of tools/pmc_run.py: one verify chunk plus the set-up that makes its proofs (request / issue / prove kernels on 256 lanes)", "kernels": {}}
for k, v in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
    out["kernels"][k] = {"launches": calls[k], "valu_wave_instructions": v.get("SQ_INSTS_VALU", 0), "waves": v.get("SQ_WAVES", 0), "wave_quad_cycles": v.get("SQ_WAVE_CYCLES", 0), "vmem_wave_instructions": v.get("SQ_INSTS_VMEM", 0)}
spend = {k: v for k, v in out["kernels"].items() if k.startswith(("k_spend", "k_hash"))}
s = sum(v["valu_wave_instructions"] for v in spend.values())
out["spend_kernels_valu_share"] = {k: v["valu_wave_instructions"] / s for k, v in spend.items()}
json.dump(out, open("gpurun_out/%s_pmc_all_kernels.json" % tag, "w"), indent=1)
print(json.dumps(out["spend_kernels_valu_share"], indent=1))
P
rm -rf gpurun_out/${tag}_pmc_all
