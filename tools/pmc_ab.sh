#!/bin/bash
# GRBM_GUI_ACTIVE / FETCH_SIZE / WRITE_SIZE of k_spend_bits for library variants (ACT_LIB_PATH), one --pmc pass each: clock and traffic A/B
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  for set in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU" "FETCH_SIZE" "WRITE_SIZE"; do
    rm -rf /tmp/pmcab; SKIP_ASSERT=1 REPS=1 ACT_LIB_PATH=$root/anonymous-credit-tokens_amd/libact_$v.so timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmcab -- python3 $root/tools/pmc_run.py > /dev/null 2>&1
    python3 - "$v" <<P
import csv, glob, sys, collections
tot = collections.defaultdict(float); dur = 0
for f in glob.glob("/tmp/pmcab/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_spend_bits" in r["Kernel_Name"]: tot[r["Counter_Name"]] += float(r["Counter_Value"])
for f in glob.glob("/tmp/pmcab/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_spend_bits" in r["Kernel_Name"]: dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
out = {"variant": sys.argv[1], "k_spend_bits_ms": round(dur, 2)}
for k, v in tot.items():
    out[k] = v
if "GRBM_GUI_ACTIVE" in tot: out["clock_ghz"] = round(tot["GRBM_GUI_ACTIVE"] / 8 / (dur * 1e6), 3)
if "FETCH_SIZE" in tot: out["fetch_GB_x2"] = round(2 * tot["FETCH_SIZE"] * 1024 / 1e9, 1)
if "WRITE_SIZE" in tot: out["write_GB"] = round(tot["WRITE_SIZE"] * 1024 / 1e9, 1)
print(out)
P
  done
done
