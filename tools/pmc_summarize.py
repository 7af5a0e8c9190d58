"""Turns the rocprofv3 counter CSVs written by tools/pmc_profile.sh into the two JSON summaries bench.py reads."""
import csv, glob, json, sys, collections
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_sha16      # bench.py cites a summary only if it was collected from the kernel sources it runs
SHA = kernel_source_sha16()
tag = sys.argv[1]; NB = int(os.environ.get("NB", "65536"))
tot = collections.defaultdict(float); dur = {}
for d in sorted(glob.glob("gpurun_out/%s_pmc/p*" % tag)):
    names = set()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_spend_bits" in r["Kernel_Name"]:
                tot[r["Counter_Name"]] += float(r["Counter_Value"]); names.add(r["Counter_Name"])
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_spend_bits" in r["Kernel_Name"]:
                for n in names: dur[n] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
waves = tot["SQ_WAVES"]
valu = {"kernel": "k_spend_bits", "kernel_source_sha16": SHA, "range_bits": 128, "proofs_per_launch": NB, "waves": waves,
        "valu_instructions_per_wave": tot["SQ_INSTS_VALU"] / waves,
        "wave_lifetime_cycles": 4 * tot["SQ_WAVE_CYCLES"] / waves,
        "cycles_per_valu_instruction_per_simd_2waves": 4 * tot["SQ_WAVE_CYCLES"] / (2 * tot["SQ_INSTS_VALU"]),
        # GRBM_GUI_ACTIVE sums the busy cycles of the 8 XCDs
        "effective_clock_ghz": tot["GRBM_GUI_ACTIVE"] / 8.0 / (dur["GRBM_GUI_ACTIVE"] * 1e6),
        "launch_ms_under_pmc": dur,
        "l2_hit_rate": tot["TCC_HIT_sum"] / max(1.0, tot["TCC_HIT_sum"] + tot["TCC_MISS_sum"]),
        "icache_hit_rate": 1.0 - tot["SQC_ICACHE_MISSES"] / max(1.0, tot["SQC_ICACHE_REQ"]),
        "wait_any_frac_of_wave_cycles": tot["SQ_WAIT_ANY"] / max(1.0, tot["SQ_WAVE_CYCLES"]),
        "raw": dict(tot),
        "note": "rocprofv3 --pmc (tools/pmc_profile.sh, one chunk in flight); SQ_WAVE_CYCLES counts quad-cycles; two waves share a SIMD"}
json.dump(valu, open("gpurun_out/%s_pmc_valu.json" % tag, "w"), indent=1)
f, w = tot["FETCH_SIZE"] * 1024, tot["WRITE_SIZE"] * 1024
json.dump({"kernel": "k_spend_bits", "kernel_source_sha16": SHA, "range_bits": 128, "proofs_per_launch": NB, "FETCH_SIZE_KB": tot["FETCH_SIZE"], "WRITE_SIZE_KB": tot["WRITE_SIZE"],
           "hbm_bytes_per_launch_uncorrected": f + w, "hbm_bytes_per_launch_fetch_x2": 2 * f + w,
           # calibrated on a known byte count in THIS access pattern (tools/calib_fetch.hip, profiles/r03_calib_fetch_144.txt: one
           # 144-byte entry per lane as nine dwordx4 at a 1 296-byte lane stride): FETCH_SIZE reports 128 B per lane (0.889 of the
           # bytes asked for; the two 128-byte lines an entry straddles would be 256 B = 2 x FETCH_SIZE), WRITE_SIZE 160 B per
           # lane (32-byte sectors: 1.11 x the bytes written)
           "fetch_calibration_factor": 1.125, "write_calibration_factor": 0.90, "hbm_bytes_per_launch_calibrated": 1.125 * f + 0.90 * w,
           "note": "separate --pmc passes (tools/pmc_profile.sh, NB=%d); gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md), both figures given; includes the per-lane Pippenger bucket and scratch traffic, counted at the L2's memory side (Infinity-Cache hits are not excluded)" % NB},
          open("gpurun_out/%s_pmc_hbm_traffic.json" % tag, "w"), indent=1)
print(json.dumps({k: v for k, v in valu.items() if k != "raw"}))

# ---- the prover's range kernel from the same passes (pmc_run.py makes its proofs with the engine's prover: one k_prove_bits launch) ----
pt = collections.defaultdict(float); pdur = {}; plaunches = collections.defaultdict(int)
for d in sorted(glob.glob("gpurun_out/%s_pmc/p*" % tag)):
    names = set()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_prove_bits" in r["Kernel_Name"]:
                pt[r["Counter_Name"]] += float(r["Counter_Value"]); names.add(r["Counter_Name"])
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_prove_bits" in r["Kernel_Name"]:
                for n in names: pdur[n] = pdur.get(n, 0.0) + (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6; plaunches[n] += 1
if pt.get("SQ_WAVES"):
    pw = pt["SQ_WAVES"]; nl = max(1, plaunches.get("SQ_INSTS_VALU", 1))
    pf, pwr = pt["FETCH_SIZE"] * 1024, pt["WRITE_SIZE"] * 1024
    prover = {"kernel": "k_prove_bits", "kernel_source_sha16": SHA, "range_bits": 128, "proofs_per_launch": NB, "launches": nl, "waves": pw,
              "valu_instructions_per_wave": pt["SQ_INSTS_VALU"] / pw, "wave_lifetime_cycles": 4 * pt["SQ_WAVE_CYCLES"] / pw,
              "cycles_per_valu_instruction_per_simd_2waves": 4 * pt["SQ_WAVE_CYCLES"] / (2 * pt["SQ_INSTS_VALU"]),
              "wait_any_frac_of_wave_cycles": pt["SQ_WAIT_ANY"] / max(1.0, pt["SQ_WAVE_CYCLES"]),
              "l2_hit_rate": pt["TCC_HIT_sum"] / max(1.0, pt["TCC_HIT_sum"] + pt["TCC_MISS_sum"]),
              "launch_ms_under_pmc": {k: v / max(1, plaunches[k]) for k, v in pdur.items()},
              "FETCH_SIZE_bytes_per_launch": pf / max(1, plaunches.get("FETCH_SIZE", 1)), "WRITE_SIZE_bytes_per_launch": pwr / max(1, plaunches.get("WRITE_SIZE", 1)),
              "raw": dict(pt),
              "note": "k_prove_bits (the input generation of tools/pmc_run.py: one launch of NB proofs per pass); FETCH_SIZE counts 128-byte requests at the L2's memory side: "
                      "every table entry is one 128-byte line (no wide-coalescing under-count applies to single-line reads)"}
    json.dump(prover, open("gpurun_out/%s_pmc_prover.json" % tag, "w"), indent=1)
    print(json.dumps({k: v for k, v in prover.items() if k != "raw"}))
