#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --marker-trace output of the -DACT_ROCTX build -> a readable timeline: the named host-side ranges
(spend.phaseA / transcript.hash_end / spend.phaseB per chunk) interleaved with the kernels, times in ms from the first event.
usage: roctx_summarize.py <rocprofv3 output dir> [max_lines]"""
import csv
import glob
import os
import sys


def rows(pattern, d):
    out = []
    for f in glob.glob(os.path.join(d, "**", pattern), recursive=True):
        with open(f, newline="") as fh:
            out += list(csv.DictReader(fh))
    return out


def pick(r, *names):
    for n in names:
        if n in r and r[n] != "":
            return r[n]
    return ""


def main():
    d = sys.argv[1]
    limit = int(sys.argv[2]) if len(sys.argv) > 2 else 120
    ev = []
    for r in rows("*marker_api_trace.csv", d):
        name = pick(r, "Function", "Message", "Name")
        s, e = pick(r, "Start_Timestamp", "Start"), pick(r, "End_Timestamp", "End")
        if s and e:
            ev.append((int(s), int(e), "range ", name))
    for r in rows("*kernel_trace.csv", d):
        name = pick(r, "Kernel_Name", "Name")
        s, e = pick(r, "Start_Timestamp", "Start"), pick(r, "End_Timestamp", "End")
        if s and e and ("k_spend" in name or "k_hash" in name):
            short = name.split("(")[0].replace("act::", "").replace("void ", "")
            ev.append((int(s), int(e), "kernel", short))
    if not ev:
        print("no events found under", d)
        return
    ev.sort()
    # the part of the trace that shows the pipeline in steady state: the LAST pass over the batch (the first one grows the pinned
    # transcript buffers: hipHostMalloc inside hash_begin), i.e. from the last `spend.phaseA off=0` range on
    starts = [i for i, x in enumerate(ev) if x[2] == "range " and x[3].startswith("spend.phaseA off=0 ")]
    first = starts[-1] if starts else 0
    t0 = ev[first][0]
    counts = {}
    for _, _, kind, name in ev:
        key = kind + name.split(" ")[0]
        counts[key] = counts.get(key, 0) + 1
    print("events:", ", ".join("%s x%d" % (k, v) for k, v in sorted(counts.items())))
    print("%10s %10s %9s  %s" % ("start ms", "end ms", "ms", "event"))
    for s, e, kind, name in ev[first:first + limit]:
        print("%10.3f %10.3f %9.3f  %s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, kind, name[:110]))


if __name__ == "__main__":
    main()
