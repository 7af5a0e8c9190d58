#!/usr/bin/env python3
"""Is the range kernel bound by the chip's power budget?  Runs the bench workload (2^20 verifies per step, default library or
ACT_LIB_PATH) in a child process and samples `rocm-smi --showpower --showclocks --showmaxpower -t --json` five times a second beside it:
socket power against the cap, engine clock against its maximum, while k_spend_bits owns the GPU (99 % of a step).
usage: python3 tools/power_probe.py [steps]   -> one JSON line (median / max of the samples taken while the timed region ran)"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
steps = sys.argv[1] if len(sys.argv) > 1 else "6"


def sample():
    try:
        r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower", "--showtemp", "--json"], capture_output=True, text=True, timeout=10)
        d = json.loads(r.stdout)
        return d.get("card0", d)
    except Exception as e:      # the probe must not die of one bad sample
        return {"error": repr(e)}


def num(v):
    import re
    m = re.search(r"[-+]?\d+(\.\d+)?", str(v))
    return float(m.group(0)) if m else None


idle = sample()
child = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", steps, "--warmup", "2", "--no-extras", "--no-cpu-baseline", "--no-node-multi"],
                         stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
samples = []
t0 = time.time()
while child.poll() is None:
    s = sample(); s["t"] = round(time.time() - t0, 2); samples.append(s)
    time.sleep(0.2)
out = child.stdout.read()
line = [l for l in out.splitlines() if l.startswith("{")]
bench = json.loads(line[-1]) if line else {}
keys = sorted({k for s in samples for k in s if k not in ("t", "error")})
# the timed region is the last steps * ms_per_step of the child's life (minus ~1 s of teardown): take the samples of its middle
T = samples[-1]["t"] if samples else 0
span = bench.get("ms_per_step", 2000) * int(steps) / 1e3
busy = [s for s in samples if T - 1.0 - span * 0.9 <= s["t"] <= T - 1.0 - span * 0.1]
res = {"verifies_per_s": round(bench.get("value", 0)), "ms_per_step": bench.get("ms_per_step"), "samples_in_timed_region": len(busy), "idle": {k: idle.get(k) for k in keys}}
for k in keys:
    vals = sorted(v for v in (num(s.get(k)) for s in busy) if v is not None)
    if vals:
        res[k] = {"median": vals[len(vals) // 2], "min": vals[0], "max": vals[-1]}
print(json.dumps(res))
