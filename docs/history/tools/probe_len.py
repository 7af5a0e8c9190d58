import os, sys, subprocess
for it in (512, 4096, 32768, 65536, 262144):
    out = subprocess.run([sys.executable, "-c", "from act_amd import capi; p, ms = capi.ubench_mad(0); print(%d, round(p/1e12,2), round(ms,2))" % it], env=dict(os.environ, ACT_UBENCH_ITERS=str(it), PYTHONPATH=os.getcwd()), capture_output=True, text=True)
    print(out.stdout.strip().split("\n")[-1], out.stderr.strip()[-200:] if out.returncode else "")
