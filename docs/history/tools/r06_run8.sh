set -x
mkdir -p gpurun_out/r06e
rocm-smi --showpower --showclocks --showmaxpower --json | head -c 1500; echo
python3 tools/power_probe.py 5 > gpurun_out/r06e/power_probe.json 2> gpurun_out/r06e/power_probe.err; cat gpurun_out/r06e/power_probe.json | head -c 3000; echo
bash tools/pmc_ab.sh base3 pair2 fused > gpurun_out/r06e/pmc_ab.txt 2>&1; cat gpurun_out/r06e/pmc_ab.txt
