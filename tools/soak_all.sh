set -x
( time ACT_SOAK_THREADS=4 ACT_SOAK_TINY=1 timeout 1500 python3 tools/soak.py 3000 66 ) > gpurun_out/r06_soak.log 2>&1; tail -6 gpurun_out/r06_soak.log
( time ACT_SOAK_MAX_BATCH=4000 timeout 900 python3 tools/soak.py 1500 67 ) >> gpurun_out/r06_soak.log 2>&1; tail -4 gpurun_out/r06_soak.log
( time timeout 900 python3 tools/soak_wire.py 600 68 ) > gpurun_out/r06_soak_wire2.log 2>&1; tail -4 gpurun_out/r06_soak_wire2.log
( time ACT_SOAK_L=128 ACT_SOAK_MAX_BATCH=65536 timeout 1500 python3 tools/soak.py 12000 71 ) > gpurun_out/r06_soak_midsize.log 2>&1; tail -5 gpurun_out/r06_soak_midsize.log
( time ACT_SOAK_L=128 ACT_SOAK_MAX_BATCH=65536 timeout 1500 python3 tools/soak.py 20000 72 ) >> gpurun_out/r06_soak_midsize.log 2>&1; tail -5 gpurun_out/r06_soak_midsize.log
