#!/bin/bash
# first GPU contact: parity tests, smoke, bench, kernel trace
set -x
mkdir -p gpurun_out
nproc; rocminfo | grep -m1 gfx; free -g | head -2
python -m pytest tests -m gpu -x -q 2>&1 | tail -40 | tee gpurun_out/pytest_gpu.log
python __graft_entry__.py smoke 2>&1 | tail -5 | tee gpurun_out/smoke.log
python bench.py --steps 10 --warmup 2 2>&1 | tee gpurun_out/bench1.json
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof1 -o r01 -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof1.log 2>&1
ls -R $R/gpurun_out/prof1 | head -30
