#!/bin/bash
# full GPU validation: parity tests, smoke, bench, kernel-trace + PMC profile of the default kernel
mkdir -p gpurun_out gpurun_out/prof_$1
timeout 300 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee gpurun_out/pytest_gpu.log
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -3 | tee gpurun_out/smoke.log
timeout 300 python bench.py 2>gpurun_out/bench_stderr.log | tee gpurun_out/bench.json
R=$PWD
(cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/prof_$1/bench_trace -o b -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_$1_bench_trace.log 2>&1)
# the same with strictly one frame at a time (the kernel with the GPU to itself)
timeout 300 python bench.py --no-cpu-baseline --frames-in-flight 1 --kernel-iters 0 2>/dev/null | tee gpurun_out/bench_fif1.json
(cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/prof_$1/bench_trace_fif1 -o b -- python $R/bench.py --no-cpu-baseline --frames-in-flight 1 --kernel-iters 0 > $R/gpurun_out/prof_$1_bench_trace_fif1.log 2>&1)
bash tools/profile.sh $1 --config C3 --frames 10
