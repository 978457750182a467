#!/bin/bash
# Round-6 validation + evidence run (GPU box).  usage: bash tools/gpu_full_r06.sh   -> everything lands under gpurun_out/r06final/
R=$PWD; O=$R/gpurun_out/r06final; mkdir -p $O
timeout -s KILL 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu_full.log 2>&1; tail -6 $O/pytest_gpu_full.log | tee $O/pytest_gpu.log
timeout -s KILL 200 python __graft_entry__.py smoke 2>&1 | tail -1 | tee $O/smoke.log
# the driver's own invocation FIRST (a box that has done little): value = the protocol as asked (cold), value_prewarmed beside it
timeout -s KILL 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_C3_n1_driver_invocation.json 2> $O/bench_C3_n1_driver_invocation.err
cp gpurun_out/bench_pmc_C3.json $O/clouds_C3_pmc_live_from_bench.json 2>/dev/null
timeout -s KILL 400 python bench.py > $O/bench_C3_n1.json 2> $O/bench_C3_n1.err
timeout -s KILL 300 python bench.py --frames-in-flight 1 --no-cpu-baseline --no-pmc > $O/bench_C3_n1_one_frame_at_a_time.json 2>/dev/null
# the rocprofv3 summaries the bench line's durations must agree with: the timed region as bench.py runs it, and one frame at a time (the dominant kernel alone)
(cd /tmp && export TMPDIR=/tmp && timeout -s KILL 300 rocprofv3 --kernel-trace --stats -f csv -d $O/bench_trace -o b -- python $R/bench.py --no-cpu-baseline --no-pmc --no-host-form --no-early-out-leg > $O/bench_trace.log 2>&1)
(cd /tmp && export TMPDIR=/tmp && timeout -s KILL 300 rocprofv3 --kernel-trace --stats -f csv -d $O/bench_trace_fif1 -o b -- python $R/bench.py --no-cpu-baseline --no-pmc --no-host-form --no-early-out-leg --frames-in-flight 1 > $O/bench_trace_fif1.log 2>&1)
cp $O/bench_trace/b_kernel_stats.csv $O/bench_py_C3_n1_kernel_stats.csv 2>/dev/null
cp $O/bench_trace_fif1/b_kernel_stats.csv $O/bench_py_C3_n1_one_frame_at_a_time_kernel_stats.csv 2>/dev/null
{
for C in C2 C5frame C5; do echo "== bench $C"; timeout -s KILL 400 python bench.py --config $C --no-cpu-baseline 2>/dev/null; done
echo "== single-process form, 8 contexts on this one GPU (CSKY_BENCH_ONE_GPU_DEBUG=1: exercises the path, the number is meaningless; multi_stats = csky_multi_get_stats)"
CSKY_BENCH_ONE_GPU_DEBUG=1 timeout -s KILL 300 python bench.py --gpus 8 --single-process --steps 40 2>&1 | tail -2
echo "== process form, 8 ranks on this one GPU through gloo (the SCALE command rehearsed; the number is meaningless)"
CSKY_BENCH_ONE_GPU_DEBUG=1 timeout -s KILL 600 python bench.py --gpus 8 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -2
echo "== variants"; timeout -s KILL 120 python tools/prof_kernel.py --time --frames 20 2>/dev/null
} > $O/secondary_numbers.log
NS=1,2,4,8 timeout -s KILL 500 python tools/share_matrix.py 1 2 4 8 2>/dev/null > $O/share_matrix.txt
python tools/isa_profile.py run --config C3 --out $O/census_counts_C3.json 2>&1 | tail -3 > $O/census_run.log
python tools/isa_profile.py report $O/census_counts_C3.json --out $O/census_report_C3.json > $O/census_report_C3.txt 2>&1
timeout -s KILL 300 python tools/parity_stats.py 2>&1 | grep '^{' > $O/parity_stats.txt
rm -rf $O/bench_trace $O/bench_trace_fif1
ls -la $O
