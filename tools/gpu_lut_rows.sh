#!/bin/bash
# round 4, sky LUT split over the ranks of a frame split: tests, shares with the whole LUT / the rank's rows / no LUT at all, and the
# N = 8 path of bench.py with all ranks on this one GPU (CSKY_BENCH_ONE_GPU_DEBUG=1: exercises gather + interleave of bands and LUT rows)
O=gpurun_out/lut_rows; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -s KILL 500 python -m pytest tests/test_gpu_round4.py tests/test_gpu_parity.py -q -m gpu -x -k "lut or sky or transmittance" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
{
for S in 8 4 2; do
  LUT=whole NS=1,2,4 timeout -s KILL 200 python tools/share_matrix.py $S 2>/dev/null | grep "seg [01]"
  LUT=rows NS=1,2,4 timeout -s KILL 200 python tools/share_matrix.py $S 2>/dev/null | grep "seg [01]"
  LUT=none NS=1,2,4 timeout -s KILL 200 python tools/share_matrix.py $S 2>/dev/null | grep "seg [01]"
done
} > $O/share_matrix_lut.txt 2>&1
cat $O/share_matrix_lut.txt
for N in 8 2; do
CSKY_BENCH_ONE_GPU_DEBUG=1 timeout -s KILL 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus $N --steps 6 --warmup 2 --no-cpu-baseline --no-pmc > $O/bench_debug_N$N.json 2> $O/bench_debug_N$N.err
tail -c 1500 $O/bench_debug_N$N.json; grep -i "gathered\|error\|Traceback" $O/bench_debug_N$N.err | head -5
done
CSKY_BENCH_ONE_GPU_DEBUG=1 timeout -s KILL 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 8 --groups 2 --steps 6 --warmup 2 --no-cpu-baseline --no-pmc > $O/bench_debug_N8_G2.json 2> $O/bench_debug_N8_G2.err
tail -c 600 $O/bench_debug_N8_G2.json; grep -i "gathered\|error\|Traceback" $O/bench_debug_N8_G2.err | head -5
