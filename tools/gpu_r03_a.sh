#!/bin/bash
# Round-3 batch A (GPU box): the compact-ilp kernel's tests + share matrix with and without it + profiler capability probes.
R=$PWD; O=$R/gpurun_out/r03a; mkdir -p $O
timeout -s KILL 400 python -m pytest tests/test_gpu_round3.py -x -q 2>&1 | tail -15 | tee $O/pytest_round3.log
{
echo "== baseline policy (ray segments for small shares)"
NS=1,2,4 TAG="base " timeout -s KILL 400 python tools/share_matrix.py 1 2 4 8 16 2>/dev/null | grep "seg 0"
echo "== compact-ilp for every launch (CSKY_ILP_MAX_WAVES=100000)"
CSKY_ILP_MAX_WAVES=100000 NS=1,2,4 TAG="ilp  " timeout -s KILL 400 python tools/share_matrix.py 1 2 4 8 16 2>/dev/null | grep "seg 0"
} | tee $O/share_matrix_ilp_ab.txt
{
echo "== rocprofv3 --list-avail (pc sampling section)"
(cd /tmp && export TMPDIR=/tmp && timeout -s KILL 60 rocprofv3 -L 2>&1 | grep -i -B2 -A12 "pc.sampl" | head -60)
for M in host_trap stochastic; do
  echo "== pc sampling probe: $M"
  U=time; I=1; if [ $M = stochastic ]; then U=cycles; I=65536; fi
  (cd /tmp && export TMPDIR=/tmp && timeout -s KILL 120 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $M --pc-sampling-unit $U --pc-sampling-interval $I --kernel-trace -f csv -d $O/pcs_$M -o p -- python $R/tools/prof_kernel.py --config C3 --frames 3 2>&1 | tail -8)
  ls -la $O/pcs_$M 2>/dev/null | head; find $O/pcs_$M -name "*.csv" | head
done
} > $O/pc_sampling_probe.txt 2>&1
for f in $(find $O -name "*pc_sampling*.csv" | head -4); do echo "$f: $(wc -l < $f) lines"; head -5 $f; done >> $O/pc_sampling_probe.txt 2>&1
# keep the merged output small
find $O -name "*.csv" -size +8M -delete
ls -la $O
