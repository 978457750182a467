#!/bin/bash
R=$PWD; O=$R/gpurun_out/r02; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "variants_and_schedules or error_behaviour or frames_in_flight" 2>&1 | tail -4
( python tools/tail_ab.py 5 9
  for f in "0.05 0.0" "0.05 0.10" "0.10 0.0" "0.10 0.20" "0.15 0.15" "0.20 0.0" "0.20 0.20" "0.30 0.30" "0.0 0.30"; do set -- $f; CSKY_TAIL_SEG4=$1 CSKY_TAIL_SEG2=$2 python tools/tail_ab.py 9 | sed "s/^/f4=$1 f2=$2 /"; done ) 2>&1 | grep beta | tee $O/tail_mixed_ab.txt
