#!/bin/bash
# A/B of the persistent launch form (CSKY_PERSISTENT: 0 never, 1 the library's policy, 2 every whole-ray launch)
# against plain launches -> gpurun_out/persistent_ab.txt (committed numbers: profiles/r02/persistent_launch_ab.txt).
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
{
  timeout -s KILL 200 python -m pytest tests/test_gpu_round2.py -m gpu -q -k persistent 2>&1 | tail -1
  for rep in 1 2 3; do
    for P in 0 1; do
      CSKY_PERSISTENT=$P timeout -s KILL 120 python bench.py --no-pmc --no-cpu-baseline --steps 100 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('P$P bench ms/step %.4f value %.1f' % (d['ms_per_step'], d['value']))"
    done
  done
  for P in 0 1 2; do
    CSKY_PERSISTENT=$P TAG="P$P " timeout -s KILL 200 python tools/share_matrix.py 1 2 4 8 2>&1 | grep "seg 0"
  done
  for P in 0 1; do
    CSKY_PERSISTENT=$P timeout -s KILL 120 python bench.py --config C5 --no-pmc --no-cpu-baseline --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C5 P$P bench ms/step %.4f value %.1f' % (d['ms_per_step'], d['value']))"
  done
} > gpurun_out/persistent_ab.txt 2>&1
cat gpurun_out/persistent_ab.txt
