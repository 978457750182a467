#!/bin/bash
# Round-3 batch C (GPU box): async host form + shim throughput path + bench line with value_host_form
R=$PWD; O=$R/gpurun_out/r03c; mkdir -p $O
timeout -s KILL 600 python -m pytest tests/test_gpu_round3.py::test_submit_collect_over_the_pinned_ring tests/test_gdext.py -x -q 2>&1 | tail -15 | tee $O/pytest.log
timeout -s KILL 400 python bench.py --steps 100 --no-pmc > $O/bench_C3_n1_nopmc.json 2> $O/bench.err; tail -3 $O/bench.err
python tools/show_bench.py $O/bench_C3_n1_nopmc.json 2>/dev/null | head -30
python - <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1] if len(sys.argv)>1 else 'gpurun_out/r03c/bench_C3_n1_nopmc.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('value','ms_per_step','value_one_frame_at_a_time','value_host_form')})
PY
