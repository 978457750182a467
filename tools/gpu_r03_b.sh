#!/bin/bash
# Round-3 batch B (GPU box): multi-GPU plumbing: C4-shape parity, csky_multi groups / staged / 4 in flight, bench.py self-launch + single-process forms.
R=$PWD; O=$R/gpurun_out/r03b; mkdir -p $O
timeout -s KILL 900 python -m pytest tests/test_gpu_round3.py tests/test_bench_contract.py "tests/test_gpu_round2.py::test_multi_device_handle_matches_single_context" "tests/test_gpu_round2.py::test_full_frame_c3_vs_oracle_tight" -x -q -s 2>&1 | tail -40 | tee $O/pytest.log
