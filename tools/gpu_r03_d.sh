#!/bin/bash
# Round-3 batch D (GPU box): basic-block census of the cloud kernels (tools/isa_profile.py) + hardware counters of the same workload
R=$PWD; O=$R/gpurun_out/r03d; mkdir -p $O
timeout -s KILL 300 python tools/isa_profile.py run --config C3 --out $O/census_counts_C3.json 2>&1 | tail -5 | tee $O/census_run.log
timeout -s KILL 300 python tools/isa_profile.py run --config C5frame --out $O/census_counts_C5frame.json 2>&1 | tail -5 | tee -a $O/census_run.log
timeout -s KILL 400 python tools/pmc_collect.py --config C3 --out $O/pmc_C3.json 2>&1 | tail -3
timeout -s KILL 300 python -m pytest tests/test_gpu_round3.py::test_external_frame_import_error_paths -x -q 2>&1 | tail -3
ls -la $O
