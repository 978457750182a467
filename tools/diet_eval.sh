#!/bin/bash
# diet_eval.sh <tag> -- one GPU call per kernel-diet step: frame hash + kernel ms alone + two frames in flight (tools/ab_frame.py), whole-frame parity statistics
# against the oracle (tools/parity_stats.py), and the basic-block census of the C3 frame (tools/isa_profile.py run + report) -> gpurun_out/diet/<tag>/
T=${1:-step}; O=gpurun_out/diet/$T; mkdir -p $O
timeout 200 python tools/ab_frame.py > $O/ab_frame.txt 2>&1
timeout 300 python tools/parity_stats.py > $O/parity_stats.txt 2>&1
timeout 200 python tools/isa_profile.py run --config C3 --out $O/census_counts_C3.json > $O/census_run.txt 2>&1
timeout 100 python tools/isa_profile.py report $O/census_counts_C3.json --out $O/census_report_C3.json > $O/census_report_C3.txt 2>&1
grep -v amdgpu.ids $O/ab_frame.txt; python - <<PY
import json
for l in open("$O/parity_stats.txt"):
    if l.startswith("{"):
        d=json.loads(l); print("%-36s bit-identical %.4f  within1 %.5f  beyond2 px %d  psnr %.1f  in-cloud %d vs %d" % (d["case"], d["within0"], d["within1"], d["beyond2_pixels"], d["psnr"], d["incloud_gpu"], d["incloud_oracle"]))
PY
grep -A2 "== plain" $O/census_report_C3.txt
