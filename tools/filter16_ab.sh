#!/bin/bash
# filter16_ab.sh -- ONE GPU call for the round-6 precision experiment: the product library against the `make filter16` build
# (y / z filter stages of every texture cell in packed fp16, cloud_core.h CSKY_FILTER16): kernel-alone ms and two-frames-in-flight ms (A B A B, so that
# box drift shows), whole-frame parity of BOTH builds under BOTH gates (tools/parity_stats.py), and the basic-block census of both -> gpurun_out/filter16/
O=gpurun_out/filter16; mkdir -p $O
P=godot-volumetric-cloud-demo-v2_amd
for rep in 1 2; do
  timeout 200 python tools/ab_frame.py 2>&1 | grep -v amdgpu.ids >> $O/ab_frame.txt
  CSKY_LIBRARY=$PWD/$P/libcloudsky_filter16.so timeout 200 python tools/ab_frame.py 2>&1 | grep -v amdgpu.ids >> $O/ab_frame.txt
done
timeout 400 python tools/parity_stats.py > $O/parity_product.txt 2>&1
CSKY_LIBRARY=$PWD/$P/libcloudsky_filter16.so timeout 400 python tools/parity_stats.py > $O/parity_filter16.txt 2>&1
timeout 200 python tools/isa_profile.py run --config C3 --out $O/census_counts_product.json > $O/census_run_product.txt 2>&1
timeout 100 python tools/isa_profile.py report $O/census_counts_product.json --out $O/census_report_product.json > $O/census_report_product.txt 2>&1
CSKY_CENSUS_VARIANT=filter16 timeout 200 python tools/isa_profile.py run --config C3 --out $O/census_counts_filter16.json > $O/census_run_filter16.txt 2>&1
CSKY_CENSUS_VARIANT=filter16 timeout 100 python tools/isa_profile.py report $O/census_counts_filter16.json --out $O/census_report_filter16.json > $O/census_report_filter16.txt 2>&1
cat $O/ab_frame.txt
python - <<PY
import json
for f in ("$O/parity_product.txt", "$O/parity_filter16.txt"):
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); print("%-26s %-36s bit-identical %.4f within1 %.5f beyond2 px %d max|d| %.2e psnr %.1f 8c-within %.5f tight %s survey-8c %s in-cloud %d vs %d" % (d["library"], d["case"], d["within0"], d["within1"], d["beyond2_pixels"], d["max_err"], d["psnr"], d["survey_8c_within_frac"], d["gate_tight_2ulp"], d["gate_survey_8c"], d["incloud_gpu"], d["incloud_oracle"]))
PY
grep -A3 "== plain" $O/census_report_product.txt $O/census_report_filter16.txt
