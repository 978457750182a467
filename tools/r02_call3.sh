#!/bin/bash
R=$PWD; O=$R/gpurun_out/r02; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > $O/pytest_call3.log
tail -3 $O/pytest_call3.log
( python tools/tail_ab.py 5 7 8; for b in 0.3 0.5 0.8 1.0; do CSKY_TAIL_BETA=$b python tools/tail_ab.py 8; done ) 2>&1 | grep beta | tee $O/tail_ab.txt
make -C godot-volumetric-cloud-demo-v2_amd/csrc timeline -s > /dev/null 2>&1
for sch in 5 8; do echo "== schedule $sch"; CSKY_LIBRARY=$R/godot-volumetric-cloud-demo-v2_amd/libcloudsky_timeline.so python tools/timeline.py 1 $sch; done 2>&1 | grep -v amdgpu.ids | tee $O/timeline_s5_s8.txt
python tools/two_streams.py 5 8 2>&1 | grep -E "schedule|1/1 frame" | tee $O/two_streams_s5_s8.txt
