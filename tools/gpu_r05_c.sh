#!/bin/bash
# Round-5 GPU call C: unrolled light loop A/B, LUT rows beside the march A/B on a 1/8 share, the whole -m gpu suite, the demo-scene fit, one bench line.
O=gpurun_out/r05c; mkdir -p $O
P=$PWD/godot-volumetric-cloud-demo-v2_amd
for pass in 1 2; do for L in libcloudsky.so libcloudsky_unroll.so; do
  CSKY_LIBRARY=$P/$L timeout 200 python tools/ab_frame.py 2>&1 | grep -v amdgpu.ids
done; done | tee $O/unroll_ab.txt
for pass in 1 2; do
  TAG="rows in order:    " NS=1,8 timeout 300 python tools/share_matrix.py 8 2>/dev/null | grep "seg 0"
  TAG="rows beside march:" CSKY_ROWS_OVERLAP=1 NS=1,8 timeout 300 python tools/share_matrix.py 8 2>/dev/null | grep "seg 0"
done | tee $O/rows_overlap_ab.txt
timeout -s KILL 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
timeout -s KILL 400 python tools/demo_scene.py > $O/demo_scene.log 2>&1; cp -r profiles/r05 $O/profiles_r05
timeout -s KILL 400 python bench.py > $O/bench_C3_n1.json 2> $O/bench_C3_n1.err; python tools/show_bench.py $O/bench_C3_n1.json 2>/dev/null | head -20
