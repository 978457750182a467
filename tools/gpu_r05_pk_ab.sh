#!/bin/bash
# Round-5 A/B: the packed-fp32 experiment builds (cloud_core.h CSKY_PK_FILTER / CSKY_PK_COORD) against the product build, two alternating passes of
# tools/ab_frame.py each (frame hash, kernel alone, two frames in flight).  usage: bash tools/gpu_r05_pk_ab.sh  -> gpurun_out/r05b/pk_ab.txt
O=gpurun_out/r05b; mkdir -p $O
P=$PWD/godot-volumetric-cloud-demo-v2_amd
for pass in 1 2; do for L in libcloudsky.so libcloudsky_pkf.so libcloudsky_pkc.so libcloudsky_pkfc.so; do
  CSKY_LIBRARY=$P/$L timeout 200 python tools/ab_frame.py 2>&1 | grep -v amdgpu.ids
done; done | tee $O/pk_ab.txt
