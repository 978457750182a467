#!/bin/bash
# A/B timing of alternative builds of libcloudsky (CSKY_LIBRARY): full-frame and 1/8-frame cloud kernel time per build
# usage: CFGS="C3 C2" tools/ab_libs.sh lib1.so lib2.so ...
mkdir -p gpurun_out
for L in "$@"; do
  for CFG in ${CFGS:-C3}; do
    echo "== $L $CFG"
    CSKY_LIBRARY=$PWD/godot-volumetric-cloud-demo-v2_amd/$L timeout 120 python tools/prof_kernel.py --time --frames 20 --config $CFG 2>&1 | grep -E "variant (${VARIANTS:-1}) "
  done
done | tee gpurun_out/ab_libs.log
