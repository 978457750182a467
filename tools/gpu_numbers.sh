#!/bin/bash
# the secondary numbers quoted in DESIGN.md / README.md (not the headline bench line)
mkdir -p gpurun_out
{
for C in C2 C5frame C5; do echo "== bench $C"; timeout 300 python bench.py --config $C --no-cpu-baseline 2>/dev/null; done
echo "== bench C3 --also-early-out"; timeout 300 python bench.py --also-early-out --no-cpu-baseline 2>/dev/null
echo "== tiny_launch"; timeout 120 python tools/tiny_launch.py 2>/dev/null
echo "== phase_split"; timeout 120 python tools/phase_split.py 2>/dev/null
echo "== variants"; timeout 120 python tools/prof_kernel.py --time --frames 20 2>/dev/null
} | tee gpurun_out/numbers.log
