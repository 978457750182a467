#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > /root/repo/gpurun_out/counters_list.txt 2>&1
rocprofv3 --help > /root/repo/gpurun_out/rocprofv3_help.txt 2>&1
