#!/bin/bash
# Calibration of the two measured roofs bench.py prices the cloud kernel against (GPU box):
#   * which SQ_INSTS_VALU_* counter each instruction kind lands in and what a kind costs to issue (SQ_BUSY_CYCLES / 32 SEs per instruction per SIMD)
#   * what one TCP (vector L1) cache access costs: TCP_TOTAL_CACHE_ACCESSES and TA_TA_BUSY of the gather patterns against their run time
# Output: gpurun_out/r03/calib/*.csv ; tools/ubench/calibration_table.py turns them into profiles/r03/issue_cost_calibration.json
R=$PWD; O=$R/gpurun_out/r03/calib; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_BUSY_CYCLES --kernel-trace -f csv -d $O/valu -o v -- $R/tools/ubench/valu_rates2 > $O/valu.log 2>&1
timeout 300 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD --kernel-trace -f csv -d $O/gather -o g -- $R/tools/ubench/gather_rates > $O/gather.log 2>&1
# FETCH_SIZE against KNOWN byte counts: the 1 GiB-footprint patterns miss the L2 on every line; a coalesced wave load is 1 KiB of whole lines,
# a same-line load one 128-byte line: what does FETCH_SIZE (KiB) report per TCC miss, and how many bytes does a miss really bring in?
# (FETCH_SIZE calibration: tools/ubench/calibrate_fetch.sh)
ls -la $O/valu $O/gather
