#!/bin/bash
# FETCH_SIZE against KNOWN byte counts, separate from calibrate.sh (the TCC counters do not fit one pass with the others):
# the 1 GiB-footprint patterns miss the L2 on every line; a coalesced wave load is 1 KiB of whole lines, a same-line load one 128-byte line.
R=$PWD; O=$R/gpurun_out/r03/calib; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum SQ_INSTS_VMEM_RD --kernel-trace -f csv -d $O/fetch -o f -- $R/tools/ubench/gather_rates big > $O/fetch.log 2>&1
ls -la $O/fetch; tail -3 $O/fetch.log
