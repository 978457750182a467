#!/bin/bash
# usage: tools/profile.sh <tag> [prof_kernel.py args...]
# kernel-trace + stats pass, then separate PMC passes (never combined with other trace domains).
TAG=$1; shift
R=/root/repo
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
APP="python $R/tools/prof_kernel.py $*"
timeout 120 rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o t -- $APP > $OUT/trace.log 2>&1
pass() { n=$1; shift; timeout 90 rocprofv3 --pmc "$@" -f csv -d $OUT/pmc_$n -o p -- $APP > $OUT/pmc_$n.log 2>&1; }
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM
pass sq2 SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_LDS
pass sq3 SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
find $OUT -name "*.csv" | head -40
