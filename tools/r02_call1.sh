#!/bin/bash
# round-2 GPU call 1: baseline validation + the measurements the new gates / rooflines are written against
R=$PWD; O=$R/gpurun_out/r02; mkdir -p $O
timeout 300 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > $O/pytest_baseline.log
timeout 120 tools/ubench/valu_rates2 > $O/valu_rates2.txt 2>&1
timeout 180 tools/ubench/gather_rates > $O/gather_rates.txt 2>&1
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -f csv -d $O/valu_pmc -o v -- $R/tools/ubench/valu_rates2 > $O/valu_pmc.log 2>&1)
timeout 600 python tools/parity_stats.py > $O/parity_stats.txt 2> $O/parity_stats.err
timeout 300 python bench.py > $O/bench_base.json 2> $O/bench_base.err
timeout 300 python bench.py --frames-in-flight 1 --no-cpu-baseline > $O/bench_base_fif1.json 2>/dev/null
bash tools/profile.sh r02base --config C3 --frames 10 > $O/profile.log 2>&1
python tools/summarize_prof.py gpurun_out/prof_r02base > $O/prof_r02base_summary.json 2>/dev/null
# keep only the csv files of the PMC dirs small enough to pull back
find $O/valu_pmc -name "*.csv" -size +20M -delete
ls -la $O
