#!/bin/bash
R=$PWD; O=$R/gpurun_out/r02; mkdir -p $O
rm -f $O/parity_log.jsonl
CSKY_PARITY_LOG=$O/parity_log.jsonl timeout 900 python -m pytest tests -m gpu -q -x -s 2>&1 | tail -40 > $O/pytest_call2.log
tail -5 $O/pytest_call2.log
