O=gpurun_out/split_tail_ab.txt; : > $O
for rep in 1 2; do
python tools/split_tail_ab.py -1 >> $O 2>&1
python tools/split_tail_ab.py 7 >> $O 2>&1
for pct in 10 25 40; do for prio in 0 1; do CSKY_SPLIT_PCT=$pct CSKY_SPLIT_PRIO=$prio python tools/split_tail_ab.py 7 >> $O 2>&1; done; done
done
grep -v amdgpu.ids $O
