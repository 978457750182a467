#!/bin/bash
# Round-3 batch E (GPU box): bench line with the census roofline + sampled clocks; FETCH_SIZE calibration
R=$PWD; O=$R/gpurun_out/r03e; mkdir -p $O
timeout -s KILL 600 python bench.py --steps 200 > $O/bench_C3_n1.json 2> $O/bench.err; tail -5 $O/bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03e/bench_C3_n1.json') if l.startswith('{')][-1])
r=d['roofline']
print('value', d['value'], 'ms', d['ms_per_step'], 'one', d['value_one_frame_at_a_time'])
for k in ('achieved','peak','frac','frac_bounds','peak_source','headline','census_note'): print(k, r.get(k))
print('clocks', json.dumps(r.get('clocks'))[:1500])
print('xcheck', (r.get('valu_issue') or {}).get('census_over_hardware_counters'))
print('class model', {k:(r.get('valu_issue_class_counter_model') or {}).get(k) for k in ('frac','kernel_cycles','issue_cycles_per_simd')})
print('l1', (r.get('l1_gather') or {}).get('frac'), 'hbm', (r.get('hbm') or {}).get('frac'), 'pmc note', r['pmc']['note'])
PY
timeout -s KILL 400 bash tools/ubench/calibrate_fetch.sh 2>&1 | tail -4
