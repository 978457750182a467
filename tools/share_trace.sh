mkdir -p gpurun_out/r02
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d /root/repo/gpurun_out/r02/share8_trace -o s -- python /root/repo/tools/share_matrix.py 8 > /root/repo/gpurun_out/r02/share8_trace.log 2>&1
cd /root/repo
grep "seg 0" gpurun_out/r02/share8_trace.log
python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/r02/share8_trace/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        print("%-60s calls %6s avg %10.1f ns  total %6.2f%%" % (r['Name'][:60], r['Calls'], float(r['AverageNs']), float(r['Percentage'])))
PY
python tools/share_matrix.py 2 4 8 2>/dev/null | tee gpurun_out/r02/share_matrix_final.txt
