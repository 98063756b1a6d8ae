export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
python -m pytest tests -m gpu -q -x 2>&1 | tail -2
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_p2 -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-fed4k > $O/prof_p2.log 2>&1
grep '^{' $O/prof_p2.log | python3 -c 'import json,sys; d=json.loads(sys.stdin.readline()); print("profiled:", d["value"], d["ms_per_step"])'
python3 - <<'PY'
import csv,glob,os
rows=list(csv.DictReader(open(glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/prof_p2/*/*_kernel_stats.csv')[0])))
for r in rows[:14]:
    print(f"{r['Name'][:60]:60s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:8.2f} pct={r['Percentage']}")
PY
cd $R; python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-fed4k 2>/dev/null | grep '^{' | python3 -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(d["value"], d["ms_per_step"], {k:round(v,2) for k,v in d["stage_ms_per_step"].items()})'
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-fed4k --frames 1 2>/dev/null | grep '^{' | python3 -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(d["value"], d["ms_per_step"])'
