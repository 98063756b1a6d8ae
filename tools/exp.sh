cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c5 -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-fed4k --width 3840 --height 2160 --frames 8 --sublevels 5 --octaves 5 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob,re
f=sorted(glob.glob('gpurun_out/c5/**/*kernel_stats.csv',recursive=True))[-1]
rows=list(csv.DictReader(open(f)))
tot=sum(int(r['TotalDurationNs']) for r in rows)
for r in rows[:18]:
    n=re.sub(r'\(anonymous namespace\)::|akz::|void ','',r['Name']).split('(')[0][:46]
    print(f"{n:48s} calls {r['Calls']:>5s} total {int(r['TotalDurationNs'])/1e6:8.2f} ms avg {float(r['AverageNs'])/1e3:8.1f} us {float(r['Percentage']):.1f}%")
print(tot/1e6/6)
PY
