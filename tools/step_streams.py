"""One steady-state step of the pipelined bench, stream by stream, from a rocprofv3 kernel trace of bench.py: for every
queue the busy time inside the step window and the largest gaps; then the step's launches in start order."""
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
blur = [i for i, r in enumerate(rows) if 'k_blur5' in r['Kernel_Name']]
a, b = blur[len(blur) // 2], blur[len(blur) // 2 + 1]
t0, t1 = int(rows[a]['Start_Timestamp']), int(rows[b]['Start_Timestamp'])
print(f"step window {(t1 - t0) / 1e3:.1f} us")
sel = [r for r in rows if int(r['End_Timestamp']) > t0 and int(r['Start_Timestamp']) < t1]
byq = collections.defaultdict(list)
for r in sel:
    byq[r['Queue_Id']].append(r)
def nm(r):
    return r['Kernel_Name'].replace('void akz::(anonymous namespace)::', '').replace('akz::(anonymous namespace)::', '').split('(')[0][:44]
for q, rs in byq.items():
    busy = sum(min(int(r['End_Timestamp']), t1) - max(int(r['Start_Timestamp']), t0) for r in rs)
    print(f"queue {q}: {len(rs)} launches, busy {busy / 1e3:.0f} us")
if len(sys.argv) > 2:
    for r in sel:
        s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
        print(f"{s / 1e3:9.1f} {e / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{r['Queue_Id']} {nm(r)}")
