import csv,sys,glob
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last batch: find last k_blur5_stream
idx=[i for i,r in enumerate(rows) if 'k_blur5' in r['Kernel_Name']][-1]
t0=int(rows[idx]['Start_Timestamp'])
for r in rows[idx:]:
    n=r['Kernel_Name'].replace('void akz::(anonymous namespace)::','').split('(')[0]
    s=int(r['Start_Timestamp'])-t0; e=int(r['End_Timestamp'])-t0
    print(f"{s/1e3:9.1f} {e/1e3:9.1f} {(e-s)/1e3:8.1f}  {n[:50]:50s} grid {r.get('Grid_Size_X','?')} wg {r.get('Workgroup_Size_X','?')}")
import collections
tot = collections.Counter()
for r in rows[idx:]:
    n = r['Kernel_Name'].replace('void akz::(anonymous namespace)::', '').split('<')[0].split('(')[0]
    tot[n] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
print("SUM " + "  ".join(f"{k} {v:.0f}" for k, v in tot.most_common(8)))
