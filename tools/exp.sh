J='import json,sys; d=json.loads(sys.stdin.readline()); r=d["roofline"]; print(d["value"], d["ms_per_step"], r["frac"], {k:(round(v,2) if not isinstance(v,str) else "") for k,v in d["stage_ms_per_step"].items()})'
b() { python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-fed4k "$@" 2>/dev/null | grep '^{' | python3 -c "$J"; }
for i in 1 2; do b --frames 1 --det-overlap; b --frames 1; done
b --frames 4 --det-overlap; b --frames 4
