python -m pytest tests -m gpu -q -x 2>&1 | tail -3
J='import json,sys; d=json.loads(sys.stdin.readline()); r=d["roofline"]; print(d["value"], d["ms_per_step"], r["frac"], {k:(round(v,2) if not isinstance(v,str) else "") for k,v in d["stage_ms_per_step"].items()})'
for a in "" "" "--frames 1"; do
echo "args: $a"; python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-fed4k $a 2>/dev/null | grep '^{' | python3 -c "$J"
done
