J='import json,sys; d=json.loads(sys.stdin.readline()); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["config"]["keypoints_per_step_rank0"], {k:round(v,2) for k,v in d["stage_ms_per_step"].items()})'
echo "4K x8 4x4"; python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fed4k --width 3840 --height 2160 --frames 8 2>/dev/null | grep '^{' | python3 -c "$J"
echo "4K x1 4x4"; python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fed4k --width 3840 --height 2160 --frames 1 2>/dev/null | grep '^{' | python3 -c "$J"
echo "4K x8 5x5"; python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fed4k --width 3840 --height 2160 --frames 8 --sublevels 5 --octaves 5 2>/dev/null | grep '^{' | python3 -c "$J"
echo "2016x1512 x16"; python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fed4k --width 2016 --height 1512 --frames 16 2>/dev/null | grep '^{' | python3 -c "$J"
