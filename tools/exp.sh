J='import json,sys; d=json.loads(sys.stdin.readline()); print(d["value"], d["ms_per_step"], d["roofline"], d["config"].get("host_ms_in_gather_per_step"))'
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 1 --steps 5 --warmup 2 --force-dist --no-cpu-baseline 2>&1 | grep '^{' | python3 -c "$J"
python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{' | python3 -c "$J"
