"""Single frame (PROBE_W x PROBE_H, default 1080p): host time inside begin / finish when streaming, and how long the GPU
needs for the chain that begin enqueues.  python tools/single_probe.py [unused] [jobs in flight]; PROBE_LANES=k and
PROBE_EAGER=1 select akz_ctx_set_lanes / akz_ctx_set_eager_finish."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "akaze-rust_amd", "python"))
import numpy as np, torch
import akaze_amd as A
dev = torch.device("cuda", 0)
W_, H_ = (int(os.environ.get("PROBE_W", 1920)), int(os.environ.get("PROBE_H", 1080)))
frames = torch.from_numpy(np.stack([A.synth_frame(W_, H_, i) for i in range(4)])).to(dev)
cfg = A.Config()
st = torch.cuda.Stream(dev)
ctx = A.Context(0, st.cuda_stream)
one = frames[0:1]
LANES, EAGER = int(os.environ.get("PROBE_LANES", 1)), int(os.environ.get("PROBE_EAGER", 0))
if LANES > 1:
    ctx.set_lanes(LANES)
    ctx.set_eager_finish(bool(EAGER))
for _ in range(10):
    ctx.extract_begin(one, cfg).finish().close()
torch.cuda.synchronize()
# streaming with DEPTH jobs in flight: host cost of begin / finish per frame
DEPTH = int(sys.argv[2]) if len(sys.argv) > 2 else 2
tb = tf = 0.0
reps = 300
t0 = time.perf_counter()
q = []
for _ in range(reps):
    a = time.perf_counter(); q.append(ctx.extract_begin(one, cfg)); tb += time.perf_counter() - a
    if len(q) >= DEPTH:
        a = time.perf_counter(); q.pop(0).finish().close(); tf += time.perf_counter() - a
while q:
    q.pop(0).finish().close()
el = time.perf_counter() - t0
print(f"lanes {LANES} eager {EAGER} streamed, {DEPTH} in flight: {el/reps*1e3:.3f} ms/frame; host in begin {tb/reps*1e3:.3f}, in finish {tf/reps*1e3:.3f}")
# GPU chain duration of begin alone: enqueue, sync
torch.cuda.synchronize()
ts = []
for _ in range(20):
    torch.cuda.synchronize(); a = time.perf_counter(); job = ctx.extract_begin(one, cfg); b = time.perf_counter(); torch.cuda.synchronize(); c = time.perf_counter()
    ts.append((b - a, c - a)); job.finish().close()
print("begin alone: host %.3f ms, until GPU idle %.3f ms" % (np.median([x for x, _ in ts]) * 1e3, np.median([y for _, y in ts]) * 1e3))
