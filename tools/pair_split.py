"""Host-side split of the BASELINE configs[2] call (2 x 4K from pinned host memory + match_features): where the wall
time of one synchronous pair goes between the C entry points and the Python marshalling around them."""
import os, sys, time
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "akaze-rust_amd", "python"))
import numpy as np, torch
import akaze_amd as A
dev = torch.device("cuda", 0)
pair = np.stack([A.synth_frame(3840, 2160, 0), A.synth_frame(3840, 2160, 0, shift=(17, 9))])
h_pair = torch.from_numpy(pair).pin_memory()
cfg = A.Config()
st = torch.cuda.Stream(dev); torch.cuda.set_stream(st)
ctx = A.Context(0, st.cuda_stream); ctx.warmup()
def one():
    t = [time.perf_counter()]
    job = ctx.extract_begin_host(h_pair, cfg, keep_all_planes=True); t.append(time.perf_counter())
    rp = job.finish(); t.append(time.perf_counter())
    k0, k1 = rp.keypoints(0), rp.keypoints(1); t.append(time.perf_counter())
    q0, q1 = rp.descriptors(0), rp.descriptors(1); t.append(time.perf_counter())
    m = ctx.descriptor_match(q0, q1, 10000, 0.86); t.append(time.perf_counter())
    m2 = A.match_features(k0, q0, k1, q1, 0.86, 1000, 3.0, ctx=ctx); t.append(time.perf_counter())
    rp.close(); t.append(time.perf_counter())
    return np.diff(np.array(t)) * 1e3, len(k0), len(k1), len(m), len(m2)
for _ in range(6): one()
rows = [one() for _ in range(20)]
med = np.median(np.array([r[0] for r in rows]), axis=0)
names = ["begin_host", "finish", "keypoints()", "descriptors()", "descriptor_match alone", "match_features (match + RANSAC)", "close"]
for n, v in zip(names, med): print("%-34s %.3f ms" % (n, v))
print("keypoints", rows[0][1], rows[0][2], "matches", rows[0][3], "after RANSAC", rows[0][4])
ctx.set_profiling(2); ctx.get_profile(reset=True)
for _ in range(5): one()
p = ctx.get_profile(reset=True)
print({k: round(v / 5, 3) for k, v in p.items() if isinstance(v, float)})
