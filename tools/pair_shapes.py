import os, sys, time
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "akaze-rust_amd", "python"))
import numpy as np, torch
import akaze_amd as A
dev = torch.device("cuda", 0)
pair = np.stack([A.synth_frame(3840, 2160, 0), A.synth_frame(3840, 2160, 0, shift=(17, 9))])
h_pair = torch.from_numpy(pair).pin_memory()
h_a, h_b = h_pair[0:1], h_pair[1:2]
cfg = A.Config()
st = torch.cuda.Stream(dev); torch.cuda.set_stream(st)
ctx = A.Context(0, st.cuda_stream); ctx.warmup()
def batch():
    t = time.perf_counter()
    rp = ctx.extract_begin_host(h_pair, cfg, keep_all_planes=True).finish()
    k0, k1, q0, q1 = rp.keypoints(0), rp.keypoints(1), rp.descriptors(0), rp.descriptors(1)
    t1 = time.perf_counter()
    m = A.match_features(k0, q0, k1, q1, 0.86, 1000, 3.0, ctx=ctx)
    t2 = time.perf_counter(); rp.close()
    return (t1 - t) * 1e3, (t2 - t1) * 1e3, len(m)
def two():
    t = time.perf_counter()
    ja = ctx.extract_begin_host(h_a, cfg, keep_all_planes=True); jb = ctx.extract_begin_host(h_b, cfg, keep_all_planes=True)
    ra = ja.finish(); rb = jb.finish()
    k0, k1, q0, q1 = ra.keypoints(0), rb.keypoints(0), ra.descriptors(0), rb.descriptors(0)
    t1 = time.perf_counter()
    m = A.match_features(k0, q0, k1, q1, 0.86, 1000, 3.0, ctx=ctx)
    t2 = time.perf_counter(); ra.close(); rb.close()
    return (t1 - t) * 1e3, (t2 - t1) * 1e3, len(m)
def seq():
    t = time.perf_counter()
    ra = ctx.extract_begin_host(h_a, cfg, keep_all_planes=True).finish(); rb = ctx.extract_begin_host(h_b, cfg, keep_all_planes=True).finish()
    k0, k1, q0, q1 = ra.keypoints(0), rb.keypoints(0), ra.descriptors(0), rb.descriptors(0)
    t1 = time.perf_counter()
    m = A.match_features(k0, q0, k1, q1, 0.86, 1000, 3.0, ctx=ctx)
    t2 = time.perf_counter(); ra.close(); rb.close()
    return (t1 - t) * 1e3, (t2 - t1) * 1e3, len(m)
for name, fn in (("one 2-frame call", batch), ("two jobs begun together", two), ("two calls one after the other", seq)) * 2:
    for _ in range(5): fn()
    ts = np.array([fn()[:2] for _ in range(20)])
    print("%-30s extract %.3f ms  match_features %.3f ms  pair %.3f ms  (matches %d)" % (name, np.median(ts[:, 0]), np.median(ts[:, 1]), np.median(ts.sum(1)), fn()[2]))
