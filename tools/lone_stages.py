import os, sys, time
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "akaze-rust_amd", "python"))
import numpy as np, torch
import akaze_amd as A
dev = torch.device("cuda", 0)
for (W, H) in ((1920, 1080), (3840, 2160)):
    frame = torch.from_numpy(A.synth_frame(W, H, 0)[None]).to(dev)
    cfg = A.Config()
    st = torch.cuda.Stream(dev); torch.cuda.set_stream(st)
    ctx = A.Context(0, st.cuda_stream); ctx.warmup()
    for _ in range(10): ctx.extract_features(frame, cfg).close()
    ctx.set_profiling(2); ctx.get_profile(reset=True)
    n = 50
    t = time.perf_counter()
    for _ in range(n): ctx.extract_features(frame, cfg).close()
    el = (time.perf_counter() - t) / n
    p = ctx.get_profile(reset=True)
    print(W, H, "sync call with stage profiling %.3f ms" % (el * 1e3), {k: round(v / n, 3) for k, v in p.items() if isinstance(v, float)})
    ctx.set_profiling(0)
    r = ctx.extract_features(frame, cfg); print("keypoints", r.counts(0)[1]); r.close()
    ctx.close()
