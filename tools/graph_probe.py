import sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "akaze-rust_amd", "python"))
import numpy as np, torch
import akaze_amd as A
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
ctx = A.Context(0, st.cuda_stream)
for (w, h, n) in ((1920, 1080, 1), (3840, 2160, 1), (1920, 1080, 4)):
    fr = torch.from_numpy(np.stack([A.synth_frame(w, h, i) for i in range(n)])).cuda()
    print(w, h, n, ctx.graph_probe(fr, reps=30))
