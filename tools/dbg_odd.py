import os, sys
sys.path.insert(0, "akaze-rust_amd/python"); sys.path.insert(0, "oracle")
import numpy as np, torch
import akaze_amd as A
import akaze_ref as R
R.build()
ctx = A.Context(0, torch.cuda.current_stream().cuda_stream)
ctx.set_detector_mode(5); ctx.set_prep_mode(3)
w, h = 517, 389
frame = A.synth_frame(w, h, 2)
res = ctx.extract_features(frame)
rf = R.extract(frame)
for lvl in (0, 1):
    for pl in ("Lx", "Ldet", "Lsmooth", "Lt", "Lflow"):
        a, b = res.plane(lvl, pl), rf.plane(lvl, pl)
        if a.size == 0: continue
        bad = np.argwhere(a.view(np.uint32) != b.view(np.uint32))
        print(lvl, pl, len(bad), "rows", np.unique(bad[:, 0])[:12], "cols", np.unique(bad[:, 1])[:12], np.unique(bad[:,1])[-5:] if len(bad) else "")
