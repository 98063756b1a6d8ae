import os, sys
sys.path.insert(0, "akaze-rust_amd/python"); sys.path.insert(0, "oracle")
import numpy as np, torch
import akaze_amd as A
import akaze_ref as R
R.build()
ctx = A.Context(0, torch.cuda.current_stream().cuda_stream)
ctx.set_detector_mode(5); ctx.set_prep_mode(3)
for (w, h) in ((320, 240), (517, 389)):
    frame = A.synth_frame(w, h, 0)
    res = ctx.extract_features(frame)
    rf = R.extract(frame)
    a, b = res.plane(0, "Lt"), rf.plane(0, "Lt")
    bad = np.argwhere(a.view(np.uint32) != b.view(np.uint32))
    print(w, h, "Lt0 bad", len(bad), "rows", np.unique(bad[:, 0])[:10], np.unique(bad[:, 0])[-5:], "cols", np.unique(bad[:, 1])[:10], np.unique(bad[:, 1])[-5:])
    if len(bad):
        y, x = bad[0]
        print("first", y, x, a[y, x], b[y, x])
