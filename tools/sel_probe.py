import os, sys
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "akaze-rust_amd", "python"))
import numpy as np, torch
import akaze_amd as A
c = A.Context(0, torch.cuda.current_stream().cuda_stream)
c.debug_set_select(2)
for name, fr in [("1080p_%d" % i, A.synth_frame(1920, 1080, i)) for i in range(4)] + [("4k_%d" % i, A.synth_frame(3840, 2160, i)) for i in range(3)] + [("640_%d" % i, A.synth_frame(640, 360, 70+i)) for i in range(3)]:
    r = c.extract_features(torch.from_numpy(fr[None]).cuda(), keep_all_planes=False)
    info = c.debug_select_info()
    print(name, "mode", info[0], "looks", info[1], "fallen", info[2] & 0xffff, "status bits", info[2] >> 16, "cands", info[3], "kps", r.counts(0)[1], "phase us", [round(x / 100.0, 1) for x in info[4:8]])
    r.close()
