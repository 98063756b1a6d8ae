"""Stand-alone cost of the keypoint kernels (k_orientation, k_mldb): plain extract_features calls on a 32-frame 1080p
batch, nothing in flight beside them.  Run under `rocprofv3 --kernel-trace --stats` and read the two kernels' averages."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "akaze-rust_amd", "python"))
import numpy as np, torch
import akaze_amd as A
ctx = A.Context(0, torch.cuda.current_stream().cuda_stream)
if os.environ.get("KP_NOISE"):
    frames = torch.randint(0, 256, (32, 1080, 1920), dtype=torch.uint8, device="cuda")
else:
    frames = torch.from_numpy(np.stack([A.synth_frame(1920, 1080, i) for i in range(32)])).cuda()
for _ in range(4):
    r = ctx.extract_begin(frames, keep_all_planes=True, host_descriptors=False).finish()
    n = sum(r.counts(i)[1] for i in range(r.num_images))
    r.close()
print("keypoints per batch", n)
