"""Regenerates tests/golden/synthetic_small.npz from the CPU oracle.

The reference is Rust and cannot run in this image, so these vectors are produced by the
oracle (oracle/akaze_ref.cpp), not by the reference: they pin the oracle and the HIP path
against drift, they do not pin the oracle to the reference ("parity unpinned").
Inputs: two deterministic synthetic frames (akz_synth_frame_u8, frame 0 at 320x240 and its
(+5,+3)-shifted second view).  Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "akaze-rust_amd", "python")]
import akaze_amd  # noqa: E402  (host-side frame generator only)
import akaze_ref  # noqa: E402

out = {}
frames = {"a": akaze_amd.synth_frame(320, 240, 0), "b": akaze_amd.synth_frame(320, 240, 0, shift=(5, 3))}
res = {}
for name, frame in frames.items():
    r = akaze_ref.extract(frame)
    res[name] = r
    out[f"{name}_frame"] = frame
    out[f"{name}_num_keypoints"] = np.int64(r.num_keypoints)
    out[f"{name}_keypoints"] = r.keypoints().view(np.uint8)
    out[f"{name}_descriptors"] = r.descriptors()
    out[f"{name}_contrast"] = np.float64(r.contrast)
    sums = []
    for lvl in range(r.num_levels):
        for pl in ("Lt", "Lsmooth", "Lflow", "Ldet"):
            sums.append(hashlib.sha256(r.plane(lvl, pl).tobytes()).hexdigest()[:16])
    out[f"{name}_plane_sha"] = np.array(sums)
    print(name, frame.shape, "levels", r.num_levels, "keypoints", r.num_keypoints, "k", r.contrast)
m = akaze_ref.descriptor_match(res["a"].descriptors(), res["b"].descriptors(), 10000, 0.86)
out["matches_ab"] = m.view(np.uint8)
print("matches", len(m))
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "synthetic_small.npz"), **out)
