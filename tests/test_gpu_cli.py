"""File-level entry points and the command-line tools on the GPU: the shape of the
reference's integration tests (akaze/tests/integration-test.rs:41-123) on its own test images, with the CPU
oracle as the checker on the same decoded pixels."""
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
BIN = os.path.join(ROOT, "akaze-rust_amd", "bin")
IMG0, IMG1 = os.path.join(GOLDEN, "1.jpg"), os.path.join(GOLDEN, "2.jpg")


def run(tool, *args):
    p = subprocess.run([os.path.join(BIN, tool), *args], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    return p.stderr


def test_extract_features_file_matches_oracle_on_reference_images(ctx, amd, ref):
    """extract_features(test-data/1.jpg) and match_features against 2.jpg (integration-test.rs:41-93)."""
    res, feats = [], []
    for path in (IMG0, IMG1):
        r = ctx.extract_features_file(path, keep_all_planes=False)
        q = ref.extract(amd.load_image_luma(path), threads=8)
        assert q.num_keypoints > 100
        assert r.counts() == (16, q.num_keypoints, 61)
        assert r.keypoints().tobytes() == q.keypoints().tobytes()
        assert np.array_equal(r.descriptors(), q.descriptors())
        res.append(r)
        feats.append(q)
    got = ctx.descriptor_match(res[0].descriptors(), res[1].descriptors(), 10000, 0.86)
    exp = ref.descriptor_match(feats[0].descriptors(), feats[1].descriptors(), 10000, 0.86)
    assert len(exp) > 10 and np.array_equal(got, exp)
    full = amd.match_features(res[0].keypoints(), res[0].descriptors(), res[1].keypoints(), res[1].descriptors(),
                              0.86, 1000, 3.0, ctx=ctx)
    assert 0 < len(full) <= len(exp)


def test_cli_extract_features(ctx, amd, tmp_path):
    out, opts, dbg = str(tmp_path / "f.cbor"), str(tmp_path / "options.json"), str(tmp_path / "debug")
    log = run("extract_features", IMG0, out, "-o", opts, "--debug_path", dbg)
    assert "Done, extracted" in log and "Writing options file" in log
    assert json.load(open(opts))["num_sublevels"] == 4             # the defaults were written (extract_features.rs:76-80)
    kp, desc = amd.deserialize_features_from_file(out)
    r = ctx.extract_features_file(IMG0, keep_all_planes=False)
    assert kp.tobytes() == r.keypoints().tobytes() and np.array_equal(desc, r.descriptors())
    names = sorted(os.listdir(dbg))
    assert "keypoints.png" in names and "Lt_00000..png" in names and "Ldet_00015..png" in names  # build_path's set_extension(".png") quirk (evolution.rs:163-168)
    assert len(names) == 1 + 16 * 10 - 2                           # Lflow_00000 / Lstep_00000 are 0x0 and skipped
    assert amd.load_image(os.path.join(dbg, "Lt_00007..png")).shape == (756, 1008)
    marked = amd.load_image(os.path.join(dbg, "keypoints.png"))
    assert marked.shape == (1512, 2016, 3) and not np.array_equal(marked, amd.load_image_rgb(IMG0))
    # second run: the options file now exists and is read; a JSON output path selects serde_json
    json.dump(dict(json.load(open(opts)), detector_threshold=0.002), open(opts, "w"))
    out2 = str(tmp_path / "f.json")
    log = run("extract_features", IMG0, out2, "--options", opts)
    assert "Reading options file" in log
    kp2, _ = amd.deserialize_features_from_file(out2)
    r2 = ctx.extract_features_file(IMG0, amd.Config(detector_threshold=0.002), keep_all_planes=False)
    assert 0 < len(kp2) < len(kp) and kp2.tobytes() == r2.keypoints().tobytes()


def test_cli_extract_and_match_then_match_features(ctx, amd, tmp_path):
    prefix, mimg = str(tmp_path / "pair"), str(tmp_path / "matches.png")
    log = run("extract_and_match", IMG0, IMG1, prefix, "-m", mimg)
    assert "Got" in log
    kp0, d0 = amd.deserialize_features_from_file(prefix + "-extractions_0.cbor")
    kp1, d1 = amd.deserialize_features_from_file(prefix + "-extractions_1.cbor")
    m = amd.deserialize_matches_from_file(prefix + "-matches.cbor")
    amd.random_seed(42, 69)   # the tool is a fresh process: its RANSAC draws from the start of the stream
    exp = amd.match_features(kp0, d0, kp1, d1, 0.86, 1000, 3.0, ctx=ctx)
    assert len(m) > 0 and np.array_equal(m, exp)
    assert amd.load_image(mimg).shape == (1512, 4032, 3)
    out = str(tmp_path / "m.json")
    run("match_features", prefix + "-extractions_0.cbor", prefix + "-extractions_1.cbor", out, "-t", "12")
    assert np.array_equal(amd.deserialize_matches_from_file(out), m)


def test_cli_usage_errors():
    p = subprocess.run([os.path.join(BIN, "extract_features"), "only_one_arg"], capture_output=True, text=True)
    assert p.returncode != 0 and "required arguments" in p.stderr
    p = subprocess.run([os.path.join(BIN, "match_features"), "--help"], capture_output=True, text=True)
    assert p.returncode == 0 and "INPUT_EXTRACTIONS_0" in p.stdout
    p = subprocess.run([os.path.join(BIN, "extract_features"), "/nonexistent.jpg", "/tmp/x.cbor"], capture_output=True, text=True)
    assert p.returncode == 1 and "cannot read" in p.stderr
