"""The host-side parsers (image decoders, akaze-util file readers) under AddressSanitizer + UBSan on corrupted
inputs: they must reject or decode, never read or write out of bounds.  CPU build only (tools/fuzz/fuzz_host.cpp)."""
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_parsers_survive_corrupted_files_under_sanitizers(amd, tmp_path):
    PIL = pytest.importorskip("PIL.Image")
    exe = str(tmp_path / "fuzz_host")
    csrc = os.path.join(ROOT, "akaze-rust_amd", "csrc")
    build = subprocess.run(
        ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-D__HIP_PLATFORM_AMD__",
         "-I/opt/rocm/include", "-I" + csrc, os.path.join(ROOT, "tools", "fuzz", "fuzz_host.cpp"), os.path.join(csrc, "akz_image.cpp"),
         os.path.join(csrc, "akz_io.cpp"), "-o", exe, "-lz"], capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("no sanitizer runtime in this toolchain")
    assert build.returncode == 0, build.stderr[-2000:]
    im = PIL.open(os.path.join(ROOT, "tests", "golden", "1.jpg")).crop((300, 200, 460, 320))
    seeds = []
    for name, kw in (("base.jpg", dict(quality=85, subsampling=2)), ("prog.jpg", dict(quality=85, subsampling=2, progressive=True)),
                     ("rst.jpg", dict(quality=85, subsampling=1, restart_marker_blocks=5)), ("rgb.png", {}), ("rgb.ppm", {})):
        p = str(tmp_path / name)
        im.save(p, **kw)
        seeds.append(p)
    p = str(tmp_path / "pal.png")
    im.convert("P").save(p)
    seeds.append(p)
    kp = np.zeros(20, amd.KEYPOINT_DTYPE)
    kp["x"], kp["size"], kp["octave"] = np.arange(20), 3.5, 2
    desc = np.random.default_rng(0).integers(0, 256, (20, 61), dtype=np.uint8)
    m = np.zeros(7, amd.MATCH_DTYPE)
    m["index_0"], m["distance"] = np.arange(7), 1.5
    for ext in ("cbor", "json"):
        amd.serialize_features_to_file(kp, desc, str(tmp_path / f"features.{ext}"))
        amd.serialize_matches_to_file(m, str(tmp_path / f"matches.{ext}"))
        seeds += [str(tmp_path / f"features.{ext}"), str(tmp_path / f"matches.{ext}")]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    run = subprocess.run([exe, "120", str(tmp_path)] + seeds, capture_output=True, text=True, env=env, timeout=600)
    assert run.returncode == 0, (run.stdout[-500:], run.stderr[-3000:])
    assert "runtime error" not in run.stderr and "AddressSanitizer" not in run.stderr, run.stderr[-3000:]
    assert "decoded" in run.stdout


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_keypoint_selection_matches_linear_scans_under_sanitizers(tmp_path):
    """The host keypoint selection (counting sort, uniform grids with 2 x 2-cell queries, merged second pass) on random
    candidate sets with dense clusters and exact ties, under AddressSanitizer + UBSan, against a direct restatement of
    the reference's linear scans (tools/fuzz/fuzz_keypoints.cpp)."""
    exe = str(tmp_path / "fuzz_keypoints")
    csrc = os.path.join(ROOT, "akaze-rust_amd", "csrc")
    build = subprocess.run(
        ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-D__HIP_PLATFORM_AMD__",
         "-I/opt/rocm/include", "-I" + csrc, os.path.join(ROOT, "tools", "fuzz", "fuzz_keypoints.cpp"),
         os.path.join(csrc, "akz_keypoints.cpp"), os.path.join(csrc, "akz_plan.cpp"), "-o", exe], capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("no sanitizer runtime in this toolchain")
    assert build.returncode == 0, build.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    run = subprocess.run([exe, "400"], capture_output=True, text=True, env=env, timeout=600)
    assert run.returncode == 0, (run.stdout[-500:], run.stderr[-3000:])
    assert "runtime error" not in run.stderr and "AddressSanitizer" not in run.stderr, run.stderr[-3000:]
    assert "identical to the linear scans" in run.stdout
    m = re.search(r"\((\d+) rounds also through the neighbour lists", run.stdout)
    assert m and int(m.group(1)) >= 300, run.stdout  # select_keypoints_rel (the device's neighbour lists, brute-forced here)
    m = re.search(r"(\d+) rounds also as the device's data flow of turns", run.stdout)
    assert m and int(m.group(1)) >= 150, run.stdout  # akz_select.hpp as k_select uses it: owners in index order, interleaved at random


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_worker_pool_under_thread_sanitizer(tmp_path):
    """The per-context host worker pool (akz_pool.hpp) under ThreadSanitizer: thousands of short runs of varying size
    (every index exactly once), short-lived pools, two pools driven by two threads (tools/fuzz/pool_tsan.cpp)."""
    exe = str(tmp_path / "pool_tsan")
    build = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-pthread",
                            os.path.join(ROOT, "tools", "fuzz", "pool_tsan.cpp"), "-o", exe], capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("no ThreadSanitizer runtime in this toolchain")
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([exe, "2000"], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, (run.stdout[-500:], run.stderr[-3000:])
    assert "ThreadSanitizer" not in run.stderr and "pool ok" in run.stdout, run.stderr[-3000:]


@pytest.mark.skipif(shutil.which("g++") is None or not os.path.isdir("/opt/rocm/include"), reason="needs g++ and the HIP headers")
def test_remove_outliers_pool_under_thread_sanitizer(tmp_path):
    """akz_remove_outliers under ThreadSanitizer: its trials run on a process-wide pool of host threads, callers that arrive
    while it is busy start their own; four caller threads seeded alike get the list of a lone call
    (tools/fuzz/ransac_tsan.cpp compiles akz_ransac.cpp with the two symbols it takes from akz_api.cpp stubbed)."""
    exe = str(tmp_path / "ransac_tsan")
    build = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-pthread", "-ffp-contract=off",
                            "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                            os.path.join(ROOT, "tools", "fuzz", "ransac_tsan.cpp"), "-o", exe], capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("no ThreadSanitizer runtime in this toolchain")
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, (run.stdout[-500:], run.stderr[-3000:])
    assert "ThreadSanitizer" not in run.stderr and "ransac ok" in run.stdout, run.stderr[-3000:]
