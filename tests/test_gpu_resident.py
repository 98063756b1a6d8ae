"""k_octave_resident (akz_resident.hip): the coarse end of the pyramid as one launch with one workgroup per image.
Every EvolutionStep plane, keypoint and descriptor byte against the CPU oracle, for shapes that exercise its patch
grid: widths / heights that are not multiples of the 8 x 8 patches or of 4 (scalar global accesses), a single border
column or row in the last patch, images that are resident from level 1 on (no 2x2 mean in front), several octaves
inside one launch, batches, the lean plane set, non-default pyramids, and -- in the default mode, where only batches
of 11 Mpx and more take it -- a 1080p batch at the size the bench runs."""
import numpy as np
import pytest

from test_gpu_extract import assert_same_result

pytestmark = pytest.mark.gpu


@pytest.fixture()
def rctx(amd):
    """prep mode 3: the fused / resident kernels wherever they are supported, whatever the batch size"""
    import torch
    c = amd.Context(0, torch.cuda.current_stream().cuda_stream)
    c.set_prep_mode(3)
    c.debug_set_host_sort(False)  # and the device sort of the candidate list (default only with < 4 host threads)
    yield c
    c.close()


@pytest.mark.parametrize("w,h", [
    (64, 64),      # resident from level 1 on (clone in front), one octave
    (161, 81),     # two octaves; odd sizes; 80 x 40 second octave
    (200, 137),    # h - 1 = 136: the last row of patches holds ONE image row
    (257, 120),    # w - 1 = 256: the last column of patches holds ONE image column
    (505, 393),    # 252 x 196 does not fit, 126 x 98 does: widths that are not multiples of 4
    (480, 270),    # 240 x 135 from octave 1: the shape of a 1080p frame's last octave, 510 of 512 patches
    (512, 256),    # 64 x 32 patches exactly fill the workgroup at octave 1 (256 x 128 -> 32 x 16), multiples of 8
    (1001, 300),   # wide and flat
])
def test_resident_all_planes(rctx, amd, ref, w, h):
    frame = amd.synth_frame(w, h, (w * 7 + h) % 50)
    rf = ref.extract(frame)
    assert_same_result(rctx.extract_features(frame), rf)


def test_resident_lean_batch_and_pyramids(rctx, amd, ref):
    """Lstep not kept (null plane pointer in the launch), a batch (one workgroup per image), 5 x 5 and 3-sublevel
    pyramids (five levels per octave; 13 to 54 steps per level)."""
    import torch
    frames = np.stack([amd.synth_frame(486, 270, 20 + i) for i in range(5)])
    lean = rctx.extract_features(torch.from_numpy(frames).cuda(), keep_all_planes=False)
    full = rctx.extract_features(torch.from_numpy(frames).cuda())
    for i in range(5):
        rf = ref.extract(frames[i])
        assert_same_result(lean, rf, planes=False, img=i)
        assert_same_result(full, rf, planes=(i in (0, 4)), img=i)
        for lvl in (5, 11):  # kept planes of the lean set are exact too
            assert np.array_equal(lean.plane(lvl, "Lflow", i), rf.plane(lvl, "Lflow"))
            assert np.array_equal(lean.plane(lvl, "Lt", i), rf.plane(lvl, "Lt"))
    frame = amd.synth_frame(640, 360, 7)
    for kw in (dict(num_sublevels=5, max_octave_evolution=5), dict(num_sublevels=3, detector_threshold=0.0005),
               dict(num_sublevels=2, max_octave_evolution=6)):
        assert_same_result(rctx.extract_features(frame, amd.Config(**kw)), ref.extract(frame, ref.default_config(**kw)))


def test_resident_flat_and_noise(rctx, amd, ref):
    """A constant frame (the contrast factor is 0, Lflow NaN from level 1 on, in the reference too) and uniform noise."""
    flat = np.full((120, 200), 77, np.uint8)
    assert_same_result(rctx.extract_features(flat), ref.extract(flat), equal_nan=True)
    noise = np.random.default_rng(3).integers(0, 256, (135, 240), dtype=np.uint8)
    assert_same_result(rctx.extract_features(noise), ref.extract(noise))


def test_default_mode_1080p_batch_all_planes(ctx, amd, ref):
    """The bench's shape in the default mode: a 6-frame 1080p batch (12.4 Mpx, above tiled_prep_px = 11 Mpx: the blur,
    contrast, level and detector marches of the fine octaves, the forked coarse chain and the resident last octave all
    engage -- five frames take the tiled preparation family); EVERY plane of one frame against the oracle."""
    import torch
    frames = np.stack([amd.synth_frame(1920, 1080, 60 + i) for i in range(6)])
    ctx.set_profiling(2)  # light: counts launches without moving the batch onto one stream
    ctx.get_profile(reset=True)
    res = ctx.extract_features(torch.from_numpy(frames).cuda())
    prof = ctx.get_profile(reset=True)
    ctx.set_profiling(0)
    # octave 0's three diffusing levels went through k_level_march (preparation inside the diffusion launch)
    assert prof["fused_px"] >= 3 * 6 * 1920 * 1080, prof
    assert_same_result(res, ref.extract(frames[2], threads=16), img=2)
    assert_same_result(res, ref.extract(frames[0], threads=16), planes=False, img=0)


def test_default_mode_odd_batch_all_planes(ctx, amd, ref):
    """The default mode on a batch of odd-sized frames above the 11 Mpx threshold (8 x 1501 x 999: dword accesses in
    every march kernel, four strips with a narrow last one, octave sizes 750 x 499, 375 x 249, 187 x 124): every plane of
    one frame, keypoints and descriptors of all."""
    import torch
    frames = np.stack([amd.synth_frame(1501, 999, 80 + i) for i in range(8)])
    res = ctx.extract_features(torch.from_numpy(frames).cuda())
    for i in range(8):
        assert_same_result(res, ref.extract(frames[i], threads=16), planes=(i == 3), img=i)


def test_input_ready_batches_run_ahead(ctx, amd, ref):
    """AKZ_INPUT_READY: the level-0 blur and the contrast factor of a large batch run on their own stream, under the
    kernels of the batch begun before (three different 6-frame 1080p batches, two in flight; the contrast scratch of the
    context is shared by them).  Identical to the same batches without the flag; every plane of one frame and the
    keypoints / descriptors of others against the oracle.  Host frames (the library's own upload) take the same path."""
    import torch
    batches = [np.stack([amd.synth_frame(1920, 1080, 200 + 6 * b + i) for i in range(6)]) for b in range(3)]
    dev = [torch.from_numpy(b).cuda() for b in batches]
    torch.cuda.synchronize()
    jobs = [ctx.extract_begin(dev[0], input_ready=True), ctx.extract_begin(dev[1], input_ready=True)]
    res = [jobs[0].finish()]
    jobs.append(ctx.extract_begin(dev[2], input_ready=True))
    res += [jobs[1].finish(), jobs[2].finish()]
    plain = [ctx.extract_begin(d).finish() for d in dev]
    host = [ctx.extract_begin_host(torch.from_numpy(b).pin_memory()) for b in batches[:2]]
    host = [j.finish() for j in host]
    for b in range(3):
        for i in range(6):
            assert res[b].keypoints(i).tobytes() == plain[b].keypoints(i).tobytes(), (b, i)
            assert res[b].descriptors(i).tobytes() == plain[b].descriptors(i).tobytes(), (b, i)
            if b < 2:
                assert host[b].keypoints(i).tobytes() == plain[b].keypoints(i).tobytes(), (b, i)
                assert host[b].descriptors(i).tobytes() == plain[b].descriptors(i).tobytes(), (b, i)
        assert float(res[b].contrast(4)) == float(plain[b].contrast(4))
    assert_same_result(res[1], ref.extract(batches[1][2], threads=16), img=2)
    assert_same_result(res[2], ref.extract(batches[2][0], threads=16), planes=False, img=0)
    assert_same_result(host[1], ref.extract(batches[1][4], threads=16), planes=False, img=4)


def test_small_frames_large_batch_resident_tail_runs_once(ctx, amd, ref):
    """Small frames in a batch above the 11 Mpx gate (90 x 480x270 = 11.7 Mpx): the resident tail starts at octave 1 (240 x
    135 fits one compute unit), BEFORE the octave the coarse chain would fork at -- the chain then forks where the tail
    starts and the levels behind it are not run a second time as separate launches (round-3 advice): three fused level
    launches + one resident launch, every plane of one frame against the oracle."""
    import torch
    frames = np.stack([amd.synth_frame(480, 270, 500 + i) for i in range(90)])
    ctx.set_profiling(2)
    ctx.get_profile(reset=True)
    res = ctx.extract_features(torch.from_numpy(frames).cuda())
    prof = ctx.get_profile(reset=True)
    ctx.set_profiling(0)
    assert prof["fed_launches"] <= 5, prof
    assert_same_result(res, ref.extract(frames[41]), img=41)
    assert_same_result(res, ref.extract(frames[0]), planes=False, img=0)


def test_eager_finish_batches_three_in_flight(amd, ref):
    """akz_ctx_set_eager_finish on a context without lanes: the finish half of every batch runs on the context's own
    thread while the caller begins the next ones (three 6-frame 1080p batches begun back to back, device and host input,
    every schedule variant); finish() only collects.  Identical to the plain pipelined calls; one frame's planes and
    others' keypoints against the oracle; an abandoned job and a context destroyed with a job in flight do no harm."""
    import torch
    ctx = amd.Context(0, torch.cuda.current_stream().cuda_stream)
    batches = [np.stack([amd.synth_frame(1920, 1080, 300 + 6 * b + i) for i in range(6)]) for b in range(3)]
    dev = [torch.from_numpy(b).cuda() for b in batches]
    torch.cuda.synchronize()
    ctx.set_eager_finish(False)  # (the default is on)
    plain = [ctx.extract_begin(d).finish() for d in dev]
    ctx.set_eager_finish(True)
    for sched in ((0, 1), (1, 1), (2, 0), (3, 0), (1, 0)):
        ctx.debug_set_schedule(0, sched[0])
        ctx.debug_set_schedule(1, sched[1])
        jobs = [ctx.extract_begin(dev[0], input_ready=True), ctx.extract_begin(dev[1], input_ready=True),
                ctx.extract_begin_host(torch.from_numpy(batches[2]).pin_memory())]
        res = [j.finish() for j in jobs]
        for b in range(3):
            for i in range(6):
                assert res[b].keypoints(i).tobytes() == plain[b].keypoints(i).tobytes(), (sched, b, i)
                assert res[b].descriptors(i).tobytes() == plain[b].descriptors(i).tobytes(), (sched, b, i)
    assert_same_result(res[0], ref.extract(batches[0][1], threads=16), img=1)
    assert_same_result(res[2], ref.extract(batches[2][3], threads=16), planes=False, img=3)
    pl = ctx.debug_stream_placement()
    assert pl["probed"] and pl["streams_sharing_a_queue"] == 0, pl  # (the test process asks for eight hardware queues)
    # small jobs take the same thread; abandon one, and destroy the context with one still in flight
    small = torch.from_numpy(amd.synth_frame(640, 360, 7)).cuda()
    ja, jb = ctx.extract_begin(small), ctx.extract_begin(small)
    del ja
    rb = jb.finish()
    assert_same_result(rb, ref.extract(amd.synth_frame(640, 360, 7)), planes=False)
    jc = ctx.extract_begin(dev[1], input_ready=True)
    del jc
    ctx.close()


def test_batch_soak_short():
    """tools/batch_soak.py for a few seconds: batches of 1-7 1080p frames from device and pinned host memory, with and
    without AKZ_INPUT_READY, one or two in flight; every frame against its synchronous extraction (a 75 s run checks
    ~11 000 frames in ~2 900 batches)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "batch_soak.py"), "5"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "no mismatch" in p.stdout
