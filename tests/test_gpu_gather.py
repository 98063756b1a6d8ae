"""The path's one exchange step through the C ABI (akz_comm_* / akz_gather_*) on one
GPU: RCCL with a single rank exercises communicator creation, the copy / exchange streams, the wire format and
both call forms; the multi-rank form of the same binary is `gather_selftest RANK NRANKS ID_FILE` on a multi-GPU node."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "akaze-rust_amd", "bin")


def test_gather_selftest_binary_world1():
    p = subprocess.run([os.path.join(BIN, "gather_selftest")], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "gather selftest ok: rank 0 of 1" in p.stdout


def test_comm_gather_matches_local_rows(ctx, amd):
    import torch
    comm = amd.Comm(0, amd.comm_unique_id(), 0, 1)
    comm.place_streams(ctx)  # its two streams onto queues / pipes the context's busy streams do not use
    frames = torch.from_numpy(np.stack([amd.synth_frame(320, 240, i) for i in range(3)])).cuda()
    res = ctx.extract_features(frames, keep_all_planes=False)
    rows = sum(res.counts(i)[1] for i in range(3))
    local = torch.zeros((rows, 64), dtype=torch.uint8, device="cuda")
    assert res.copy_device_descriptors(local) == rows
    host = np.concatenate([np.pad(res.descriptors(i), ((0, 0), (0, 3))) for i in range(3)])
    assert np.array_equal(local.cpu().numpy(), host)
    # Appendix C form
    allrows, counts = comm.gather_descriptors(local)
    assert counts == [rows] and torch.equal(allrows, local)
    # pipelined form: two gathers in flight, retired in order
    cap = rows + 100
    g1 = comm.gather_begin([res], cap)
    g2 = comm.gather_begin_rows(local, cap, torch.cuda.current_stream().cuda_stream)
    for g, n_img in ((g1, 3), (g2, 1)):
        g.stream_wait(torch.cuda.current_stream().cuda_stream)
        p, block_rows, cnt, img = g.finish()
        assert block_rows == cap + 1 and cnt == [rows] and img == [n_img]
        out = torch.empty((rows, 64), dtype=torch.uint8, device="cuda")
        amd.copy_d2d(out.data_ptr(), p + 64, rows * 64)
        assert torch.equal(out, local)
        g.free()
    # a shard that does not fit: the rank still takes part in the collective, finish reports AKZ_ERR_BUFFER (on every
    # rank) and the communicator stays usable; a gather still held when the communicator closes is retired by close()
    g3 = comm.gather_begin([res], rows - 1)
    with pytest.raises(amd.AkazeError) as e:
        g3.finish()
    assert e.value.status == -7
    g3.free()
    g4 = comm.gather_begin([res], rows + 1)  # + one row for the table of rows per image (3 images)
    assert g4.finish()[2] == [rows] and g4.image_rows(0) == [res.counts(i)[1] for i in range(3)]
    pairs = g4.match_all_pairs(ctx)          # BASELINE configs[4] through the C ABI: every image against every other one
    assert (pairs.n_images, pairs.first_owned, pairs.n_owned) == (3, 0, 3)
    for q in range(3):
        for j in range(3):
            got = pairs.matches(q, j)
            exp = ctx.descriptor_match(res.descriptors(q), res.descriptors(j), 10000, 0.86) if q != j else got[:0]
            assert np.array_equal(got, exp), (q, j)
    assert sorted(pairs.held(0)) == [(a, b) for a in range(3) for b in range(3) if a != b]
    total = pairs.total_matches()
    pairs.free()
    g4.free()
    g4 = comm.gather_begin([res], rows + 1)  # the next step: the freed object's buffers are reused, the lists are the same
    pairs = g4.match_all_pairs(ctx)
    assert pairs.total_matches() == total and total > 0
    for mode in (1, 0):  # the other matcher kernels compute the second direction separately: same lists
        ctx.set_match_mode(mode)
        other = g4.match_all_pairs(ctx)
        for q, j in ((0, 1), (1, 0), (2, 0), (0, 2)):
            assert np.array_equal(other.matches(q, j), pairs.matches(q, j)), (mode, q, j)
        other.free()
    ctx.set_match_mode(2)
    pairs.free()
    g5 = comm.gather_begin([res], rows + 1)   # never finished nor freed by the caller
    res.close()
    comm.close()
    g5.free()
    del g4, g5


def test_all_pairs_steps_of_different_sizes(ctx, amd):
    """akz_match_all_pairs over consecutive steps with 5, 2, 1 and 7 images of different sizes: consecutive lead images
    run on two streams with a scratch set each (akz::match_sets_at, side 0 / 1), which grow and are reused from step to
    step; every list equals the pair's descriptor_match (feature_matching.rs:23-94) in both directions."""
    import torch
    comm = amd.Comm(0, amd.comm_unique_id(), 0, 1)
    sizes = [(320, 240), (480, 360), (256, 200), (400, 300), (352, 288), (640, 480), (300, 220)]
    ress = [ctx.extract_features(torch.from_numpy(amd.synth_frame(w, h, i)[None]).cuda(), keep_all_planes=False)
            for i, (w, h) in enumerate(sizes)]
    ref = {}
    def expected(a, b):
        if (a, b) not in ref:
            ref[(a, b)] = ctx.descriptor_match(ress[a].descriptors(0), ress[b].descriptors(0), 10000, 0.86)
        return ref[(a, b)]
    for pick in ([0, 1, 2, 3, 4], [5, 2], [6], [3, 5, 0, 6, 1, 4, 2], [0, 1, 2, 3, 4]):
        rows = sum(ress[i].counts(0)[1] for i in pick)
        g = comm.gather_begin([ress[i] for i in pick], rows + 8)
        pairs = g.match_all_pairs(ctx)
        assert pairs.n_images == len(pick)
        n = 0
        for a in range(len(pick)):
            for b in range(len(pick)):
                if a == b:
                    continue
                got = pairs.matches(a, b)
                assert np.array_equal(got, expected(pick[a], pick[b])), (pick, a, b)
                n += len(got)
        assert n == pairs.total_matches() and (n > 0 or len(pick) == 1)
        pairs.free()
        g.free()
    for r in ress:
        r.close()
    comm.close()


@pytest.mark.parametrize("world,per", [(2, 3), (3, 2)])
def test_all_pairs_ranks_sharing_one_gpu(tmp_path, world, per):
    """BASELINE configs[4] with MORE THAN ONE rank, on the hardware there is: `world` real ranks on device 0 (tests/c5_rank.py),
    frame f on rank f mod world, the C ABI's blocks carried over gloo (akz_comm_create_external), akz_match_all_pairs on every
    rank.  Each rank checks every list it holds against akz_descriptor_match of the pair; here: every ORDERED pair of the job is
    held by exactly one rank, both directions of a pair by the same one, and the work is spread over the ranks."""
    import json
    import socket
    import sys
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "c5_rank.py"), str(r), str(world), str(port), str(tmp_path), str(per)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=420)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    n_job = world * per
    seen = {}
    for r in range(world):
        d = json.load(open(tmp_path / f"held_{r}.json"))
        assert d["frames"] == list(range(r, n_job, world))
        for a, b, n in d["held"]:
            assert (a, b) not in seen, (a, b)
            seen[(a, b)] = r
    assert sorted(seen) == [(a, b) for a in range(n_job) for b in range(n_job) if a != b]
    assert all(seen[(a, b)] == seen[(b, a)] for a, b in seen)
    per_rank = [sum(1 for v in seen.values() if v == r) for r in range(world)]
    assert min(per_rank) > 0 and max(per_rank) <= 2 * min(per_rank) + 2, per_rank


def test_bench_c5_workload_two_ranks_sharing_one_gpu():
    """`bench.py --gpus 2 --share-gpu --workload c5`: the configs[4] step (extraction, exchange, all-pairs match, totals read)
    with two ranks, reduced frame size; the line carries the job's pairs, the lists held per rank and the sampled check."""
    two = _bench_json(["--gpus", "2", "--share-gpu", "--workload", "c5", "--frames", "2", "--width", "1280", "--height", "720",
                       "--steps", "2", "--warmup", "1", "--regions", "2", "--no-cpu-baseline"])
    ap = two["config"]["all_pairs"]
    assert two["n_gpus"] == 2 and "all-pairs" in two["metric"] and two["config"]["exchange_ranks_seen"] == 2
    assert ap["images_per_step"] == 4 and ap["unordered_image_pairs_per_step"] == 6
    assert sum(ap["match_lists_held_per_rank"]) == 12 and min(ap["match_lists_held_per_rank"]) > 0
    assert ap["pairs_check"]["both_directions_equal_descriptor_match"] is True and ap["descriptor_pairs_per_step"] > 0
    assert two["config"]["regions"]["n"] == 2 and len(two["config"]["regions"]["Mpix_s"]) == 2
    one = _bench_json(["--gpus", "1", "--workload", "c5", "--frames", "4", "--width", "1280", "--height", "720",
                       "--steps", "2", "--warmup", "1", "--regions", "2", "--no-cpu-baseline"])
    assert one["config"]["all_pairs"]["descriptor_pairs_per_step"] == ap["descriptor_pairs_per_step"]
    assert one["config"]["all_pairs"]["matches_per_step"] == ap["matches_per_step"]


def test_external_transport_single_rank_and_its_error_paths(ctx, amd):
    """akz_comm_create_external (the caller carries the blocks): a gather may not be finished, waited for or matched before
    akz_gather_deliver; the synchronous two-collective form is refused; with the one rank's block copied into place the
    exchange gives the local rows back and akz_match_all_pairs equals pairwise descriptor_match."""
    import torch
    comm = amd.Comm(0, None, 0, 1)
    assert comm.external
    frames = torch.from_numpy(np.stack([amd.synth_frame(320, 240, i) for i in range(3)])).cuda()
    res = ctx.extract_features(frames, keep_all_planes=False)
    rows = sum(res.counts(i)[1] for i in range(3))
    local = torch.zeros((rows, 64), dtype=torch.uint8, device="cuda")
    res.copy_device_descriptors(local)
    with pytest.raises(amd.AkazeError) as e:
        comm.gather_descriptors(local)
    assert e.value.status < 0 and "akz_comm_create" in str(e.value)
    g = comm.gather_begin([res], rows + 8)
    for call in (lambda: g.finish(), lambda: g.stream_wait(torch.cuda.current_stream().cuda_stream), lambda: g.match_all_pairs(ctx)):
        with pytest.raises(amd.AkazeError) as e:
            call()
        assert "deliver" in str(e.value)
    send, recv, nbytes = g.blocks()
    assert nbytes == (1 + rows + 8) * 64
    amd.copy_d2d(recv, send, nbytes)
    g.deliver()
    p, block_rows, cnt, img = g.finish()
    assert block_rows == rows + 9 and cnt == [rows] and img == [3]
    out = torch.empty((rows, 64), dtype=torch.uint8, device="cuda")
    amd.copy_d2d(out.data_ptr(), p + 64, rows * 64)
    assert torch.equal(out, local)
    pairs = g.match_all_pairs(ctx)
    for q in range(3):
        for j in range(3):
            if q != j:
                assert np.array_equal(pairs.matches(q, j), ctx.descriptor_match(res.descriptors(q), res.descriptors(j), 10000, 0.86)), (q, j)
    lists, matches, dists = pairs.totals()
    assert lists == 6 and matches == pairs.total_matches() and dists == sum(res.counts(a)[1] * res.counts(b)[1] for a in range(3) for b in range(a + 1, 3))
    pairs.free()
    g.free()
    # akz_comm_set_timeout: an exchange that does not complete in time is an error (a peer is missing), not a hang; here the
    # blocks are "delivered" behind half a second of work on another stream
    comm.set_timeout(0.05)
    g3 = comm.gather_begin([res], rows + 8)
    send, recv, nbytes = g3.blocks()
    amd.copy_d2d(recv, send, nbytes)
    slow = torch.cuda.Stream()
    with torch.cuda.stream(slow):
        torch.cuda._sleep(int(1.0e9))   # ~0.5 s of spinning at ~2 GHz
    g3.deliver(slow.cuda_stream)
    with pytest.raises(amd.AkazeError) as e:
        g3.finish()
    assert e.value.status == -10 and "timeout" in str(e.value)
    comm.set_timeout(0)
    assert g3.finish()[2] == [rows]   # without a limit the same gather completes
    g3.free()
    g2 = comm.gather_begin([res], rows + 8)   # never delivered: freeing it and closing the communicator must not wait
    g2.free()
    res.close()
    comm.close()


def test_pairs_outlive_their_communicator(ctx, amd):
    """Lifetime rule of akz_pairs (include/akaze_hip.h): a result that is still held when its communicator is destroyed
    stays readable and is freed afterwards without touching the communicator (round-4 advice: akz_pairs_free pushed the
    object into the pool of a deleted akz_comm); freed objects beyond the pool's two are destroyed, pooled ones reused
    by a SECOND context on the same communicator (another matcher stream: the reuse waits for the last step's event)."""
    import torch
    comm = amd.Comm(0, amd.comm_unique_id(), 0, 1)
    frames = torch.from_numpy(np.stack([amd.synth_frame(320, 240, i) for i in range(3)])).cuda()
    res = ctx.extract_features(frames, keep_all_planes=False)
    rows = sum(res.counts(i)[1] for i in range(3))
    held = []
    for _ in range(4):  # four results alive at once: more than the pool keeps
        g = comm.gather_begin([res], rows + 8)
        held.append(g.match_all_pairs(ctx))
        g.free()
    exp = {(q, j): ctx.descriptor_match(res.descriptors(q), res.descriptors(j), 10000, 0.86) for q in range(3) for j in range(3) if q != j}
    for p in held[:3]:
        p.free()            # two go to the pool, the third is destroyed
    st2 = torch.cuda.Stream()
    ctx2 = amd.Context(0, st2.cuda_stream)
    g = comm.gather_begin([res], rows + 8)
    p2 = g.match_all_pairs(ctx2)  # pooled buffers, last used on ctx's stream
    for k, v in exp.items():
        assert np.array_equal(p2.matches(*k), v), k
    g.free()
    last = held[3]
    comm.close()            # `last` and `p2` are still held
    for k, v in exp.items():
        assert np.array_equal(last.matches(*k), v), k
    last.free()
    p2.free()
    ctx2.close()
    res.close()


def test_all_pairs_while_extractions_are_in_flight(ctx, amd):
    """The matcher's side stream is the context's finish stream: an all-pairs step enqueued while two batches of the same
    context are being finished by its own thread (their keypoint kernels and copies go to that stream too) gives the lists
    of a quiet context, and the batches' results are what a lone extraction gives."""
    import torch
    comm = amd.Comm(0, amd.comm_unique_id(), 0, 1)
    small = [ctx.extract_features(torch.from_numpy(amd.synth_frame(400, 300, 20 + i)[None]).cuda(), keep_all_planes=False) for i in range(6)]
    rows = sum(r.counts(0)[1] for r in small)
    g = comm.gather_begin(small, rows + 8)
    quiet = g.match_all_pairs(ctx)
    want = {(a, b): quiet.matches(a, b) for a in range(6) for b in range(6) if a != b}
    quiet.free(); g.free()
    big = torch.from_numpy(np.stack([amd.synth_frame(1920, 1080, 40 + i) for i in range(5)])).cuda()  # 10.4 Mpx: the batch path
    lone = ctx.extract_features(big, keep_all_planes=False)
    for it in range(6):
        jobs = [ctx.extract_begin(big, keep_all_planes=False), ctx.extract_begin(big, keep_all_planes=False)]
        g = comm.gather_begin(small, rows + 8)
        pairs = g.match_all_pairs(ctx)
        for (a, b), exp in want.items():
            assert np.array_equal(pairs.matches(a, b), exp), (it, a, b)
        pairs.free(); g.free()
        for j in jobs:
            r = j.finish()
            for i in range(5):
                assert r.counts(i) == lone.counts(i) and np.array_equal(r.descriptors(i), lone.descriptors(i))
            r.close()
    for r in small:
        r.close()
    lone.close()
    comm.close()


def test_bench_force_dist_capi_world1():
    """bench.py's N > 1 code path (gloo rendezvous + C-ABI exchange, retired one step late) with one rank."""
    import json
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--steps", "3", "--warmup", "1",
                        "--frames", "4", "--no-cpu-baseline", "--no-fed4k", "--no-single", "--no-match"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 1 and "C ABI" in out["config"]["exchange"]
    assert out["self_check"]["identical_to_single_frame_extraction"] is True
    assert out["roofline"]["kernel"] and out["roofline_2"]["kernel"]
    assert out["config"]["exchange_ranks_seen"] == 1


def _bench_json(extra, timeout=900):
    import json
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--regions", "2", "--no-cpu-baseline",
                        "--no-fed4k", "--no-single", "--no-match", "--no-host-share-leg", "--width", "960", "--height", "540"] + extra,
                       env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])


def test_bench_two_real_ranks_share_one_gpu():
    """The first N > 1 run, on the hardware there is: `bench.py --gpus 2 --share-gpu` starts two REAL extraction ranks on
    device 0 (launcher, gloo rendezvous, capacity agreement, two HIP extraction processes at once, pipelined retire of
    the exchange, rank pinning before HIP starts); the rows travel over gloo because RCCL refuses two ranks on one
    device.  Frame i of the job lives on rank i mod 2: every frame's keypoints + descriptors must equal what ONE rank
    computes for the same global frames."""
    two = _bench_json(["--gpus", "2", "--share-gpu", "--frames", "2", "--c5-leg-frames", "2"])
    one = _bench_json(["--gpus", "1", "--frames", "4"])
    assert two["n_gpus"] == 2 and two["config"]["share_gpu"] is True and "rehearsal" in two
    # the default N > 1 run also measures BASELINE configs[4] over all ranks (extra leg, own clock)
    leg = two["config"]["all_pairs_c5_leg"]
    assert leg["images_per_step"] == 4 and leg["unordered_image_pairs_per_step"] == 6 and leg["Mpix_s"] > 0
    assert sum(leg["match_lists_held_per_rank"]) == 12 and leg["matches_per_step"] > 0 and leg["transport"] == "external"
    assert one["config"]["all_pairs_c5_leg"] is None
    assert two["config"]["exchange_ranks_seen"] == 2
    assert len(two["config"]["per_rank_Mpix_s"]) == 2 and all(v > 0 for v in two["config"]["per_rank_Mpix_s"])
    assert two["self_check"]["identical_to_single_frame_extraction"] is True
    sha2, sha1 = two["config"]["frame_sha256_16"], one["config"]["frame_sha256_16"]
    assert sorted(sha2) == ["0", "1", "2", "3"] and sha2 == sha1, (sha2, sha1)
    assert one["host_input"]["value"] > 0 and two["config"]["placement"] is not None


def test_bench_five_real_ranks_share_one_gpu():
    """As far towards the 8-rank node as one box goes.  The box allows six processes on its GPU and this test process is one
    of them, so FIVE rank processes: five placement probes, five finisher threads and RANSAC pools on one host, a five-way
    rendezvous, the exchange with an odd world size -- for the default workload (with its configs[4] leg) and for
    `--workload c5`.  Every rank must take part in the exchange, every ordered image pair must be held by exactly one
    rank, every frame must equal what a single rank computes, and no rank may fall far behind the others."""
    world = 5
    run = _bench_json(["--gpus", str(world), "--share-gpu", "--frames", "2", "--c5-leg-frames", "1"], timeout=1500)
    cfg = run["config"]
    assert run["n_gpus"] == world and cfg["exchange_ranks_seen"] == world
    per = cfg["per_rank_Mpix_s"]
    assert len(per) == world and all(v > 0 for v in per)
    med = sorted(per)[world // 2]
    assert min(per) * 1.3 >= med, per   # (per-rank throughput: the slowest rank within 1.3 x of the median)
    leg = cfg["all_pairs_c5_leg"]
    assert "error" not in leg, leg
    assert leg["images_per_step"] == world and leg["unordered_image_pairs_per_step"] == world * (world - 1) // 2
    assert sum(leg["match_lists_held_per_rank"]) == world * (world - 1) and min(leg["match_lists_held_per_rank"]) > 0
    assert run["self_check"]["identical_to_single_frame_extraction"] is True
    one = _bench_json(["--gpus", "1", "--frames", "4"])   # (a single rank lists its first four frames: global frames 0 .. 3)
    sha, sha1 = cfg["frame_sha256_16"], one["config"]["frame_sha256_16"]
    assert sorted(sha, key=int) == [str(i) for i in range(2 * world)] and sorted(sha1) == ["0", "1", "2", "3"]
    assert all(sha[k] == v for k, v in sha1.items()), (sha, sha1)
    c5 = _bench_json(["--gpus", str(world), "--share-gpu", "--workload", "c5", "--frames", "2", "--width", "1280", "--height", "720"], timeout=1500)
    ap = c5["config"]["all_pairs"]
    n_img = 2 * world
    assert c5["config"]["exchange_ranks_seen"] == world and ap["images_per_step"] == n_img
    assert ap["unordered_image_pairs_per_step"] == n_img * (n_img - 1) // 2
    held = ap["match_lists_held_per_rank"]
    assert sum(held) == n_img * (n_img - 1) and min(held) > 0 and max(held) <= 2 * min(held), held
    assert ap["pairs_check"]["both_directions_equal_descriptor_match"] is True


def test_bench_host_share_of_an_8_rank_node():
    """One rank with 1/8 of the host cores as its thread budget: same results, and the line carries the share it ran with."""
    full = _bench_json(["--frames", "4", "--no-host-input"])
    eighth = _bench_json(["--frames", "4", "--no-host-input", "--host-share", "8"])
    assert eighth["config"]["placement"]["host_share"] == 8 and eighth["config"]["placement"]["cpus"] >= 1
    assert eighth["config"]["frame_sha256_16"] == full["config"]["frame_sha256_16"]
