"""world_size-2 tests of the multi-GPU logic on CPU (gloo).  (a) The PRODUCT's exchange -- `akz_comm.cpp` behind the C ABI --
on a host-memory communicator (AKZ_COMM_HOST): its wire format (header row, descriptor rows, per-image table), the overflow
protocol, the capacity check and the all-pairs plan (akz_pairs_plan: lead rule, holder map) with the blocks carried by
gloo, no GPU anywhere.  (b) per-image sharding and the Python helpers `akaze_amd.gather_descriptor_rows` /
`all_pairs_match` (torch.distributed).  With a device the same C-ABI objects are driven by tests/test_gpu_gather.py -- two and
three real ranks on one GPU -- and by tests/test_comm_faults.py against a stub librccl."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, counts, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "akaze-rust_amd", "python"))
    import torch
    import torch.distributed as dist
    import akaze_amd as A
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(100 + rank)
        local = torch.from_numpy(rng.integers(0, 256, (counts[rank], 64), dtype=np.uint8))
        rows, cnts = A.gather_descriptor_rows(local)
        # fixed-capacity form (no host sync): same rows once the padding is stripped
        padded, dcnt = A.gather_descriptor_rows(local, cap_rows=16)
        assert padded.shape == (world, 16, 64) and [int(v) for v in dcnt.tolist()] == list(cnts)
        strip = torch.cat([padded[r, :cnts[r]] for r in range(world)], dim=0)
        assert torch.equal(strip, rows)
        assert all(int(padded[r, cnts[r]:].sum()) == 0 for r in range(world))
        # a capacity that the largest shard exceeds: no rank leaves the collective (that would hang the others); the
        # gathered counts tell every rank which shard did not fit, the shards that fit arrive intact
        cap = max(cnts) - 1
        small, dcnt2 = A.gather_descriptor_rows(local, cap_rows=cap)
        assert [int(v) for v in dcnt2.tolist()] == list(cnts)
        for r in range(world):
            if cnts[r] <= cap:
                assert torch.equal(small[r, :cnts[r]], rows[sum(cnts[:r]):sum(cnts[:r + 1])])
            else:
                assert int(small[r].sum()) == 0
        np.save(os.path.join(out_dir, f"rows_{rank}.npy"), rows.numpy())
        np.save(os.path.join(out_dir, f"cnts_{rank}.npy"), np.array(cnts))
        np.save(os.path.join(out_dir, f"local_{rank}.npy"), local.numpy())
        # sharding: every frame owned exactly once
        owned = A.shard_frames(11, rank, world)
        np.save(os.path.join(out_dir, f"owned_{rank}.npy"), np.array(owned))
    finally:
        dist.destroy_process_group()


def _capi_worker(rank, world, port, sizes, out_dir):
    """One rank of the product's exchange on host memory: blocks by akz_gather_blocks, carried by gloo."""
    sys.path.insert(0, os.path.join(ROOT, "akaze-rust_amd", "python"))
    import ctypes as C
    import pickle
    import torch.distributed as dist
    import akaze_amd as A
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = None
    try:
        rng = np.random.default_rng(50 + rank)
        per = list(sizes[rank])
        local = rng.integers(0, 256, (sum(per), 64), dtype=np.uint8)
        comm = A.Comm(A.Comm.HOST, None, rank, world)
        most = max(sum(s) for s in sizes)
        cap = most + 4  # rows + the per-image table's rows
        log = {}
        # ---- an exchange that fits: headers, rows, per-image tables of every rank ----
        g = comm.gather_begin_image_rows(local, per, cap)
        with pytest.raises(A.AkazeError):  # not delivered yet
            g.finish()
        g.exchange_over()
        blocks, block_rows, counts, images = g.finish()
        assert block_rows == cap + 1 and counts == [sum(s) for s in sizes] and images == [len(s) for s in sizes]
        raw = np.ctypeslib.as_array(C.cast(blocks, C.POINTER(C.c_uint8)), shape=(world, block_rows, 64)).copy()
        log["raw"] = raw
        for r in range(world):
            assert g.image_rows(r) == list(sizes[r])
            hdr = raw[r, 0].view(np.uint64)
            table_rows = (len(sizes[r]) + 7) // 8
            assert list(hdr[:6]) == [sum(sizes[r]), len(sizes[r]), cap, 1, 0, table_rows]  # rows, images, capacity, sequence, overflow, table rows
            tab = raw[r, 1 + sum(sizes[r]):1 + sum(sizes[r]) + table_rows].view(np.uint64).ravel()
            assert list(tab[:len(sizes[r])]) == list(sizes[r]) and not tab[len(sizes[r]):].any()
        assert np.array_equal(raw[rank, 1:1 + len(local)], local)
        # ---- the all-pairs plan over this gather: who matches what ----
        plan = g.plan_all_pairs()
        n_img = sum(len(s) for s in sizes)
        assert plan.n_images == n_img and plan.first_owned == sum(len(s) for s in sizes[:rank]) and plan.n_owned == len(per)
        flat = [n for s in sizes for n in s]
        owners = [r for r in range(world) for _ in sizes[r]]
        assert [plan.image_rows(j) for j in range(n_img)] == list(zip(flat, owners))
        mine = {}
        for k in range(plan.n_owned):
            q = plan.first_owned + k
            mine[q] = plan.lead_sets(q)
            assert all(A.pairs_lead(q, j) == q and plan.holder(q, j) == rank for j in mine[q])
        lists, matches, dists = plan.totals()
        assert lists == 2 * sum(len(v) for v in mine.values()) and matches == 0
        assert dists == sum(flat[q] * flat[j] for q, v in mine.items() for j in v)
        if n_img >= 2:  # a plan holds no lists (and a pair another rank leads is not this rank's to read either way)
            with pytest.raises(A.AkazeError):
                plan.matches(0, 1)
        log["mine"] = mine
        log["holder"] = {(a, b): plan.holder(a, b) for a in range(n_img) for b in range(n_img) if a != b}
        plan.free()
        g.free()
        # ---- a shard that does not fit the agreed capacity: the rank still takes part (header only), EVERY rank gets
        # AKZ_ERR_BUFFER with the counts, and the communicator stays usable ----
        small = most - 1
        g2 = comm.gather_begin_image_rows(local, per, small)
        g2.exchange_over()
        with pytest.raises(A.AkazeError) as e:
            g2.finish()
        assert e.value.status == -7
        g2.free()
        # ---- raw rows (one image, no table) ----
        g4 = comm.gather_begin_image_rows(local, [len(local)], cap)
        g4.exchange_over()
        _, _, counts4, images4 = g4.finish()
        assert counts4 == [sum(s) for s in sizes] and images4 == [1] * world
        g4.free()
        # device-only calls are refused, not attempted
        with pytest.raises(A.AkazeError) as e:
            comm.gather_begin([], cap)
        assert e.value.status == -6
        with open(os.path.join(out_dir, f"capi_{rank}.pkl"), "wb") as f:
            pickle.dump(log, f)
    finally:
        if comm is not None:
            comm.close()
        dist.destroy_process_group()


@pytest.mark.parametrize("sizes", [((5, 9, 2), (4,)), ((0, 7), (3, 0, 6, 1, 1, 2, 5, 4, 3)), ((), (8, 2))])
def test_product_wire_format_and_all_pairs_plan_gloo(tmp_path, sizes):
    """akz_comm.cpp's exchange with two CPU ranks: the blocks every rank ends up with are identical, carry the documented
    header / rows / table, and the all-pairs plans of the ranks cover every unordered image pair exactly once."""
    _run_capi(tmp_path, sizes)


def _run_capi(tmp_path, sizes):
    import pickle
    import torch.multiprocessing as mp
    world = len(sizes)
    mp.spawn(_capi_worker, args=(world, _free_port(), sizes, str(tmp_path)), nprocs=world, join=True)
    logs = [pickle.load(open(tmp_path / f"capi_{r}.pkl", "rb")) for r in range(world)]
    for lg in logs[1:]:
        assert np.array_equal(logs[0]["raw"], lg["raw"])       # the same gathered blocks on every rank
        assert logs[0]["holder"] == lg["holder"]               # every rank names the same holder for every pair
    n_img = sum(len(s) for s in sizes)
    covered = [frozenset((q, j)) for lg in logs for q, v in lg["mine"].items() for j in v]
    assert len(covered) == len(set(covered)) == n_img * (n_img - 1) // 2   # every unordered pair, exactly once
    return logs


def test_product_exchange_with_the_eight_ranks_of_a_node_gloo(tmp_path):
    """BASELINE configs[4]'s job shape -- eight ranks, two 4K frames each -- through akz_comm.cpp's exchange and all-pairs plan on
    the CPU: eight processes, one rendezvous, one fixed-size all-gather per exchange; all 120 unordered pairs of the 16
    images assigned exactly once, 14-16 per rank (the lead rule balances without a further exchange)."""
    sizes = tuple((11 + r, 7 + 2 * r) for r in range(8))
    logs = _run_capi(tmp_path, sizes)
    per_rank = [sum(len(v) for v in lg["mine"].values()) for lg in logs]
    assert sum(per_rank) == 120 and min(per_rank) >= 14 and max(per_rank) <= 16, per_rank


@pytest.mark.parametrize("counts", [(5, 9), (0, 4), (7, 0)])
def test_gather_descriptor_rows_gloo(tmp_path, counts):
    import torch.multiprocessing as mp
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, counts, str(tmp_path)), nprocs=world, join=True)
    locals_ = [np.load(tmp_path / f"local_{r}.npy") for r in range(world)]
    expect = np.concatenate(locals_, axis=0)
    for r in range(world):
        rows = np.load(tmp_path / f"rows_{r}.npy")
        assert rows.shape == (sum(counts), 64)
        assert np.array_equal(rows, expect)          # rank order, unpadded, identical on every rank
        assert list(np.load(tmp_path / f"cnts_{r}.npy")) == list(counts)
    owned = sorted(np.concatenate([np.load(tmp_path / f"owned_{r}.npy") for r in range(world)]).tolist())
    assert owned == list(range(11))


def test_shard_frames_one_image_per_gpu_slot():
    sys.path.insert(0, os.path.join(ROOT, "akaze-rust_amd", "python"))
    import akaze_amd as A
    for world in (1, 2, 4, 8):
        shards = [A.shard_frames(256, r, world) for r in range(world)]
        assert sorted(sum(shards, [])) == list(range(256))
        assert all(len(s) == 256 // world for s in shards)
        assert all(i % world == r for r, s in enumerate(shards) for i in s)


def _np_match(d0, d1, ratio=0.86):
    """Plain numpy Hamming 1-NN / 2-NN with the reference's rule (feature_matching.rs:37-50, :113-123), for CPU tensors."""
    a, b = d0.numpy(), d1.numpy()
    out = []
    if len(b) == 0:
        return out
    for i in range(len(a)):
        dist = np.unpackbits(a[i][None, :] ^ b, axis=1).sum(axis=1).astype(np.int64)
        order = np.argsort(dist, kind="stable")
        best = int(order[0])
        second = int(dist[order[1]]) if len(b) > 1 else 10 ** 9
        if dist[best] < 10000 and float(dist[best]) ** 2 < (ratio ** 2) * float(second) ** 2:
            out.append((i, best, int(dist[best])))
    return out


def _pairs_worker(rank, world, port, sizes, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "akaze-rust_amd", "python"))
    import pickle
    import torch
    import torch.distributed as dist
    import akaze_amd as A
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(7)
        everything = [torch.from_numpy(rng.integers(0, 256, (n, 64), dtype=np.uint8)) for r in range(world) for n in sizes[r]]
        first = sum(len(sizes[r]) for r in range(rank))
        local = everything[first:first + len(sizes[rank])]
        sets, owners = A.gather_descriptor_sets(local)
        assert len(sets) == len(everything) and all(torch.equal(a, b) for a, b in zip(sets, everything))
        assert owners == [r for r in range(world) for _ in sizes[r]]
        res = A.all_pairs_match(local, _np_match)

        def np_match_sets(q, cat, rows):  # the both-direction multi-set form: one call per lead image against the sets it leads
            offs = np.concatenate([[0], np.cumsum(rows)])
            return ([_np_match(q, cat[offs[k]:offs[k + 1]]) for k in range(len(rows))],
                    [_np_match(cat[offs[k]:offs[k + 1]], q) for k in range(len(rows))])

        assert A.all_pairs_match(local, None, match_sets_fn=np_match_sets) == res
        with open(os.path.join(out_dir, f"pairs_{rank}.pkl"), "wb") as f:
            pickle.dump(res, f)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("sizes", [((6, 9), (4,)), ((5,), (0, 7, 3)), ((), (8, 2))])
def test_all_pairs_match_gloo(tmp_path, sizes):
    """BASELINE configs[4]'s cross-GPU all-pairs match on two CPU ranks: every UNORDERED image pair is matched exactly once,
    in both directions, by the rank that owns the pair's lead image (akz_match_all_pairs' rule), with the results a single
    process gets for both ordered pairs."""
    import pickle
    import torch
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_pairs_worker, args=(world, _free_port(), sizes, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(7)
    everything = [torch.from_numpy(rng.integers(0, 256, (n, 64), dtype=np.uint8)) for r in range(world) for n in sizes[r]]
    owners = [r for r in range(world) for _ in sizes[r]]
    seen = {}
    for r in range(world):
        res = pickle.load(open(tmp_path / f"pairs_{r}.pkl", "rb"))
        sys.path.insert(0, os.path.join(ROOT, "akaze-rust_amd", "python"))
        import akaze_amd as A
        assert all(owners[A.pairs_lead(i, j)] == r for (i, j) in res)
        assert all((j, i) in res for (i, j) in res)  # both directions of a pair live on one rank
        assert not (set(res) & set(seen))
        seen.update(res)
    n = len(everything)
    assert set(seen) == {(i, j) for i in range(n) for j in range(n) if i != j}
    for (i, j), m in seen.items():
        assert m == _np_match(everything[i], everything[j]), (i, j)



def _bench(*args, env=None, timeout=240):
    import subprocess
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=e, capture_output=True, text=True,
                          timeout=timeout)


def test_bench_gpus_n_starts_n_ranks_itself():
    """`bench.py --gpus 2` without torchrun must run TWO ranks (stub context + gloo here: no GPU in this container),
    forward rank 0's JSON line and report n_gpus 2 with one figure per rank."""
    import json
    p = _bench("--gpus", "2", "--steps", "3", "--warmup", "1", "--frames", "4", "--stub")
    assert p.returncode == 0, p.stderr
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["stub"] is True and out["steps"] == 3
    assert len(out["config"]["per_rank_ms_per_step"]) == 2
    assert out["config"]["exchange_ms_per_step"] > 0


def test_bench_launcher_with_the_eight_ranks_of_a_node():
    """`bench.py --gpus 8` as the driver starts it on an 8-GPU node, rehearsed without GPUs (stub contexts): eight rank
    processes, an eight-way gloo rendezvous, every rank's figure on the line."""
    import json
    p = _bench("--gpus", "8", "--steps", "2", "--warmup", "1", "--frames", "2", "--stub", timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 8 and out["stub"] is True and len(out["config"]["per_rank_ms_per_step"]) == 8


def test_bench_refuses_world_size_mismatch():
    """--gpus N with a different WORLD_SIZE is an error, never a silent run at another size."""
    p = _bench("--gpus", "2", "--stub", env={"WORLD_SIZE": "4", "RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE=4" in p.stderr
    p = _bench("--gpus", "1", "--stub", env={"WORLD_SIZE": "2", "RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr


def test_bench_launcher_fails_when_a_rank_fails():
    """A rank that dies takes the launch down with a non-zero status (no hang, no JSON line)."""
    p = _bench("--gpus", "2", "--stub", "--frames", "0")  # zero frames: every rank raises
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_bench_rank_gives_up_when_a_peer_never_arrives():
    """A rank whose peers never reach the rendezvous (a crashed process, a GPU that did not initialise) ends with an error
    after AKZ_BENCH_RENDEZVOUS_S seconds instead of waiting for the default half hour -- on a real node that is the
    difference between a failed run and a hung one."""
    import socket
    import time
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    t0 = time.monotonic()
    p = _bench("--gpus", "2", "--stub", "--steps", "2", "--warmup", "1",
               env={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "2", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                    "AKZ_BENCH_RENDEZVOUS_S": "8"}, timeout=120)
    assert p.returncode != 0 and time.monotonic() - t0 < 90
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
