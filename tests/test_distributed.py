"""world_size-2 tests of the multi-GPU logic on CPU (gloo): per-image sharding and the descriptor-row
all-gather that precedes a cross-image match, as the Python helpers `akaze_amd.gather_descriptor_rows` / `all_pairs_match`
(torch.distributed) do them.  The PRODUCT's exchange is `akz_comm.cpp` behind the C ABI (its own RCCL communicator, or a
caller-carried transport): that one needs a device and is driven by tests/test_gpu_gather.py -- two and three real ranks on
one GPU with the C ABI's blocks over gloo -- and by tests/test_comm_faults.py against a stub librccl."""
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
