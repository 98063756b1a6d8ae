"""world_size-2 tests of the multi-GPU logic on CPU (gloo): per-image sharding and the descriptor-row
all-gather that precedes a cross-image match.  The same code runs over RCCL on the GPUs."""
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
