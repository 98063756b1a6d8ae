"""One rank of the cross-GPU all-pairs rehearsal (tests/test_gpu_gather.py::test_all_pairs_ranks_sharing_one_gpu): NOT a
test module.  WORLD ranks run on device 0; frame f of the job lives on rank f mod WORLD; the descriptor rows travel in
the C ABI's wire format over gloo (akz_comm_create_external -> akz_gather_begin -> akz_gather_blocks -> all-gather ->
akz_gather_deliver), then akz_match_all_pairs.  Every list this rank holds is compared with akz_descriptor_match of the
pair (ops::feature_matching::descriptor_match, feature_matching.rs:23-94); the held pairs go to OUT_DIR/held_<rank>.json.

    python tests/c5_rank.py RANK WORLD PORT OUT_DIR FRAMES_PER_RANK"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "akaze-rust_amd", "python"))


def main():
    rank, world, port, out_dir, per = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], int(sys.argv[5])
    import datetime
    import numpy as np
    import torch
    import torch.distributed as dist
    import akaze_amd as A
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", port
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    try:
        torch.cuda.set_device(0)
        st = torch.cuda.Stream()
        ctx = A.Context(0, st.cuda_stream)
        cfg = A.Config(num_sublevels=5, max_octave_evolution=5)
        sizes = [(480, 360), (400, 304), (352, 288), (512, 384)]
        n_job = world * per
        # every frame of the job, extracted here too (small frames): the expected lists need both sets' descriptors
        desc = []
        for f in range(n_job):
            w, h = sizes[f % len(sizes)]
            r = ctx.extract_features(torch.from_numpy(A.synth_frame(w, h, f)[None]).cuda(), cfg, keep_all_planes=False)
            desc.append(r.descriptors(0))
            r.close()
        mine = A.shard_frames(n_job, rank, world)  # frame f -> rank f mod world
        ress = []
        for f in mine:
            w, h = sizes[f % len(sizes)]
            ress.append(ctx.extract_begin(torch.from_numpy(A.synth_frame(w, h, f)[None]).cuda(), cfg, keep_all_planes=False,
                                          host_descriptors=False).finish())
        rows = sum(r.counts(0)[1] for r in ress)
        cap = torch.tensor([rows], dtype=torch.int64)
        dist.all_reduce(cap, op=dist.ReduceOp.MAX)
        comm = A.Comm(0, None, rank, world)
        held_all = []
        for step in range(2):  # the second step reuses the pooled gather and pairs objects
            g = comm.gather_begin(ress, int(cap.item()) + 8)
            g.exchange_over()
            pr = g.match_all_pairs(ctx)
            # image numbering of the job is rank-major: global image i of rank r is frame mine_r[i]
            order = [f for r in range(world) for f in A.shard_frames(n_job, r, world)]
            assert pr.n_images == n_job and pr.first_owned == sum(len(A.shard_frames(n_job, r, world)) for r in range(rank))
            held = []
            for a in range(n_job):
                for b in range(n_job):
                    if a == b:
                        continue
                    if pr.holder(a, b) != rank:
                        continue
                    got = pr.matches(a, b)
                    exp = ctx.descriptor_match(desc[order[a]], desc[order[b]], 10000, 0.86)
                    assert np.array_equal(got, exp), (rank, step, a, b, len(got), len(exp))
                    held.append([order[a], order[b], int(len(got))])
            lists, matches, dists = pr.totals()
            assert lists == len(held) and matches == sum(h[2] for h in held)
            held_all = held
            pr.free()
            g.free()
        for r in ress:
            r.close()
        comm.close()
        json.dump({"rank": rank, "held": held_all, "frames": list(map(int, mine))}, open(os.path.join(out_dir, f"held_{rank}.json"), "w"))
        dist.barrier()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
