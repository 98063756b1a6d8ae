#!/usr/bin/env python3
"""Headline benchmark: Mpix/s through extract_features (4 octaves x 4 sublevels) on synthetic
1920x1080 frames, one process per GPU, frames sharded one-per-GPU-slot across ranks.

    python bench.py --gpus N --steps K --warmup W [--frames F] [--width 1920 --height 1080]

A "step" is one extract_features pass over this rank's shard of F frames that are already
resident in HBM (uploaded before the timed region).  For N > 1 the step also runs the path's
one exchange: an RCCL all-gather of the shard's descriptor rows (what a following brute-force
match needs); extraction itself has no collective.  Rank 0 prints ONE JSON line.

Extra objects on that line:
  roofline      the FED diffusion kernel (the dominant kernel): algorithmic bytes (12 B per
                pixel-step) / HIP-event time of the FED launches inside the timed
                steps, against 8 TB/s HBM peak.  roofline.fed_4k is the same kernel measured on
                3840x2160 planes (the north-star's quoted point), outside the timed region.
  stage_roofline  algorithmic HBM bytes of every GPU stage / its time in one un-pipelined, fully profiled step.
  single_frame  BASELINE configs[1] taken literally — one 1920x1080 frame per extract_features call: latency of a
                lone call, the rate of a stream of such calls on one context, and on four contexts driven by four
                host threads (untimed extra leg; the headline workload is the
                32-frames-per-GPU batch of configs[3], which is what the 1/2/4/8-GPU metric shards).
  match         the brute-force Hamming matcher on two descriptor sets (untimed extra leg): pairs/s and fraction of the
                dense int8 MFMA rate (1024 operations per descriptor pair).
  cpu_baseline  the CPU oracle (C++ restatement of the reference's CPU path; the Rust reference
                cannot be built in this image) timed on the host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(ROOT, "akaze-rust_amd", "python")]

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
FED_BYTES_PER_PX_STEP = 12.0   # read Lt + read Lflow + write Lt'  
MFMA_I8_PEAK_OPS = 5.0e15      # dense int8 MFMA rate: 2 x the 2.5 PFLOP/s bf16 figure (MI355X_MICROARCH.md)


def pmc_traffic():
    """HBM bytes per FED launch from the committed rocprofv3 PMC summary (profiles/), if present."""
    p = os.path.join(ROOT, "profiles", "fed_pmc_traffic.json")
    if os.path.exists(p):
        try:
            return json.load(open(p)).get("hbm_bytes_per_launch")
        except Exception:
            return None
    return None


def effective_cpus():
    """Host cores this process may actually use: the scheduler affinity mask, cut down by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except Exception:
            continue
    return max(1, n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=32, help="frames per GPU per step (C4: 256 frames / 8 GPUs)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--lean", action="store_true", help="do not materialise Lxx/Lyy/Lxy/Lstep")
    ap.add_argument("--sublevels", type=int, default=4)
    ap.add_argument("--octaves", type=int, default=4)
    ap.add_argument("--det-mode", type=int, default=2, help="detector kernels: 2 auto, 1 streaming pair, 3 fused streaming, 4 one tiled kernel, 0 tiled pair")
    ap.add_argument("--prep-mode", type=int, default=2, help="level-preparation kernel: 2 auto, 1 streaming, 0 LDS-tiled")
    ap.add_argument("--det-overlap", type=int, default=0, choices=[0, 1, 2],
                    help="detector launches on a side stream: 1 = every level as soon as its Lsmooth exists (the FED spans of "
                         "the roofline then include the time shared with the detector kernels), 2 = the fine octaves' "
                         "detectors next to the coarse octaves' latency-bound chain only")
    ap.add_argument("--depth", type=int, default=1, choices=[1, 2],
                    help="batches begun ahead of the one being finished (the context holds at most three in flight)")
    ap.add_argument("--threshold", type=float, default=None,
                    help="detector_threshold override (tuning runs: a huge value removes every extremum candidate)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fed4k", action="store_true")
    ap.add_argument("--no-single", action="store_true", help="skip the single-frame-per-call leg")
    ap.add_argument("--no-match", action="store_true", help="skip the matcher leg")
    ap.add_argument("--parts", type=int, default=1,
                    help="batches per step; batches are software-pipelined on ONE stream (begin(batch j+1) is "
                         "enqueued before finish(batch j)), so the host keypoint phase of a batch runs under the "
                         "kernels of the next")
    ap.add_argument("--sync", action="store_true",
                    help="finish every batch right after beginning it (no software pipelining): clean per-stage times")
    ap.add_argument("--no-profile", action="store_true", help="do not record stage events in the timed region")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise RCCL and run the descriptor gather even with one rank (self-test)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import numpy as np
    import torch
    import torch.distributed as dist
    import akaze_amd as A

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    cfg = A.Config(num_sublevels=args.sublevels, max_octave_evolution=args.octaves)  # default 4 x 4, 486-bit M-LDB
    if args.threshold is not None:
        cfg.detector_threshold = args.threshold
    W, H, F = args.width, args.height, args.frames

    # this rank's shard of the world*F frames of a step: image i -> GPU i mod world (weak scaling)
    frames = np.stack([A.synth_frame(W, H, i) for i in A.shard_frames(world * F, rank, world)])
    d_frames = torch.from_numpy(frames).to(dev)
    torch.cuda.synchronize()

    # one context = one host thread + one HIP stream + an auxiliary stream.  The stream is a dedicated
    # non-blocking one, never the legacy null stream: that one synchronises implicitly with every
    # blocking stream in the process (RCCL has some) and would serialise the pipeline.
    main = torch.cuda.Stream(dev)
    torch.cuda.set_stream(main)
    ctx = A.Context(local_rank, main.cuda_stream)
    ctx.set_detector_mode(args.det_mode)
    ctx.set_prep_mode(args.prep_mode)
    ctx.set_detector_overlap(args.det_overlap)
    NP = max(1, min(args.parts, F))
    cut = [(F * i) // NP for i in range(NP + 1)]
    batches = [d_frames[cut[i]:cut[i + 1]] for i in range(NP)]

    side = torch.cuda.Stream(dev)  # collectives run here so that they never wait for the extraction stream

    gather_ms = [0.0]
    host_ms = {"begin": 0.0, "finish": 0.0, "calls": 0}  # host time inside extract_begin / extract_finish
    gather_cap = [0]  # fixed row capacity of the padded all-gather, set from the first step

    def gather_descriptors(results):
        """The path's exchange step: counts, then padded 64-byte rows over RCCL (a few MB, latency-bound).
        Fixed capacity => no host synchronisation: both collectives are only enqueued on the side stream."""
        rows = sum(res.counts(i)[1] for res in results for i in range(res.num_images))
        if gather_cap[0] == 0:
            # every rank must pad to the SAME capacity: agree on it once (first warm-up step, one host sync); the
            # frames of a shard are the same every step, so the row counts do not change afterwards
            most = torch.tensor([rows], dtype=torch.int64, device=dev)
            dist.all_reduce(most, op=dist.ReduceOp.MAX)
            most = int(most.item())
            gather_cap[0] = 1 << max(10, (most + most // 2).bit_length())
        if rows > gather_cap[0]:
            raise SystemExit(f"rank {rank}: {rows} descriptor rows exceed the agreed gather capacity {gather_cap[0]}")
        with torch.cuda.stream(side):
            local = torch.empty((rows, 64), dtype=torch.uint8, device=dev)
            side.synchronize()  # the allocation above is the only thing the aux-stream copy must wait for
            o = 0
            for res in results:
                o += res.copy_device_descriptors(local[o:])  # complete on return (aux stream)
            return A.gather_descriptor_rows(local, cap_rows=gather_cap[0])

    def run_steps(k_steps):
        """k_steps passes over the shard as one software-pipelined stream of k_steps*NP batches:
        begin(batch j+1) is enqueued before finish(batch j), also across step boundaries, so the
        candidate fetch + host keypoint logic of a batch run under the kernels of the next one.
        Every step's results are complete (and, with N > 1, gathered) before run_steps returns."""
        nk, inflight, done = 0, [], []

        def retire(res):
            nonlocal nk, done
            done.append(res)
            if len(done) == NP:  # a whole step has finished
                nk = sum(r.counts(i)[1] for r in done for i in range(r.num_images))
                if use_dist:
                    tg = time.perf_counter()
                    gather_descriptors(done)
                    gather_ms[0] += (time.perf_counter() - tg) * 1e3
                for r in done:
                    r.close()
                done = []

        for _ in range(k_steps):
            for bt in batches:
                tb = time.perf_counter()
                job = ctx.extract_begin(bt, cfg, keep_all_planes=not args.lean)
                host_ms["begin"] += (time.perf_counter() - tb) * 1e3
                host_ms["calls"] += 1
                if args.sync:
                    retire(job.finish())
                    continue
                inflight.append(job)
                if len(inflight) > args.depth:
                    tf = time.perf_counter()
                    res = inflight.pop(0).finish()
                    host_ms["finish"] += (time.perf_counter() - tf) * 1e3
                    retire(res)
        while inflight:
            retire(inflight.pop(0).finish())
        return nk

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    if args.warmup:
        run_steps(args.warmup)
    # one extra untimed step with full stage profiling (informational stage_ms_per_step); the timed region only
    # records the FED spans needed for the roofline (light mode: ~30 instead of ~190 HIP events per batch)
    ctx.set_profiling(0 if args.no_profile else 1)
    ctx.get_profile(reset=True)
    run_steps(1)
    warm_prof = ctx.get_profile(reset=True)
    ctx.set_profiling(0 if args.no_profile else 2)
    barrier()
    t0 = time.perf_counter()
    gather_ms[0] = 0.0
    host_ms.update(begin=0.0, finish=0.0, calls=0)
    nk = run_steps(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    prof = ctx.get_profile(reset=True)
    ctx.set_profiling(False)
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    total_px = float(W) * H * F * world * args.steps
    value = total_px / elapsed / 1e6

    # ---- roofline of the dominant kernel (FED step) inside the timed region -------------------
    fed_s = prof["fed"] / 1e3
    fed_bytes = FED_BYTES_PER_PX_STEP * prof["fed_px_steps"]
    achieved = fed_bytes / fed_s / 1e9 if fed_s > 0 else 0.0
    roofline = {
        "bound": "hbm", "kernel": A.lib().akz_fed_kernel_name().decode(),
        "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4),
        "traffic": pmc_traffic(),
        "launches": prof["fed_launches"],
        "avg_launch_us": round(prof["fed"] * 1e3 / max(1, prof["fed_launches"]), 2),
        "algorithmic_bytes_per_launch": round(fed_bytes / max(1, prof["fed_launches"])),
        "note": "all FED launches of the timed steps (levels 1..15, 1920x1080 down to 240x135); time = sum of "
                "HIP-event spans around each level's FED launches on the launching stream",
    }

    # ---- the same kernel on 3840x2160 planes (north-star target point), untimed leg -----------
    if rank == 0 and not args.no_fed4k:
        w4, h4, nst = 3840, 2160, 40
        lt = torch.rand((h4, w4), dtype=torch.float32, device=dev)
        lf = torch.rand((h4, w4), dtype=torch.float32, device=dev)
        taus = np.full(nst, 0.2)
        ctx.fed_steps(lt, lf, taus)  # warm
        ctx.set_profiling(True)
        ctx.get_profile(reset=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ctx.fed_steps(lt, lf, taus)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        p4 = ctx.get_profile(reset=True)
        ctx.set_profiling(False)
        launches = max(1, p4["fed_launches"])
        per = ms / launches * 1e3
        gbs = FED_BYTES_PER_PX_STEP * w4 * h4 * nst / (ms * 1e-3) / 1e9
        roofline["fed_4k"] = {"avg_launch_us": round(per, 2), "achieved": round(gbs, 1),
                              "frac": round(gbs / HBM_PEAK_GBS, 4), "steps": nst, "launches": launches,
                              "algorithmic_bytes_per_launch": round(FED_BYTES_PER_PX_STEP * w4 * h4 * nst / launches)}
        del lt, lf

    # ---- the matcher (BASELINE configs[2] / [4]: Hamming match of two descriptor sets), untimed leg, rank 0 ------
    match_leg = None
    if rank == 0 and world == 1 and not args.no_match:  # N = 1 information; the scaling runs only need `value`
        g = torch.Generator(device=dev).manual_seed(7)
        legs = []
        for n_m in (11264, 65536):  # a 4K frame's keypoint count; a gathered multi-frame set
            da = torch.randint(0, 256, (n_m, 64), dtype=torch.uint8, device=dev, generator=g)
            db = torch.randint(0, 256, (n_m, 64), dtype=torch.uint8, device=dev, generator=g)
            da[:, 61:] = 0
            db[:, 61:] = 0
            ctx.descriptor_match_device(da, db)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps_m = 10
            e0.record()
            for _ in range(reps_m):
                ctx.descriptor_match_device(da, db)
            e1.record()
            torch.cuda.synchronize()
            ms_m = e0.elapsed_time(e1) / reps_m
            pairs = float(n_m) * n_m
            legs.append({"n0": n_m, "n1": n_m, "ms": round(ms_m, 3), "Tpairs_per_s": round(pairs / ms_m / 1e9, 3),
                         "mfma_frac": round(pairs * 2 * 512 / (ms_m * 1e-3) / MFMA_I8_PEAK_OPS, 3)})
        match_leg = {"kernel": "k_match_mfma (+ unpack, merge, compaction)", "bound": "mfma",
                     "peak": "%.1f POP/s int8 dense (2 x the 2.5 PFLOP/s bf16 MFMA rate)" % (MFMA_I8_PEAK_OPS / 1e15),
                     "ops_per_pair": 1024, "sets": legs}

    # ---- BASELINE configs[1] taken literally: ONE frame per extract call (untimed extra leg, rank 0) -----------
    single = None
    if rank == 0 and world == 1 and not args.no_single:
        ctx.set_profiling(0)
        one = d_frames[:1]
        for _ in range(10):
            ctx.extract_begin(one, cfg, keep_all_planes=not args.lean).finish().close()
        torch.cuda.synchronize()
        reps = 100
        t1 = time.perf_counter()
        for _ in range(reps):                      # latency: begin + finish, nothing in flight
            ctx.extract_begin(one, cfg, keep_all_planes=not args.lean).finish().close()
        lat = (time.perf_counter() - t1) / reps
        t1 = time.perf_counter()
        prev_job = None
        for _ in range(reps):                      # stream of single frames, the next one begun before this one is finished
            job = ctx.extract_begin(one, cfg, keep_all_planes=not args.lean)
            if prev_job is not None:
                prev_job.finish().close()
            prev_job = job
        prev_job.finish().close()
        thr = (time.perf_counter() - t1) / reps
        single = {"workload": f"one {W}x{H} frame per extract_features call (BASELINE configs[1])",
                  "latency_ms": round(lat * 1e3, 3), "stream_ms_per_frame": round(thr * 1e3, 3),
                  "stream_Mpix_s": round(W * H / thr / 1e6, 1)}
        # the same calls from K host threads with one context + stream each: a lone 1080p chain is launch-latency
        # bound and leaves most of the chip idle, concurrent chains fill it (ctypes releases the GIL in the calls)
        import threading
        K, reps_k = 4, 60
        wall = [0.0] * K
        bar = threading.Barrier(K)

        def single_worker(k):
            st_k = torch.cuda.Stream(dev)
            ctx_k = A.Context(local_rank, st_k.cuda_stream)
            fr = d_frames[k % F: k % F + 1]
            for _ in range(5):
                ctx_k.extract_begin(fr, cfg, keep_all_planes=not args.lean).finish().close()
            bar.wait()
            t_k = time.perf_counter()
            prev = None
            for _ in range(reps_k):
                job = ctx_k.extract_begin(fr, cfg, keep_all_planes=not args.lean)
                if prev is not None:
                    prev.finish().close()
                prev = job
            prev.finish().close()
            wall[k] = time.perf_counter() - t_k
            bar.wait()
            ctx_k.close()

        threads = [threading.Thread(target=single_worker, args=(k,)) for k in range(K)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        per = max(wall) / (K * reps_k)
        single["multi_context"] = {"contexts": K, "ms_per_frame": round(per * 1e3, 3),
                                   "Mpix_s": round(W * H / per / 1e6, 1)}

    # ---- CPU baseline: the oracle on the host cores, bounded sample, rank 0 at N=1 only -------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import akaze_ref as R
        cores = effective_cpus()  # affinity mask and cgroup quota, not the machine's core count
        n_sample = 32  # ~10 s of host work on the GPU box
        t1 = time.perf_counter()
        kp_ref = 0
        for i in range(n_sample):
            r = R.extract(frames[i % F], threads=cores)
            kp_ref += r.num_keypoints
            r.close()
        dt = time.perf_counter() - t1
        cpu = {"value": round(W * H * n_sample / dt / 1e6, 3), "unit": "Mpix/s", "cores": cores, "kind": "port",
               "sample": f"{n_sample} of the same {W}x{H} frames through the C++ restatement of the reference CPU "
                         f"path (scale space single-threaded, detector derivatives on {cores} threads as "
                         "detector_response.rs:16-29)",
               "seconds": round(dt, 2)}

    if rank == 0:
        stage_ms = {k: round(warm_prof[k], 3) for k in A.STAGES}
        stage_ms["note"] = "one untimed step with full stage profiling, un-pipelined; timed steps record FED spans only"
        # algorithmic HBM bytes of each GPU stage of one step (SURVEY.md 8(d) model: every stage input read once, every
        # kept plane written once) against that stage's time in the profiled step: where the path stands kernel by kernel
        lv = A.plan_levels(W, H, cfg)
        px = [l["w"] * l["h"] * F for l in lv]
        half = [i > 0 and lv[i]["octave"] > lv[i - 1]["octave"] for i in range(len(lv))]
        stage_bytes = {
            "blur0": px[0] * (1 + 4),                                   # u8 in, f32 Lt0 out
            "contrast": px[0] * 4 * 2,                                  # two passes over Lt0
            "prep": sum((16 + 12 if half[i] else 4 + 8) * px[i] for i in range(1, len(lv))),  # [2x2 mean: 4 px in, Lt out] Lsmooth, Lflow out
            "fed": sum(FED_BYTES_PER_PX_STEP * len(lv[i]["tau"]) * px[i] for i in range(1, len(lv))),
            "detector": sum((4 + 8 + 8 + (4 if args.lean else 16)) * p for p in px),  # Lsmooth in; Lx, Ly out and in again; Ldet (+Lxx, Lyy, Lxy) out
        }
        stage_roofline = {k: {"algorithmic_GB": round(b / 1e9, 3), "achieved_GBps": round(b / 1e9 / (warm_prof[k] * 1e-3), 1),
                              "frac": round(b / 1e9 / (warm_prof[k] * 1e-3) / HBM_PEAK_GBS, 3)}
                          for k, b in stage_bytes.items() if warm_prof[k] > 0}
        out = {
            "metric": f"Mpix/s through extract_features ({args.octaves} oct x {args.sublevels} sub)",
            "value": round(value, 2), "unit": "Mpix/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{W}x{H} synthetic 8-bit luma frames resident in HBM, Config::default() "
                                   f"({args.octaves} octaves x {args.sublevels} sublevels, 486-bit M-LDB), {F} frames per GPU per step "
                                   "(BASELINE configs[1] frame shape; configs[3] sharding: one image per GPU slot)",
                       "frames_per_gpu": F, "width": W, "height": H, "batches_per_step": NP,
                       "pipelining": "begin(batch j+1) before finish(batch j) on one stream, across steps",
                       "planes": "lean" if args.lean else "all 10 EvolutionStep planes materialised",
                       "exchange": "RCCL all-gather of descriptor rows" if use_dist else "none (1 GPU)",
                       "keypoints_per_step_rank0": nk,
                       "host_ms_in_gather_per_step": round(gather_ms[0] / max(1, args.steps), 3),
                       "host_ms_in_begin_per_batch": round(host_ms["begin"] / max(1, host_ms["calls"]), 3),
                       "host_ms_in_finish_per_batch": round(host_ms["finish"] / max(1, host_ms["calls"]), 3)},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "stage_ms_per_step": stage_ms,
            "stage_roofline": stage_roofline,
            "single_frame": single,
            "match": match_leg,
        }
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()  # rank 0 may still be in its untimed extra leg: all ranks leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
