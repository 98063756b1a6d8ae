#!/usr/bin/env python3
"""Headline benchmark: Mpix/s through extract_features (4 octaves x 4 sublevels) on synthetic
1920x1080 frames, one process per GPU, frames sharded one-per-GPU-slot across ranks.

    python bench.py --gpus N --steps K --warmup W [--frames F] [--width 1920 --height 1080]
    python bench.py --gpus N --workload c5     (BASELINE configs[4]: 4K frames, 5 x 5, exchange + cross-GPU all-pairs match)

The timed region of K steps is run --regions times (default 5) in one process; `value` / `ms_per_step` are the MEDIAN region
(config.regions holds all of them): the pool's boxes differ by +-3 % from run to run, which is as much as a round's gain.

With N > 1 and no WORLD_SIZE in the environment this process only LAUNCHES the N ranks (children with
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*; the parent never touches a GPU), forwards rank 0's JSON line and
exits non-zero if any rank does.  Under torchrun (WORLD_SIZE set) it is one rank; --gpus must equal WORLD_SIZE.

A "step" is one extract_features pass over this rank's shard of F frames that are already resident in HBM
(uploaded and synchronised before the timed region; the calls say so with AKZ_INPUT_READY, which lets a batch's first
two stages run under the kernels of the batch before -- --no-input-ready drops the flag).  For N > 1 the step also runs the path's one exchange: an RCCL all-gather
of the shard's descriptor rows (what a following brute-force match needs), enqueued without stalling the
extraction pipeline and retired one step later (the last one inside the timed region); extraction itself has
no collective.  Rank 0 prints ONE JSON line.

Extra objects on that line:
  roofline      the kernel with the largest share of the step (the detector march or the FED kernel): ALGORITHMIC
                bytes per launch / average launch duration measured with HIP events on the launching
                stream inside the timed steps, against the 8 TB/s HBM peak.  `traffic` = HBM bytes per launch from the
                committed rocprofv3 PMC summary of the same workload (null for any other workload), `traffic_frac` =
                that traffic / launch time / peak, i.e. real DRAM utilisation.  roofline_2 = the other of the two.
  fed_standalone  the FED kernel alone (nothing else on the chip): on one 3840x2160 plane (north-star point; its
                100 MB working set sits in the Infinity Cache) and on a 32 x 1920x1080 level (an HBM number).
  detector_standalone  the detector kernel alone on a 32 x 1920x1080 level, four plane sets in turn so that its input
                comes from HBM, not from the Infinity Cache; roofline.standalone_frac repeats its sigma_size 3 figure.
  stage_roofline  algorithmic HBM bytes of every GPU stage / its time in one un-pipelined, fully profiled step.
  self_check    untimed: frames of the timed batch extracted one by one give the same keypoints and descriptor bytes.
  single_frame  BASELINE configs[1] taken literally — one 1920x1080 frame per extract_features call.
  match         the brute-force Hamming matcher (untimed extra leg): pairs/s and fraction of the dense int8 MFMA rate.
  cpu_baseline  the CPU oracle (C++ restatement of the reference's CPU path; the Rust reference cannot be built in
                this image) timed on the host cores on a bounded sample.
"""
import argparse
import json
import os
import subprocess
import sys
import time

# (GPU_MAX_HW_QUEUES is left alone: with the runtime's default of four hardware queues -- one per pipe of the command
# processor -- the context's stream-placement probe gives each of its four busy streams a queue of its own; config.
# stream_placement of the line says what it found.)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(ROOT, "akaze-rust_amd", "python")]

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
FED_BYTES_PER_PX_STEP = 12.0   # read Lt + read Lflow + write Lt'  
MFMA_FP4_PEAK_OPS = 10.0e15    # dense FP4 MFMA rate (v_mfma_scale_f32_32x32x64_f8f6f4, e2m1 operands): 4 x the 2.5 PFLOP/s bf16 figure
                               # (MI355X_MICROARCH.md, Matrix cores)


def pmc_traffic(kernel, workload):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC summary — only if it was taken on this workload."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        d = json.load(open(p))
    except Exception:
        return None
    if d.get("workload") != workload:
        return None
    e = d.get("kernels", {}).get(kernel)
    return e.get("hbm_bytes_per_launch") if e else None


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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # (the first ~100 ms of a process run 2-3 % below the steady state -- clocks, buffer growth -- so the defaults warm
    # up for ten steps and time forty: 0.3 s in all)
    ap.add_argument("--steps", type=int, default=None, help="timed steps per region (default 40; 8 with --workload c5)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps first (default 10; 3 with --workload c5)")
    ap.add_argument("--regions", type=int, default=5,
                    help="timed regions of --steps steps each, back to back in one process; the line reports the median region")
    ap.add_argument("--workload", choices=["frames", "c5"], default="frames",
                    help="frames: BASELINE configs[1]/[3] (1080p frames, 4 x 4, descriptor gather with N > 1) -- the headline; "
                         "c5: BASELINE configs[4] -- 3840x2160 frames, 5 octaves x 5 sublevels, frame f on rank f mod N, descriptors "
                         "stay on the device, exchange + all-pairs match (akz_gather_begin -> akz_match_all_pairs -> totals read) "
                         "inside the timed step")
    ap.add_argument("--frames", type=int, default=None, help="frames per GPU per step (default 32 = C4: 256 frames / 8 GPUs; 8 with --workload c5 = C5: 64 frames / 8 GPUs)")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--lean", action="store_true", help="do not materialise Lxx/Lyy/Lxy/Lstep")
    ap.add_argument("--sublevels", type=int, default=None)
    ap.add_argument("--octaves", type=int, default=None)
    ap.add_argument("--det-mode", type=int, default=2, help="detector kernels: 2 auto, 5 column march, 4 one LDS-tiled kernel, 0 LDS-tiled pair")
    ap.add_argument("--prep-mode", type=int, default=2, help="level-preparation kernel: 2 auto, 1 streaming, 0 LDS-tiled")
    ap.add_argument("--eager", type=int, default=1, help="akz_ctx_set_eager_finish (the library's default: on): the finish half on the context's own thread")
    ap.add_argument("--select", type=int, default=-1, help="akz_debug_set_select: keypoint selection on the device (2), on the host from the device's neighbour lists (1), from the host's grids (0), automatic (-1)")
    ap.add_argument("--sched", type=str, default="", help="akz_debug_set_schedule pairs, e.g. 0=1,1=1,2=0")
    ap.add_argument("--depth", type=int, default=2, choices=[1, 2],
                    help="batches begun ahead of the one being finished (the context holds at most three in flight)")
    ap.add_argument("--threshold", type=float, default=None,
                    help="detector_threshold override (tuning runs: a huge value removes every extremum candidate)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fed4k", action="store_true", help="skip the stand-alone FED legs")
    ap.add_argument("--no-single", action="store_true", help="skip the single-frame-per-call leg")
    ap.add_argument("--no-match", action="store_true", help="skip the matcher leg")
    ap.add_argument("--no-self-check", action="store_true")
    ap.add_argument("--no-host-input", action="store_true",
                    help="skip the second timed region that takes the frames from pinned HOST memory (upload overlapped)")
    ap.add_argument("--parts", type=int, default=1,
                    help="batches per step; batches are software-pipelined on ONE stream (begin(batch j+1) is "
                         "enqueued before finish(batch j)), so the host keypoint phase of a batch runs under the "
                         "kernels of the next")
    ap.add_argument("--sync", action="store_true",
                    help="finish every batch right after beginning it (no software pipelining): clean per-stage times")
    ap.add_argument("--no-input-ready", action="store_true",
                    help="do not pass AKZ_INPUT_READY for the resident frames (the first stages of a batch then wait for the batch before)")
    ap.add_argument("--no-profile", action="store_true", help="do not record stage events in the timed region")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the descriptor exchange (RCCL) even with one rank (self-test of the N > 1 path)")
    ap.add_argument("--exchange", choices=["capi", "torch"], default="capi",
                    help="capi: akz_gather_begin/finish of the C ABI (own RCCL communicator, gloo only for rendezvous and "
                         "barriers); torch: torch.distributed all_gather on the nccl backend")
    ap.add_argument("--match-ctx", choices=["shared", "own"], default="shared",
                    help="--workload c5: the all-pairs launches on the extraction context's stream (shared) or on a context and "
                         "stream of their own (own: they run beside the next steps' extraction kernels)")
    ap.add_argument("--no-c5-leg", action="store_true",
                    help="N > 1: skip the extra leg that runs BASELINE configs[4] (4K frames, 5 x 5, exchange + all-pairs match) over "
                         "all ranks after the headline measurement")
    ap.add_argument("--c5-leg-frames", type=int, default=4, help="3840x2160 frames per rank in that leg")
    ap.add_argument("--c5-leg-timeout", type=int, default=180,
                    help="seconds after which the leg is given up: rank 0 prints the line without it and every rank exits 0")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal of the N > 1 path on a ONE-GPU box: every rank runs on device 0 (real extraction ranks, "
                         "real rendezvous, capacity agreement and pipelined retire); the descriptor rows travel D2H -> gloo "
                         "all-gather -> H2D, because RCCL refuses two ranks on one device.  The line says so.")
    ap.add_argument("--host-share", type=int, default=0,
                    help="run as ONE rank of a K-rank node: the context's host worker pool gets 1/K of the usable cores "
                         "(akz_ctx_set_host_threads)")
    ap.add_argument("--no-host-share-leg", action="store_true", help="skip the child run that measures --host-share 8")
    ap.add_argument("--sort", choices=["auto", "host", "device"], default="auto",
                    help="A/B: where the extrema candidates are put into scan order (akz_debug_set_host_sort)")
    ap.add_argument("--stub", action="store_true",
                    help="launcher self-test without a GPU: a stub context (sleeps, fake rows) and the gloo backend; the "
                         "line it prints says so and is not a measurement")
    args = ap.parse_args(argv)
    c5 = args.workload == "c5"
    for name, frames_default, c5_default in (("steps", 40, 8), ("warmup", 10, 3), ("frames", 32, 8), ("width", 1920, 3840),
                                             ("height", 1080, 2160), ("sublevels", 4, 5), ("octaves", 4, 5)):
        if getattr(args, name) is None:
            setattr(args, name, c5_default if c5 else frames_default)
    args.regions = max(1, args.regions)
    return args


# --------------------------------------------------------------------------------------------------
# host placement of a rank: BEFORE anything initialises HIP (the runtime's helper threads inherit the mask)
# --------------------------------------------------------------------------------------------------
def _parse_cpulist(txt):
    cpus = set()
    for part in txt.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus.update(range(int(a), int(b or a) + 1))
    return cpus


def gpu_numa_node(local_rank):
    """NUMA node of the local_rank-th GPU from the KFD topology in sysfs (no HIP call): the GPU nodes in node order are
    the runtime's device order unless *_VISIBLE_DEVICES reorders them, which is honoured for plain index lists."""
    base = "/sys/class/kfd/kfd/topology/nodes"
    gpus = []
    for name in sorted(os.listdir(base), key=lambda s: int(s) if s.isdigit() else 1 << 30):
        try:
            props = dict(ln.split(None, 1) for ln in open(os.path.join(base, name, "properties")).read().splitlines() if " " in ln)
        except Exception:
            continue
        if int(props.get("simd_count", "0")) > 0:
            gpus.append(props)
    order = list(range(len(gpus)))
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v and all(p.strip().isdigit() for p in v.split(",")):
            order = [order[int(p)] for p in v.split(",") if int(p) < len(order)]
    if local_rank >= len(order):
        return None
    p = gpus[order[local_rank]]
    loc, dom = int(p.get("location_id", "0")), int(p.get("domain", "0"))
    bdf = "%04x:%02x:%02x.%x" % (dom, (loc >> 8) & 0xff, (loc >> 3) & 0x1f, loc & 7)
    try:
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
    except Exception:
        return None
    return node if node >= 0 else None


def place_rank(args):
    """Affinity of this rank, set before the first GPU call: --host-share K keeps 1/K of the allowed cores (and sets
    LOCAL_WORLD_SIZE=K); a rank of a multi-rank job takes its share of the cores of ITS GPU's NUMA node."""
    info = {}
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except Exception:
        return info
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    usable = effective_cpus()  # the mask cut down by the cgroup CPU quota: what the whole node's ranks share
    if args.host_share > 1:
        # 1/K of the usable cores as a THREAD budget for the context's host pool.  The mask itself is left alone: on a
        # shared box "the first usable/K CPUs" are as likely as not busy with somebody else's work (seen: 0.62 against
        # 0.87 for the same build), and a rank of a real node floats over its node's cores the same way.
        k = args.host_share
        info = {"host_share": k, "cpus": max(1, usable // k), "of": usable}
    elif local_world > 1 and not args.stub:
        node = None
        try:
            node = gpu_numa_node(0 if args.share_gpu else local_rank)
            if node is not None:
                cpus = sorted(_parse_cpulist(open(f"/sys/devices/system/node/node{node}/cpulist").read()) & set(allowed))
                # the ranks whose GPUs hang off this node split its cores
                peers = [r for r in range(local_world) if gpu_numa_node(0 if args.share_gpu else r) == node]
                if cpus and local_rank in peers:
                    per = max(1, len(cpus) // len(peers))
                    i = peers.index(local_rank)
                    keep = cpus[i * per:(i + 1) * per] or cpus
                    os.sched_setaffinity(0, keep)
                    info = {"numa_node": node, "cpus": max(1, min(len(keep), usable // local_world)), "mask": len(keep),
                            "ranks_on_node": len(peers)}
        except Exception as e:  # placement is best effort: never a reason to fail the run
            info = {"numa_node": node, "error": str(e)}
    return info


# --------------------------------------------------------------------------------------------------
# launcher: --gpus N without torchrun
# --------------------------------------------------------------------------------------------------
def launch_ranks(args, argv):
    """Start one child per GPU and wait.  The parent imports nothing that could initialise a GPU."""
    import socket
    n = args.gpus
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.pop("AKZ_BENCH_PINNED", None)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))
    out0, failed = None, None
    try:
        # rank 0's stdout is small (one JSON line); read it while polling so that nobody blocks on a full pipe
        import threading
        buf = []
        t = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
        t.start()
        alive = set(range(n))
        while alive and failed is None:
            for r in sorted(alive):
                rc = procs[r].poll()
                if rc is None:
                    continue
                alive.discard(r)
                if rc != 0:
                    failed = (r, rc)
                    break
            time.sleep(0.05)
        if failed is None:
            t.join(timeout=10)
            out0 = b"".join(buf).decode("utf-8", "replace")
    finally:
        for p in procs:  # exactly the children started above, by handle
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
    if failed is not None:
        sys.stderr.write(f"bench.py: rank {failed[0]} exited with status {failed[1]}; the other ranks were stopped\n")
        return 1
    line = [ln for ln in (out0 or "").splitlines() if ln.startswith("{")]
    if not line:
        sys.stderr.write("bench.py: rank 0 printed no JSON line\n")
        return 1
    sys.stdout.write(line[-1] + "\n")
    sys.stdout.flush()
    return 0


# --------------------------------------------------------------------------------------------------
# stub context (launcher / exchange self-test on a machine without GPUs)
# --------------------------------------------------------------------------------------------------
class _StubResult:
    def __init__(self, rows):
        self.num_images, self._rows = len(rows), rows

    def counts(self, i):
        return (16, self._rows[i], 61)

    def close(self):
        pass


class _StubJob:
    def __init__(self, rows):
        self._rows = rows

    def finish(self):
        time.sleep(0.002)
        return _StubResult(self._rows)


class _StubContext:
    """Stands in for akaze_amd.Context when --stub is given: no HIP library, no GPU, fixed row counts."""

    def __init__(self, rank):
        self.rank = rank

    def extract_begin(self, batch, cfg, keep_all_planes=True, input_ready=False, host_descriptors=True):
        return _StubJob([100 + 7 * self.rank + i for i in range(int(batch.shape[0]))])

    def set_eager_finish(self, *_):
        pass

    def set_lanes(self, *_):
        pass

    def set_profiling(self, *_):
        pass

    def get_profile(self, reset=True):
        return {k: 0.0 for k in ("blur0", "contrast", "prep", "fed", "detector", "nms", "host_kp", "orient", "mldb", "total")} | {
            "fed_launches": 0, "fed_px_steps": 0, "det_launches": 0, "det_px": 0, "calls": 0, "pixels": 0}


# --------------------------------------------------------------------------------------------------
# one rank
# --------------------------------------------------------------------------------------------------
def main_rank(args):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    placement = place_rank(args)  # before torch / HIP exist in this process
    # stdout carries exactly ONE line, the JSON: whatever libraries print there (RCCL writes a version banner to stdout
    # when a communicator is created) goes to stderr instead
    json_out = os.fdopen(os.dup(1), "w")
    sys.stdout.flush()
    os.dup2(2, 1)
    dev_index = 0 if args.share_gpu else local_rank

    import numpy as np
    import torch
    import torch.distributed as dist

    stub = args.stub
    c5 = args.workload == "c5"
    use_dist = world > 1 or args.force_dist          # a process group exists (rendezvous, barriers, reductions)
    use_xch = use_dist or c5                          # a step has an exchange (c5: also at N = 1, the all-pairs match reads gathered blocks)
    # capi: RCCL behind the C ABI; external: the C ABI's blocks carried by gloo (--share-gpu: RCCL refuses two ranks on one device)
    exchange = "gloo-stub" if stub else ("external" if args.share_gpu else args.exchange)
    if c5 and (stub or exchange == "torch"):
        raise SystemExit("--workload c5 runs through the C ABI (akz_gather_begin -> akz_match_all_pairs): not with --stub / --exchange torch")
    W, H, F = args.width, args.height, args.frames
    if F < 1:
        raise SystemExit(f"rank {rank}: --frames must be >= 1")

    if stub:
        A = None
        dev = torch.device("cpu")
    else:
        import akaze_amd as A
        torch.cuda.set_device(dev_index)
        dev = torch.device("cuda", dev_index)

    if use_dist:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # a rank that never arrives (a GPU that failed to initialise, a crashed peer) must not leave the others waiting
        # for the default half hour: the rendezvous and every later collective of the group give up after five minutes
        tmo = datetime.timedelta(seconds=int(os.environ.get("AKZ_BENCH_RENDEZVOUS_S", "300")))
        if exchange == "torch":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=tmo)
        else:  # rendezvous, barriers and the two scalar reductions need no GPU: gloo
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=tmo)

    def host_max(v):
        if not use_dist:
            return v
        t = torch.tensor([v], dtype=torch.float64, device=dev if exchange == "torch" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def host_sum(v):
        if not use_dist:
            return v
        t = torch.tensor([v], dtype=torch.float64, device=dev if exchange == "torch" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    def host_allgather(v):
        if not use_dist:
            return [v]
        t = torch.tensor([v], dtype=torch.float64, device=dev if exchange == "torch" else "cpu")
        out = torch.zeros(world, dtype=torch.float64, device=t.device)
        dist.all_gather_into_tensor(out, t)
        return [float(x) for x in out.tolist()]

    # ---- this rank's shard of the world*F frames of a step: image i -> GPU i mod world (weak scaling) ----
    if stub:
        frames = np.zeros((F, 8, 8), np.uint8)
        d_frames = torch.from_numpy(frames)
        cfg = None
        ctx = _StubContext(rank)
    else:
        cfg = A.Config(num_sublevels=args.sublevels, max_octave_evolution=args.octaves)  # default 4 x 4, 486-bit M-LDB
        if args.threshold is not None:
            cfg.detector_threshold = args.threshold
        frames = np.stack([A.synth_frame(W, H, i) for i in A.shard_frames(world * F, rank, world)])
        d_frames = torch.from_numpy(frames).to(dev)
        torch.cuda.synchronize()
        # one context = one host thread + one HIP stream + an auxiliary stream.  The stream is a dedicated
        # non-blocking one, never the legacy null stream: that one synchronises implicitly with every
        # blocking stream in the process (RCCL has some) and would serialise the pipeline.
        main = torch.cuda.Stream(dev)
        torch.cuda.set_stream(main)
        ctx = A.Context(dev_index, main.cuda_stream)
        if placement.get("cpus"):  # this rank owns exactly these cores: no second division by LOCAL_WORLD_SIZE
            ctx.set_host_threads(placement["cpus"])
        ctx.set_detector_mode(args.det_mode)
        ctx.set_eager_finish(bool(args.eager))
        ctx.debug_set_select(None if args.select < 0 else args.select)
        for kv in filter(None, args.sched.split(",")):
            k, v = kv.split("=")
            ctx.debug_set_schedule(int(k), int(v))
        ctx.set_prep_mode(args.prep_mode)
        if args.sort != "auto":
            ctx.debug_set_host_sort(args.sort == "host")
        # the stream-placement probe now, on idle streams, instead of inside the first large extract_begin (which then
        # stays asynchronous); config.stream_placement says what it found
        ctx.warmup()
    NP = max(1, min(args.parts, F))
    cut = [(F * i) // NP for i in range(NP + 1)]
    batches = [d_frames[cut[i]:cut[i + 1]] for i in range(NP)]
    # the same batches in pinned host memory (the reference's extract_features starts from host data, lib.rs:171-178):
    # second timed region, upload of batch j+1 on the context's copy stream under the kernels of batch j
    h_batches = None
    if not stub and not args.no_host_input and not c5:
        h_frames = torch.from_numpy(frames).pin_memory()
        h_batches = [h_frames[cut[i]:cut[i + 1]] for i in range(NP)]
    frame_src = {"host": False}

    # ---- the exchange: one all-gather of descriptor rows per step, begun when the step's results exist, retired when
    # the next step's results exist (or at the end of the timed region).  --workload c5: the all-pairs match of the step
    # (akz_match_all_pairs over the gathered blocks) is enqueued right behind the exchange and its totals are read at
    # retire time -- exchange, match and the read of the lists' counts are all inside the timed region ----
    xch = {"host_ms": 0.0, "wait_ms": 0.0, "cap": 0, "pending": None, "comm": None, "mode": exchange if use_xch else "none",
           "ranks_seen": None, "seq": 0, "lists": 0, "matches": 0, "dist": 0, "dist_sum": 0, "steps": 0}
    if use_xch and exchange in ("capi", "external"):
        try:
            if exchange == "external":
                xch["comm"] = A.Comm(dev_index, None, rank, world)
            else:
                uid = torch.zeros(A.COMM_ID_BYTES, dtype=torch.uint8)
                if rank == 0:
                    uid = torch.frombuffer(bytearray(A.comm_unique_id()), dtype=torch.uint8).clone()
                if use_dist:
                    dist.broadcast(uid, src=0)
                xch["comm"] = A.Comm(dev_index, bytes(uid.numpy().tobytes()), rank, world)
            # the collective of a step runs beside the next batch's kernels: its streams go where the context's busy ones are not
            xch["comm"].place_streams(ctx)
            # a peer that died leaves a collective waiting for ever: this rank then fails with a message instead of hanging
            xch["comm"].set_timeout(float(os.environ.get("AKZ_BENCH_RENDEZVOUS_S", "300")))
            ok = 1.0
        except Exception as e:  # RCCL cannot be loaded / initialised
            sys.stderr.write(f"rank {rank}: C-ABI exchange unavailable ({e})\n")
            ok = 0.0
        if -host_max(-ok) < 0.5:  # min over ranks: every rank takes the same fallback
            if xch["comm"] is not None:
                xch["comm"].close()
                xch["comm"] = None
            if c5:
                # the all-pairs match needs the C ABI's gather objects: the blocks travel over the process group instead
                sys.stderr.write(f"rank {rank}: falling back to the caller-carried transport (akz_comm_create_external over the process group)\n")
                xch["mode"] = exchange = "external"
                xch["comm"] = A.Comm(dev_index, None, rank, world)
            else:
                sys.stderr.write(f"rank {rank}: falling back to torch.distributed\n")
                xch["mode"] = exchange = "torch"
                xch["group"] = dist.new_group(backend="nccl")
    if use_dist and exchange == "torch" and "group" not in xch:
        xch["group"] = None
    side = None if stub or not use_dist else torch.cuda.Stream(dev)  # torch collectives run here, never on the extraction stream
    # --match-ctx own: the all-pairs launches of a step on a context (stream, scratch) of their own, beside the next steps'
    # extraction kernels (matrix cores beside HBM-bound stencils) instead of in line with them on the extraction stream
    match_ctx = ctx
    if c5 and args.match_ctx == "own":
        match_stream = torch.cuda.Stream(dev)
        match_ctx = A.Context(dev_index, match_stream.cuda_stream)

    def agree_capacity(rows):
        most = int(host_max(float(rows)))
        xch["cap"] = max(1024, (most + most // 4 + 4095) // 4096 * 4096)  # 25 % head-room, whole 256 KiB blocks

    def exchange_retire():
        g = xch["pending"]
        if g is None:
            return
        t0 = time.perf_counter()
        if exchange in ("capi", "external"):
            gth, pr = g
            if pr is not None:
                xch["lists"], xch["matches"], xch["dist"] = pr.totals()  # waits for the step's launches, reads every list's count
                xch["dist_sum"] += xch["dist"]
                xch["steps"] += 1
                pr.free()
            _, _, cnts_r, imgs_r = gth.finish(want_counts=True)  # the headers of all blocks: who took part
            xch["ranks_seen"] = sum(1 for v in imgs_r if v > 0)
            gth.free()
        elif exchange == "torch":
            g[0].synchronize()  # event behind the collectives on the side stream
            xch["ranks_seen"] = int((g[1] > 0).sum().item())
        xch["wait_ms"] += (time.perf_counter() - t0) * 1e3
        xch["pending"] = None

    def exchange_begin(results):
        rows = sum(res.counts(i)[1] for res in results for i in range(res.num_images))
        if xch["cap"] == 0:
            # every rank must pad to the SAME capacity: agree on it once (first warm-up step); the frames of a shard
            # are the same every step, so the row counts do not change afterwards
            agree_capacity(rows)
        if rows > xch["cap"]:
            raise SystemExit(f"rank {rank}: {rows} descriptor rows exceed the agreed gather capacity {xch['cap']}")
        exchange_retire()  # step k-1's gather (and all-pairs match): long complete
        t0 = time.perf_counter()
        if exchange in ("capi", "external"):
            gth = xch["comm"].gather_begin(results, xch["cap"])
            if exchange == "external":
                # the C ABI's wire format over the process group: this rank's block D2H, one gloo all-gather of the
                # fixed-size blocks, H2D into the gather's receive blocks, akz_gather_deliver (synchronous)
                gth.exchange_over()
            pr = gth.match_all_pairs(match_ctx) if c5 else None  # enqueues and returns
            xch["pending"] = (gth, pr)
        elif exchange == "torch":
            cap = xch["cap"]
            if "bufs" not in xch:  # two sets of preallocated buffers: step k's collective may still run while k+1 fills
                xch["bufs"] = [(torch.zeros((cap, 64), dtype=torch.uint8, device=dev),
                                torch.empty((world * cap, 64), dtype=torch.uint8, device=dev),
                                torch.zeros(1, dtype=torch.int64, device=dev),
                                torch.zeros(world, dtype=torch.int64, device=dev)) for _ in range(2)]
                xch["flip"] = 0
            local, gathered, cnt, cnts = xch["bufs"][xch["flip"]]
            xch["flip"] ^= 1
            o = 0
            for res in results:
                o += res.copy_device_descriptors(local[o:])  # D2D on the context's aux stream, complete on return
            with torch.cuda.stream(side):
                cnt.fill_(rows)
                dist.all_gather_into_tensor(cnts, cnt, group=xch["group"])
                dist.all_gather_into_tensor(gathered, local, group=xch["group"])
                ev = torch.cuda.Event()
                ev.record(side)
            xch["pending"] = (ev, cnts)
        xch["host_ms"] += (time.perf_counter() - t0) * 1e3

    if stub:
        def exchange_begin(results):  # noqa: F811  gloo all-gather of fake rows, synchronous
            rows = sum(res.counts(i)[1] for res in results for i in range(res.num_images))
            t0 = time.perf_counter()
            cnt = torch.tensor([rows], dtype=torch.int64)
            cnts = torch.zeros(world, dtype=torch.int64)
            dist.all_gather_into_tensor(cnts, cnt)
            cap = int(cnts.max().item())
            local = torch.zeros((cap, 64), dtype=torch.uint8)
            gathered = torch.empty((world * cap, 64), dtype=torch.uint8)
            dist.all_gather_into_tensor(gathered, local)
            xch["host_ms"] += (time.perf_counter() - t0) * 1e3

    host_ms = {"begin": 0.0, "finish": 0.0, "calls": 0}  # host time inside extract_begin / extract_finish
    keep_last = {"results": None}

    def run_steps(k_steps, keep_final=False):
        """k_steps passes over the shard as one software-pipelined stream of k_steps*NP batches:
        begin(batch j+1) is enqueued before finish(batch j), also across step boundaries, so the
        candidate fetch + host keypoint logic of a batch run under the kernels of the next one.
        Every step's results are complete (and, with N > 1, gathered) before run_steps returns."""
        nk, inflight, done, steps_done = 0, [], [], 0

        def retire(res):
            nonlocal nk, done, steps_done
            done.append(res)
            if len(done) == NP:  # a whole step has finished
                nk = sum(r.counts(i)[1] for r in done for i in range(r.num_images))
                if use_xch:
                    exchange_begin(done)
                steps_done += 1
                if keep_final and steps_done == k_steps:
                    keep_last["results"] = done  # only the LAST step's results outlive the timed region (self-check)
                else:
                    for r in done:
                        r.close()
                done = []

        for _ in range(k_steps):
            for bi, bt in enumerate(batches):
                tb = time.perf_counter()
                if frame_src["host"]:
                    job = ctx.extract_begin_host(h_batches[bi], cfg, keep_all_planes=not args.lean)
                else:
                    # (the frames were uploaded and synchronised before the first step: AKZ_INPUT_READY holds)
                    # (c5: the descriptor rows stay on the device -- the exchange and the matcher read them there)
                    job = ctx.extract_begin(bt, cfg, keep_all_planes=not args.lean, input_ready=not args.no_input_ready,
                                            host_descriptors=not c5)
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
        if use_xch:
            exchange_retire()  # the last step's gather (and all-pairs match) completes inside the timed region
        return nk

    def barrier():
        if not stub:
            torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        if not stub:
            torch.cuda.synchronize()

    if args.warmup:
        run_steps(args.warmup)
    # one extra untimed step with full stage profiling (informational stage_ms_per_step); the timed region only
    # records the FED and detector spans needed for the rooflines (light mode: ~40 instead of ~190 HIP events per batch)
    ctx.set_profiling(0 if args.no_profile else 1)
    ctx.get_profile(reset=True)
    run_steps(1)
    warm_prof = ctx.get_profile(reset=True)
    if not stub:
        ctx.kernel_rows(reset=True)  # (the rows below cover the timed regions only)
    ctx.set_profiling(0 if args.no_profile else 2)
    want_check = not stub and not c5 and ((rank == 0 and not args.no_self_check) or use_dist or F <= 4)
    # The timed region -- exactly args.steps steps between two barriers -- is run args.regions times back to back; the
    # line reports the MEDIAN region (max over ranks per region) and lists them all: one 0.2 s region per run cannot
    # resolve a 2 % change on boxes that differ by 3 % from run to run.
    xch["host_ms"] = xch["wait_ms"] = 0.0
    xch["dist_sum"], xch["steps"] = 0, 0
    host_ms.update(begin=0.0, finish=0.0, calls=0)
    regions = []  # (elapsed max over ranks, this rank's elapsed)
    for reg in range(args.regions):
        barrier()
        t0 = time.perf_counter()
        nk = run_steps(args.steps, keep_final=want_check and reg == args.regions - 1)
        barrier()
        el = time.perf_counter() - t0
        regions.append((host_max(el), el))
    prof = ctx.get_profile(reset=True)  # spans of ALL regions: the rooflines average over args.regions * args.steps steps
    kernel_rows = [] if stub else ctx.kernel_rows(reset=True)  # the same spans by kernel variant and launch shape
    ctx.set_profiling(False)
    timed_steps = args.steps * args.regions
    host_ms_timed = dict(host_ms)  # the accumulators go on counting in the host-input leg below: the line reports the timed regions
    order = sorted(range(len(regions)), key=lambda i: regions[i][0])
    med = order[(len(order) - 1) // 2]   # the median region (the lower one of an even count: a run that happened)
    elapsed, elapsed_rank = regions[med]
    per_rank_s = host_allgather(elapsed_rank)

    total_px = float(W) * H * F * world * args.steps
    value = total_px / elapsed / 1e6
    region_values = [round(total_px / r[0] / 1e6, 1) for r in regions]

    # ---- the same steps with the frames in pinned HOST memory, H2D inside the timed region (informational: `value`
    # stays the HBM-resident figure the bench contract asks for) ----
    host_input = None
    if h_batches is not None:
        ctx.set_profiling(0)
        frame_src["host"] = True
        run_steps(min(3, max(1, args.warmup)))
        els = []
        for _ in range(min(3, args.regions)):
            barrier()
            t0 = time.perf_counter()
            nk_h = run_steps(args.steps)
            barrier()
            els.append(host_max(time.perf_counter() - t0))
        el_h = sorted(els)[(len(els) - 1) // 2]
        frame_src["host"] = False
        v_h = total_px / el_h / 1e6
        host_input = {"value": round(v_h, 2), "unit": "Mpix/s", "ms_per_step": round(el_h / max(1, args.steps) * 1e3, 3),
                      "of_hbm_resident": round(v_h / value, 4), "keypoints_per_step_rank0": nk_h,
                      "h2d_MB_per_step": round(float(W) * H * F / 1e6, 1),
                      "input": "pinned host memory; akz_extract_begin_host_u8 uploads batch j+1 on the context's copy "
                               "stream under the kernels of batch j (same steps, same pipelining, inside the timed region)"}

    if stub:
        if rank == 0:
            print(json.dumps({
                "metric": "launcher self-test (stub context, no GPU work) - NOT a measurement", "value": 0.0, "unit": "Mpix/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(elapsed / max(1, args.steps) * 1e3, 3), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f32", "data": "stub", "stub": True,
                "config": {"workload": "stub", "frames_per_gpu": F, "exchange": "gloo all-gather of fake rows",
                           "exchange_ms_per_step": round(xch["host_ms"] / max(1, timed_steps), 3),
                           "regions": {"n": args.regions, "reported": "median"},
                           "per_rank_ms_per_step": [round(s / max(1, args.steps) * 1e3, 3) for s in per_rank_s]}}), file=json_out, flush=True)
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return 0

    workload_key = {"width": W, "height": H, "frames": F, "octaves": args.octaves, "sublevels": args.sublevels,
                    "lean": bool(args.lean)}

    # ---- rooflines of the two kernels that dominate the step, from HIP-event spans inside the timed region ----
    def roof(kernel, ms, launches, alg_bytes, note):
        s = ms / 1e3
        achieved = alg_bytes / s / 1e9 if s > 0 else 0.0
        avg_us = ms * 1e3 / max(1, launches)
        tr = pmc_traffic(kernel, workload_key)
        out = {"bound": "hbm", "kernel": kernel, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": tr,
               "traffic_frac": round(tr / (avg_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if tr and avg_us > 0 else None,
               "launches": launches, "avg_launch_us": round(avg_us, 2),
               "algorithmic_bytes_per_launch": round(alg_bytes / max(1, launches)),
               "ms_per_step": round(ms / max(1, timed_steps), 3), "note": note}
        return out

    det_bpp = 4 + 8 + 4 + (0 if args.lean else 12)  # Lsmooth in; Lx, Ly, Ldet (+ Lxx, Lyy, Lxy) out: each once
    roof_det = roof(A.lib().akz_detector_kernel_name().decode(), prof["detector"], prof["det_launches"],
                    float(det_bpp) * prof["det_px"],
                    f"all detector launches of the timed steps ({det_bpp} B per level pixel: Lsmooth read once, every output "
                    "plane written once); time = sum of HIP-event spans around each launch on the launching stream. In the "
                    "step the fine-octave launches share the chip with the coarse octaves' chain and the previous batch's "
                    "keypoint kernels (that overlap is worth +7 % throughput), which stretches the spans: standalone_frac "
                    "is the same kernel alone on HBM-cold planes (detector_standalone)")
    roof_fed = roof(A.lib().akz_fed_kernel_name().decode(), prof["fed"], prof["fed_launches"],
                    FED_BYTES_PER_PX_STEP * prof["fed_px_steps"] + 12.0 * prof["fused_px"],
                    "all diffusion launches of the timed steps (levels 1..15, 1920x1080 down to 240x135): k_level_march "
                    "(level preparation + the level's first <= 4 FED steps in one launch, levels of 4 Mpx and more) and "
                    "k_fed_own (<= 8 steps per launch); 12 B per pixel-step ALGORITHMIC plus 12 B per pixel for a "
                    "preparation that runs inside the launch — steps are fused, so frac can exceed 1; traffic_frac is "
                    "the DRAM figure")
    # ---- the rows behind the two groups: one per kernel NAME (as rocprofv3's kernel stats list them, so that avg_launch_us can
    # be held against profiles/*_kernel_stats.csv), with the launch shapes it ran on ----
    def per_kernel_traffic(name):
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            if d.get("workload") != workload_key:
                return None
            e = d.get("per_kernel", {}).get(name)
            return e.get("hbm_bytes_per_launch") if e else None
        except Exception:
            return None

    def kernel_table(kinds):
        by_name = {}
        for r in kernel_rows:
            if r["kind"] not in kinds:
                continue
            odd = "true" if (r["w"] & 1) else "false"
            if r["kind"] == 1:    # k_level_march<N, KEEPSTEP, ODDW, HALF>
                name = f"k_level_march<{r['param'] & 15}, {'true' if r['param'] & 32 else 'false'}, {odd}, {'true' if r['param'] & 16 else 'false'}>"
                alg = 12.0 * r["px"] + FED_BYTES_PER_PX_STEP * r["px_steps"]
            elif r["kind"] == 2:  # the k_fed_own launches of one level (their template arguments depend on the launch size)
                name, alg = "k_fed_own", FED_BYTES_PER_PX_STEP * r["px_steps"]
            elif r["kind"] == 3:
                name, alg = "k_octave_resident<true>", FED_BYTES_PER_PX_STEP * r["px_steps"]
            elif r["kind"] == 4:  # k_detector_tiled<S, NMS, KEEP>
                name, alg = f"k_detector_tiled<{r['param']}, true, {'false' if args.lean else 'true'}>", float(det_bpp) * r["px"]
            else:                 # k_detector_march<S, NMS, KEEP, ODDW>
                name, alg = f"k_detector_march<{r['param']}, true, {'false' if args.lean else 'true'}, {odd}>", float(det_bpp) * r["px"]
            e = by_name.setdefault(name, {"kernel": name, "launches": 0, "ms": 0.0, "alg": 0.0, "shapes": []})
            e["launches"] += r["launches"]
            e["ms"] += r["ms"]
            e["alg"] += alg
            e["shapes"].append({"level": f"{r['n']} x {r['w']}x{r['h']}", "launches": r["launches"],
                                "avg_launch_us": round(r["ms"] * 1e3 / max(1, r["launches"]), 1)})
        out_rows = []
        for e in sorted(by_name.values(), key=lambda e: -e["ms"]):
            us = e["ms"] * 1e3 / max(1, e["launches"])
            gbs = e["alg"] / (e["ms"] * 1e-3) / 1e9 if e["ms"] > 0 else 0.0
            tr = per_kernel_traffic(e["kernel"])
            out_rows.append({"kernel": e["kernel"], "launches": e["launches"], "launches_per_step": round(e["launches"] / max(1, timed_steps), 2),
                             "avg_launch_us": round(us, 1), "ms_per_step": round(e["ms"] / max(1, timed_steps), 3),
                             "algorithmic_bytes_per_launch": round(e["alg"] / max(1, e["launches"])),
                             "achieved": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": tr,
                             "traffic_frac": round(tr / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if tr and us > 0 else None,
                             "shapes": e["shapes"]})
        return out_rows
    roof_det["kernels"] = kernel_table((4, 5))
    roof_fed["kernels"] = kernel_table((1, 2, 3))
    # the detector march is the kernel with the largest share of the step when each kernel runs alone (2.2 of 5.7 ms of
    # kernel time); since the fork the two groups' spans overlap in the step, so their sums no longer rank them
    roofline, roofline_2 = roof_det, roof_fed

    # ---- the FED kernel alone: 3840x2160 plane (north-star target point) and a 32 x 1080p level (HBM-resident) ----
    fed_alone = None
    if rank == 0 and world == 1 and not args.no_fed4k and not c5:  # N = 1 information
        fed_alone = {}
        for name, shape, nst in (("4k_plane", (2160, 3840), 40), ("level_32x1080p", (32, 1080, 1920), 8)):
            lt = torch.rand(shape, dtype=torch.float32, device=dev)
            lf = torch.rand(shape, dtype=torch.float32, device=dev)
            taus = np.full(nst, 0.2)
            ctx.fed_steps(lt, lf, taus)  # warm
            ctx.set_profiling(True)
            ctx.get_profile(reset=True)
            reps = 3
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                ctx.fed_steps(lt, lf, taus)
            e1.record()
            torch.cuda.synchronize()
            p4 = ctx.get_profile(reset=True)
            ctx.set_profiling(False)
            launches = max(1, p4["fed_launches"])
            ms = p4["fed"]  # spans around the FED launches only (the entry point's input copy is outside them)
            px = float(np.prod(shape))
            gbs = FED_BYTES_PER_PX_STEP * px * nst * reps / (ms * 1e-3) / 1e9
            fed_alone[name] = {"avg_launch_us": round(ms / launches * 1e3, 2), "achieved": round(gbs, 1),
                               "frac": round(gbs / HBM_PEAK_GBS, 4), "steps": nst, "launches_per_pass": launches // reps,
                               "algorithmic_bytes_per_launch": round(FED_BYTES_PER_PX_STEP * px * nst * reps / launches)}
            if name == "4k_plane":
                # the same launch under the FETCH_SIZE / WRITE_SIZE passes (tools/pmc_calib.py, committed in profiles/pmc_traffic.json):
                # what it really moves -- `frac` above counts 12 B per pixel-step, the launch fuses 8 steps and its 100 MB
                # working set sits in the Infinity Cache
                try:
                    cal = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["calibration"]["fed_own_4k"]
                    tr = (cal["fetch_kib_per_launch"] + cal["write_kib_per_launch"]) * 1024.0
                    us = ms / launches * 1e3
                    fed_alone[name].update({"traffic": round(tr), "traffic_frac": round(tr / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                            "bound": "vector issue / LDS latency, not HBM: frac > 1 is 8-step temporal fusion on a cache-resident plane"})
                except Exception:
                    fed_alone[name].update({"traffic": None, "traffic_frac": None})
            del lt, lf
        # the north star's own point -- the FED inner loop on one 3840x2160 plane -- inside `roofline` (the record the driver
        # parses): algorithmic_frac counts SURVEY 8(d)'s 12 B per pixel-step, traffic_frac what the launch really moves
        f4 = fed_alone.get("4k_plane")
        if f4:
            roofline["fed_4k"] = {"kernel": "k_fed_own<64, 32, 8, 512>", "plane": "1 x 3840x2160", "steps_per_launch": f4["steps"] // max(1, f4["launches_per_pass"]),
                                  "avg_launch_us": f4["avg_launch_us"], "us_per_step": round(f4["avg_launch_us"] * f4["launches_per_pass"] / f4["steps"], 2),
                                  "ideal_unfused_us_per_step": round(FED_BYTES_PER_PX_STEP * 3840 * 2160 / (HBM_PEAK_GBS * 1e9) * 1e6, 2),
                                  "algorithmic_frac": f4["frac"], "traffic": f4.get("traffic"), "traffic_frac": f4.get("traffic_frac"),
                                  "bound": "vector issue / LDS latency (8 steps fused per launch on a plane that sits in the Infinity Cache): "
                                           "algorithmic_frac > 0.40 is the north-star target in SURVEY 8(d)'s terms; traffic_frac is the DRAM figure"}

    # ---- the detector kernel alone, on HBM-cold inputs: four plane sets of a 32 x 1080p level in turn (a launch that
    # re-reads the plane it read last time finds most of it in the 256 MB Infinity Cache and looks 25 % faster than any
    # launch of the pyramid, where every level's Lsmooth was written long before) ----
    det_alone = None
    if rank == 0 and world == 1 and not args.no_fed4k and not args.lean and not c5:
        n_, h_, w_ = 32, 1080, 1920
        pb = n_ * h_ * w_ * 4
        big = torch.empty(4 * 7 * pb + (1 << 21), dtype=torch.uint8, device=dev)  # carved like the product's slab
        base = (big.data_ptr() + (1 << 21) - 1) >> 21 << 21
        sets = [[base + (7 * k + i) * pb for i in range(7)] for k in range(4)]
        src = torch.rand((n_, h_, w_), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()  # (the copies below run on the null stream, which does not wait for this stream's kernels)
        for k in range(4):
            A.copy_d2d(sets[k][0], src.data_ptr(), pb)
        det_alone = {}
        for S in (2, 3, 4):
            def call(k):
                p_ = sets[k]
                A._check(A.lib().akz_op_detector_response(ctx._h, p_[0], S, p_[1], p_[2], p_[3], p_[4], p_[5], p_[6], w_, h_, n_))
            for k in range(4):
                call(k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 3
            e0.record()
            for _ in range(reps):
                for k in range(4):
                    call(k)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / (4 * reps) * 1e3
            gbs = 28.0 * n_ * h_ * w_ / us / 1e3
            det_alone[f"sigma_size_{S}"] = {"avg_launch_us": round(us, 1), "achieved": round(gbs, 1),
                                            "frac": round(gbs / HBM_PEAK_GBS, 4)}
        det_alone["note"] = ("k_detector_march alone on one 32 x 1080p level, 28 B/px, inputs not in any cache; a plain "
                             "streaming kernel with the same 1 read : 6 writes reaches 5.0-5.5 TB/s on this part "
                             "(tools/membw/hbm_ceiling.hip)")
        del big, src
        roof_det["standalone_frac"] = det_alone["sigma_size_3"]["frac"]

    # ---- what every rank computed, by GLOBAL frame index (frame i of the job lives on rank i mod world): sha256 of the
    # keypoints + descriptor bytes of the first frames of each shard, gathered to rank 0.  A --gpus 2 --share-gpu run and
    # a single-rank run over the same frames must print the same table (tests/test_gpu_gather.py). ----
    frame_sha = None
    if not stub and keep_last["results"]:
        import hashlib
        mine = {}
        gidx = A.shard_frames(world * F, rank, world)
        for k in range(min(F, 4 if F <= 4 else 2)):
            part = next(i for i in range(NP) if cut[i] <= k < cut[i + 1])
            rb, kb = keep_last["results"][part], k - cut[part]
            mine[int(gidx[k])] = hashlib.sha256(rb.keypoints(kb).tobytes() + rb.descriptors(kb).tobytes()).hexdigest()[:16]
        if use_dist:
            allsha = [None] * world
            dist.all_gather_object(allsha, mine)
            frame_sha = {str(k): v for d in allsha for k, v in sorted(d.items())}
        else:
            frame_sha = {str(k): v for k, v in sorted(mine.items())}
        if not (rank == 0 and not args.no_self_check):
            for r in keep_last["results"]:
                r.close()
            keep_last["results"] = None

    # ---- self-check of the timed configuration (untimed): frame k of the batch == the same frame extracted alone ----
    self_check = None
    if rank == 0 and keep_last["results"]:
        import hashlib
        res_b = keep_last["results"]
        picks = sorted(set([0, F // 2, F - 1]))
        ok, sums = True, []
        for k in picks:
            part = next(i for i in range(NP) if cut[i] <= k < cut[i + 1])
            rb, kb = res_b[part], k - cut[part]
            single = ctx.extract_begin(d_frames[k:k + 1], cfg, keep_all_planes=not args.lean).finish()
            a = (rb.keypoints(kb).tobytes(), rb.descriptors(kb).tobytes())
            b = (single.keypoints(0).tobytes(), single.descriptors(0).tobytes())
            single.close()
            ok = ok and a == b and len(a[0]) > 0
            sums.append(hashlib.sha256(a[0] + a[1]).hexdigest()[:16])
        for r in res_b:
            r.close()
        keep_last["results"] = None
        self_check = {"frames": picks, "identical_to_single_frame_extraction": ok, "sha256_16": sums}
        if not ok:
            raise SystemExit("bench.py self-check failed: a frame of the timed batch differs from its single-frame extraction")

    # ---- the matcher (BASELINE configs[2] / [4]: Hamming match of two descriptor sets), untimed leg, rank 0 ------
    match_leg = None
    if rank == 0 and world == 1 and not args.no_match and not c5:  # N = 1 information; the scaling runs only need `value`
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
                         "mfma_frac": round(pairs * 2 * 512 / (ms_m * 1e-3) / MFMA_FP4_PEAK_OPS, 3)})
        # all-pairs shape (BASELINE configs[4]): one query image against 16 train sets in one launch
        n_q, n_sets = 11264, 16
        dq = torch.randint(0, 256, (n_q, 64), dtype=torch.uint8, device=dev, generator=g)
        dt = torch.randint(0, 256, (n_q * n_sets, 64), dtype=torch.uint8, device=dev, generator=g)
        dq[:, 61:] = 0
        dt[:, 61:] = 0
        rows = [n_q] * n_sets
        ctx.descriptor_match_sets_device(dq, dt, rows)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ctx.descriptor_match_sets_device(dq, dt, rows)
        e1.record()
        torch.cuda.synchronize()
        ms_s = e0.elapsed_time(e1) / 5
        pairs = float(n_q) * n_q * n_sets
        legs.append({"n0": n_q, "n1": n_q, "train_sets": n_sets, "ms": round(ms_s, 3),
                     "Tpairs_per_s": round(pairs / ms_s / 1e9, 3),
                     "mfma_frac": round(pairs * 2 * 512 / (ms_s * 1e-3) / MFMA_FP4_PEAK_OPS, 3)})
        match_leg = {"kernel": "k_match_fp4 (+ unpack, merge, compaction): +-1 operands in FP4 on v_mfma_scale_f32_32x32x64_f8f6f4", "bound": "mfma",
                     "peak": "mfma_frac is against %.0f PFLOP/s, the dense FP4 rate of the instruction at 2.4 GHz; under this instruction "
                             "the chip holds ~2.0 GHz, and chains of it with one ds_read_b128 each and nothing else sustain 7.0-7.7 PFLOP/s "
                             "(tools/mfma_probe/fp4_rate.hip, profiles/r04_fp4_rate.txt)" % (MFMA_FP4_PEAK_OPS / 1e15),
                     "ops_per_pair": 1024, "sets": legs}

    # ---- BASELINE configs[2] and configs[4] end to end (extra legs, rank 0, own clocks) --------------------------------
    c3_leg = c5_leg = None
    if rank == 0 and world == 1 and not args.no_match and not c5:
        ctx.set_profiling(0)
        # configs[2]: a 3840x2160 pair -- ONE extract call on the two-frame batch (host frames, descriptors back on the
        # host) + match_features with its reference signature (descriptor_match on the GPU, RANSAC filter on the host)
        pair = np.stack([A.synth_frame(3840, 2160, 0), A.synth_frame(3840, 2160, 0, shift=(17, 9))])
        h_pair = torch.from_numpy(pair).pin_memory()
        cfg44 = A.Config()
        t_ex = t_mt = t_dm = 0.0
        n_m = n_k = 0
        reps_p = 6
        for it in range(reps_p + 2):
            torch.cuda.synchronize()
            a0 = time.perf_counter()
            rp = ctx.extract_begin_host(h_pair, cfg44, keep_all_planes=not args.lean).finish()
            k0, k1, q0, q1 = rp.keypoints(0), rp.keypoints(1), rp.descriptors(0), rp.descriptors(1)
            a1 = time.perf_counter()
            mm = A.match_features(k0, q0, k1, q1, 0.86, 1000, 3.0, ctx=ctx)
            a2 = time.perf_counter()
            ctx.descriptor_match(q0, q1, 10000, 0.86)  # the GPU half of match_features alone (upload, scan, compaction, download)
            a3 = time.perf_counter()
            rp.close()
            if it >= 2:
                t_ex += a1 - a0
                t_mt += a2 - a1
                t_dm += a3 - a2
                n_m, n_k = len(mm), len(k0) + len(k1)
        # the same pairs as a stream: pair j + 1 is begun before pair j is collected and matched (its finish half runs on the
        # context's own thread under the caller's match_features of the pair before)
        prev_job = None
        n_stream, n_warm = 12, 3
        for it in range(n_warm + n_stream + 1):
            if it == n_warm:  # (the first pairs of the stream grow the pooled blocks of two pyramids in flight)
                t_s0 = time.perf_counter()
            job = ctx.extract_begin_host(h_pair, cfg44, keep_all_planes=not args.lean) if it < n_warm + n_stream else None
            if prev_job is not None:
                rs = prev_job.finish()
                A.match_features(rs.keypoints(0), rs.descriptors(0), rs.keypoints(1), rs.descriptors(1), 0.86, 1000, 3.0, ctx=ctx)
                rs.close()
            prev_job = job
        t_stream = (time.perf_counter() - t_s0) / n_stream
        c3_leg = {"workload": "BASELINE configs[2]: 3840x2160 synthetic pair from pinned host memory, one extract call on the "
                              "2-frame batch (upload, scale space, keypoints, descriptors to the host) + match_features(0.86, "
                              "1000 RANSAC trials, eps 3.0)",
                  "ms_per_pair": round((t_ex + t_mt) / reps_p * 1e3, 3), "extract_ms": round(t_ex / reps_p * 1e3, 3),
                  "match_features_ms": round(t_mt / reps_p * 1e3, 3),
                  "match_features_split_ms": {"descriptor_match (GPU scan incl. descriptor upload, match download)": round(t_dm / reps_p * 1e3, 3),
                                              "remove_outliers (1000 RANSAC trials: samples, winner and final filter on the host, models + inlier counts on the device)": round((t_mt - t_dm) / reps_p * 1e3, 3)},
                  "streamed_ms_per_pair": round(t_stream * 1e3, 3),
                  "streamed_note": "pair j+1 begun before pair j is collected: its extraction runs under match_features of the pair before",
                  "Mpix_s": round(2 * 3840 * 2160 / ((t_ex + t_mt) / reps_p) / 1e6, 1), "keypoints": n_k, "matches": n_m}
        del h_pair, pair
        # configs[4] on one GPU: a stream of 3840x2160 frames, 5 octaves x 5 sublevels, then the exchange and the all-pairs
        # match of every frame against every other one through the C ABI (akz_gather_begin -> akz_match_all_pairs)
        try:
            n_fr, per = 16, 8
            cfg55 = A.Config(num_sublevels=5, max_octave_evolution=5)
            fr4k = torch.from_numpy(np.stack([A.synth_frame(3840, 2160, i) for i in range(n_fr)])).pin_memory()
            comm1 = A.Comm(dev_index, A.comm_unique_id(), 0, 1)
            best = None
            for it in range(3):
                torch.cuda.synchronize()
                b0 = time.perf_counter()
                jobs, ress = [], []
                for k in range(0, n_fr, per):
                    jobs.append(ctx.extract_begin_host(fr4k[k:k + per], cfg55, keep_all_planes=not args.lean, host_descriptors=False))
                    if len(jobs) > 1:
                        ress.append(jobs.pop(0).finish())
                while jobs:
                    ress.append(jobs.pop(0).finish())
                b1 = time.perf_counter()
                rows = sum(r.counts(i)[1] for r in ress for i in range(r.num_images))
                g = comm1.gather_begin(ress, rows + 64)
                for r in ress:
                    r.close()
                pr = g.match_all_pairs(ctx)  # (enqueues and returns)
                pr.count(0, 1)               # the first read of a list waits for the launches
                b2 = time.perf_counter()
                tot = pr.total_matches()
                pairs_n = sum(pr.image_rows(q)[0] * pr.image_rows(j)[0] for q in range(pr.n_images) for j in range(pr.n_images) if q != j)
                pr.free()
                g.free()
                cur = {"extract_ms": (b1 - b0) * 1e3, "match_ms": (b2 - b1) * 1e3, "rows": rows, "matches": tot, "pairs": pairs_n}
                if it > 0 and (best is None or cur["extract_ms"] + cur["match_ms"] < best["extract_ms"] + best["match_ms"]):
                    best = cur
            comm1.close()
            tt = (best["extract_ms"] + best["match_ms"]) / 1e3
            c5_leg = {"workload": f"BASELINE configs[4] on one GPU: {n_fr} 3840x2160 frames from pinned host memory in batches of {per}, "
                                  "5 octaves x 5 sublevels, descriptors kept on the device; RCCL gather (world 1) and the all-pairs "
                                  "match of every frame against every other (akz_match_all_pairs: every unordered pair once, both directions from one "
                                  "pass over its distances -- one both-direction multi-set launch per lead frame)",
                      "extract_ms": round(best["extract_ms"], 2), "extract_Mpix_s": round(n_fr * 3840 * 2160 / best["extract_ms"] / 1e3, 1),
                      "gather_and_match_ms": round(best["match_ms"], 2),
                      "Tpairs_per_s": round(best["pairs"] / best["match_ms"] / 1e9, 3),
                      "end_to_end_Mpix_s": round(n_fr * 3840 * 2160 / tt / 1e6, 1), "descriptor_rows": best["rows"],
                      "image_pairs": n_fr * (n_fr - 1), "matches": best["matches"]}
            del fr4k
        except Exception as e:
            c5_leg = {"error": str(e)[:300]}

    # ---- BASELINE configs[1] taken literally: ONE frame per extract call (untimed extra leg, rank 0) -----------
    single = None
    if rank == 0 and world == 1 and not args.no_single and not c5:
        ctx.set_profiling(0)
        one = d_frames[:1]
        for _ in range(10):
            ctx.extract_begin(one, cfg, keep_all_planes=not args.lean).finish().close()
        torch.cuda.synchronize()
        reps = 100
        t1 = time.perf_counter()
        for _ in range(reps):                      # latency of the synchronous entry point (akz_extract_device_*), nothing in flight
            ctx.extract_features(one, cfg, keep_all_planes=not args.lean).close()
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
        if not stub:
            try:  # where the order-dependent keypoint selection of a synchronous call ran (akz_debug_select_info)
                ctx.extract_features(one, cfg, keep_all_planes=not args.lean).close()
                si = ctx.debug_select_info()
                single["selection"] = {"where": {0: "host, spatial grids", 1: "host, device neighbour lists", 2: "device (k_select)"}.get(si[0], str(si[0])),
                                       "candidates": si[3], "looks_of_slowest_thread": si[1],
                                       "k_select_phase_us": {"first_states": si[4] / 100.0, "turns": si[5] / 100.0, "second_pass": si[6] / 100.0,
                                                             "output": si[7] / 100.0}}
            except Exception as e:
                single["selection"] = {"error": str(e)[:200]}
            try:  # where that call's angles (atan2f) and their cosines / sines came from (akz_debug_device_libm)
                avail, last = ctx.debug_device_libm()
                names = {0: "host libm (one host round trip between orientation and descriptors)",
                         1: "device: glibc's atan2f + the FMA build of cosf / sinf as IEEE arithmetic, proven against this process's libm",
                         2: "device: glibc's atan2f + the SSE2 build of cosf / sinf as IEEE arithmetic, proven against this process's libm"}
                single["orientation_libm"] = {"available": names.get(avail, str(avail)), "this_call": names.get(last, str(last))}
            except Exception as e:
                single["orientation_libm"] = {"error": str(e)[:200]}
        # the same stream on ONE context with lanes: frames dealt to 2 .. 4 child contexts, each of which finishes its frames
        # on its own thread; the caller's thread only enqueues the next frames and collects results; lone 4K frames the same way
        # (on a context of its own, as tools/lanes_ab.py: the bench's context has been through every other leg -- its four
        # streams, host-thread settings and probes -- and its lanes then measure 0.5-0.7 ms per frame where a fresh context's
        # measure 0.33-0.40)
        def stream_eager(lctx, frame, lanes, reps_e):
            lctx.set_lanes(lanes)
            for _ in range(2 * lanes + 4):
                lctx.extract_begin(frame, cfg, keep_all_planes=not args.lean).finish().close()
            torch.cuda.synchronize()
            t_e = time.perf_counter()
            pend = []
            for _ in range(reps_e):
                pend.append(lctx.extract_begin(frame, cfg, keep_all_planes=not args.lean))
                if len(pend) >= lanes:
                    pend.pop(0).finish().close()
            while pend:
                pend.pop(0).finish().close()
            dt = (time.perf_counter() - t_e) / reps_e
            return dt
        single["lanes"] = {}
        try:
            if stub:
                raise RuntimeError("no lanes on the stub context")
            st_l = torch.cuda.Stream(dev)
            with torch.cuda.stream(st_l):
                ctx_l = A.Context(dev_index, st_l.cuda_stream)
                if hasattr(ctx_l, "warmup"):
                    ctx_l.warmup()
                for lanes in (2, 3, 4):
                    thr_e = stream_eager(ctx_l, one, lanes, 2 * reps)
                    single["lanes"][str(lanes)] = {"stream_ms_per_frame": round(thr_e * 1e3, 3), "stream_Mpix_s": round(W * H / thr_e / 1e6, 1)}
                ctx_l.set_lanes(1)
                ctx_l.close()
        except Exception as e:
            single["lanes"]["error"] = str(e)[:300]
        if not stub:
            try:
                one4k = torch.from_numpy(A.synth_frame(3840, 2160, 3)[None]).to(dev)
                ctx.set_lanes(1)
                def stream_plain(frame, reps_p):
                    t_p = time.perf_counter()
                    prev = None
                    for _ in range(reps_p):
                        job = ctx.extract_begin(frame, cfg, keep_all_planes=not args.lean)
                        if prev is not None:
                            prev.finish().close()
                        prev = job
                    prev.finish().close()
                    return (time.perf_counter() - t_p) / reps_p
                stream_plain(one4k, 6)  # (warm-up in the same pattern: two pyramids alive at a time)
                thr4 = stream_plain(one4k, 30)
                # (a lone 4K frame is a batch-path job since round 5 -- it is not dealt to lanes; what helps a stream of them is a
                # second job begun ahead)
                def stream_ahead2(frame, reps_p):
                    t_p = time.perf_counter()
                    pend = []
                    for _ in range(reps_p):
                        pend.append(ctx.extract_begin(frame, cfg, keep_all_planes=not args.lean))
                        if len(pend) > 2:
                            pend.pop(0).finish().close()
                    while pend:
                        pend.pop(0).finish().close()
                    return (time.perf_counter() - t_p) / reps_p
                stream_ahead2(one4k, 6)
                thr4e = stream_ahead2(one4k, 40)
                t_l = time.perf_counter()
                for _ in range(20):
                    ctx.extract_features(one4k, cfg, keep_all_planes=not args.lean).close()
                lat4 = (time.perf_counter() - t_l) / 20
                single["lone_4k"] = {"workload": "one 3840x2160 frame per extract_features call",
                                     "sync_call_ms": round(lat4 * 1e3, 3),
                                     "stream_ms_per_frame": round(thr4 * 1e3, 3), "stream_Mpix_s": round(3840 * 2160 / thr4 / 1e6, 1),
                                     "two_begun_ahead": {"stream_ms_per_frame": round(thr4e * 1e3, 3),
                                                         "stream_Mpix_s": round(3840 * 2160 / thr4e / 1e6, 1)}}
                del one4k
            except Exception as e:
                single["lone_4k"] = {"error": str(e)[:300]}
        ctx.set_lanes(1)
        try:  # the begin phase (scale space, detector, extrema) as one hipGraph launch against the plain launch chain
            g_ms, p_ms, nodes = ctx.graph_probe(one, cfg, keep_all_planes=not args.lean, reps=50)
            single["graph"] = {"begin_phase_graph_ms": round(g_ms, 3), "begin_phase_plain_ms": round(p_ms, 3), "graph_nodes": nodes,
                               "note": "GPU time of the begin phase from an idle stream (HIP events), graph launch vs "
                                       "plain enqueue of the same kernels"}
        except Exception as e:
            single["graph"] = {"error": str(e)}
        # the same calls from K host threads with one context + stream each: a lone 1080p chain is launch-latency
        # bound and leaves most of the chip idle, concurrent chains fill it (ctypes releases the GIL in the calls)
        import threading
        K, reps_k = 4, 60
        wall = [0.0] * K
        bar = threading.Barrier(K)

        def single_worker(k):
            st_k = torch.cuda.Stream(dev)
            ctx_k = A.Context(local_rank, st_k.cuda_stream)
            ctx_k.debug_set_select(None if args.select < 0 else args.select)
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

    # ---- one rank of an 8-rank node: a child run of the same workload whose context gets 1/8 of the usable host cores
    # as its thread budget; this process is idle meanwhile.  The host half of a batch (candidate
    # bucketing, keypoint selection, libm) must still hide under the next batch's kernels. ----
    host_share_leg = None
    if rank == 0 and world == 1 and not args.no_host_share_leg and not args.host_share and not stub and not c5:
        try:
            torch.cuda.synchronize()
            cmd = [sys.executable, os.path.abspath(__file__), "--host-share", "8", "--steps", str(min(args.steps, 20)),
                   "--warmup", str(min(args.warmup, 5)), "--frames", str(F), "--width", str(W), "--height", str(H),
                   "--octaves", str(args.octaves), "--sublevels", str(args.sublevels), "--no-cpu-baseline", "--no-fed4k",
                   "--no-single", "--no-match", "--no-self-check", "--no-host-input", "--no-host-share-leg"] + (["--lean"] if args.lean else [])
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE")}
            p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
            ln = [x for x in p.stdout.splitlines() if x.startswith("{")]
            d = json.loads(ln[-1])
            host_share_leg = {"value": d["value"], "of_full_host": round(d["value"] / value, 4),
                              "cpus": d["config"]["placement"]["cpus"], "cpus_full": effective_cpus(),
                              "host_ms_in_finish_per_batch": d["config"]["host_ms_in_finish_per_batch"]}
        except Exception as e:
            host_share_leg = {"error": str(e)[:200]}

    # ---- --workload c5: what the job matched, and an untimed check of sampled pairs against the pairwise matcher ----
    c5_info = None
    if c5:
        steps_seen = max(1, xch["steps"])
        dist_job = host_sum(float(xch["dist_sum"]) / steps_seen)       # descriptor pairs whose distance a step forms, all ranks
        lists_all = host_allgather(float(xch["lists"]))
        matches_job = host_sum(float(xch["matches"]))
        # one more step, synchronously, keeping the gather and the pairs object
        ress = [ctx.extract_begin(bt, cfg, keep_all_planes=not args.lean, host_descriptors=False).finish() for bt in batches]
        gth = xch["comm"].gather_begin(ress, xch["cap"])
        if exchange == "external":
            gth.exchange_over()
        for r_ in ress:
            r_.close()
        pr = gth.match_all_pairs(match_ctx)
        base, block_rows, cnts_c, imgs_c = gth.finish()
        table = []
        for r_ in range(world):
            o = 0
            for n_ in gth.image_rows(r_):
                table.append((r_, o, n_))
                o += n_

        def rows_of(i):
            r_, o, n_ = table[i]
            # (torch.empty, not zeros: a fill kernel on the non-blocking stream is not ordered against the synchronous copy)
            t = torch.empty((max(n_, 1), 64), dtype=torch.uint8, device=dev)
            torch.cuda.current_stream().synchronize()
            if n_:
                A.copy_d2d(t.data_ptr(), base + ((r_ * block_rows + 1 + o) * 64), n_ * 64)
            return t[:n_]
        held = [(a, b) for (a, b) in pr.held(rank) if a < b]
        picks = held[:: max(1, len(held) // 3)][:3]
        ok_pairs = True
        for a, b in picks:
            ta, tb = rows_of(a), rows_of(b)
            for q, j, tq, tj in ((a, b, ta, tb), (b, a, tb, ta)):
                m_t, m_n = ctx.descriptor_match_device(tq, tj)
                torch.cuda.synchronize()
                exp = m_t[:int(m_n.item())].cpu().numpy().view(A.MATCH_DTYPE).reshape(-1)
                got = pr.matches(q, j)
                if not np.array_equal(got, exp):
                    ok_pairs = False
                    nd = int((got[:min(len(got), len(exp))] != exp[:min(len(got), len(exp))]).sum())
                    sys.stderr.write(f"rank {rank}: pair ({q}, {j}): {len(got)} matches held, {len(exp)} from the pairwise call, "
                                     f"{nd} of the common prefix differ; rows {len(tq)} x {len(tj)}\n")
        every = sorted((a, b) for a in range(pr.n_images) for b in range(pr.n_images) if a != b)
        holders_ok = all(0 <= pr.holder(a, b) < world for a, b in every[:: max(1, len(every) // 64)])
        n_images = pr.n_images
        pr.free()
        gth.free()
        ok_all = host_sum(0.0 if (ok_pairs and holders_ok) else 1.0) == 0.0
        c5_info = {"images_per_step": n_images, "unordered_image_pairs_per_step": n_images * (n_images - 1) // 2,
                   "descriptor_pairs_per_step": int(dist_job),
                   "Tdistances_per_s": round(dist_job * args.steps / elapsed / 1e12, 3),
                   "match_lists_held_per_rank": [int(v) for v in lists_all],
                   "matches_per_step": int(matches_job),
                   "pairs_check": {"sampled_pairs_per_rank": len(picks), "both_directions_equal_descriptor_match": bool(ok_all)},
                   "match_context": args.match_ctx}
        if not ok_all:
            raise SystemExit("bench.py --workload c5: a sampled pair's lists differ from akz_descriptor_match_device of the pair")

    # ---- CPU baseline: the oracle on the host cores, bounded sample, rank 0 at N=1 only -------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import akaze_ref as R
        cores = effective_cpus()  # affinity mask and cgroup quota, not the machine's core count
        cfg_ref = R.default_config()
        cfg_ref.num_sublevels, cfg_ref.max_octave_evolution = args.sublevels, args.octaves
        n_sample = 2 if c5 else 8  # ~2.5 s of host work on the GPU box (32 frames give the same figure in 10 s)
        t1 = time.perf_counter()
        kp_ref = 0
        for i in range(n_sample):
            r = R.extract(frames[i % F], cfg=cfg_ref, threads=cores)
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
        stage_ms["note"] = ("one untimed step with full stage profiling, un-pipelined and on ONE stream (the coarse octaves' chain is "
                            "not forked, so that every stage's time is its own); timed steps record FED and detector spans only")
        # algorithmic HBM bytes of each GPU stage of one step (SURVEY.md 8(d) model: every stage input read once, every
        # kept plane written once) against that stage's time in the profiled step: where the path stands kernel by kernel
        lv = A.plan_levels(W, H, cfg)
        px = [l["w"] * l["h"] * F for l in lv]
        half = [i > 0 and lv[i]["octave"] > lv[i - 1]["octave"] for i in range(len(lv))]
        stage_bytes = {
            "blur0": px[0] * (1 + 4),                                   # u8 in, f32 Lt0 out
            "contrast": px[0] * 4 * 2,                                  # two passes over Lt0
            # level preparation ([2x2 mean: 4 px in, Lt out,] Lsmooth, Lflow out) + diffusion: one line, because large levels run
            # both in one kernel (k_level_march)
            "prep+fed": sum((16 + 12 if half[i] else 4 + 8) * px[i] for i in range(1, len(lv))) +
                        sum(FED_BYTES_PER_PX_STEP * len(lv[i]["tau"]) * px[i] for i in range(1, len(lv))),
            "detector": sum(det_bpp * p for p in px),                   # Lsmooth in; Lx, Ly, Ldet (+ Lxx, Lyy, Lxy) out, each once
        }
        stage_t = dict(warm_prof)
        stage_t["prep+fed"] = warm_prof["prep"] + warm_prof["fed"]
        stage_roofline = {k: {"algorithmic_GB": round(b / 1e9, 3), "ms": round(stage_t[k], 3),
                              "achieved_GBps": round(b / 1e9 / (stage_t[k] * 1e-3), 1),
                              "frac": round(b / 1e9 / (stage_t[k] * 1e-3) / HBM_PEAK_GBS, 3)}
                          for k, b in stage_bytes.items() if stage_t[k] > 0}
        steps = max(1, args.steps)
        # THE whole-path byte model is SURVEY.md 8(d)'s: level 0 60 B/px; a level that continues its octave 64 + 12 n_l; the first
        # level of an octave 76 + 12 n_l (n_l diffusion steps) -- 575.8 B per input pixel at 4 x 4, 704.3 at 5 x 5, all planes kept
        whole = sum((60 if i == 0 else (76 if half[i] else 64) + 12 * len(lv[i]["tau"])) * px[i] for i in range(len(lv)))
        whole_gbs = whole / 1e9 / (elapsed / steps) if world == 1 else None
        stage_sum = sum(stage_bytes.values())
        stage_roofline["whole_path"] = {"model": "SURVEY.md 8(d)", "algorithmic_GB": round(whole / 1e9, 3),
                                        "B_per_input_pixel": round(whole / (float(W) * H * F), 1),
                                        "achieved_GBps": round(whole_gbs, 1) if whole_gbs else None,
                                        "frac": round(whole_gbs / HBM_PEAK_GBS, 3) if whole_gbs else None,
                                        "stage_sum_B_per_input_pixel": round(stage_sum / (float(W) * H * F), 1),
                                        "note": "SURVEY.md 8(d)'s bytes per input pixel x the step's pixels / the timed step (pipelined, keypoint "
                                                "kernels included).  The model counts every diffusion step as a pass over HBM and the "
                                                "derivative planes as written and read again: the kernels here fuse both, so the path can sit "
                                                "above 'its' roofline -- the DRAM figure is roofline.traffic_frac and profiles/*_pmc_per_kernel.txt.  "
                                                "stage_sum_B_per_input_pixel is the sum of the per-stage rows above (each stage's own inputs and "
                                                "outputs once), a smaller model kept for the per-stage fractions only"}
        out = {
            "metric": f"Mpix/s through extract_features ({args.octaves} oct x {args.sublevels} sub)" +
                      (" + exchange + cross-GPU all-pairs match" if c5 else ""),
            "value": round(value, 2), "unit": "Mpix/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / steps * 1e3, 3),
            **({"rehearsal": "all ranks share ONE GPU (--share-gpu): a functional run of the N > 1 path, not a scaling number"}
               if args.share_gpu else {}),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": (f"BASELINE configs[4]: {W}x{H} synthetic 8-bit luma frames resident in HBM, {args.octaves} octaves x "
                                    f"{args.sublevels} sublevels, 486-bit M-LDB, {F} frames per GPU per step (frame f on rank f mod N), "
                                    "descriptor rows stay on the device; per step: extraction, exchange of the rows (akz_gather_begin), "
                                    "all-pairs match of every frame of the job against every other (akz_match_all_pairs: every unordered "
                                    "pair once, both directions, on the rank that owns its lead frame), totals of all held lists read")
                                   if c5 else
                                   (f"{W}x{H} synthetic 8-bit luma frames resident in HBM, Config::default() "
                                    f"({args.octaves} octaves x {args.sublevels} sublevels, 486-bit M-LDB), {F} frames per GPU per step "
                                    "(BASELINE configs[1] frame shape; configs[3] sharding: one image per GPU slot)"),
                       "regions": {"n": args.regions, "reported": "median", "Mpix_s": region_values,
                                   "min": min(region_values), "max": max(region_values)},
                       "all_pairs": c5_info,
                       "frames_per_gpu": F, "width": W, "height": H, "batches_per_step": NP,
                       "pipelining": f"begin(batch j+{args.depth}) before finish(batch j) on one context, across steps" + (", finish half on the context's own thread" if args.eager else ""),
                       "planes": "lean" if args.lean else "all 10 EvolutionStep planes materialised",
                       "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                       "stream_placement": ctx.get_profile(reset=False)["placement"],
                       "input_ready_flag": not args.no_input_ready,
                       "exchange_ranks_seen": xch["ranks_seen"],
                       "placement": placement or None,
                       "share_gpu": bool(args.share_gpu),
                       "frame_sha256_16": frame_sha,
                       "host_share_8ranks_Mpix_s": host_share_leg,
                       "exchange": {"external": "the C ABI's blocks carried by the caller (akz_comm_create_external: akz_gather_begin -> block "
                                                "D2H -> gloo all-gather -> H2D -> akz_gather_deliver): --share-gpu rehearsal with every rank "
                                                "on device 0, or the fallback when RCCL cannot be initialised",
                                    "capi": "RCCL all-gather of descriptor rows through the C ABI (akz_gather_begin / _finish), "
                                            "one fixed-size collective per step, retired one step later",
                                    "torch": "RCCL all-gather of descriptor rows through torch.distributed, retired one step later",
                                    "none": "none (1 GPU)"}[xch["mode"]],
                       "exchange_ms_per_step": round((xch["host_ms"] + xch["wait_ms"]) / timed_steps, 3),
                       "exchange_host_ms_in_begin_per_step": round(xch["host_ms"] / timed_steps, 3),
                       "exchange_host_ms_waiting_per_step": round(xch["wait_ms"] / timed_steps, 3),
                       "per_rank_Mpix_s": [round(float(W) * H * F * args.steps / s / 1e6, 1) for s in per_rank_s],
                       "keypoints_per_step_rank0": nk,
                       "host_ms_in_begin_per_batch": round(host_ms_timed["begin"] / max(1, host_ms_timed["calls"]), 3),
                       "host_ms_in_finish_per_batch": round(host_ms_timed["finish"] / max(1, host_ms_timed["calls"]), 3)},
            "host_input": host_input,
            "roofline": roofline,
            "roofline_2": roofline_2,
            "fed_standalone": fed_alone,
            "detector_standalone": det_alone,
            "cpu_baseline": cpu,
            "self_check": self_check,
            "stage_ms_per_step": stage_ms,
            "stage_roofline": stage_roofline,
            "single_frame": single,
            "pair_4k": c3_leg,
            "stream_5x5_all_pairs": c5_leg,
            "match": match_leg,
        }
    # ---- N > 1: BASELINE configs[4] over ALL ranks as an extra leg (own clock, after the headline figures are complete): 4K
    # frames, 5 octaves x 5 sublevels, frame f on rank f mod N, descriptors stay on the device; per step extraction -> exchange
    # (akz_gather_begin) -> akz_match_all_pairs -> totals of every held list read.  Un-pipelined (one step at a time): the figure
    # of `--workload c5`, which pipelines the steps, is higher.  The driver's `bench.py --gpus N` thereby measures configs[4]
    # too.  A watchdog guards the headline: if the leg does not finish, rank 0 prints the line with the error in the leg's place and
    # every rank exits NON-zero (a hung collective or matcher is a failed run, whatever the line says).
    leg = None
    if world > 1 and not stub and not c5 and not args.no_c5_leg and xch["comm"] is not None and exchange in ("capi", "external"):
        import threading
        wd_lock = threading.Lock()
        wd_state = {"over": False}

        def give_up(why, status=3):
            # The leg cannot complete on this rank (it hangs, or it failed here while the peers sit in a collective).  The
            # headline figures were complete before the leg began: rank 0 still prints them, with the error -- and EVERY rank
            # then leaves with a non-zero status, without the runtime's exit handlers (a stuck collective would hold them).
            with wd_lock:
                if wd_state["over"]:
                    return
                wd_state["over"] = True
                if rank == 0:
                    out["config"]["all_pairs_c5_leg"] = {"error": why}
                    print(json.dumps(out), file=json_out, flush=True)
                sys.stderr.write(f"rank {rank}: all-pairs leg: {why}; exiting with status {status}\n")
                sys.stderr.flush()
                os._exit(status)
        wd = threading.Timer(args.c5_leg_timeout, give_up,
                             (f"given up after {args.c5_leg_timeout} s; the figures above were complete before it began",))
        wd.daemon = True
        wd.start()
        # local preparation first; whether it worked is agreed on by all ranks BEFORE the first collective of the leg, so that
        # a rank that cannot take part does not leave the others waiting in one
        prep_error = None
        try:
            n5, W5, H5, steps5 = max(1, args.c5_leg_frames), 3840, 2160, 3
            cfg55 = A.Config(num_sublevels=5, max_octave_evolution=5)
            d5 = torch.from_numpy(np.stack([A.synth_frame(W5, H5, i) for i in A.shard_frames(world * n5, rank, world)])).to(dev)
            torch.cuda.synchronize()
            ctx.set_profiling(0)

            def extract5():
                return ctx.extract_begin(d5, cfg55, keep_all_planes=not args.lean, input_ready=True, host_descriptors=False).finish()
            r5 = extract5()
            rows5 = sum(r5.counts(i)[1] for i in range(r5.num_images))
            r5.close()
        except Exception as e:
            prep_error = str(e)[:300]
        if host_max(0.0 if prep_error is None else 1.0) > 0.5:
            leg = {"error": f"skipped on every rank: preparation failed on at least one ({prep_error or 'another rank'})"}
        else:
            try:
                most = int(host_max(float(rows5)))
                cap5 = max(1024, (most + most // 4 + 4095) // 4096 * 4096)

                def step5():
                    r_ = extract5()
                    g_ = xch["comm"].gather_begin([r_], cap5)
                    if exchange == "external":
                        g_.exchange_over()
                    r_.close()
                    p_ = g_.match_all_pairs(ctx)
                    tot = p_.totals()  # waits for the launches, reads every held list's count
                    n_img = p_.n_images
                    p_.free()
                    g_.free()
                    return tot, n_img
                step5()
                barrier()
                t5 = time.perf_counter()
                for _ in range(steps5):
                    tot5, n_img5 = step5()
                barrier()
                el5 = host_max(time.perf_counter() - t5)
                dist5, match5 = host_sum(float(tot5[2])), host_sum(float(tot5[1]))
                lists5 = host_allgather(float(tot5[0]))
                leg = {"workload": f"BASELINE configs[4] over {world} ranks: {n5} 3840x2160 frames per rank per step, 5 octaves x 5 sublevels, "
                                   "descriptors on the device; extraction -> exchange -> akz_match_all_pairs -> totals read, one step at a "
                                   "time (un-pipelined; `--workload c5` pipelines)",
                       "Mpix_s": round(float(W5) * H5 * n5 * world * steps5 / el5 / 1e6, 1), "ms_per_step": round(el5 / steps5 * 1e3, 2),
                       "images_per_step": int(n_img5), "unordered_image_pairs_per_step": int(n_img5) * (int(n_img5) - 1) // 2,
                       "Tdistances_per_s": round(dist5 * steps5 / el5 / 1e12, 3), "matches_per_step": int(match5),
                       "match_lists_held_per_rank": [int(v) for v in lists5], "transport": exchange}
                del d5
            except Exception as e:
                # between collectives there is no way to tell the peers: this rank reports and leaves non-zero; theirs end
                # through the communicator's timeout or their own watchdog, non-zero as well
                give_up(f"failed on rank {rank}: {str(e)[:300]}")
        with wd_lock:  # the timer either has fired (and ended the process) or never will
            wd_state["over"] = True
            wd.cancel()
    if rank == 0:
        out["config"]["all_pairs_c5_leg"] = leg
        print(json.dumps(out), file=json_out, flush=True)
    if use_dist:
        dist.barrier()  # rank 0 may still be in its untimed extra legs: all ranks leave together
    if xch["comm"] is not None:
        xch["comm"].close()
    if use_dist:
        dist.destroy_process_group()
    return 0


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None:
        if args.gpus > 1:  # no launcher around us: be the launcher
            sys.exit(launch_ranks(args, argv))
    elif int(env_world) != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={env_world}: start bench.py with --gpus equal to the number of "
                         "ranks (or without WORLD_SIZE in the environment, and it starts the ranks itself)")
    try:
        rc = main_rank(args)
    except Exception as e:
        # An exchange that timed out (AKZ_ERR_TIMEOUT: a peer is missing or hung) leaves a collective on the communicator's
        # stream that may never complete; interpreter shutdown would run the GPU runtime's exit handlers behind it.  Report and
        # leave without them, non-zero.
        if getattr(e, "status", None) == -10:
            sys.stderr.write(f"bench.py: {e}\nbench.py: the exchange timed out; exiting with status 3 without runtime teardown\n")
            sys.stderr.flush()
            os._exit(3)
        raise
    sys.exit(rc)


if __name__ == "__main__":
    main()
