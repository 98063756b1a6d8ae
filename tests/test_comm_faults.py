"""akz_comm.cpp against a stand-in for librccl.so.1 (tests/stub_rccl/stub_rccl.cpp, built here): what the real library
cannot show on a one-GPU box.  (a) the communicator's RCCL code path with 2 and 3 RANKS -- id hand-over, ncclCommInitRank
with nranks > 1, the fixed-size all-gather, headers, per-image tables, the overflow protocol, akz_match_all_pairs' lead rule
-- by `gather_selftest RANK NRANKS ID_FILE 0`, every rank on device 0, the stub carrying the blocks between the processes;
(b) fault paths: communicator creation fails, the collective returns an error, a symbol is missing, a peer never arrives --
every rank ends non-zero with the library's message, none hangs.  The hosts are C++ (no PyTorch in the process, so the
stub is the only librccl there)."""
import os
import subprocess
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "akaze-rust_amd", "bin", "gather_selftest")
SRC = os.path.join(ROOT, "tests", "stub_rccl", "stub_rccl.cpp")


def _build(dirpath, *defs):
    os.makedirs(dirpath, exist_ok=True)
    out = os.path.join(dirpath, "librccl.so.1")
    cmd = ["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", *defs, "-o", out, SRC,
           "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-Wl,-soname,librccl.so.1", "-Wl,-rpath,/opt/rocm/lib"]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    return out


def test_stub_rccl_builds_and_exports_what_akz_comm_binds(tmp_path):
    lib = _build(str(tmp_path / "full"))
    syms = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True).stdout
    for s in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclAllGather", "ncclGetErrorString"):
        assert f" T {s}" in syms
    lib2 = _build(str(tmp_path / "noag"), "-DSTUB_NO_ALLGATHER")
    assert " T ncclAllGather" not in subprocess.run(["nm", "-D", "--defined-only", lib2], capture_output=True, text=True).stdout


def _ranks(stub_dir, world, tmp_path, mode="ok", only=None, timeout_s="30", limit=300, extra_env=None):
    env = dict(os.environ, LD_LIBRARY_PATH=stub_dir + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""), AKZ_STUB_RCCL=mode,
               AKZ_STUB_RCCL_TIMEOUT_S=timeout_s, **(extra_env or {}))
    idf = str(tmp_path / f"id_{mode}_{world}_{time.monotonic_ns()}")
    procs = [subprocess.Popen([BIN, str(r), str(world), idf, "0"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in (range(world) if only is None else only)]
    outs = []
    t0 = time.monotonic()
    try:
        for p in procs:
            outs.append(p.communicate(timeout=limit)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return [p.returncode for p in procs], outs, time.monotonic() - t0


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 2, 3])
def test_rccl_code_path_with_several_ranks_on_one_gpu(tmp_path, world):
    stub = os.path.dirname(_build(str(tmp_path / "stub")))
    rcs, outs, _ = _ranks(stub, world, tmp_path)
    assert rcs == [0] * world, "\n".join(outs)
    for r in range(world):
        assert f"gather selftest ok: rank {r} of {world}" in outs[r]


@pytest.mark.gpu
def test_comm_fault_paths_end_with_a_message_not_a_hang(tmp_path):
    stub = os.path.dirname(_build(str(tmp_path / "stub")))
    # communicator creation fails on every rank
    rcs, outs, dt = _ranks(stub, 2, tmp_path, mode="init_fail")
    assert all(rc != 0 for rc in rcs) and all("ncclCommInitRank failed" in o for o in outs), outs
    # the collective returns an error on every rank (the first exchange of the self-test)
    rcs, outs, dt = _ranks(stub, 2, tmp_path, mode="allgather_fail")
    assert all(rc != 0 for rc in rcs) and all("AllGather" in o and "failed" in o for o in outs), outs
    # a librccl without a symbol the library binds: refused when the communicator id is asked for / the communicator is made
    noag = os.path.dirname(_build(str(tmp_path / "noag"), "-DSTUB_NO_ALLGATHER"))
    rcs, outs, dt = _ranks(noag, 1, tmp_path)
    assert rcs[0] != 0 and "RCCL symbol missing: ncclAllGather" in outs[0], outs
    # a peer that never arrives: rank 0 of 2 alone gives up (the stub's collective init times out), it does not hang
    rcs, outs, dt = _ranks(stub, 2, tmp_path, only=[0], timeout_s="3", limit=120)
    assert rcs[0] != 0 and "ncclCommInitRank failed" in outs[0] and dt < 60, (outs, dt)


@pytest.mark.gpu
def test_a_peer_lost_after_init_ends_the_rank_with_an_error_within_the_timeout(tmp_path):
    """The collective is enqueued and never completes (the stub keeps the stream busy for 25 s behind a sleeping host
    function -- what a rank sees when its peer dies mid-job).  With akz_comm_set_timeout(2 s) akz_gather_finish gives up
    with AKZ_ERR_TIMEOUT, the communicator is abandoned: akz_comm_destroy returns at once (it must not synchronise the
    stuck stream, free device memory or destroy the RCCL communicator) and the host exits non-zero -- long before the
    25 s are over.  Both a stall of the very first exchange and one after several good ones (pooled gathers in flight)."""
    stub = os.path.dirname(_build(str(tmp_path / "stub")))
    for after in (0, 3):
        rcs, outs, dt = _ranks(stub, 1, tmp_path, mode=f"stall_after_{after}", limit=120,
                               extra_env=dict(AKZ_SELFTEST_TIMEOUT_S="2", AKZ_STUB_RCCL_STALL_S="25"))
        assert rcs[0] == 3, outs
        assert "did not complete within the communicator's timeout" in outs[0] and "exiting with status 3" in outs[0], outs
        assert dt < 15, (dt, outs)  # 2 s timeout + start-up; nothing waited for the 25 s stall
