"""CPU: first-contact insurance for the RCCL transport (csrc/comm.hip).  No multi-GPU node is available to the build, so
the multi-rank code path of `ctsi_halo_exchange_reduce` / `ctsi_comm_allgather` is driven here, rank by rank, against a
recording stand-in for librccl (tests/stubs/rccl_stub.c, loaded through CTSI_RCCL_LIB; LD_PRELOADed as well so the
volume-end memsets are recorded instead of needing a device).  For rank in {0, 3, 7} of 8 the log must show:
peers r-1 / r+1 only, byte counts = the slice bytes, exactly ONE group per sync point with every transfer inside it,
the statistics all-reduce in that same group, and memsets only towards a volume end."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "video-to-video-diffusion_amd", "libctsi.so")
NCCL_UINT8, NCCL_F32, NCCL_F64 = 1, 7, 8

DRIVER = textwrap.dedent("""
    import ctypes as C, sys
    lib = C.CDLL(sys.argv[1])
    rank, world = int(sys.argv[2]), int(sys.argv[3])
    lib.ctsi_last_error.restype = C.c_char_p
    vp = C.c_void_p
    lib.ctsi_comm_init.argtypes = [C.POINTER(vp), vp, C.c_int, C.c_int]
    lib.ctsi_halo_exchange_reduce.argtypes = [vp, vp, vp, vp, vp, C.c_size_t, vp, C.c_int, vp, C.c_longlong, vp]
    lib.ctsi_halo_exchange.argtypes = [vp, vp, vp, vp, vp, C.c_size_t, vp]
    lib.ctsi_gn_allreduce.argtypes = [vp, vp, C.c_int, vp, C.c_longlong, vp]
    lib.ctsi_comm_allgather.argtypes = [vp, vp, vp, C.c_size_t, vp]
    lib.ctsi_comm_destroy.argtypes = [vp]
    uid = (C.c_ubyte * 128)()
    assert lib.ctsi_comm_unique_id(C.cast(uid, vp)) == 0, lib.ctsi_last_error()
    h = vp()
    assert lib.ctsi_comm_init(C.byref(h), C.cast(uid, vp), rank, world) == 0, lib.ctsi_last_error()
    assert lib.ctsi_comm_rank(h) == rank and lib.ctsi_comm_world(h) == world
    ST = 0x7700
    # sync point 1: statistics + raw boundary slices + the attention depth sum (GroupNorm / attention sync)
    rc = lib.ctsi_halo_exchange_reduce(h, 0x1000, 0x2000, 0x3000, 0x4000, 4194304, 0x5000, 64, 0x6000, 1048576, ST)
    print("rc1", rc, lib.ctsi_last_error())
    # sync point 2: a plain halo exchange (conv_in / Downsample / Upsample outputs)
    print("rc2", lib.ctsi_halo_exchange(h, 0x1100, 0x2100, 0x3100, 0x4100, 12582912, ST))
    # sync point 3: statistics only
    print("rc3", lib.ctsi_gn_allreduce(h, 0x5100, 16, None, 0, ST))
    # result gather
    print("rc4", lib.ctsi_comm_allgather(h, 0x8000, 0x9000, 786432, ST))
    lib.ctsi_comm_destroy(h)
""")


@pytest.fixture(scope="module")
def stub(tmp_path_factory):
    d = tmp_path_factory.mktemp("rccl_stub")
    so = d / "librccl_stub.so"
    subprocess.run(["gcc", "-shared", "-fPIC", "-O1", "-o", str(so), os.path.join(ROOT, "tests", "stubs", "rccl_stub.c")],
                   check=True)
    return d, so


def _run(stub, rank, world, extra_env=None):
    d, so = stub
    log = d / f"log_{rank}_{world}_{len(extra_env or {})}.txt"
    if log.exists():
        log.unlink()
    env = dict(os.environ, CTSI_RCCL_LIB=str(so), LD_PRELOAD=str(so), RCCL_STUB_LOG=str(log))
    env.update(extra_env or {})
    proc = subprocess.run([sys.executable, "-c", DRIVER, LIB, str(rank), str(world)], env=env, capture_output=True,
                          text=True, timeout=120)
    assert proc.returncode == 0, proc.stdout + proc.stderr
    recs = []
    for line in log.read_text().splitlines():
        kind, *kv = line.split()
        recs.append((kind, dict(x.split("=", 1) for x in kv if "=" in x)))
    return proc.stdout, recs


def _sync_points(recs):
    """Split the log into sync points: memsets issued right before a group belong to it."""
    points, cur = [], []
    for kind, kv in recs:
        if kind in ("uid", "init", "destroy", "allgather", "memcpy"):
            continue
        cur.append((kind, kv))
        if kind == "group_end":
            points.append(cur)
            cur = []
    assert not cur, f"calls outside any group: {cur}"
    return points


@pytest.mark.parametrize("rank", [0, 3, 7])
def test_sync_points_of_one_rank_of_eight(stub, rank):
    world = 8
    out, recs = _run(stub, rank, world)
    assert "rc1 0" in out and "rc2 0" in out and "rc3 0" in out and "rc4 0" in out, out
    assert ("init", dict(rank=str(rank), world=str(world))) in recs
    points = _sync_points(recs)
    assert len(points) == 3, "exactly one RCCL group per sync point"
    expect = [dict(bytes=4194304, lo_own="0x1000", hi_own="0x2000", lo_halo="0x3000", hi_halo="0x4000",
                   reduces=[("0x5000", 64, NCCL_F64), ("0x6000", 1048576, NCCL_F32)]),
              dict(bytes=12582912, lo_own="0x1100", hi_own="0x2100", lo_halo="0x3100", hi_halo="0x4100", reduces=[]),
              dict(bytes=0, reduces=[("0x5100", 16, NCCL_F64)])]
    for pt, ex in zip(points, expect):
        kinds = [k for k, _ in pt]
        assert kinds.count("group_start") == 1 and kinds.count("group_end") == 1
        gs, ge = kinds.index("group_start"), kinds.index("group_end")
        assert pt[gs][1]["depth"] == "0" and pt[ge][1]["depth"] == "0", "groups are not nested"
        inside = pt[gs + 1:ge]
        before = pt[:gs]
        # memsets: only towards a volume end, before the group, of exactly one slice
        memsets = [kv for k, kv in before if k == "memset"]
        assert all(k == "memset" for k, _ in before)
        want = []
        if ex["bytes"]:
            if rank == 0:
                want.append(ex["lo_halo"])
            if rank == world - 1:
                want.append(ex["hi_halo"])
        assert sorted(m["ptr"] for m in memsets) == sorted(want)
        assert all(m["bytes"] == str(ex["bytes"]) and m["value"] == "0" and m["stream"] == "0x7700" for m in memsets)
        # transfers: neighbours only, slice bytes, uint8, all inside the group and on the caller's stream
        sends = [kv for k, kv in inside if k == "send"]
        recvs = [kv for k, kv in inside if k == "recv"]
        peers = ([rank - 1] if rank > 0 else []) + ([rank + 1] if rank < world - 1 else [])
        if not ex["bytes"]:
            peers = []
        assert sorted(int(s["peer"]) for s in sends) == peers and sorted(int(r["peer"]) for r in recvs) == peers
        for s in sends:
            assert s["ptr"] == (ex["lo_own"] if int(s["peer"]) == rank - 1 else ex["hi_own"])
        for r in recvs:
            assert r["ptr"] == (ex["lo_halo"] if int(r["peer"]) == rank - 1 else ex["hi_halo"])
        for x in sends + recvs:
            assert x["count"] == str(ex["bytes"]) and x["dtype"] == str(NCCL_UINT8) and x["grouped"] == "1"
            assert x["stream"] == "0x7700"
        reds = [kv for k, kv in inside if k == "allreduce"]
        assert [(r["src"], int(r["count"]), int(r["dtype"])) for r in reds] == ex["reduces"]
        assert all(r["src"] == r["dst"] and r["op"] == "0" and r["grouped"] == "1" for r in reds)
    # the result gather: one call, slab bytes, outside any group
    ag = [kv for k, kv in recs if k == "allgather"]
    assert len(ag) == 1 and ag[0]["count"] == "786432" and ag[0]["grouped"] == "0" and ag[0]["dtype"] == str(NCCL_UINT8)


def test_failed_call_inside_a_group_closes_the_group(stub):
    """an error between ncclGroupStart and ncclGroupEnd must not leave the thread's group open."""
    out, recs = _run(stub, 3, 8, extra_env={"RCCL_STUB_FAIL_SEND": "0"})
    assert "rc1 -" in out and "stub failure" in out, out          # the first sync point reports the failure ...
    kinds = [k for k, _ in recs]
    first_end = kinds.index("group_end")
    assert kinds.index("send") < first_end and recs[first_end][1]["depth"] == "0"   # ... with its group closed
    assert "rc2 0" in out and "rc3 0" in out                     # and later sync points are issued normally
    assert kinds.count("group_start") == kinds.count("group_end") == 3


def test_world_two_has_one_neighbour_each(stub):
    for rank in (0, 1):
        _, recs = _run(stub, rank, 2)
        sends = [int(kv["peer"]) for k, kv in recs if k == "send"]
        assert sends == [1 - rank, 1 - rank]        # two exchanging sync points, one neighbour
        assert sum(1 for k, _ in recs if k == "memset") == 2
