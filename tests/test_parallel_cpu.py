"""CPU, world_size 2 over gloo: the N>1 host logic (unit partition, ragged gather, depth-slab halo
exchange, statistics all-reduce, depth gather) exactly as the GPU ranks run it over RCCL."""
import importlib
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

P = importlib.import_module("video-to-video-diffusion_amd.parallel")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        comm = P.DistComm()
        assert (comm.rank, comm.world) == (rank, world)
        # one volume of depth 6 cut into `world` slabs, bf16 activations with a halo slice on each side
        D, H, W, C = 6, 3, 4, 8
        full = torch.arange(D * H * W * C, dtype=torch.float32).reshape(D, H * W * C) / 7.0
        slab = P.depth_slab(D, rank, world)
        dl = len(slab)
        buf = torch.full(((dl + 2), H * W * C), -1.0).to(torch.bfloat16)
        buf[1:dl + 1] = full[slab.start:slab.stop].to(torch.bfloat16)
        flat = buf.reshape(-1)
        se = H * W * C
        comm.exchange(rank, flat[se:2 * se], flat[dl * se:(dl + 1) * se], flat[0:se], flat[(dl + 1) * se:(dl + 2) * se])
        want = torch.zeros((dl + 2), H * W * C)
        lo, hi = slab.start - 1, slab.stop + 1
        for i, dd in enumerate(range(lo, hi)):
            if 0 <= dd < D:
                want[i] = full[dd]
        assert torch.equal(buf.float(), want.to(torch.bfloat16).float()), rank
        # GroupNorm statistics: fp64 (sum, sumsq) all-reduced over the ranks
        sums = torch.tensor([float(rank + 1), 10.0 * (rank + 1)], dtype=torch.float64)
        comm.all_reduce(rank, sums)
        tot = sum(range(1, world + 1))
        assert sums.tolist() == [float(tot), 10.0 * tot]
        # the coalesced sync point of the sharded programs: boundary slices + fp64 statistics + fp32 depth sum together
        buf2 = torch.full(((dl + 2), H * W * C), -1.0).to(torch.bfloat16)
        buf2[1:dl + 1] = (2.0 * full[slab.start:slab.stop]).to(torch.bfloat16)
        f2 = buf2.reshape(-1)
        sums2 = torch.tensor([1.5 * (rank + 1), -2.0 * (rank + 1)], dtype=torch.float64)
        dsum = torch.full((5,), float(rank + 1))
        comm.exchange(rank, f2[se:2 * se], f2[dl * se:(dl + 1) * se], f2[0:se], f2[(dl + 1) * se:(dl + 2) * se],
                      sums=sums2, f32=dsum)
        assert torch.equal(buf2.float(), (2.0 * want).to(torch.bfloat16).float()), rank
        assert sums2.tolist() == [1.5 * tot, -2.0 * tot] and dsum.tolist() == [float(tot)] * 5
        comm.exchange(rank, None, None, None, None, sums=sums2)         # statistics only
        assert sums2.tolist() == [1.5 * tot * world, -2.0 * tot * world]
        # result gather along depth
        vol = full.reshape(1, 1, D, H * W, C)
        got = comm.gather_depth(rank, vol[:, :, slab.start:slab.stop].clone())
        assert torch.equal(got, vol)
        # ragged slabs (depth 7 over the ranks): every rank sends max(counts) slices, the gather trims them
        vol7 = torch.arange(7 * 5, dtype=torch.float32).reshape(1, 1, 7, 5, 1)
        sl7 = P.depth_slab(7, rank, world)
        counts = [len(P.depth_slab(7, r, world)) for r in range(world)]
        assert torch.equal(comm.gather_depth(rank, vol7[:, :, sl7.start:sl7.stop].clone(), counts=counts), vol7)
        # data parallel: 5 independent volumes over 2 ranks -> 3 + 2, gathered back in order
        units = P.shard_units(5, rank, world)
        counts = [len(P.shard_units(5, r, world)) for r in range(world)]
        local = torch.stack([torch.full((2, 2), float(u)) for u in units])
        allv = P.gather_units(local, counts)
        assert allv[:, 0, 0].tolist() == [0.0, 1.0, 2.0, 3.0, 4.0]
        # data-parallel training: bucketed gradient all-reduce (ragged tensors, several buckets, averaged)
        params = [torch.nn.Parameter(torch.zeros(shape)) for shape in ((7, 3), (5,), (2, 2, 2), (1,), (33,))]
        for i, p_ in enumerate(params):
            p_.grad = torch.full(p_.shape, float((rank + 1) * (i + 1)))
        params.append(torch.nn.Parameter(torch.zeros(3)))        # a parameter without gradient is skipped
        nb = P.allreduce_gradients(params, bucket_bytes=64)
        assert nb >= 3
        mean_rank = sum(range(1, world + 1)) / world
        for i, p_ in enumerate(params[:-1]):
            assert torch.allclose(p_.grad, torch.full(p_.shape, mean_rank * (i + 1))), (rank, i)
        assert params[-1].grad is None
        # sliding-window stitching: window data parallelism is opt-in.  With torch.distributed initialised but no
        # dp_group, every rank keeps ALL windows (ranks may be validating different volumes, or only rank 0 may be
        # calling); with dp_group=True the windows are split; depth sharding + dp_group is refused.
        S = importlib.import_module("video-to-video-diffusion_amd.sampler")
        wins_by_rank = [[(0, 0, w) for w in range(3 + rank)]][0]      # a different volume per rank
        mine, w_, g_ = S._window_partition(wins_by_rank, None, object())
        assert mine == wins_by_rank and w_ == 1 and g_ is None
        same = [(0, h, w) for h in range(3) for w in range(3)]
        mine, w_, g_ = S._window_partition(same, True, object())
        assert w_ == world and mine == [same[i] for i in P.shard_units(len(same), rank, world)]

        class _Shard:
            depth_shard_comm = comm

        class _Sampler:
            model = _Shard()

        with pytest.raises(P.CtsiError, match="cannot be combined with depth sharding"):
            S._window_partition(same, True, _Sampler())
        with open(os.path.join(out_dir, f"ok{rank}"), "w") as f:
            f.write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_distributed_collectives_gloo(tmp_path, world):
    port = _free_port()
    mp.start_processes(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def test_window_partition_without_process_group():
    S = importlib.import_module("video-to-video-diffusion_amd.sampler")
    wins = [(0, 0, 0), (0, 0, 8)]
    assert S._window_partition(wins, None, object()) == (wins, 1, None)
    with pytest.raises(P.CtsiError, match="initialised torch.distributed"):
        S._window_partition(wins, True, object())


def test_partitions():
    assert [list(P.shard_units(10, r, 4)) for r in range(4)] == [[0, 1, 2], [3, 4, 5], [6, 7], [8, 9]]
    assert [len(P.shard_units(3, r, 8)) for r in range(8)] == [1, 1, 1, 0, 0, 0, 0, 0]
    assert list(P.depth_slab(48, 7, 8)) == list(range(42, 48))
    # ragged slabs: the reference's real volumes are 50 thick -> 300 thin slices, 300 is not a multiple of 8
    assert [len(P.depth_slab(300, r, 8)) for r in range(8)] == [38, 38, 38, 38, 37, 37, 37, 37]
    assert P.depth_slab(300, 4, 8).start == 152 and P.depth_slab(300, 7, 8).stop == 300
    spec = P.ShardSpec(5, 8, None, 300)
    assert (spec.depth_local, spec.depth_start, sum(spec.depth_counts)) == (37, 189, 300)
    with pytest.raises(P.CtsiError):
        P.depth_slab(5, 0, 8)


def test_local_comm_matches_dist_semantics():
    """LocalComm (virtual ranks, used for single-GPU parity tests) == the gloo behaviour above."""
    world, se = 3, 4
    comm = P.LocalComm(world)
    bufs = [torch.full((4 * se,), -1.0) for _ in range(world)]
    for r, b in enumerate(bufs):
        b[se:3 * se] = torch.arange(2 * se, dtype=torch.float32) + 100 * r
    for r, b in enumerate(bufs):
        comm.exchange(r, b[se:2 * se], b[2 * se:3 * se], b[0:se], b[3 * se:4 * se])
    assert torch.equal(bufs[0][0:se], torch.zeros(se)) and torch.equal(bufs[2][3 * se:], torch.zeros(se))
    assert torch.equal(bufs[1][0:se], bufs[0][2 * se:3 * se]) and torch.equal(bufs[1][3 * se:], bufs[2][se:2 * se])
    vals = [torch.tensor([float(r)]) for r in range(world)]
    for r, v in enumerate(vals):
        comm.all_reduce(r, v)
    assert [v.item() for v in vals] == [3.0, 3.0, 3.0]
    # coalesced sync point: slices + fp64 statistics + fp32 buffer in one call
    sums = [torch.tensor([1.0 + r, 2.0], dtype=torch.float64) for r in range(world)]
    f32 = [torch.full((3,), float(r)) for r in range(world)]
    for r, b in enumerate(bufs):
        b[0:se] = -7.0
        b[3 * se:] = -7.0
        comm.exchange(r, b[se:2 * se], b[2 * se:3 * se], b[0:se], b[3 * se:4 * se], sums=sums[r], f32=f32[r])
    assert torch.equal(bufs[1][0:se], bufs[0][2 * se:3 * se]) and torch.equal(bufs[0][0:se], torch.zeros(se))
    assert all(s_.tolist() == [6.0, 6.0] for s_ in sums) and all(f.tolist() == [3.0] * 3 for f in f32)
