"""N>1 path on CPU: two gloo ranks shard a global batch, score their shard, all-gather.  The model callable here is the
ORACLE (tests may use it); what is under test is vtamiq_amd.dist's sharding and gather order."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, global_batch, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import vtamiq_oracle as O
    from vtamiq_amd import synth
    from vtamiq_amd.dist import shard_range, sharded_forward
    from vtamiq_amd.spec import make_spec
    torch.set_num_threads(2)
    spec = make_spec(dict(variant="ViT-B16", num_keep_layers=1), num_rgs=1, num_rcabs=1)
    sd = O.to_torch(synth.make_state_dict(spec, 9))
    patches, pos, _ = synth.make_inputs(spec, global_batch, 12, 99)
    lo, hi = shard_range(global_batch, rank, world)
    tp, tq = torch.from_numpy(patches[lo:hi]), torch.from_numpy(pos[lo:hi])
    model = lambda p, q, s: O.vtamiq_forward(sd, spec, p, q, s)
    qg = sharded_forward(model, (tp[:, 0], tp[:, 1]), (tq[:, 0], tq[:, 1]), (None, None), global_batch)
    np.save(os.path.join(out_dir, f"q{rank}.npy"), qg.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("global_batch", [4, 5])
def test_two_rank_shard_and_gather(tmp_path, global_batch):
    world = 2
    port = 29600 + (os.getpid() % 200) + global_batch
    mp.spawn(_worker, args=(world, port, global_batch, str(tmp_path)), nprocs=world, join=True)
    from oracle import vtamiq_oracle as O
    from vtamiq_amd import synth
    from vtamiq_amd.spec import make_spec
    spec = make_spec(dict(variant="ViT-B16", num_keep_layers=1), num_rgs=1, num_rcabs=1)
    sd = O.to_torch(synth.make_state_dict(spec, 9))
    patches, pos, _ = synth.make_inputs(spec, global_batch, 12, 99)
    tp, tq = torch.from_numpy(patches), torch.from_numpy(pos)
    want = O.vtamiq_forward(sd, spec, (tp[:, 0], tp[:, 1]), (tq[:, 0], tq[:, 1]), (None, None))[0].numpy()
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), f"q{r}.npy"))
        assert got.shape == (global_batch,)
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-7)


def _worker8(rank, world, port, global_batch, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.bench_stub import StubModel
    from vtamiq_amd.dist import shard_range, sharded_forward
    torch.set_num_threads(1)
    g = torch.Generator().manual_seed(5)
    ref = torch.rand(global_batch, 3, 3, 4, 4, generator=g)
    dst = torch.rand(global_batch, 3, 3, 4, 4, generator=g)
    pos = torch.rand(global_batch, 3, 2, generator=g)
    lo, hi = shard_range(global_batch, rank, world)
    qg = sharded_forward(StubModel(), (ref[lo:hi], dst[lo:hi]), (pos[lo:hi], pos[lo:hi]), (None, None), global_batch)
    if rank in (0, world - 1):
        np.save(os.path.join(out_dir, f"q{rank}.npy"), qg.numpy())
        if rank == 0:
            np.save(os.path.join(out_dir, "want.npy"), StubModel()((ref, dst), (pos, pos), (None, None))[0].numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("global_batch", [256, 257])
def test_eight_rank_shard_and_gather(tmp_path, global_batch):
    """BASELINE configs[2] in shape: a global batch over 8 ranks, even (256 = 8 x 32) and uneven (257: rank 0 owns one pair more and the
    other ranks' shards are padded inside the fixed-size all-gather); every rank ends with all scores in global pair order."""
    world = 8
    mp.spawn(_worker8, args=(world, 29300 + (os.getpid() % 200) + global_batch % 7, global_batch, str(tmp_path)), nprocs=world, join=True)
    want = np.load(os.path.join(str(tmp_path), "want.npy"))
    for r in (0, world - 1):
        got = np.load(os.path.join(str(tmp_path), f"q{r}.npy"))
        assert got.shape == (global_batch,)
        np.testing.assert_array_equal(got, want)


def test_shard_ranges_cover():
    from vtamiq_amd.dist import shard_range
    for gb in (1, 7, 8, 256, 257):
        for w in (1, 2, 4, 8):
            spans = [shard_range(gb, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == gb
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def _solo_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=0, world_size=1)
    from vtamiq_amd.dist import gather_scores
    q = torch.arange(5, dtype=torch.float32)
    same = gather_scores(q, 5)                                   # world of one: nothing to gather
    forced = gather_scores(q, 5, force_collective=True)          # the collective itself on one rank
    np.save(os.path.join(out_dir, "solo.npy"), np.stack([same.numpy(), forced.numpy()]))
    assert same.data_ptr() == q.data_ptr() and forced.data_ptr() != q.data_ptr()
    dist.destroy_process_group()


def test_single_rank_forced_collective(tmp_path):
    """bench.py's world-size-1 RCCL block: gather_scores(force_collective=True) runs the all-gather on ONE rank (gloo here)."""
    mp.spawn(_solo_worker, args=(1, 29850 + (os.getpid() % 100), str(tmp_path)), nprocs=1, join=True)
    got = np.load(os.path.join(str(tmp_path), "solo.npy"))
    np.testing.assert_array_equal(got[0], np.arange(5, dtype=np.float32))
    np.testing.assert_array_equal(got[1], got[0])
