"""CPU tests of the N>1 path: the agent-major partition, the fusion plans
of every rank, and the feature all-gather itself over a real world_size-2 `gloo` process group."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from v2x_sim_amd.parallel import AgentShard, exchange_features


def test_partition_covers_every_item_once():
    A, Bt = 5, 8
    for world in (1, 2, 4, 8):
        rows, items = [], []
        for r in range(world):
            s = AgentShard(A, Bt, r, world)
            assert len(s.rows) == A * Bt // world
            rows += s.rows
            items += s.items
        assert rows == list(range(A * Bt))
        assert items == [(a, f) for a in range(A) for f in range(Bt)]  # agent-major, as the reference batches
    with pytest.raises(ValueError):
        AgentShard(5, 3, 0, 2)  # 15 items do not divide over 2 ranks


def test_fusion_plans_union_equals_single_rank_plan():
    A, Bt = 5, 4
    nat = torch.tensor([[5] * A, [3] * A, [2] * A, [5] * A])
    one = AgentShard(A, Bt, 0, 1).fusion_plan(nat, "cpu")
    for world in (2, 4):
        its, cf = [], []
        for r in range(world):
            p = AgentShard(A, Bt, r, world).fusion_plan(nat, "cpu")
            its.append(p["items"])
            cf.append(p["coef"])
            if p["local_rows"] is not None:
                assert p["local_rows"].numel() == p["n"]
        assert torch.equal(torch.cat(its), one["items"]) and torch.equal(torch.cat(cf), one["coef"])
    # padding agents (a >= count) are never fused; neighbours beyond the count never contribute
    assert one["n"] == 5 + 3 + 2 + 5
    m = one["items"].tolist().index([1, 2])
    assert one["coef"][m].tolist() == [1, 0, 0, 0, 0]
    with pytest.raises(RuntimeError, match="non-empty"):
        AgentShard(A, 1, 0, 1).fusion_plan(torch.tensor([[1] * A]), "cpu")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, A, Bt, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shard = AgentShard(A, Bt, rank, world)
        H, W, C = 4, 4, 8
        # every item's map is filled with its global row id (+ channel ramp) so misplacement shows
        local = torch.stack([torch.full((H, W, C), float(r)) + torch.arange(C) / 16.0 for r in shard.rows])
        for dtype in (torch.float32, torch.bfloat16):
            g = exchange_features(local.to(dtype), world)
            assert g.shape == (A * Bt, H, W, C) and g.dtype == dtype
            expect = torch.stack([torch.full((H, W, C), float(r)) + torch.arange(C) / 16.0 for r in range(A * Bt)])
            assert torch.equal(g.float(), expect.to(dtype).float()), "gathered maps are not in agent-major row order"
        # the neighbour map of (agent j, frame f) sits at row j*Bt + f on every rank
        for (a, f) in shard.items:
            for j in range(A):
                assert float(g[j * Bt + f, 0, 0, 0]) == float(j * Bt + f)
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_all_gather_exchange_gloo_world2():
    world, A, Bt = 2, 5, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, A, Bt, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_exchange_world1_is_identity():
    t = torch.randn(3, 2, 2, 8)
    assert exchange_features(t, 1) is t
