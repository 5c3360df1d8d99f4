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


def _spawn(target, world, *args, timeout=240):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port) + args + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=timeout) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, "ok") for r in range(world)], res


@pytest.mark.parametrize("world,Bt", [(2, 2), (4, 4), (8, 8)])
def test_all_gather_exchange_gloo(world, Bt):
    """world 2, 4 and 8 real gloo ranks (nothing above world 2 had run); 5 agents, Bt frames."""
    _spawn(_worker, world, 5, Bt)


def test_exchange_world1_is_identity():
    t = torch.randn(3, 2, 2, 8)
    assert exchange_features(t, 1) is t


# ------------------------------------------------------------------ point-to-point transports (SURVEY.md 8e "comm-sparsity path")
def _check_plans(plans, needs, per_rank):
    world = len(plans)
    for r in range(world):
        got = []
        for src, lo, hi in plans[r]["recv"]:
            assert lo // per_rank == (hi - 1) // per_rank == src != r, "a range crosses an owner boundary"
            got += list(range(lo, hi))
        own = set(range(r * per_rank, (r + 1) * per_rank))
        assert sorted(got) == sorted(set(needs[r]) - own), "rank %d does not receive exactly what it needs" % r
        assert plans[r]["rows"] == len(got)
        for dst in range(world):  # pairwise: src's send list == dst's recv list, same order
            s = [(lo, hi) for d, lo, hi in plans[r]["send"] if d == dst]
            q = [(lo, hi) for sr, lo, hi in plans[dst]["recv"] if sr == r]
            assert s == q and s == sorted(s)


def test_row_exchange_plan_v2vnet_needed_rows():
    from v2x_sim_amd.parallel import ShardedV2VNet
    A = 5
    for world, Bt in ((2, 2), (4, 4), (8, 8), (8, 64)):
        class _M:
            gnn_iter_num, neighbor_source = 1, "initial"
        rn = ShardedV2VNet(_M(), AgentShard(A, Bt, 0, world), transport="needed")
        plans = rn.needed_plan()
        sh = rn.shard
        needs = []
        for r in range(world):
            fr = {row % Bt for row in range(r * sh.per_rank, (r + 1) * sh.per_rank)}
            needs.append([j * Bt + f for f in fr for j in range(A)])
        _check_plans(plans, needs, sh.per_rank)
        dense = (world - 1) * sh.per_rank
        assert all(p["rows"] <= dense for p in plans)
        if world == 8 and Bt == 64:  # 40 rows per rank cover 40 frames of one agent: 4 x 40 foreign maps instead of 7 x 40
            assert max(p["rows"] for p in plans) == 160 < dense == 280
    # ragged frame: agents beyond the count are neither egos nor sources
    rn = ShardedV2VNet(_M(), AgentShard(A, 2, 0, 2), transport="needed")
    plans = rn.needed_plan(counts=[5, 3])
    assert not any(lo <= 3 * 2 + 1 < hi or lo <= 4 * 2 + 1 < hi for p in plans for _, lo, hi in p["recv"])


def test_row_exchange_plan_when2com_sparsity():
    from v2x_sim_amd.parallel import plan_row_exchange, when2com_needs
    A, Bt, world = 5, 8, 4
    g = torch.Generator().manual_seed(3)
    coef = (torch.rand(Bt, A, A, generator=g) > 0.7).float() * torch.rand(Bt, A, A, generator=g)
    counts = [5, 5, 4, 5, 2, 5, 5, 3]
    sh = AgentShard(A, Bt, 0, world)
    needs = when2com_needs(sh, coef, counts)
    plans = plan_row_exchange(needs, sh.per_rank)
    _check_plans(plans, needs, sh.per_rank)
    # every needed row is justified by a non-zero coefficient of an ego the rank owns; nothing else travels
    for r in range(world):
        for row in needs[r]:
            k, f = divmod(row, Bt)
            egos = [q for q in range(counts[f]) if (q * Bt + f) // sh.per_rank == r and q != k]
            assert k < counts[f] and any(float(coef[f, k, q]) != 0 for q in egos)
    who2com = torch.zeros(Bt, A, A)
    who2com[:, 0, :] = 1.0                     # everybody listens to agent 0 only
    plans = plan_row_exchange(when2com_needs(sh, who2com, [A] * Bt), sh.per_rank)
    assert sum(p["rows"] for p in plans) == sum(len({0 * Bt + (row % Bt) for row in range(r * 10, r * 10 + 10)} - set(range(r * 10, r * 10 + 10)))
                                                for r in range(world))
    assert plans[0]["rows"] == 0 and not plans[0]["recv"]  # rank 0 owns agent 0's maps itself


def _sparse_worker(rank, world, port, A, Bt, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from v2x_sim_amd.parallel import ShardedV2VNet, ShardedWhen2com, plan_row_exchange, sparse_exchange, when2com_needs
        shard = AgentShard(A, Bt, rank, world)
        H, W, C = 4, 4, 8
        mk = lambda rows: torch.stack([torch.full((H, W, C), float(r)) + torch.arange(C) / 16.0 for r in rows]).to(torch.bfloat16)  # noqa: E731
        local = mk(shard.rows)
        dense = exchange_features(local, world)                                   # the all-gather path = reference
        # --- V2VNet "needed" transport
        class _M:
            gnn_iter_num, neighbor_source = 1, "initial"
        rn = ShardedV2VNet(_M(), shard, transport="needed")
        full, work = rn.start_exchange(local, out=torch.full((A * Bt, H, W, C), -1.0).to(torch.bfloat16))
        rn.wait(work)
        need = set(r for p in [rn.needed_plan()[rank]] for _, lo, hi in p["recv"] for r in range(lo, hi)) | set(shard.rows)
        for r in range(A * Bt):
            if r in need:
                assert torch.equal(full[r].view(torch.int16), dense[r].view(torch.int16)), "row %d differs from the all-gather" % r
            else:
                assert float(full[r, 0, 0, 0]) == -1.0, "row %d travelled although nobody reads it" % r
        # --- when2com sparse transport, byte-counted
        g = torch.Generator().manual_seed(11)
        coef = (torch.rand(Bt, A, A, generator=g) > 0.6).float()
        counts = [A] * (Bt - 1) + [3]
        rw = ShardedWhen2com(None, shard)
        full = rw.fetch_maps(local, coef, counts, "activated")
        needs = when2com_needs(shard, coef, counts)[rank]
        for r in needs | set(shard.rows):
            assert torch.equal(full[r].view(torch.int16), dense[r].view(torch.int16))
        st = rw.last_comm
        foreign = len(needs - set(shard.rows))
        assert st["transport"] == "sparse" and st["rows_received"] == foreign
        assert st["bytes_received"] == foreign * H * W * C * 2 < st["rows_allgather"] * H * W * C * 2
        dense2 = rw.fetch_maps(local, coef, counts, "softmax")                    # dense weights: falls back to the all-gather
        assert rw.last_comm["transport"] == "allgather" and torch.equal(dense2.view(torch.int16), dense.view(torch.int16))
        sent = torch.tensor([st["bytes_sent"], st["bytes_received"]], dtype=torch.int64)
        dist.all_reduce(sent)
        assert int(sent[0]) == int(sent[1]), "bytes sent != bytes received over the job"
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,Bt", [(2, 4), (4, 8), (8, 8)])
def test_sparse_transports_gloo(world, Bt):
    """'needed' (V2VNet) and 'sparse' (when2com) point-to-point transports on 2, 4 and 8 real gloo ranks."""
    _spawn(_sparse_worker, world, 5, Bt)


def test_bench_geometry_plans_world8():
    """The driver's 8-GPU run: 128 frames per GPU = two half-batches of 512 frames -> 2 560 items per half, 320 per rank.  Host logic of
    every rank at that size: the partition is agent-major and complete, each rank's slice lies inside ONE agent (so the 'needed' plan
    fetches the 4 other agents' maps of its 320 frames: 1 280 rows instead of the all-gather's 2 240), and the plans pair up."""
    from v2x_sim_amd.parallel import ShardedV2VNet
    A, Bh, world = 5, 512, 8
    class _M:
        gnn_iter_num, neighbor_source = 1, "initial"
    rows = []
    for r in range(world):
        sh = AgentShard(A, Bh, r, world)
        assert sh.per_rank == 320 and len({a for a, _ in sh.items}) <= 2
        rows += sh.rows
        plan = sh.fusion_plan(torch.full((Bh, A), A), "cpu")
        assert plan["n"] == 320 and plan["local_rows"] is None and plan["coef"].sum().item() == 320 * 4
    assert rows == list(range(A * Bh))
    rn = ShardedV2VNet(_M(), AgentShard(A, Bh, 0, world), transport="needed")
    plans = rn.needed_plan()
    needs = []
    for r in range(world):
        fr = {row % Bh for row in range(r * 320, (r + 1) * 320)}
        needs.append([j * Bh + f for f in fr for j in range(A)])
    _check_plans(plans, needs, 320)
    assert max(p["rows"] for p in plans) <= 4 * 320 + 320 < 7 * 320
    # the literal north_star layout: 5 ranks, one agent each -> every rank reads ALL other agents' maps: needed == all-gather
    rn5 = ShardedV2VNet(_M(), AgentShard(A, Bh, 0, 5), transport="needed")
    assert all(p["rows"] == 4 * Bh for p in rn5.needed_plan())
    for r in range(5):
        assert {a for a, _ in AgentShard(A, Bh, r, 5).items} == {r}


def _subgroup_worker(rank, world, port, q):
    """Shard on a SUB-GROUP whose members are not ranks 0..R-1 of the job (P2POp peers are GLOBAL ranks), and the
    synchronous later-round exchange of the 'needed' transport (neighbor_source='updated')."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from v2x_sim_amd.parallel import ShardedV2VNet
        members = [1, 3]
        group = dist.new_group(ranks=members)          # collective over ALL ranks of the job
        if rank in members:
            A, Bt = 5, 4
            srank = members.index(rank)
            shard = AgentShard(A, Bt, srank, 2)
            H, W, C = 2, 2, 8
            mk = lambda rows, off: torch.stack([torch.full((H, W, C), float(r) + off) for r in rows]).to(torch.bfloat16)  # noqa: E731
            class _M:
                gnn_iter_num, neighbor_source = 2, "updated"
            for transport in ("allgather", "needed"):
                rn = ShardedV2VNet(_M(), shard, group=group, transport=transport)
                for off in (0.0, 64.0):                # first round (start_exchange) and a later round (exchange_round)
                    local = mk(shard.rows, off)
                    if off == 0.0:
                        full, work = rn.start_exchange(local)
                        rn.wait(work)
                    else:
                        full = rn.exchange_round(local)
                    need = set(shard.rows)
                    if transport == "needed":
                        need |= {r for _, lo, hi in rn.needed_plan()[srank]["recv"] for r in range(lo, hi)}
                    else:
                        need = set(range(A * Bt))
                    for r in need:
                        assert float(full[r, 0, 0, 0]) == float(r) + off, (transport, off, r, float(full[r, 0, 0, 0]))
        dist.barrier()
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_shard_on_a_subgroup_and_later_round_exchange_gloo_world4():
    _spawn(_subgroup_worker, 4)
