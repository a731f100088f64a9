"""The N>1 path of bench.py on CPU: two gloo processes shard a pair list and gather ragged match arrays."""
import os
import socket

import pytest

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from patch2pix_amd.gather import gather_matches, pack_results, shard_pairs


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


BIG = (1 << 40) + 1                            # pair ids beyond float precision must survive the gather


def _fake_rows(pair_id):
    n = (pair_id * 7) % 5                      # ragged, includes empty results
    g = torch.Generator().manual_seed(pair_id)
    return torch.rand(n, 9, generator=g) + pair_id


def _worker(rank, world, port, num_pairs, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard_pairs(num_pairs, rank, world)
    rows = [_fake_rows(p) for p in mine]
    ids = [torch.full((r.shape[0],), p + BIG, dtype=torch.int64) for r, p in zip(rows, mine)]
    rows_all, ids_all = gather_matches(torch.cat(rows) if rows else torch.zeros(0, 9),
                                       torch.cat(ids) if ids else torch.zeros(0, dtype=torch.int64))
    ret[rank] = (rows_all, ids_all)
    dist.destroy_process_group()


def test_shard_covers_every_pair_once():
    for world in (1, 2, 3, 8):
        seen = sorted(p for r in range(world) for p in shard_pairs(11, r, world))
        assert seen == list(range(11))


def test_two_rank_gather_of_ragged_matches():
    world, num_pairs = 2, 9
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, num_pairs, ret), nprocs=world, join=True)
    assert set(ret.keys()) == {0, 1}
    r0, i0 = ret[0]
    r1, i1 = ret[1]
    assert torch.equal(r0, r1) and torch.equal(i0, i1)          # every rank holds the same gathered set
    for p in range(num_pairs):                                   # and it is exactly the union of the shards
        assert torch.equal(r0[i0 == p + BIG], _fake_rows(p))
    assert r0.shape[0] == sum((p * 7) % 5 for p in range(num_pairs)) and i0.dtype == torch.int64


def test_single_process_is_identity():
    rows = torch.rand(4, 9)
    ids = torch.arange(4)
    a, b = gather_matches(rows, ids)
    assert a is rows and b is ids


def _fake_step(rank, step, B):
    """(fine, score, coarse) lists like Patch2Pix.fine_from_ticket returns, sizes depending on (rank, step, pair)."""
    g = torch.Generator().manual_seed(1000 * rank + step)
    n = [(3 * rank + 2 * step + b) % 4 for b in range(B)]
    return ([torch.rand(k, 4, generator=g) for k in n], [torch.rand(k, generator=g) for k in n],
            [torch.randint(0, 640, (k, 4), generator=g) for k in n])


def _bench_worker(rank, world, port, steps, B, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    results = [_fake_step(rank, i, B) for i in range(steps)]
    ret[rank] = gather_matches(*pack_results(results, rank, world, B))        # exactly bench.py's final exchange
    dist.destroy_process_group()


def test_bench_final_exchange_on_two_ranks():
    """bench.py --gpus 2: every rank packs its steps with pack_results and the ranks exchange them with gather_matches;
    the gathered set must hold every (rank, step, pair) result exactly once under its global pair id."""
    world, steps, B = 2, 3, 2
    port = _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_bench_worker, args=(world, port, steps, B, ret), nprocs=world, join=True)
    rows, ids = ret[0]
    assert torch.equal(rows, ret[1][0]) and torch.equal(ids, ret[1][1]) and ids.dtype == torch.int64
    total = 0
    for r in range(world):
        for i in range(steps):
            fine, score, coarse = _fake_step(r, i, B)
            for b in range(B):
                got = rows[ids == (i * world + r) * B + b]
                want = torch.cat([fine[b], score[b][:, None], coarse[b].float()], dim=1)
                assert torch.equal(got, want)
                total += want.shape[0]
    assert rows.shape[0] == total and len(set(ids.tolist())) <= world * steps * B


def _weak_worker(rank, world, port, steps, B, ret):
    """bench.py's own timed region (bench.timed_steps) and per-rank bookkeeping (bench.exchange_rank_stats) with a stand-in
    for the matcher: what `python -m torch.distributed.run ... bench.py --gpus 2` executes, minus the GPU work."""
    import sys
    import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")

    def run(nsteps):
        time.sleep(0.02 * (rank + 1))                   # ranks finish their own steps at different times
        return [_fake_step(rank, i, B) for i in range(nsteps)]

    t = bench.timed_steps(run, steps, True, rank, world, B, dev, dist)
    roof = bench.roofline_of("fp16x2", t["events"])
    stats = bench.exchange_rank_stats(dist, dev, world, [t["elapsed"], steps * B / t["local_elapsed"], roof["avg_launch_ms"], roof["frac"]])
    ret[rank] = (t["nrows"], t["elapsed"], t["local_elapsed"], stats)
    dist.destroy_process_group()


def test_bench_weak_scaling_path_on_two_ranks():
    """The weak-scaling leg of bench.py end to end on two gloo ranks: every rank times its own steps between the two
    barriers, the match arrays of all ranks are gathered inside the timed region, and rank 0 receives every rank's own
    rate and regress-launch figures (per_rank_pairs_per_s, per_rank_roofline_frac of the JSON line)."""
    world, steps, B = 2, 3, 2
    port = _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_weak_worker, args=(world, port, steps, B, ret), nprocs=world, join=True)
    total = sum(_fake_step(r, i, B)[0][b].shape[0] for r in range(world) for i in range(steps) for b in range(B))
    for r in range(world):
        nrows, elapsed, local, stats = ret[r]
        assert nrows == total                                           # every rank holds the whole gathered set
        assert len(stats) == world and all(len(x) == 4 for x in stats)
        assert local <= elapsed + 1e-6
    assert ret[0][3] == ret[1][3]                                       # the same table on every rank
    rates = [x[1] for x in ret[0][3]]
    assert rates[0] > rates[1] > 0                                      # rank 1 slept twice as long: its own rate is lower
    assert max(x[0] for x in ret[0][3]) >= 0.04                         # the slowest rank's clock covers its sleep


def _stream_worker(rank, world, port, num_pairs, chunk, gather_every, ret):
    from patch2pix_amd.gather import run_pair_stream
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    log = []

    def submit(pair_ids):                       # stand-in for "generate the chunk's pyramids + coarse_async"
        log.append(("submit", tuple(pair_ids)))
        return list(pair_ids)

    def finish(ticket):                         # stand-in matcher: pair id -> (id % 4) rows that encode the id
        log.append(("finish", tuple(ticket)))
        n = [pid % 4 for pid in ticket]
        return ([torch.full((k, 4), float(pid)) for k, pid in zip(n, ticket)], [torch.full((k,), 0.5) for k in n],
                [torch.full((k, 4), pid, dtype=torch.int64) for k, pid in zip(n, ticket)])

    rows, ids, done = run_pair_stream(num_pairs, rank, world, chunk, submit, finish, gather_every=gather_every)
    ret[rank] = (rows, ids, done, log)
    dist.destroy_process_group()


@pytest.mark.parametrize("num_pairs,chunk,gather_every", [(37, 4, 2), (5, 4, 3), (1, 2, 2), (64, 8, 8)])
def test_pair_stream_on_two_ranks(num_pairs, chunk, gather_every):
    """bench.py --pairs N (BASELINE configs[3]): pair i goes to rank i % world, chunks are pipelined one ahead, results
    are exchanged every few chunks.  Every pair id must arrive exactly once on every rank, with its own rows, and a rank
    whose share is empty (1 pair on 2 ranks) still takes part in every collective."""
    world = 2
    port = _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_stream_worker, args=(world, port, num_pairs, chunk, gather_every, ret), nprocs=world, join=True)
    rows, ids, _, _ = ret[0]
    assert torch.equal(rows, ret[1][0]) and torch.equal(ids, ret[1][1])
    assert ret[0][2] + ret[1][2] == num_pairs and ret[0][2] == len(range(0, num_pairs, 2))
    for pid in range(num_pairs):
        sel = ids == pid
        assert int(sel.sum()) == pid % 4
        assert bool((rows[sel][:, :4] == float(pid)).all()) and bool((rows[sel][:, 5:] == float(pid)).all())
    assert int(ids.numel()) == sum(p % 4 for p in range(num_pairs))
    for r in range(world):                      # one chunk ahead: submit(k+1) is issued before finish(k)
        log = ret[r][3]
        subs = [i for i, e in enumerate(log) if e[0] == "submit"]
        fins = [i for i, e in enumerate(log) if e[0] == "finish"]
        assert len(subs) == len(fins) and all(e[1] == f[1] for e, f in zip([log[i] for i in subs], [log[i] for i in fins]))
        assert all(subs[k + 1] < fins[k] for k in range(len(fins) - 1))


def test_plain_bench_command_starts_its_own_ranks():
    """`python bench.py --gpus 2` as a PLAIN command (no launcher, no WORLD_SIZE): bench.self_launch re-executes it under
    torch.distributed.run with two ranks on 127.0.0.1; --rendezvous-check stops after the ranks have found each other
    (gloo stands in for RCCL without a GPU)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rendezvous-check"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["rendezvous"] == "ok" and out["world"] == 2 and out["ranks"] == [0, 1]
