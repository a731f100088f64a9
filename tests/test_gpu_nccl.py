"""RCCL on the hardware that is available: a world-size-1 `nccl` process group on the 1-GPU lease.

SURVEY 8(e): the only exchange of the path is the gather of the match arrays (gather.gather_matches).  The 2-rank gloo
tests (test_distributed_gloo.py) cover its control flow on CPU tensors; this file runs the SAME functions on device
tensors through torch's `nccl` backend (= RCCL on ROCm) -- process-group creation with `device_id`, all_gather of
int64 / float32 / float64 device tensors, barrier -- next to the library's own HIP kernels in one process (the
library is built with hipcc 7.2, torch carries its own HIP runtime and RCCL).  The reference is single-device
(utils/eval/model_helper.py:30), so nothing but a run can vouch for this part.
"""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def nccl_world1():
    import bench
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist, info = bench.init_dist(1, dev, force=True)
    assert dist is not None, info
    assert info["backend"] == "nccl" and info["world"] == 1
    yield dist, dev, info
    dist.destroy_process_group()


def test_world1_group_reports_rccl(nccl_world1):
    dist, dev, info = nccl_world1
    assert dist.is_initialized() and dist.get_world_size() == 1 and dist.get_rank() == 0
    assert info["rccl_version"] and info["hip_runtime"]
    t = torch.arange(5, device=dev, dtype=torch.float64)
    out = [torch.zeros_like(t)]
    dist.all_gather(out, t)
    dist.barrier()
    torch.cuda.synchronize()
    assert torch.equal(out[0], t)


@pytest.mark.parametrize("n", [0, 1, 7, 6400])
def test_gather_matches_over_rccl(nccl_world1, n):
    """gather.gather_matches with device tensors: empty, ragged and bench-sized contributions; ids beyond 2^40 survive."""
    from patch2pix_amd.gather import gather_matches
    dist, dev, _ = nccl_world1
    g = torch.Generator().manual_seed(n)
    rows = torch.rand(n, 9, generator=g).to(dev)
    ids = (torch.arange(n, dtype=torch.int64) + (1 << 40) + 1).to(dev)
    a, b = gather_matches(rows, ids)
    torch.cuda.synchronize()
    assert a.device.type == "cuda" and b.dtype == torch.int64
    assert torch.equal(a, rows) and torch.equal(b, ids)


def test_bench_timed_region_with_rccl_and_hip_kernels(nccl_world1):
    """bench.timed_steps (barrier, K steps of the REAL hot path through libp2p_hip, pack_results, gather_matches over
    RCCL, barrier) + bench.exchange_rank_stats on the world-1 group: the driver's N = 1 line goes through exactly this."""
    import bench
    from patch2pix_amd.utils import synthetic
    from patch2pix_amd.utils.eval import model_helper
    dist, dev, _ = nccl_world1
    cfg = dict(bench.CONFIGS["A"])
    cfg.update(H=128, W=160, pairs_per_step=2, ptmax=32)
    ckpt = synthetic.make_checkpoint(0)
    net = model_helper.load_model(ckpt, lprint=lambda *a: None)
    _, batches = bench.resident_batches(cfg, 0, dev, 1)
    np.random.seed(7)
    runner = bench.Runner(net, batches, cfg["ptmax"])
    with torch.no_grad():
        runner.run(1)
        t = bench.timed_steps(runner.run, 2, True, 0, 1, cfg["pairs_per_step"], dev, dist)
    assert t["nrows"] == 2 * cfg["pairs_per_step"] * cfg["ptmax"]
    assert t["elapsed"] >= t["local_elapsed"] > 0 and len(t["events"]) == 2
    stats = bench.exchange_rank_stats(dist, dev, 1, [t["elapsed"], 1.0, 2.0, 0.5])
    assert stats == [[t["elapsed"], 1.0, 2.0, 0.5]]


def test_pair_stream_exchanges_over_rccl(nccl_world1):
    """gather.run_pair_stream (bench.py --pairs N): every round's exchange on device tensors, a round with no rows included."""
    from patch2pix_amd.gather import run_pair_stream
    dist, dev, _ = nccl_world1

    def submit(pair_ids):
        return list(pair_ids)

    def finish(ticket):
        n = [pid % 3 for pid in ticket]
        return ([torch.full((k, 4), float(pid), device=dev) for k, pid in zip(n, ticket)],
                [torch.full((k,), 0.5, device=dev) for k in n],
                [torch.full((k, 4), pid, dtype=torch.int64, device=dev) for k, pid in zip(n, ticket)])

    rows, ids, done = run_pair_stream(11, 0, 1, 2, submit, finish, gather_every=2, device=dev)
    torch.cuda.synchronize()
    assert done == 11 and int(ids.numel()) == sum(p % 3 for p in range(11))
    for pid in range(11):
        assert int((ids == pid).sum()) == pid % 3
    # a stream whose only pair yields no rows: the exchange runs on empty device tensors
    rows, ids, done = run_pair_stream(1, 0, 1, 2, submit, finish, gather_every=2, device=dev)
    assert done == 1 and rows.shape == (0, 9) and ids.shape == (0,)


def test_bench_under_the_launcher_with_one_rank():
    """The driver's N > 1 command line with N = 1: `python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr
    127.0.0.1 --master-port P bench.py --gpus 1 ...` -- the rendezvous comes from the launcher's environment (RANK / WORLD_SIZE /
    MASTER_*), the process group is `nccl` on the rank's GPU, the JSON line says so."""
    import json
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["P2P_BENCH_SPINUP"] = "0.2"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--pairs-per-step", "2",
           "--no-parity", "--no-other-modes", "--no-e2e", "--no-other-configs", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["steps"] == 2
    assert out["dist"]["backend"] == "nccl" and out["dist"]["world"] == 1
    assert len(out["per_rank_pairs_per_s"]) == 1
