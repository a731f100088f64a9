#!/usr/bin/env python
"""Throughput of the Patch2Pix matching hot path on MI355X (BASELINE.json metric: image-pairs/sec,
480x640, ptmax=400).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = one pass of the hot path over a batch of image pairs whose feature pyramids are already resident in
HBM: coarse stage (normalise, 4-D correlation + pool, mutual matching, consensus, matches) ->
filter_coarse(ptmax=400) -> mid + fine regressors -> match arrays on the device.  Pairs are
independent, so with N ranks every rank runs its own K steps (weak scaling) and the match arrays are
gathered once over RCCL inside the timed region.  Prints ONE JSON line on rank 0.

The headline (`value`, `dtype`, `roofline`) is measured in the library's default arithmetic, which is
fp32-equivalent (bf16x3: every fp32 operand as three bf16 planes = 24 significant bits, fp32 accumulation;
see include/p2p_hip.h).  Outside the timed region rank 0 also (i) pushes one of the benched pairs -- through the
same batched calls -- and the CPU oracle and reports the differences (`parity`), (ii) times the other
arithmetic modes for a few steps (`other_modes`, informational), (iii) times the oracle on the host (`cpu_baseline`).
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

KSIZE = 2
CONFIGS = {
    # BASELINE.json configs[1]: the configuration the metric is quoted on
    "A": dict(H=480, W=640, ptmax=400, panc=1, pairs_per_step=16,
              workload="480x640 pairs, ksize=2, ptmax=400 proposals per pair, panc=1, coarse (NCNet 4D) + mid/fine "
                       "regressors; configs[1] of BASELINE.json, several pairs per step"),
    # BASELINE.json configs[4]: Aachen-size pairs with the training-time proposal options (non-reference at eval)
    "E": dict(H=960, W=1280, ptmax=800, panc=8, pairs_per_step=2,
              workload="960x1280 pairs, ksize=2, ptmax=800 x panc=8 = 6400 proposals per pair, coarse (NCNet 4D) + "
                       "mid/fine regressors; configs[4] of BASELINE.json"),
}
# algorithmic work of one regress launch: per proposal per level
#   conv1 2*64*512*(518*9) + conv2 2*64*512*(512*9) + fc 2*(512*512+512*256+256*5) flop
FLOP_PER_PROPOSAL_LEVEL = 2 * 64 * 512 * (518 * 9) + 2 * 64 * 512 * (512 * 9) + 2 * (512 * 512 + 512 * 256 + 256 * 5)
PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_BF16_MFMA_TFLOPS = 2500.0
# bf16 kernels: matrix-core work ISSUED per proposal and level = 8 waves x (584 conv1 units + 576 conv2 units)
# x (products x 2 m-tiles) x v_mfma_f32_32x32x16_bf16 (32768 flop each); + 0.1 % K padding
MODES = {
    "bf16x3": dict(kernel="regress_x3_kernel", products=6, peak=PEAK_BF16_MFMA_TFLOPS / 6.0,
                   dtype="f32-equivalent: bf16x3 (every f32 operand = exact sum of 3 bf16 planes, 24 significant bits; "
                         "6 bf16 MFMA products per f32 product, f32 accumulate)",
                   peak_note="peak = 2500 TFLOP/s dense bf16 MFMA / 6 MFMA products per fp32 product"),
    "f32": dict(kernel="regress_kernel", products=None, peak=PEAK_F32_MFMA_TFLOPS, dtype="f32",
                peak_note="peak = 157.3 TFLOP/s dense fp32 MFMA"),
    "bf16x2": dict(kernel="regress_split_kernel", products=3, peak=PEAK_BF16_MFMA_TFLOPS / 3.0,
                   dtype="bf16x2 (REDUCED precision: f32 operands split hi+lo = 16 significant bits, f32 accumulate)",
                   peak_note="peak = 2500 TFLOP/s dense bf16 MFMA / 3 MFMA products per fp32 product"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="A", help="A = the metric's configuration (default)")
    ap.add_argument("--pairs-per-step", type=int, default=None,
                    help="image pairs per step; their proposals share one regress launch (fills the 256 CUs)")
    ap.add_argument("--mode", choices=sorted(MODES), default=None, help="regressor arithmetic (default: library default)")
    ap.add_argument("--overlap", type=int, default=0,
                    help="1: coarse stage of step i+1 on a second stream beside the regress launch of step i; measured "
                         "neutral (444-446 vs 444-457 pairs/s: the regress launch stretches from 29.2 to 35.3 ms), so 0 is the default")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the in-run oracle check of a benched pair")
    ap.add_argument("--no-other-modes", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end estimate_matches timing")
    return ap.parse_args()


def source_hash():
    """Identifies the kernel sources a profile was taken with (profiles/regress_traffic.json)."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "patch2pix_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h")):
            h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()[:16]


def oracle_pair(ckpt, pyr1, pyr2, ptmax, panc, seed, gpu_mid=None):
    """One full pair through the CPU oracle (oracle/p2p_oracle.py): every stage boundary."""
    from oracle import p2p_oracle as orc
    ncn, mid_p, fine_p = orc.split_params(ckpt["state_dict"])
    corr, delta = orc.coarse_forward(pyr1[4], pyr2[4], KSIZE, ncn)
    m, s = orc.cal_coarse_matches(corr, delta, KSIZE, 8)
    cm, _ = orc.filter_coarse(m, s, 0.0, True, ptmax=ptmax, rng=np.random.RandomState(seed))
    cm = orc.shift_to_anchors(cm, 8, panc)
    mid, mid_s, _ = orc.fine_level(pyr1[:4], pyr2[:4], cm, mid_p)
    fine, fine_s, _ = orc.fine_level(pyr1[:4], pyr2[:4], mid if gpu_mid is None else gpu_mid, fine_p)
    return dict(all_rows=m, proposals=cm, mid=mid, mid_scores=mid_s, fine=fine, fine_scores=fine_s)


def cpu_baseline(ckpt, pyr1, pyr2, cfg):
    """The CPU oracle (a port of the reference algorithm, oracle/p2p_oracle.py) on the host cores of this box, on a
    bounded sample of the same workload: one warm-up + 5 repetitions of one full pair (coarse stage + filter_coarse(ptmax)
    + both regressors on all proposals), median.  32 threads: measured on the 2x64-core host, torch-CPU is fastest at 32
    threads for these op sizes (8: 0.88 s, 32: 0.64 s, 128: 2.4 s for the coarse stage).  The port is conservative
    for the comparison: its batched conv3d is faster than the reference's Python loop over conv3d slices; the
    unmodified reference itself cannot run on the GPU box (/root/reference is not there)."""
    threads = min(32, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    reps = 5 if cfg["H"] <= 480 else 1
    times = []
    with torch.no_grad():
        oracle_pair(ckpt, pyr1, pyr2, 16, 1, 0)                         # warm-up (thread pools, oneDNN primitives)
        for _ in range(reps):
            t0 = time.perf_counter()
            oracle_pair(ckpt, pyr1, pyr2, cfg["ptmax"], cfg["panc"], 0)
            times.append(time.perf_counter() - t0)
    t_pair = sorted(times)[len(times) // 2]
    return {"value": 1.0 / t_pair, "unit": "pairs/s", "cores": threads, "kind": "port",
            "sample": f"1 warm-up + {reps} x one full {cfg['H']}x{cfg['W']} pair (coarse + filter ptmax={cfg['ptmax']} + mid/fine "
                      f"regressors on {cfg['ptmax'] * cfg['panc']} proposals), median {t_pair:.2f} s; torch-CPU fp32 port of the "
                      f"reference algorithm (oracle/p2p_oracle.py), {threads} threads of {os.cpu_count()} logical cores, "
                      f"torch {torch.__version__}"}


def parity_check(net, ckpt, batch, cpu_pair, cfg):
    """Push one benched pair through the SAME batched calls bench times (coarse_async / fine_from_ticket over the
    whole batch) and through the CPU oracle: coarse rows bit-exact?, regressed coordinates / scores max |diff|.
    The fine level of the oracle is fed the kernel's own mid matches (a 1e-6 px wobble across an integer would move
    the whole fine patch by one pixel, networks/utils.py:19); `fine_px_chain` is the plain end-to-end difference."""
    seed = 4242
    np.random.seed(seed)                       # pair 0 is the first one filter_coarse samples for
    f1, f2 = batch
    with torch.no_grad():
        ticket = net.coarse_async(f1, f2, ksize=KSIZE)
        fine, fine_s, mid, mid_s, coarse = net.fine_from_ticket(ticket, ncn_thres=0.0, mutual=True, return_all=True,
                                                                ptmax=cfg["ptmax"])
        torch.cuda.synchronize()
        all_rows = ticket["matches"][0].cpu()
        g_mid = mid[0].cpu()
        ref = oracle_pair(ckpt, cpu_pair[0], cpu_pair[1], cfg["ptmax"], cfg["panc"], seed, gpu_mid=g_mid)
        ref_chain = oracle_pair(ckpt, cpu_pair[0], cpu_pair[1], cfg["ptmax"], cfg["panc"], seed)
    rows_equal = bool(torch.equal(all_rows, ref["all_rows"]))
    props_equal = bool(torch.equal(coarse[0].cpu(), ref["proposals"]))
    out = {"pair": "pair 0 of the benched batch, batched calls", "coarse_indices_equal": rows_equal,
           "coarse_rows": int(all_rows.shape[0]),
           "coarse_rows_differing": int((all_rows != ref["all_rows"]).any(dim=1).sum()),
           "proposals_equal": props_equal, "proposals": int(coarse[0].shape[0])}
    if props_equal:
        out.update({
            "max_px_err_mid": float((g_mid - ref["mid"]).abs().max()),
            "max_px_err": float((fine[0].cpu() - ref["fine"]).abs().max()),
            "max_score_err": float(max((mid_s[0].cpu() - ref["mid_scores"]).abs().max(),
                                       (fine_s[0].cpu() - ref["fine_scores"]).abs().max())),
            "max_px_err_fine_chain": float((fine[0].cpu() - ref_chain["fine"]).abs().max()),
            "tolerance_px": 1e-3, "tolerance_score": 1e-5})
    return out


def main():
    args = parse()
    cfg = dict(CONFIGS[args.config])
    if args.pairs_per_step:
        cfg["pairs_per_step"] = args.pairs_per_step
    H, W, PTMAX, B = cfg["H"], cfg["W"], cfg["ptmax"], cfg["pairs_per_step"]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # host side of a rank is one Python thread plus small torch-CPU ops: keep N ranks from oversubscribing the host
    torch.set_num_threads(max(1, min(16, (os.cpu_count() or 16) // max(world, 1))))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    from patch2pix_amd import ops
    from patch2pix_amd.gather import gather_matches, pack_results
    from patch2pix_amd.utils import synthetic
    from patch2pix_amd.utils.eval import model_helper

    ckpt = synthetic.make_checkpoint(0)
    ckpt["regressor_config"].panc = cfg["panc"]
    net = model_helper.load_model(ckpt, lprint=lambda *a: None)
    net.panc = cfg["panc"]                       # load_model forces panc = 1 like the reference (model_helper.py:46)
    if args.mode:
        for w in net._weights()[1:]:
            w.set_mode(args.mode)
    mode = net._weights()[1].mode
    # a few distinct synthetic batches per rank, resident in HBM before the clock starts
    nbatches = 2
    cpu_pairs = [synthetic.make_correlated_pyramids(1000 + rank * 64 + i, H, W) for i in range(nbatches * B)]
    batches = []
    for k in range(nbatches):
        chunk = cpu_pairs[k * B:(k + 1) * B]
        f1 = [torch.stack([p[0][j] for p in chunk]).to(dev) for j in range(5)]
        f2 = [torch.stack([p[1][j] for p in chunk]).to(dev) for j in range(5)]
        batches.append((f1, f2))
    np.random.seed(1234 + rank)

    coarse_stream = torch.cuda.Stream(device=dev) if args.overlap else None

    def submit(i):
        f1, f2 = batches[i % nbatches]
        if coarse_stream is None:
            return net.coarse_async(f1, f2, ksize=KSIZE)
        # experiment: the coarse stage of the NEXT step on its own stream, beside the regress launch of the current one
        with torch.cuda.stream(coarse_stream):
            return net.coarse_async(f1, f2, ksize=KSIZE)

    def finish(ticket):
        return net.fine_from_ticket(ticket, ncn_thres=0.0, mutual=True, ptmax=PTMAX)

    def run(nsteps):
        """nsteps steps, software-pipelined on one stream: the coarse stage of step i+1 is enqueued
        before the host filters step i, so the GPU never waits for the host."""
        out = []
        ticket = submit(0)
        for i in range(nsteps):
            nxt = submit(i + 1) if i + 1 < nsteps else None
            out.append(finish(ticket))
            ticket = nxt
        return out

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(nsteps, with_gather):
        barrier()
        ops.regress_events = []
        t0 = time.perf_counter()
        results = run(nsteps)
        nrows = None
        if with_gather:
            # final gather of the match arrays (the only inter-GPU exchange of the path)
            all_rows, all_ids = gather_matches(*pack_results(results, rank, world, B))
            nrows = all_rows.shape[0]
            assert all_ids.dtype == torch.int64 and all_ids.shape[0] == nrows
        barrier()
        elapsed = time.perf_counter() - t0
        events = ops.regress_events
        ops.regress_events = None
        return elapsed, events, nrows

    with torch.no_grad():
        # untimed spin-up: a fresh box needs ~1 s of work before clocks / allocator / page cache settle
        # (the first process on a cold box otherwise measures ~30 % low), then the W warm-up steps
        t_spin = time.perf_counter()
        while time.perf_counter() - t_spin < 1.5:
            run(2)
            torch.cuda.synchronize()
        run(args.warmup)
        elapsed, events, nrows = timed(args.steps, True)
    assert nrows == world * args.steps * B * PTMAX * cfg["panc"]
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    def roofline_of(mode, events):
        M = MODES[mode]
        kern_ms = [a.elapsed_time(b) for a, b, _, _ in events]
        flop = sum(n * lv * FLOP_PER_PROPOSAL_LEVEL for _, _, n, lv in events) / max(len(events), 1)
        avg_ms = sum(kern_ms) / max(len(kern_ms), 1)
        # achieved = ALGORITHMIC flop of one regress launch (608.3 MFLOP per proposal and level) / its
        # average duration (HIP events on the launch stream)
        achieved = flop / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        r = {"kernel": M["kernel"], "bound": "mfma", "achieved": achieved, "peak": M["peak"], "unit": "TFLOP/s",
             "frac": achieved / M["peak"], "traffic": None, "avg_launch_ms": avg_ms, "algorithmic_flop_per_launch": flop,
             "peak_note": M["peak_note"],
             "note": "achieved = algorithmic flop of the launch (608.3 MFLOP x proposals x levels) / launch time "
                     "measured with HIP events on the launch stream"}
        return r

    value = world * args.steps * B / elapsed
    if rank == 0:
        roof = roofline_of(mode, events)
        # HBM traffic of the dominant kernel: PMC measurement (tools/collect_profiles.sh) of THIS source tree, same
        # kernel and launch size -- otherwise null
        tf = os.path.join(ROOT, "profiles", "regress_traffic.json")
        if os.path.exists(tf):
            rec = json.load(open(tf)).get(mode, {})
            if (rec.get("kernel") == MODES[mode]["kernel"] and rec.get("proposals_per_launch") == B * PTMAX * cfg["panc"]
                    and rec.get("config", "A") == args.config and rec.get("source_hash") == source_hash()):
                roof["traffic"] = rec.get("hbm_bytes_per_launch")
                roof["traffic_source"] = rec.get("source")
                roof["traffic_note"] = ("FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE of the kernel per launch = requests on the "
                                        "L2's fabric side: Infinity-Cache hits are included, so this is an upper bound of the HBM "
                                        "bytes; the compulsory bytes of the launch are ~1 GB (16 x 61 MB of pyramids + 57 MB of "
                                        "weights), the rest are L2 capacity misses of the 28.5 MB-per-level weight stream, which "
                                        "lives in the 256 MB Infinity Cache")
        # the path's compulsory HBM bytes per pair against the 8 TB/s roofline, as north_star asks
        alg_bytes = 118e6 if args.config == "A" else 0.42e9
        out = {
            "metric": f"image-pairs/sec ({H}x{W}, ptmax={PTMAX}), matching hot path, feature pyramids resident in HBM",
            "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": MODES[mode]["dtype"], "data": "synthetic",
            "config": {"workload": cfg["workload"], "pairs_per_step": B, "regress_mode": mode,
                       "parallelism": f"pairs sharded over {world} GPU(s), one final RCCL all_gather"},
            "roofline": roof,
            "hbm_roofline": {"algorithmic_bytes_per_pair": alg_bytes, "achieved_GBps": value / world * alg_bytes / 1e9,
                             "peak_GBps": 8000.0, "frac": value / world * alg_bytes / 8e12,
                             "note": "compulsory bytes of the whole path per pair (SURVEY 8d) x pairs/s per GPU; the path "
                                     "is MFMA-bound (4300 flop/B), so this fraction is << 1 by construction"},
        }
    # ---- outside the timed region (single-GPU runs only) ----
    if rank == 0 and world == 1:
        if not args.no_parity:
            out["parity"] = parity_check(net, ckpt, batches[0], cpu_pairs[0], cfg)
        if not args.no_other_modes:
            other = {}
            for m2 in MODES:
                if m2 == mode:
                    continue
                for w in net._weights()[1:]:
                    w.set_mode(m2)
                with torch.no_grad():
                    run(2)
                    n2 = max(3, args.steps // 4)
                    e2, ev2, _ = timed(n2, False)
                r2 = roofline_of(m2, ev2)
                other[m2] = {"value": n2 * B / e2, "unit": "pairs/s", "dtype": MODES[m2]["dtype"],
                             "roofline_frac": r2["frac"], "achieved": r2["achieved"], "peak": r2["peak"],
                             "avg_launch_ms": r2["avg_launch_ms"]}
                if not args.no_parity:
                    p2 = parity_check(net, ckpt, batches[0], cpu_pairs[0], cfg)
                    other[m2].update({k: p2.get(k) for k in ("max_px_err_mid", "max_px_err", "max_score_err")})
            for w in net._weights()[1:]:
                w.set_mode(mode)
            out["other_modes"] = other
        if not args.no_e2e and args.config == "A":
            try:
                from tools import e2e_bench
                out["e2e"] = e2e_bench.measure(net, H, W)
            except Exception as e:       # informational leg; never fail the bench line on it
                out["e2e"] = {"error": repr(e)}
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(ckpt, cpu_pairs[0][0], cpu_pairs[0][1], cfg)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
