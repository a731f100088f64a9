#!/usr/bin/env python
"""Throughput of the Patch2Pix matching hot path on MI355X (BASELINE.json metric: image-pairs/sec,
480x640, ptmax=400).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: either under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...`, or as the
  plain command above, which then starts its own N ranks that way on 127.0.0.1 -- `self_launch`)

A step = one pass of the hot path over a batch of image pairs whose feature pyramids are already resident in
HBM: coarse stage (normalise, 4-D correlation + pool, mutual matching, consensus, matches) ->
filter_coarse(ptmax=400) -> mid + fine regressors -> match arrays on the device.  Pairs are
independent, so with N ranks every rank runs its own K steps (weak scaling) and the match arrays are
gathered once over RCCL inside the timed region.  Prints ONE JSON line on rank 0.

The headline (`value`, `dtype`, `roofline`) is measured in the library's default arithmetic, which is
fp32-equivalent (fp16x2: every fp32 operand, scaled by an exact power of two, as two fp16 planes to within 2^-24; three
MFMA products per fp32 product, fp32 accumulation; see include/p2p_hip.h).  Outside the timed region rank 0 also (i) pushes one of the benched pairs -- through the
same batched calls -- and the CPU oracle and reports the differences (`parity`), (ii) times the other
arithmetic modes for a few steps (`other_modes`, informational), (iii) times BASELINE configs[4] for a few steps
(`other_configs.E`: 960x1280, ptmax 800 x panc 8 -- GPU legs only), (iv) times the oracle on the host (`cpu_baseline`).

  python bench.py --pairs 10000 [--gpus N]     BASELINE configs[3]: a stream of seeded pairs, pair i on rank i % N,
                                                strong scaling (`scaling_efficiency` against rank 0 running alone)
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
# fp16 MFMA flop the default mode ISSUES per proposal and level (regress_h2.hip / regress_wino.hip): conv1 after the cell-row
# de-duplication = per wave 48 + 18 x 72 v_mfma_f32_32x32x16_f16 (32 768 flop) and 18 x 96 v_mfma_f32_16x16x32_f16 (16 384 flop),
# 8 waves; conv2 as Winograd = 16 positions x 16 tile rows x 512 x 512 x 2 flop x 3 products; the FC tail runs on fp32 MFMA
ISSUED_FP16_FLOP_PER_PROPOSAL_LEVEL = 8 * ((48 + 18 * 72) * 32768 + 18 * 96 * 16384) + 16 * 16 * 512 * 512 * 2 * 3
PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_BF16_MFMA_TFLOPS = 2500.0
MODES = {
    "fp16x2w": dict(kernel="p2p_regress_batch", products=3, peak=PEAK_BF16_MFMA_TFLOPS / 3.0,
                    kernels=["regress_h2_kernel<true>", "wino_gemm_kernel", "regress_fc_kernel"],
                    dtype="f32-equivalent: fp16x2 (every f32 operand, scaled by an exact power of two, = sum of 2 fp16 planes to within "
                          "2^-24 of its magnitude; 3 fp16 MFMA products per f32 product, f32 accumulate); second convolution as "
                          "Winograd F(2x2,3x3) (transforms in f32, filters transformed in f64)",
                    peak_note="peak = 2500 TFLOP/s dense fp16 MFMA / 3 MFMA products per fp32 product; the flop count is the "
                              "DIRECT convolutions' (608.3 MFLOP per proposal and level) although conv2 issues 2.25x fewer"),
    "fp16x2": dict(kernel="regress_h2_kernel", products=3, peak=PEAK_BF16_MFMA_TFLOPS / 3.0,
                   dtype="f32-equivalent: fp16x2 (every f32 operand, scaled by an exact power of two, = sum of 2 fp16 planes to within "
                         "2^-24 of its magnitude; 3 fp16 MFMA products per f32 product, f32 accumulate)",
                   peak_note="peak = 2500 TFLOP/s dense fp16 MFMA / 3 MFMA products per fp32 product"),
    "f32": dict(kernel="regress_kernel", products=None, peak=PEAK_F32_MFMA_TFLOPS, dtype="f32",
                peak_note="peak = 157.3 TFLOP/s dense fp32 MFMA"),
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
    ap.add_argument("--depth", type=int, default=2,
                    help="coarse stages enqueued ahead of the step whose proposals the host samples (2: a host-side hiccup of a "
                         "few ms -- shared test boxes have them, profiles/r03_host_effects.txt -- does not idle the GPU; "
                         "no difference on a quiet host)")
    ap.add_argument("--overlap", type=int, default=0,
                    help="1: coarse stage of step i+1 on a second stream beside the regress launch of step i; measured "
                         "neutral (444-446 vs 444-457 pairs/s: the regress launch stretches from 29.2 to 35.3 ms), so 0 is the default")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the in-run oracle check of a benched pair")
    ap.add_argument("--no-other-modes", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end estimate_matches timing")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the config-E leg of the default run")
    ap.add_argument("--pairs", type=int, default=0,
                    help="stream mode (BASELINE configs[3]): this many seeded pairs in total, sharded pair_id %% world, "
                         "generated on the device chunk by chunk; --steps / --warmup are ignored")
    ap.add_argument("--rendezvous-check", action="store_true",
                    help="only initialise the ranks (RCCL, or gloo without a GPU), all_gather their ranks, print them and exit")
    return ap.parse_args()


def self_launch(ngpus):
    """`python bench.py --gpus N` as a PLAIN command (no WORLD_SIZE in the environment): become
    `python -m torch.distributed.run --standalone --local-addr 127.0.0.1 --nnodes=1 --nproc-per-node N bench.py <argv>`,
    one rank per GPU; the ranks read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* like under an external launcher."""
    # --standalone: the launcher binds its own rendezvous endpoint on a free port of --local-addr (no port is picked here and
    # closed again before the launcher binds it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={ngpus}", os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL between processes needs it on these hosts
    env.setdefault("OMP_NUM_THREADS", "4")
    sys.stdout.flush()
    os.execvpe(cmd[0], cmd, env)


def init_dist(world, dev, force=False):
    """The run's process group: `nccl` (= RCCL on ROCm) with the rank's GPU as `device_id`.  Also at N = 1 (unless
    P2P_BENCH_DIST=0): a world-size-1 group, so that the driver's single-GPU line goes through process-group creation, the
    barriers and the three all_gathers of gather_matches ON THE GPU -- the multi-GPU code path minus the xGMI links.  Under a
    launcher the rendezvous comes from the environment (MASTER_ADDR / MASTER_PORT / RANK / WORLD_SIZE); a plain
    single-process run uses an in-process store (nothing to resolve, no port to race for).
    -> (torch.distributed or None, info dict for the JSON line)."""
    if world == 1 and not force and os.environ.get("P2P_BENCH_DIST", "1") == "0":
        return None, {"backend": None, "note": "P2P_BENCH_DIST=0: no process group at N = 1"}
    import torch.distributed as dist
    try:
        if "MASTER_ADDR" in os.environ and "RANK" in os.environ and "WORLD_SIZE" in os.environ:
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("nccl", store=dist.HashStore(), rank=0, world_size=1, device_id=dev)
        t = torch.ones(1, device=dev)
        dist.all_reduce(t)                       # the communicator is created lazily: fail here, not inside the timed region
        torch.cuda.synchronize()
        assert float(t) == float(world)
    except Exception as e:
        if world > 1:
            raise
        if dist.is_initialized():
            dist.destroy_process_group()
        return None, {"backend": None, "error": repr(e)}
    try:
        rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception as e:                       # informational only
        rccl = repr(e)
    return dist, {"backend": dist.get_backend(), "world": world, "rccl_version": rccl, "hip_runtime": torch.version.hip,
                  "torch": torch.__version__,
                  "collectives_in_timed_region": "2 barriers + all_gather of counts (int64), rows (f32 [M,9]) and pair ids (int64)"}


def rendezvous_check(world, rank, local_rank):
    """--rendezvous-check: the ranks find each other and exchange one number each (what the first seconds of a real run do)."""
    import torch.distributed as dist
    use_gpu = torch.cuda.is_available() and torch.cuda.device_count() >= world
    if use_gpu:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        dist.init_process_group("nccl", device_id=dev)
    else:
        dev = torch.device("cpu")
        dist.init_process_group("gloo")
    t = torch.tensor([float(rank)], device=dev, dtype=torch.float64)
    allt = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(allt, t)
    if rank == 0:
        print(json.dumps({"rendezvous": "ok", "backend": dist.get_backend(), "world": world,
                          "ranks": [int(x.item()) for x in allt]}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


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


def usable_cores():
    """Logical cores this process may actually use: os.cpu_count() capped by the cgroup CPU quota (the GPU boxes expose
    256 logical cores under a 16-core quota; thread teams wider than the quota get the whole process throttled)."""
    n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(ckpt, pyr1, pyr2, cfg):
    """The CPU oracle (a port of the reference algorithm, oracle/p2p_oracle.py) on the host cores of this box, on a
    bounded sample of the same workload: one warm-up + 5 repetitions of one full pair (coarse stage + filter_coarse(ptmax)
    + both regressors on all proposals), median.  min(32, cores the cgroup quota allows) threads: measured on the 2x64-core
    host, torch-CPU is fastest at 32 threads for these op sizes (8: 0.88 s, 32: 0.64 s, 128: 2.4 s for the coarse stage);
    more threads than the quota only get the process throttled.  The port is conservative
    for the comparison: its batched conv3d is faster than the reference's Python loop over conv3d slices; the
    unmodified reference itself cannot run on the GPU box (/root/reference is not there)."""
    threads = min(32, usable_cores())
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
    # what the UNMODIFIED reference would read: its time over the port's, both measured on the same cores of the build container
    # (tools/cpu_reference_time.py; /root/reference does not exist on the GPU box, so the ratio is a committed record)
    ratio = {}
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "cpu_reference_vs_port.json")))
        ratio = {"reference_over_port": rec["reference_over_port"],
                 "reference_equivalent_value": 1.0 / t_pair / rec["reference_over_port"],
                 "reference_over_port_source": f"{rec['source']} ({rec['host']}, {rec['threads']} threads: reference "
                                               f"{rec['reference_s_per_pair']} s, port {rec['port_s_per_pair']} s per pair)"}
    except (OSError, KeyError, ValueError):
        pass
    return {**ratio, "value": 1.0 / t_pair, "unit": "pairs/s", "cores": threads, "kind": "port",
            "sample": f"1 warm-up + {reps} x one full {cfg['H']}x{cfg['W']} pair (coarse + filter ptmax={cfg['ptmax']} + mid/fine "
                      f"regressors on {cfg['ptmax'] * cfg['panc']} proposals), median {t_pair:.2f} s; torch-CPU fp32 port of the "
                      f"reference algorithm (oracle/p2p_oracle.py), {threads} threads ({os.cpu_count()} logical cores, CPU quota "
                      f"{usable_cores()}), "
                      f"torch {torch.__version__}"}


def parity_check(net, ckpt, batch, cpu_pairs, cfg, pairs=None):
    """Push the benched batch through the SAME batched calls bench times (coarse_async / fine_from_ticket over the
    whole batch) and every pair of it (the first `pairs` of them) through the CPU oracle: coarse rows bit-exact?,
    sampled proposals equal?, regressed coordinates / scores max |diff|.  The fine level of the oracle is fed the
    kernel's own mid matches (a 1e-6 px wobble across an integer would move the whole fine patch by one pixel,
    networks/utils.py:19); `max_px_err_fine_chain` is the plain end-to-end difference over ALL checked pairs for the matches
    whose truncated mid coordinates equal the oracle's, `fine_patches_moved_by_trunc` counts the others (their fine patch
    sits one pixel off the oracle's; `max_px_err_fine_chain_moved` is their difference).  The product samples the
    ptmax proposals pair after pair from the global numpy RNG (networks/utils.py:55-63), the oracle from an equally
    seeded stream in the same order.  Where a coarse row differs from the fp32 oracle, the pair goes through the fp32
    error model of oracle/error_model.py: the row must be UNDECIDABLE in fp32 (the two candidates closer in an fp64
    evaluation than the rounding-error bound of an fp32 one -- the reference's own answer for such a row is an artefact
    of its summation order) and every decidable row of the pair must hold the fp64 winner; `coarse_indices_equal` is
    "every row decidable in fp32 is equal".  The proposals of such a pair differ legitimately and its fine stage is
    compared on the kernel's own proposals."""
    from oracle import p2p_oracle as orc
    from oracle.error_model import ErrorModel, assert_decidable_rows, differing_rows_are_near_ties, differing_rows_are_near_ties_local
    seed = 4242
    np.random.seed(seed)
    rng = np.random.RandomState(seed)
    f1, f2 = batch
    B = f1[0].shape[0] if pairs is None else min(pairs, f1[0].shape[0])
    ncn, mid_p, fine_p = orc.split_params(ckpt["state_dict"])
    with torch.no_grad():
        ticket = net.coarse_async(f1, f2, ksize=KSIZE)
        fine, fine_s, mid, mid_s, coarse = net.fine_from_ticket(ticket, ncn_thres=0.0, mutual=True, return_all=True,
                                                                ptmax=cfg["ptmax"])
        torch.cuda.synchronize()
        out = {"pairs_checked": B, "what": f"the first {B} pairs of the benched batch, through the batched calls the bench times",
               "coarse_rows": 0, "coarse_rows_differing": 0, "pairs_with_differing_rows": 0, "proposals": 0,
               "pairs_with_equal_proposals": 0, "max_px_err_mid": 0.0, "max_px_err": 0.0, "max_score_err": 0.0,
               "coarse_rows_differing_decidable": 0, "worst_gap_over_fp32_error_bound": 0.0}
        for b in range(B):
            p1, p2 = cpu_pairs[b]
            corr, delta = orc.coarse_forward(p1[4], p2[4], KSIZE, ncn)
            rows, sc = orc.cal_coarse_matches(corr, delta, KSIZE, 8)
            cm, _ = orc.filter_coarse(rows, sc, 0.0, True, ptmax=cfg["ptmax"], rng=rng)
            cm = orc.shift_to_anchors(cm, 8, cfg["panc"])
            got_rows, got_props, g_mid = ticket["matches"][b].cpu(), coarse[b].cpu(), mid[b].cpu()
            ndiff = int((got_rows != rows).any(dim=1).sum())
            out["coarse_rows"] += int(rows.shape[0])
            out["coarse_rows_differing"] += ndiff
            out["pairs_with_differing_rows"] += int(ndiff > 0)
            if ndiff:
                try:
                    if cfg["H"] > 480:       # the full fp64 model costs minutes at 960x1280: evaluated where the differing rows need it
                        _, worst = differing_rows_are_near_ties_local(got_rows, rows, p1[4], p2[4], ckpt["state_dict"], KSIZE)
                    else:
                        em = ErrorModel(p1[4], p2[4], ckpt["state_dict"], KSIZE)
                        em.check(corr, "oracle fp32 volume")
                        _, worst = differing_rows_are_near_ties(got_rows, rows, em)
                        assert_decidable_rows(got_rows, em)
                    out["worst_gap_over_fp32_error_bound"] = max(out["worst_gap_over_fp32_error_bound"], worst)
                except AssertionError as e:
                    out["coarse_rows_differing_decidable"] += ndiff
                    out.setdefault("errors", []).append(f"pair {b}: {e}")
            out["proposals"] += int(got_props.shape[0])
            same = bool(torch.equal(got_props, cm))
            out["pairs_with_equal_proposals"] += int(same)
            props = cm if same else got_props
            r_mid, r_ms, _ = orc.fine_level(p1[:4], p2[:4], props, mid_p)
            r_fine, r_fs, _ = orc.fine_level(p1[:4], p2[:4], g_mid, fine_p)
            out["max_px_err_mid"] = max(out["max_px_err_mid"], float((g_mid - r_mid).abs().max()))
            out["max_px_err"] = max(out["max_px_err"], float((fine[b].cpu() - r_fine).abs().max()))
            out["max_score_err"] = max(out["max_score_err"], float((mid_s[b].cpu() - r_ms).abs().max()),
                                       float((fine_s[b].cpu() - r_fs).abs().max()))
            # the plain chain: the oracle's fine level on the ORACLE's mid matches.  A mid coordinate within the rounding
            # error of an integer truncates differently (networks/utils.py:19) and moves that match's whole fine patch by
            # one pixel: such matches are counted, the others compared
            chain, _, _ = orc.fine_level(p1[:4], p2[:4], r_mid, fine_p)
            moved = (g_mid.long() != r_mid.long()).any(dim=1)
            out["fine_patches_moved_by_trunc"] = out.get("fine_patches_moved_by_trunc", 0) + int(moved.sum())
            if bool((~moved).any()):
                out["max_px_err_fine_chain"] = max(out.get("max_px_err_fine_chain", 0.0),
                                                   float((fine[b].cpu() - chain)[~moved].abs().max()))
            if bool(moved.any()):
                out["max_px_err_fine_chain_moved"] = max(out.get("max_px_err_fine_chain_moved", 0.0),
                                                         float((fine[b].cpu() - chain)[moved].abs().max()))
    out["coarse_indices_equal"] = out["coarse_rows_differing_decidable"] == 0
    out["proposals_equal"] = out["pairs_with_equal_proposals"] == B
    out.update({"tolerance_px": 1e-3, "tolerance_score": 1e-5})
    return out


def roofline_of(mode, events):
    M = MODES[mode]
    kern_ms = [a.elapsed_time(b) for a, b, _, _ in events]
    flop = sum(n * lv * FLOP_PER_PROPOSAL_LEVEL for _, _, n, lv in events) / max(len(events), 1)
    avg_ms = sum(kern_ms) / max(len(kern_ms), 1)
    # achieved = ALGORITHMIC flop of one regress launch (608.3 MFLOP per proposal and level) / its
    # average duration (HIP events on the launch stream)
    achieved = flop / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
    out = {"kernel": M["kernel"], "bound": "mfma", "achieved": achieved, "peak": M["peak"], "unit": "TFLOP/s",
           "frac": achieved / M["peak"], "traffic": None, "avg_launch_ms": avg_ms, "algorithmic_flop_per_launch": flop,
           "peak_note": M["peak_note"],
           "note": "achieved = algorithmic flop of the launch (608.3 MFLOP x proposals x levels) / launch time "
                   "measured with HIP events on the launch stream"}
    if M["products"]:
        out["power_ceiling_note"] = ("the nominal peak is not reachable on random data under the 1400 W package limit: a register-only "
                                     "v_mfma_f32_32x32x16_f16 loop measures 1.69 PFLOP/s (0.68 of 2.5; zeros: 2.49) = 563 TFLOP/s in "
                                     "three-product fp32-equivalent terms, and kernels that also move their operands sit at "
                                     "1.0-1.1 GHz of matrix-pipe issue against its 1.61 (profiles/r05_mfma_power_ceiling.txt, "
                                     "DESIGN.md section 4)")
    if mode == "fp16x2w" and avg_ms > 0:
        issued = sum(n * lv * ISSUED_FP16_FLOP_PER_PROPOSAL_LEVEL for _, _, n, lv in events) / max(len(events), 1)
        out["issued_fp16_flop_per_launch"] = issued
        out["issued_frac"] = issued / (avg_ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS
        out["issued_note"] = ("issued_frac = fp16 MFMA flop the call actually issues (conv1 with its cell-row de-duplication: 578.7 "
                              "MFLOP, conv2 as Winograd F(2x2,3x3): 402.7 MFLOP per proposal and level; three products per fp32 "
                              "product included) / time / 2500 TFLOP/s nominal: what the matrix pipe is asked to do, next to "
                              "`frac`, which prices the DIRECT convolutions' flop count against 2500 / 3")
    if "kernels" in M:
        out["kernels"] = M["kernels"]
        out["note"] = ("the fine stage of one step = ONE p2p_regress_batch call = per regressor level and chunk of <= 2560 proposals "
                       "regress_h2_kernel<true> (gather + conv1 -> transformed conv2 input) and wino_gemm_kernel (conv2 as 16 "
                       "batched GEMMs + BN + max-pool), then regress_fc_kernel per level; achieved = algorithmic flop of the call "
                       "(608.3 MFLOP x proposals x levels) / its duration between HIP events on the launch stream = the sum of "
                       "its kernels in the rocprofv3 summary")
    return out


def add_traffic(roof, mode, config, proposals_per_launch):
    """HBM traffic of the dominant kernel: PMC measurement (tools/collect_profiles.sh) of THIS source tree, same kernel,
    configuration and launch size -- otherwise it stays null."""
    tf = os.path.join(ROOT, "profiles", "regress_traffic.json")
    if not os.path.exists(tf):
        return
    rec = json.load(open(tf)).get(mode if config == "A" else f"{mode}@{config}", {})
    if (rec.get("kernel") == "+".join(MODES[mode].get("kernels", [MODES[mode]["kernel"]])) and rec.get("proposals_per_launch") == proposals_per_launch
            and rec.get("config", "A") == config and rec.get("source_hash") == source_hash()):
        roof["traffic"] = rec.get("hbm_bytes_per_launch")
        roof["traffic_source"] = rec.get("source")
        roof["traffic_note"] = ("FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE of the call's kernels = requests on the "
                                "L2's fabric side: Infinity-Cache hits are included, so this is an upper bound of the HBM "
                                "bytes; the compulsory bytes of a 16-pair launch are ~1 GB (16 x 61 MB of pyramids + 70 MB of "
                                "weights) -- plus, in the Winograd mode, the transformed conv2 input written once and read once "
                                "(512 KiB per proposal and level); the rest are L2 capacity misses of the conv1 weight stream, "
                                "which lives in the 256 MB Infinity Cache")
        if rec.get("per_kernel"):
            roof["traffic_per_kernel"] = rec["per_kernel"]


class Runner:
    """The benched loop over resident batches: `run(n)` = n steps, software-pipelined on one stream (the coarse stage of
    step i+1 is enqueued before the host samples the proposals of step i, so the GPU never waits for the host)."""

    def __init__(self, net, batches, ptmax, overlap=False, depth=1):
        self.net, self.batches, self.ptmax, self.depth = net, batches, ptmax, max(1, depth)
        self.coarse_stream = torch.cuda.Stream(device=net.device) if overlap else None

    def submit(self, i):
        f1, f2 = self.batches[i % len(self.batches)]
        if self.coarse_stream is None:
            return self.net.coarse_async(f1, f2, ksize=KSIZE)
        # experiment: the coarse stage of the NEXT step on its own stream, beside the regress launch of the current one
        with torch.cuda.stream(self.coarse_stream):
            return self.net.coarse_async(f1, f2, ksize=KSIZE)

    def finish(self, ticket):
        return self.net.fine_from_ticket(ticket, ncn_thres=0.0, mutual=True, ptmax=self.ptmax)

    def run(self, nsteps):
        out, tickets = [], [self.submit(j) for j in range(min(self.depth, nsteps))]
        for i in range(nsteps):
            if i + self.depth < nsteps:
                tickets.append(self.submit(i + self.depth))      # `depth` coarse stages enqueued ahead of the step being finished
            out.append(self.finish(tickets.pop(0)))
        return out


def resident_batches(cfg, rank, dev, nbatches, on_device=False):
    """`nbatches` distinct synthetic batches of cfg['pairs_per_step'] pairs, resident in HBM; the CPU copies are kept for
    the oracle legs (on_device: generated on the GPU, no CPU copy -- legs without an oracle)."""
    from patch2pix_amd.utils import synthetic
    B, H, W = cfg["pairs_per_step"], cfg["H"], cfg["W"]
    if on_device:
        cpu_pairs = None
        pairs = [synthetic.make_correlated_pyramids_device(1000 + rank * 64 + i, H, W, dev) for i in range(nbatches * B)]
    else:
        pairs = cpu_pairs = [synthetic.make_correlated_pyramids(1000 + rank * 64 + i, H, W) for i in range(nbatches * B)]
    batches = []
    for k in range(nbatches):
        chunk = pairs[k * B:(k + 1) * B]
        batches.append(([torch.stack([p[0][j] for p in chunk]).to(dev) for j in range(5)],
                        [torch.stack([p[1][j] for p in chunk]).to(dev) for j in range(5)]))
    return cpu_pairs, batches


def coarse_stage_ms(net, batch, reps=3):
    """Per-pair time of the coarse stage alone (forward_coarse_match + cal_coarse_matches), HIP events on the launch stream."""
    f1, f2 = batch
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    with torch.no_grad():
        corr, delta = net.forward_coarse_match(f1[4], f2[4], ksize=KSIZE)
        torch.cuda.synchronize()
        ev[0].record()
        for _ in range(reps):
            corr, delta = net.forward_coarse_match(f1[4], f2[4], ksize=KSIZE)
            net.cal_coarse_matches(corr, delta, ksize=KSIZE, upsample=net.upsample, center=True)
        ev[1].record()
        torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps / f1[4].shape[0]


# algorithmic flop per pair of the two MFMA kernels of the coarse stage: correlation 2 nA nB 256; consensus
# 4 layer applications x 2 x cells x 16 x 81
def coarse_roofline(net, batch, config):
    """Per-kernel roofline objects of the coarse stage.  nc_fused_kernel is timed LIVE (HIP events around
    p2p_neigh_consensus_batch on the batch's pooled volumes: that call is exactly one launch of it); corr_pool_kernel has no
    entry point of its own, its time comes from the rocprofv3 record of this source tree (profiles/kernel_times.json, keyed by
    the hash of csrc/ like roofline.traffic; null when the sources changed since).  MFMA busy x clock from the PMC pass of the
    same record."""
    from patch2pix_amd import ops
    f1, f2 = batch
    B = f1[4].shape[0]
    ha, wa = f1[4].shape[-2] // KSIZE, f1[4].shape[-1] // KSIZE
    hb, wb = f2[4].shape[-2] // KSIZE, f2[4].shape[-1] // KSIZE
    cells = ha * wa * hb * wb
    flop_corr = 2.0 * (ha * wa * KSIZE * KSIZE) * (hb * wb * KSIZE * KSIZE) * f1[4].shape[1]
    flop_nc = 4 * 2.0 * cells * 16 * 81
    peak = PEAK_BF16_MFMA_TFLOPS / 3.0
    out = {"peak": peak, "unit": "TFLOP/s", "bound": "mfma",
           "peak_note": "2500 TFLOP/s dense fp16 MFMA / 3 MFMA products per fp32 product (both kernels compute in fp16x2)"}
    with torch.no_grad():
        corr, _ = net.forward_coarse_match(f1[4], f2[4], ksize=KSIZE)
        x = corr.reshape(B, ha, wa, hb, wb).contiguous()
        ncn = net._weights()[0]
        for _ in range(2):
            ops.neigh_consensus_batch(x, ncn)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        torch.cuda.synchronize()
        reps = 5
        ev[0].record()
        for _ in range(reps):
            ops.neigh_consensus_batch(x, ncn)
        ev[1].record()
        torch.cuda.synchronize()
    nc_ms = ev[0].elapsed_time(ev[1]) / reps
    out["nc_fused_kernel"] = {"avg_launch_ms": nc_ms, "pairs_per_launch": B, "algorithmic_flop_per_launch": flop_nc * B,
                              "achieved": flop_nc * B / (nc_ms * 1e-3) / 1e12, "frac": flop_nc * B / (nc_ms * 1e-3) / 1e12 / peak,
                              "timing": "live: HIP events around p2p_neigh_consensus_batch (one launch per call), alone on the GPU"}
    rec = {}
    try:
        allrec = json.load(open(os.path.join(ROOT, "profiles", "kernel_times.json")))
        rec = allrec.get(config, {})
        if rec.get("source_hash") != source_hash():
            rec = {}
    except (OSError, ValueError):
        pass
    k = rec.get("kernels", {})
    if "corr_pool_kernel" in k:
        us = k["corr_pool_kernel"]["avg_us"]
        out["corr_pool_kernel"] = {"avg_launch_ms": us / 1e3, "pairs_per_launch": rec.get("pairs_per_step"),
                                   "algorithmic_flop_per_launch": flop_corr * rec.get("pairs_per_step", B),
                                   "achieved": flop_corr * rec.get("pairs_per_step", B) / (us * 1e-6) / 1e12,
                                   "frac": flop_corr * rec.get("pairs_per_step", B) / (us * 1e-6) / 1e12 / peak,
                                   "timing": f"rocprofv3 kernel trace of bench.py inside the step ({rec.get('source')})"}
    else:
        out["corr_pool_kernel"] = None
    for name in ("nc_fused_kernel", "corr_pool_kernel"):
        if out.get(name) and name in k:
            for f in ("mfma_busy_fraction", "effective_clock_ghz", "lds_bank_conflict_fraction"):
                if k[name].get(f) is not None:
                    out[name][f] = k[name][f]
            if "nc_fused_kernel" == name:
                out[name]["avg_us_inside_the_step"] = k[name]["avg_us"]
    return out


def config_leg(net, name, mode, rank, dev, steps=3, warmup=1):
    """A short GPU-only measurement of another BASELINE configuration with the same loop (no oracle, no CPU legs)."""
    from patch2pix_amd import ops
    cfg = dict(CONFIGS[name])
    panc0 = net.panc
    net.panc = cfg["panc"]
    try:
        _, batches = resident_batches(cfg, rank, dev, 1, on_device=True)
        r = Runner(net, batches, cfg["ptmax"])
        np.random.seed(4321)
        with torch.no_grad():
            r.run(warmup)
            torch.cuda.synchronize()
            ops.regress_events = []
            t0 = time.perf_counter()
            r.run(steps)
            torch.cuda.synchronize()
            elapsed = time.perf_counter() - t0
            events, ops.regress_events = ops.regress_events, None
        roof = roofline_of(mode, events)
        add_traffic(roof, mode, name, cfg["pairs_per_step"] * cfg["ptmax"] * cfg["panc"])
        coarse_ms = coarse_stage_ms(net, batches[0])
        B = cfg["pairs_per_step"]
        return {"workload": cfg["workload"], "value": steps * B / elapsed, "unit": "pairs/s", "steps": steps, "warmup": warmup,
                "pairs_per_step": B, "ms_per_step": elapsed / steps * 1e3, "regress_mode": mode,
                "coarse_stage_ms_per_pair": coarse_ms, "regress_launch_ms": roof["avg_launch_ms"],
                "roofline": {k: roof[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms",
                                                 "algorithmic_flop_per_launch")},
                "data": "synthetic (pyramids generated on the device), no oracle leg: parity at this size is "
                        "tests/test_gpu_parity.py::test_config_E_vs_oracle"}
    finally:
        net.panc = panc0
        del batches
        torch.cuda.empty_cache()


def stream_mode(args, net, cfg, mode, rank, world, dev, dist):
    """BASELINE configs[3]: `--pairs N` seeded pairs, pair i on rank i % world, each chunk's pyramids generated on the
    device from the pair ids (10 000 x 71 MB cannot be resident), results exchanged every 8 chunks.  Strong scaling:
    total work fixed; `scaling_efficiency` = value / (world x the rate of rank 0 running ALONE on a few chunks of the same
    stream before the timed region)."""
    from patch2pix_amd.gather import run_pair_stream
    from patch2pix_amd.utils import synthetic
    H, W, B, PTMAX = cfg["H"], cfg["W"], cfg["pairs_per_step"], cfg["ptmax"]

    def submit(pair_ids):
        pairs = [synthetic.make_correlated_pyramids_device(pid, H, W, dev) for pid in pair_ids]
        f1 = [torch.stack([p[0][j] for p in pairs]) for j in range(5)]
        f2 = [torch.stack([p[1][j] for p in pairs]) for j in range(5)]
        t = net.coarse_async(f1, f2, ksize=KSIZE)
        t["ids"] = pair_ids
        return t

    def finish(ticket):
        np.random.seed(ticket["ids"][0])          # the ptmax sample of a chunk depends on its pair ids only, not on the sharding
        return net.fine_from_ticket(ticket, ncn_thres=0.0, mutual=True, ptmax=PTMAX)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        t_spin = time.perf_counter()
        while time.perf_counter() - t_spin < float(os.environ.get("P2P_BENCH_SPINUP", "1.5")):                      # clocks / allocator / page cache (see main)
            run_pair_stream(2 * B, 0, 1, B, submit, finish, exchange=False)
            torch.cuda.synchronize()
        # the exchange once on every rank before the clock starts (packing kernels, RCCL channels: first use costs tens of ms)
        run_pair_stream(min(args.pairs, 2 * B * world), rank, world, B, submit, finish, gather_every=1, device=dev)
        barrier()
        solo = None
        if rank == 0:                                                  # single-GPU reference: rank 0 alone, the others idle
            n_solo = min(args.pairs, 6 * B)
            t0 = time.perf_counter()
            run_pair_stream(n_solo, 0, 1, B, submit, finish, exchange=False)
            torch.cuda.synchronize()
            solo = n_solo / (time.perf_counter() - t0)
        barrier()
        t0 = time.perf_counter()
        rows, ids, mine = run_pair_stream(args.pairs, rank, world, B, submit, finish, gather_every=8, device=dev)
        torch.cuda.synchronize()
        busy = time.perf_counter() - t0
        barrier()
        elapsed = time.perf_counter() - t0
    assert int(torch.unique(ids).numel()) == args.pairs and int(rows.shape[0]) == args.pairs * PTMAX * cfg["panc"]
    per_rank = [mine / busy]
    if dist is not None:
        t = torch.tensor([elapsed, mine / busy], device=dev, dtype=torch.float64)
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        elapsed = max(float(x[0]) for x in allt)
        per_rank = [float(x[1]) for x in allt]
    if rank != 0:
        return None
    value = args.pairs / elapsed
    return {"metric": f"image-pairs/sec ({H}x{W}, ptmax={PTMAX}), matching hot path over a stream of {args.pairs} seeded pairs",
            "value": value, "unit": "pairs/s", "n_gpus": world, "steps": -(-args.pairs // (B * world)), "warmup": 0,
            "ms_per_step": elapsed / max(1, -(-args.pairs // (B * world))) * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": MODES[mode]["dtype"], "data": "synthetic",
            "config": {"workload": f"BASELINE configs[3]: stream of {args.pairs} seeded {H}x{W} pairs (pyramids generated on the "
                                   f"device from the pair id, chunk by chunk), pair i on rank i % {world}, {B} pairs per chunk, "
                                   f"ptmax={PTMAX}; match arrays exchanged every 8 chunks (RCCL all_gather)",
                       "pairs": args.pairs, "pairs_per_step": B, "regress_mode": mode,
                       "parallelism": f"pair_id % {world}, no data-path collective"},
            "per_rank_pairs_per_s": per_rank, "single_gpu_reference_pairs_per_s": solo,
            "scaling_efficiency": value / (world * solo) if solo else None,
            "note": "a step here = one chunk per rank; the timed region includes the on-device generation of every chunk's "
                    "pyramids (the stand-in for the backbone producer) and every exchange"}


def _sync(dev):
    if torch.device(dev).type == "cuda":
        torch.cuda.synchronize()


def timed_steps(run, nsteps, with_gather, rank, world, pairs_per_step, dev, dist):
    """The timed region of the weak-scaling contract: barrier + synchronize, `run(nsteps)` on this rank, the final gather
    of the match arrays (the only inter-GPU exchange of the path), barrier + synchronize.  -> elapsed (this rank's clock
    over the whole region), local_elapsed (up to the end of this rank's own steps), the regress launch events, rows gathered."""
    from patch2pix_amd import ops
    from patch2pix_amd.gather import gather_matches, pack_results

    def barrier():
        if dist is not None:
            dist.barrier()
        _sync(dev)

    barrier()
    ops.regress_events = []
    t0 = time.perf_counter()
    results = run(nsteps)
    _sync(dev)
    local_elapsed = time.perf_counter() - t0
    nrows = None
    if with_gather:
        all_rows, all_ids = gather_matches(*pack_results(results, rank, world, pairs_per_step, device=dev))
        nrows = all_rows.shape[0]
        assert all_ids.dtype == torch.int64 and all_ids.shape[0] == nrows
    barrier()
    elapsed = time.perf_counter() - t0
    events, ops.regress_events = ops.regress_events, None
    return {"elapsed": elapsed, "local_elapsed": local_elapsed, "events": events, "nrows": nrows}


def exchange_rank_stats(dist, dev, world, values):
    """all_gather of a few float64 numbers per rank -> list (by rank) of lists."""
    if dist is None:
        return [list(values)]
    t = torch.tensor(values, device=dev, dtype=torch.float64)
    allt = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(allt, t)
    return [[float(v) for v in x] for x in allt]


def main():
    args = parse()
    cfg = dict(CONFIGS[args.config])
    if args.pairs_per_step:
        cfg["pairs_per_step"] = args.pairs_per_step
    H, W, PTMAX, B = cfg["H"], cfg["W"], cfg["ptmax"], cfg["pairs_per_step"]
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)                   # does not return: the plain command becomes its own launcher
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node and --gpus must agree")
    if args.rendezvous_check:
        if world > 1:
            rendezvous_check(world, rank, local_rank)
        elif rank == 0:
            print(json.dumps({"rendezvous": "ok", "backend": None, "world": 1, "ranks": [0]}), flush=True)
        return
    # host side of a rank is one Python thread plus small torch-CPU ops: keep N ranks from oversubscribing the host, and
    # every thread team well inside the CPU quota (a team as wide as the quota, spinning after a parallel region, gets the
    # whole process throttled for the rest of the scheduler period: one such stall costs a 20-step run 7 %)
    torch.set_num_threads(max(1, min(8, usable_cores() // 2 // max(world, 1))))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # the process that drives a GPU belongs on the GPU's socket (utils/host.py; profiles/r03_host_effects.txt)
    pinned_cpus = None
    if os.environ.get("P2P_NUMA_PIN", "1") != "0":
        from patch2pix_amd.utils.host import pin_process_to_gpu
        pinned_cpus = pin_process_to_gpu(local_rank)
    dist, dist_info = init_dist(world, dev)

    from patch2pix_amd import ops
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

    if args.pairs:
        out = stream_mode(args, net, cfg, mode, rank, world, dev, dist)
        if rank == 0:
            print(json.dumps(out), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return

    # a few distinct synthetic batches per rank, resident in HBM before the clock starts
    cpu_pairs, batches = resident_batches(cfg, rank, dev, 2)
    np.random.seed(1234 + rank)
    runner = Runner(net, batches, PTMAX, overlap=bool(args.overlap), depth=args.depth)
    run = runner.run

    def timed(nsteps, with_gather):
        t = timed_steps(run, nsteps, with_gather, rank, world, B, dev, dist)
        return t["elapsed"], t["events"], t["nrows"], t["local_elapsed"]

    with torch.no_grad():
        # untimed spin-up: a fresh box needs ~1 s of work before clocks / allocator / page cache settle
        # (the first process on a cold box otherwise measures ~30 % low), then the W warm-up steps
        t_spin = time.perf_counter()
        while time.perf_counter() - t_spin < float(os.environ.get("P2P_BENCH_SPINUP", "1.5")):
            run(2)
            torch.cuda.synchronize()
        # the W warm-up steps go through the same region as the timed ones, exchange included: the first use of the
        # packing kernels and of the collectives (lazy code-object loads, RCCL channel set-up) costs tens of milliseconds
        if args.warmup > 0:
            timed(args.warmup, True)
        elapsed, events, nrows, local_elapsed = timed(args.steps, True)
    assert nrows == world * args.steps * B * PTMAX * cfg["panc"]
    # every rank's own numbers (weak scaling: each ran its own K steps): its rate up to the end of its own work, and the
    # roofline of its regress launches -- gathered so that rank 0 can print them; `elapsed` = the slowest rank's time
    my_roof = roofline_of(mode, events)
    per_rank = exchange_rank_stats(dist, dev, world, [elapsed, args.steps * B / local_elapsed, my_roof["avg_launch_ms"], my_roof["frac"]])
    elapsed = max(r[0] for r in per_rank)

    value = world * args.steps * B / elapsed
    if rank == 0:
        roof = roofline_of(mode, events)
        add_traffic(roof, mode, args.config, B * PTMAX * cfg["panc"])
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
            "per_gpu_pairs_per_s": value / world,
            "dist": dist_info,
            "per_rank_pairs_per_s": [r[1] for r in per_rank],
            "per_rank_regress_launch_ms": [r[2] for r in per_rank],
            "per_rank_roofline_frac": [r[3] for r in per_rank],
            "host": {"threads_pinned_to_gpu_local_cpus": len(pinned_cpus) if pinned_cpus else 0,
                     "torch_cpu_threads": torch.get_num_threads(), "usable_cores": usable_cores(),
                     "coarse_stages_enqueued_ahead": args.depth},
        }
    # ---- outside the timed region (single-GPU runs only) ----
    if rank == 0 and world == 1:
        out["coarse_stage_ms_per_pair"] = coarse_stage_ms(net, batches[0])
        try:
            out["coarse_roofline"] = coarse_roofline(net, batches[0], args.config)
        except Exception as e:       # informational leg; never fail the bench line on it
            out["coarse_roofline"] = {"error": repr(e)}
        if not args.no_parity:
            out["parity"] = parity_check(net, ckpt, batches[0], cpu_pairs, cfg)
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
                    e2, ev2, _, _ = timed(n2, False)
                r2 = roofline_of(m2, ev2)
                other[m2] = {"value": n2 * B / e2, "unit": "pairs/s", "dtype": MODES[m2]["dtype"],
                             "roofline_frac": r2["frac"], "achieved": r2["achieved"], "peak": r2["peak"],
                             "avg_launch_ms": r2["avg_launch_ms"]}
                if not args.no_parity:
                    p2 = parity_check(net, ckpt, batches[0], cpu_pairs, cfg, pairs=1)
                    other[m2].update({k: p2.get(k) for k in ("max_px_err_mid", "max_px_err", "max_score_err")})
            for w in net._weights()[1:]:
                w.set_mode(mode)
            out["other_modes"] = other
        if not args.no_other_configs and args.config == "A":
            try:
                out["other_configs"] = {"E": config_leg(net, "E", mode, rank, dev)}
            except Exception as e:       # informational leg; never fail the bench line on it
                out["other_configs"] = {"E": {"error": repr(e)}}
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
