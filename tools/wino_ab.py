"""A/B of the regressor arithmetic modes on one box: launch time of p2p_regress_batch (both levels) for B pairs x N proposals,
agreement between the modes, and the error of each against the fp64 oracle on a few proposals.
usage: python tools/wino_ab.py [B=16] [N=400] [H=480] [W=640] [modes=fp16x2,fp16x2w]"""
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from patch2pix_amd import ops  # noqa: E402
from patch2pix_amd.utils import synthetic  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    H = int(sys.argv[3]) if len(sys.argv) > 3 else 480
    W = int(sys.argv[4]) if len(sys.argv) > 4 else 640
    modes = (sys.argv[5] if len(sys.argv) > 5 else "fp16x2,fp16x2w").split(",")
    reps = int(os.environ.get("REPS", "12"))
    dev = torch.device("cuda:0")
    sd = synthetic.make_state_dict(0, backbone=False)
    sub = lambda p: {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}
    mid, fine = ops.RegressorWeights(sub("regress_mid."), dev), ops.RegressorWeights(sub("regress_fine."), dev)
    npyr = min(B, 4)
    pyr1 = [[t.to(dev) for t in synthetic.make_pyramid(100 + i, H, W)[:4]] for i in range(npyr)]
    pyr2 = [[t.to(dev) for t in synthetic.make_pyramid(200 + i, H, W)[:4]] for i in range(npyr)]
    g = torch.Generator().manual_seed(9)
    props = [torch.stack([torch.randint(0, W + 1, (N,), generator=g), torch.randint(0, H + 1, (N,), generator=g),
                          torch.randint(0, W + 1, (N,), generator=g), torch.randint(0, H + 1, (N,), generator=g)], 1).to(dev)
             for _ in range(B)]
    p1 = [pyr1[i % npyr] for i in range(B)]
    p2 = [pyr2[i % npyr] for i in range(B)]
    outs = {}
    for rnd in range(2):                 # two rounds: the order of the modes must not matter
        for mode in modes:
            mid.set_mode(mode); fine.set_mode(mode)
            for _ in range(2):
                out = ops.regress_batch(mid, fine, p1, p2, props, want_raw=True)
            torch.cuda.synchronize()
            ts = []
            for _ in range(reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                out = ops.regress_batch(mid, fine, p1, p2, props, want_raw=True)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            ts.sort()
            outs[mode] = {k: torch.cat([o[k] for o in out]).cpu() for k in ("matches1", "matches2", "probs1", "probs2", "raw1", "raw2")}
            print(f"round {rnd} mode {mode:8s}: {B} x {N} proposals x 2 levels: median {ts[len(ts) // 2]:.3f} ms, min {ts[0]:.3f}, max {ts[-1]:.3f}",
                  flush=True)
    ref = outs[modes[0]]
    for mode in modes[1:]:
        o = outs[mode]
        print(f"{mode} vs {modes[0]}: " + ", ".join(f"{k} {float((o[k] - ref[k]).abs().max()):.2e}" for k in o))
    if os.environ.get("ORACLE", "1") != "0":
        from oracle import p2p_oracle as orc
        _, mid_p, fine_p = orc.split_params(sd, torch.float64)
        k = 24
        cp1 = [t.double().cpu() for t in p1[0]]
        cp2 = [t.double().cpu() for t in p2[0]]
        pr = props[0][:k].cpu()
        t0 = time.time()
        ref_mid, ref_midp, ref_raw = orc.fine_level(cp1, cp2, pr, mid_p)
        for mode in modes:
            o = outs[mode]
            ref_fine, ref_finep, _ = orc.fine_level(cp1, cp2, o["matches1"][:k].double(), fine_p)
            print(f"{mode} vs fp64 oracle ({k} proposals): mid {float((o['matches1'][:k] - ref_mid).abs().max()):.2e} px, "
                  f"fine (fed the kernel's mid) {float((o['matches2'][:k] - ref_fine).abs().max()):.2e} px, "
                  f"scores {float((o['probs1'][:k] - ref_midp).abs().max()):.1e} / {float((o['probs2'][:k] - ref_finep).abs().max()):.1e}")
        print(f"(oracle {time.time() - t0:.1f} s)")


if __name__ == "__main__":
    main()
