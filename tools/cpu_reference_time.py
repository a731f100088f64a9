"""The UNMODIFIED reference (through oracle/ref_shim.py) against the oracle port on the SAME host cores, hot path of one
480x640 pair (coarse stage + filter_coarse(ptmax=400) + both regressors): shows what bench.py's `cpu_baseline` (kind
"port"; the reference tree does not exist on the GPU box) stands for.  Build container only (needs /root/reference)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import make_golden as mg, p2p_oracle as orc
from patch2pix_amd.utils import synthetic
threads = int(os.environ.get("THREADS", os.cpu_count()))
torch.set_num_threads(threads)
H, W, PTMAX = 480, 640, 400
sd = synthetic.make_state_dict(0)
ref = mg.load_reference()
net = mg.build_reference_net(sd, synthetic.default_regressor_config())
p1, p2 = synthetic.make_correlated_pyramids(1000, H, W)
f1, f2 = [t[None] for t in p1], [t[None] for t in p2]

def reference_pair():
    with torch.no_grad():
        corr, delta = net.forward_coarse_match(p1[4][None], p2[4][None], ksize=2)
        m, s = net.cal_coarse_matches(corr, delta, ksize=2, upsample=8, center=True)
        np.random.seed(0)
        cm, cs = ref.utils.filter_coarse(m, s, 0.0, True, ptmax=PTMAX)
        mid, _ = net.forward_fine_match(f1, f2, cm, 16, "center", net.regress_mid)
        fine, _ = net.forward_fine_match(f1, f2, mid, 16, "center", net.regress_fine)
    return fine

ncn, mid_p, fine_p = orc.split_params(sd)
def port_pair():
    with torch.no_grad():
        corr, delta = orc.coarse_forward(p1[4], p2[4], 2, ncn)
        m, s = orc.cal_coarse_matches(corr, delta, 2, 8)
        cm, _ = orc.filter_coarse(m, s, 0.0, True, ptmax=PTMAX, rng=np.random.RandomState(0))
        mid, _, _ = orc.fine_level(p1[:4], p2[:4], cm, mid_p)
        fine, _, _ = orc.fine_level(p1[:4], p2[:4], mid, fine_p)
    return fine

for name, fn in (("reference (unmodified, shimmed)", reference_pair), ("oracle port", port_pair)):
    fn()
    ts = []
    for _ in range(int(os.environ.get("REPS", "5"))):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    med = sorted(ts)[len(ts) // 2]
    print(f"{name:34s}: median {med:.2f} s per pair = {1 / med:.3f} pairs/s  ({threads} threads, torch {torch.__version__})", flush=True)
