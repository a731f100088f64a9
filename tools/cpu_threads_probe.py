"""Time the CPU oracle's coarse stage and a small fine-stage sample at several thread counts
(used once to pick a fair thread count for bench.py's cpu_baseline leg)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import p2p_oracle as orc
from patch2pix_amd.utils import synthetic

sd = synthetic.make_state_dict(0, backbone=False)
ncn, mid_p, _ = orc.split_params(sd)
p1, p2 = synthetic.make_correlated_pyramids(1000, 480, 640)
props = torch.randint(8, 400, (16, 4))
for th in (8, 16, 32, 64, 128):
    torch.set_num_threads(th)
    with torch.no_grad():
        t0 = time.perf_counter(); orc.coarse_forward(p1[4], p2[4], 2, ncn); t1 = time.perf_counter()
        orc.fine_level(p1[:4], p2[:4], props, mid_p); t2 = time.perf_counter()
    print(f"threads {th:4d}: coarse {t1-t0:6.2f} s   fine(16 proposals, 1 level) {t2-t1:6.2f} s", flush=True)
