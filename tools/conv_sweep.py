"""Time every tile variant of the backbone convolution (ops.FORCED_CONV_TILE -> p2p_conv_set_tile) per layer shape and batch size."""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from patch2pix_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
H, W = int(os.environ.get("H", 480)), int(os.environ.get("W", 640))


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for ci, co, ks, st, h, w in ((64, 64, 3, 1, H // 4, W // 4), (64, 128, 3, 2, H // 4, W // 4), (128, 128, 3, 1, H // 8, W // 8),
                             (128, 256, 3, 1, H // 8, W // 8), (256, 256, 3, 1, H // 8, W // 8), (128, 256, 1, 1, H // 8, W // 8)):
    conv = torch.nn.Conv2d(ci, co, ks, st, ks // 2, bias=False).to(dev)
    bn = torch.nn.BatchNorm2d(co).to(dev).eval()
    cv = ops.ConvBN(conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, st, dev)
    tiles = {256: ["2,4,2", "1,4,2", "2,2,2", "1,2,2"], 128: ["2,2,2", "1,2,2"], 64: ["1,2,1"]}[co]
    for nb in (2, 4, 8, 16, 32):
        x = torch.relu(torch.randn(nb, h, w, ci, device=dev))
        xm = ops.absmax_batch(x)
        fl = 2.0 * nb * (h // st) * (w // st) * co * ci * ks * ks
        res = []
        for t in tiles + ["auto"]:
            ops.FORCED_CONV_TILE = None if t == "auto" else tuple(int(v) for v in t.split(","))
            ms = timed(lambda: cv.forward(x, xm))
            res.append(f"{t}: {ms:.3f} ms ({fl / ms / 1e9:.0f} TF)")
        print(f"conv {ci:3d}->{co:3d} k{ks} s{st} {h}x{w} x{nb:2d}:  " + "   ".join(res), flush=True)
