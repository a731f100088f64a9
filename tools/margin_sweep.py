"""Error margin of the regressor arithmetic modes against an fp64 evaluation over a sweep of checkpoint statistics
(VERDICT item 8): convolution / FC weight scale x0.25 ... x4, BN running_var spread, feature magnitude.  Prints, per setting
and mode, the max |error| of the regressed coordinates (px) of both levels; the bar is 1e-3 px."""
import sys, os; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from oracle import p2p_oracle as orc
from patch2pix_amd import ops
from patch2pix_amd.utils import synthetic
dev = torch.device("cuda:0")
H, W, n = 96, 128, 200
p1 = synthetic.make_pyramid(7, H, W); p2 = synthetic.make_pyramid(8, H, W)
g = torch.Generator().manual_seed(9)
props = torch.stack([torch.randint(0, W + 1, (n,), generator=g), torch.randint(0, H + 1, (n,), generator=g),
                     torch.randint(0, W + 1, (n,), generator=g), torch.randint(0, H + 1, (n,), generator=g)], 1)
print(f"{'setting':34s} " + " ".join(f"{m:>21s}" for m in ("f32", "fp16x2", "fp16x2w")) + "   (max |err| mid px / fine px vs fp64)")
worst = {}
# the regular grid, then the range extremes that matter for the fp16 planes (features 1e-3 ... 3e3, weights 0.01 ... 50)
GRID = [(w, v, f) for w in (0.25, 0.5, 1.0, 2.0, 4.0) for v in (1.0, 8.0) for f in (1.0, 30.0)]
GRID += [(1.0, 1.0, 1e-3), (1.0, 1.0, 3e3), (0.01, 1.0, 1.0), (50.0, 1.0, 1.0), (0.05, 8.0, 1e-2)]
for wscale, var_spread, fscale in GRID:
    if True:
        if True:
            sd = synthetic.make_state_dict(0, backbone=False)
            gen = torch.Generator().manual_seed(123)
            for k in list(sd):
                if k.startswith("regress_") and (".conv.0.weight" in k or ".conv.2.weight" in k or (".fc." in k and k.endswith("weight") and sd[k].dim() == 2)):
                    sd[k] = sd[k] * wscale
                if k.startswith("regress_") and k.endswith("running_var"):
                    sd[k] = sd[k] * torch.exp((torch.rand(sd[k].shape, generator=gen) - 0.5) * 2 * torch.log(torch.tensor(var_spread)))
            q1 = [t * (fscale if i else 1.0) for i, t in enumerate(p1[:4])]      # backbone feature magnitude (image level untouched)
            q2 = [t * (fscale if i else 1.0) for i, t in enumerate(p2[:4])]
            sub = lambda p: {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}
            mid = ops.RegressorWeights(sub("regress_mid."), dev); fine = ops.RegressorWeights(sub("regress_fine."), dev)
            _, mid64, fine64 = orc.split_params(sd, torch.float64)
            d1, d2 = [t.double() for t in q1], [t.double() for t in q2]
            ref_m, _, _ = orc.fine_level(d1, d2, props, mid64)
            g1 = [t.to(dev) for t in q1]; g2 = [t.to(dev) for t in q2]
            cells = []
            for mode in ("f32", "fp16x2", "fp16x2w"):
                mid.set_mode(mode); fine.set_mode(mode)
                out = ops.regress(mid, fine, g1, g2, props.to(dev))
                torch.cuda.synchronize()
                e_m = (out["matches1"].cpu().double() - ref_m).abs().max().item()
                rf, _, _ = orc.fine_level(d1, d2, out["matches1"].cpu().double(), fine64)
                e_f = (out["matches2"].cpu().double() - rf).abs().max().item()
                worst[mode] = max(worst.get(mode, 0.0), e_m, e_f)
                cells.append(f"{e_m:9.2e} /{e_f:9.2e}")
            print(f"w x{wscale:<4} var spread x{var_spread:<3} feat x{fscale:<4}   " + "  ".join(cells), flush=True)
print("worst over the sweep:", {k: f"{v:.2e}" for k, v in worst.items()}, " bar 1e-3 px")
