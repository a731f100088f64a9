import sys, os; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from oracle import p2p_oracle as orc
from patch2pix_amd import ops
from patch2pix_amd.utils import synthetic
dev = torch.device("cuda:0")
sd = synthetic.make_state_dict(0, backbone=False)
sub = lambda p: {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}
mid = ops.RegressorWeights(sub("regress_mid."), dev); fine = ops.RegressorWeights(sub("regress_fine."), dev)
_, mid_p, fine_p = orc.split_params(sd)
_, mid_p64, fine_p64 = orc.split_params(sd, torch.float64)
for (H, W, n) in [(96, 128, 257), (480, 640, 400)]:
    p1 = synthetic.make_pyramid(7, H, W); p2 = synthetic.make_pyramid(8, H, W)
    g = torch.Generator().manual_seed(9)
    props = torch.stack([torch.randint(0, W + 1, (n,), generator=g), torch.randint(0, H + 1, (n,), generator=g),
                         torch.randint(0, W + 1, (n,), generator=g), torch.randint(0, H + 1, (n,), generator=g)], 1)
    props[0] = torch.tensor([0, 0, W, H]); props[1] = torch.tensor([W, H, 0, 0])
    g1 = [t.to(dev) for t in p1[:4]]; g2 = [t.to(dev) for t in p2[:4]]
    ref_m, ref_p, ref_raw = orc.fine_level([t.double() for t in p1[:4]], [t.double() for t in p2[:4]], props, mid_p64)
    for mode in ("f32", "fp16x2", "fp16x2w"):
        mid.set_mode(mode); fine.set_mode(mode)
        out = ops.regress(mid, fine, g1, g2, props.to(dev), want_raw=True)
        torch.cuda.synchronize()
        e_raw = (out["raw1"].cpu().double() - ref_raw).abs().max().item()
        e_m = (out["matches1"].cpu().double() - ref_m).abs().max().item()
        e_p = (out["probs1"].cpu().double() - ref_p).abs().max().item()
        rf, rp, _ = orc.fine_level([t.double() for t in p1[:4]], [t.double() for t in p2[:4]], out["matches1"].cpu().double(), fine_p64)
        e_f = (out["matches2"].cpu().double() - rf).abs().max().item()
        e_fp = (out["probs2"].cpu().double() - rp).abs().max().item()
        print(f"{H}x{W} n={n} {mode:7s}: raw {e_raw:.2e}  mid px {e_m:.2e}  mid score {e_p:.2e}  fine px {e_f:.2e} fine score {e_fp:.2e}", flush=True)
