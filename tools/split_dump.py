import sys, os; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from patch2pix_amd import ops
from patch2pix_amd.utils import synthetic
dev = torch.device("cuda:0")
sd = synthetic.make_state_dict(0, backbone=False)
sub = lambda p: {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}
mid = ops.RegressorWeights(sub("regress_mid."), dev)
H, W, n = 96, 128, 64
p1 = synthetic.make_pyramid(7, H, W); p2 = synthetic.make_pyramid(8, H, W)
g = torch.Generator().manual_seed(9)
props = torch.stack([torch.randint(0, W + 1, (n,), generator=g), torch.randint(0, H + 1, (n,), generator=g),
                     torch.randint(0, W + 1, (n,), generator=g), torch.randint(0, H + 1, (n,), generator=g)], 1)
g1 = [t.to(dev) for t in p1[:4]]; g2 = [t.to(dev) for t in p2[:4]]
res = {}
for mode in ("f32", "bf16x2"):
    mid.set_mode(mode)
    out = ops.regress(mid, None, g1, g2, props.to(dev), want_raw=True)
    res[mode] = out["raw1"].cpu()
os.makedirs("gpurun_out", exist_ok=True)
torch.save({"props": props, **res}, "gpurun_out/raw_split.pt")
print((res["f32"] - res["bf16x2"]).abs().max())
