"""Accuracy of the three regressor arithmetic modes against an fp64 evaluation of the oracle, with the kernels
executed by the CPU stand-in of the test-suite (tests/hipemu).  Development tool; the GPU form is tools/split_check.py.
  python tools/emu_precision.py [n_proposals]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))
import torch  # noqa: E402
import emu_lib  # noqa: E402
from oracle import p2p_oracle as orc  # noqa: E402
from patch2pix_amd.utils import synthetic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
emu = emu_lib.load()
sd = synthetic.make_state_dict(0, backbone=False)
sub = lambda p: {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}
_, mid64, fine64 = orc.split_params(sd, torch.float64)
_, mid32, fine32 = orc.split_params(sd)
H, W = 96, 128
p1, p2 = synthetic.make_pyramid(7, H, W), synthetic.make_pyramid(8, H, W)
g = torch.Generator().manual_seed(9)
props = torch.stack([torch.randint(0, W + 1, (n,), generator=g), torch.randint(0, H + 1, (n,), generator=g),
                     torch.randint(0, W + 1, (n,), generator=g), torch.randint(0, H + 1, (n,), generator=g)], 1)
d1, d2 = [t.double() for t in p1[:4]], [t.double() for t in p2[:4]]
ref_m, ref_p, ref_raw = orc.fine_level(d1, d2, props, mid64)
o_m, o_p, o_raw = orc.fine_level(p1[:4], p2[:4], props, mid32)
print(f"torch-CPU fp32 oracle: raw {(o_raw.double() - ref_raw).abs().max():.2e}  px {(o_m.double() - ref_m).abs().max():.2e}")
for mode in ("f32", "fp16x2", "fp16x2w"):
    mid = emu_lib.regressor_create(emu, sub("regress_mid."), mode)
    fine = emu_lib.regressor_create(emu, sub("regress_fine."), mode)
    out = emu_lib.regress(emu, mid, fine, p1[:4], p2[:4], props)
    e_raw = (out["raw1"].double() - ref_raw).abs().max().item()
    e_m = (out["matches1"].double() - ref_m).abs().max().item()
    e_p = (out["probs1"].double() - ref_p).abs().max().item()
    rf, rp, rr = orc.fine_level(d1, d2, out["matches1"].double(), fine64)
    e_f = (out["matches2"].double() - rf).abs().max().item()
    e_r2 = (out["raw2"].double() - rr).abs().max().item()
    print(f"{mode:7s}: raw {e_raw:.2e}  mid px {e_m:.2e}  mid score {e_p:.2e}  fine raw {e_r2:.2e}  fine px {e_f:.2e}", flush=True)
