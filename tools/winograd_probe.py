"""Would a Winograd F(2x2, 3x3) conv2 stay inside the parity bars?  (CPU only; NOTES.md round-5 step 1a)

conv2 of FeatRegressNet (networks/modules.py:80-84: 512 -> 512, 3x3, stride 1, pad 1 on the 8x8 map) as 16 GEMMs over the
transformed tiles, in fp32 with fp32 accumulation (what the fp16x2 matrix path delivers), against the direct fp32 convolution
and the fp64 evaluation, through the rest of the regressor to the coordinates.  Sweeps the same axes as tools/margin_sweep.py.
usage: python tools/winograd_probe.py [n_proposals]"""
import os
import sys
import numpy as np
import torch
import torch.nn.functional as F
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import p2p_oracle as orc  # noqa: E402
from patch2pix_amd.utils import synthetic  # noqa: E402

G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


# F(4x4, 3x3) (Lavin & Gray): 6 x 6 windows at stride 4, 36 GEMMs, 2.25 products per output instead of F(2x2)'s 4
G4 = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
                   [0, 0, 1]], dtype=torch.float64)
BT4 = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                    [0, 4, 0, -5, 0, 1]], dtype=torch.float64)
AT4 = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float64)


def winograd_conv(u, w, dtype, f4=False):
    """u [N,C,8,8], w [O,C,3,3] -> [N,O,8,8]; input / output transforms and the GEMMs in `dtype`, the filter transform in
    fp64 (it is done once, at pack time) and then rounded to `dtype`.  f4: F(4x4,3x3) instead of F(2x2,3x3)."""
    g, bt, at = (G4, BT4, AT4) if f4 else (G, BT, AT)
    win, step, m = (6, 4, 4) if f4 else (4, 2, 2)
    n, c = u.shape[:2]
    up = F.pad(u.to(dtype), (1, 1, 1, 1))                                    # 10 x 10
    tiles = up.unfold(2, win, step).unfold(3, win, step)                     # [N,C,T,T,win,win]: tile (ty,tx)
    V = torch.einsum("ij,nctujk,lk->nctuil", bt.to(dtype), tiles, bt.to(dtype))      # B^T d B
    U = torch.einsum("ij,ocjk,lk->ocil", g, w.double(), g).to(dtype)         # G g G^T   [O,C,win,win]
    M = torch.einsum("nctuil,ocil->notuil", V, U)                            # win^2 GEMMs over c
    Y = torch.einsum("ij,notujk,lk->notuil", at.to(dtype), M, at.to(dtype))  # A^T m A   [N,O,T,T,m,m]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(n, -1, 8, 8)


def forward(f1, f2, p, dtype, conv2):
    q = {k: v.to(dtype) for k, v in p.items()}
    z = torch.cat([f1, f2], dim=1).to(dtype)
    u = orc._bn(F.conv2d(z, q["conv.0.weight"], None, stride=2, padding=1), q, "conv.1", (1, -1, 1, 1))
    u = conv2(u, q["conv.2.weight"])
    u = orc._bn(u, q, "conv.3", (1, -1, 1, 1))
    v = F.relu(u).amax(dim=(2, 3))
    v = F.relu(orc._bn(F.linear(v, q["fc.0.weight"], q["fc.0.bias"]), q, "fc.1", (1, -1)))
    v = F.relu(orc._bn(F.linear(v, q["fc.3.weight"], q["fc.3.bias"]), q, "fc.4", (1, -1)))
    return F.linear(v, q["fc.6.weight"], q["fc.6.bias"])


def main(n=96):
    torch.set_num_threads(8)
    print("setting: |coordinate error| in px against the fp64 evaluation, max over proposals: direct fp32 / F(2x2,3x3) fp32 / "
          "F(4x4,3x3) fp32 (+ conv2 output error relative to its rms)")
    for wscale, bnspread, fmag in ((1, 1, 1), (2, 1, 1), (1, 2, 1), (1, 1, 4), (2, 2, 4)):
        sd = synthetic.make_state_dict(0, backbone=False)
        for k in sd:
            if ".conv.0.weight" in k or ".conv.2.weight" in k:
                sd[k] = sd[k] * wscale
            if "running_var" in k:
                sd[k] = sd[k] ** bnspread
        _, mid_p, _ = orc.split_params(sd, torch.float64)
        p1 = [t * fmag for t in synthetic.make_pyramid(7, 240, 320)]
        p2 = [t * fmag for t in synthetic.make_pyramid(8, 240, 320)]
        g = torch.Generator().manual_seed(5)
        props = torch.stack([torch.randint(0, 320, (n,), generator=g), torch.randint(0, 240, (n,), generator=g),
                             torch.randint(0, 320, (n,), generator=g), torch.randint(0, 240, (n,), generator=g)], 1)
        f1 = orc.gather_patch_feats([t.double() for t in p1[:4]], props[:, 0], props[:, 1])
        f2 = orc.gather_patch_feats([t.double() for t in p2[:4]], props[:, 2], props[:, 3])
        f1, f2 = orc.l2_normalize(f1, 1) if False else f1, f2
        direct = lambda u, w: F.conv2d(u, w, None, stride=1, padding=1)
        ref = forward(f1, f2, mid_p, torch.float64, direct)
        coords = lambda out: orc.parse_regressor_out(out.double(), props, 320, 240, 320, 240)[0]
        e_dir = (coords(forward(f1, f2, mid_p, torch.float32, direct)) - coords(ref)).abs().max().item()
        e_win = (coords(forward(f1, f2, mid_p, torch.float32, lambda u, w: winograd_conv(u, w, torch.float32))) - coords(ref)).abs().max().item()
        e_w64 = (coords(forward(f1, f2, mid_p, torch.float64, lambda u, w: winograd_conv(u, w, torch.float64))) - coords(ref)).abs().max().item()
        e_f4 = (coords(forward(f1, f2, mid_p, torch.float32, lambda u, w: winograd_conv(u, w, torch.float32, True))) - coords(ref)).abs().max().item()
        # conv2 alone
        q = {k: v.double() for k, v in mid_p.items()}
        u = orc._bn(F.conv2d(torch.cat([f1, f2], 1), q["conv.0.weight"], None, stride=2, padding=1), q, "conv.1", (1, -1, 1, 1))
        c64 = direct(u, q["conv.2.weight"])
        rms = c64.pow(2).mean().sqrt()
        c_dir = ((direct(u.float(), q["conv.2.weight"].float()).double() - c64).abs().max() / rms).item()
        c_win = ((winograd_conv(u, q["conv.2.weight"], torch.float32).double() - c64).abs().max() / rms).item()
        c_f4 = ((winograd_conv(u, q["conv.2.weight"], torch.float32, True).double() - c64).abs().max() / rms).item()
        print(f"weights x{wscale} bn^{bnspread} feats x{fmag}: {e_dir:.2e} / {e_win:.2e} / {e_f4:.2e} px (F(2x2) in fp64: {e_w64:.1e});  "
              f"conv2 max err / rms: direct {c_dir:.2e}, F(2x2) {c_win:.2e}, F(4x4) {c_f4:.2e}", flush=True)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 96)
