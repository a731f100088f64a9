"""Near-tie adjudication for argmax-derived match rows (test infrastructure).

Two correct fp32 evaluations of the coarse stage (e.g. the oracle and the unmodified reference: batched conv3d vs a loop
of conv3d slices) agree on the volume to ~3e-6 relative, which is enough to flip an argmax whose top two candidates are
closer than that.  On the synthetic "correlated" inputs such ties do not occur; on real photographs with the random-init
checkpoint the consensus output is nearly flat (scores ~ 1/cells) and ties are common.  A differing row is therefore
accepted only if BOTH candidates are within `tol` (relative) of each other in an fp64 evaluation of the volume."""
import numpy as np
import torch


def _cells(rows, ksize, upsample=8):
    """pixel rows (xA,yA,xB,yB) = upsample*(ksize*cell + delta) + upsample//2  ->  pooled cells (a,b,c,d)."""
    idx = (rows - upsample // 2) // upsample // ksize
    return idx[:, 1], idx[:, 0], idx[:, 3], idx[:, 2]


def differing_rows_are_near_ties(rows_got, rows_ref, corr64, ksize=2, tol=3e-5):
    """rows_*: [nB+nA,4] int64 (B->A rows first, then A->B, networks/patch2pix.py:351-355); corr64: fp64 final volume
    [hA',wA',hB',wB'].  Returns (number of differing rows, worst relative gap); raises AssertionError on a real difference."""
    rows_got, rows_ref = torch.as_tensor(rows_got), torch.as_tensor(rows_ref)
    bad = torch.nonzero((rows_got != rows_ref).any(dim=1)).flatten()
    if bad.numel() == 0:
        return 0, 0.0
    ag, bg, cg, dg = _cells(rows_got[bad], ksize)
    ar, br, cr, dr = _cells(rows_ref[bad], ksize)
    nB = corr64.shape[2] * corr64.shape[3]
    first = bad < nB                                   # B->A rows: B cell fixed, A cell chosen; else the converse
    assert bool(((cg == cr) & (dg == dr))[first].all()) and bool(((ag == ar) & (bg == br))[~first].all()), \
        "a differing row does not even belong to the same query cell"
    vg, vr = corr64[ag, bg, cg, dg], corr64[ar, br, cr, dr]
    gap = ((vg - vr).abs() / torch.maximum(vg.abs(), vr.abs()).clamp_min(1e-300))
    worst = float(gap.max())
    assert worst <= tol, f"{int((gap > tol).sum())} differing rows are not near-ties (worst relative gap {worst:.2e} > {tol})"
    return int(bad.numel()), worst
