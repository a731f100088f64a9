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


def differing_rows_are_near_ties(rows_got, rows_ref, corr64, ksize=2, tol=3e-5, feats=None, upsample=8):
    """rows_*: [nB+nA,4] int64 (B->A rows first, then A->B, networks/patch2pix.py:351-355); corr64: fp64 final volume
    [hA',wA',hB',wB'].  Returns (number of differing rows, worst relative gap); raises AssertionError on a real difference.
    A row can differ in two ways: another pooled cell won the softmax argmax (checked in corr64), or the same cell won
    but its relocalisation (the 4-D max-pool argmax, modules.py:11-34) picked another of the k^4 positions -- checked in
    an fp64 evaluation of the full-resolution correlation of the two positions, which needs `feats` = (featA, featB)."""
    rows_got, rows_ref = torch.as_tensor(rows_got), torch.as_tensor(rows_ref)
    bad = torch.nonzero((rows_got != rows_ref).any(dim=1)).flatten()
    if bad.numel() == 0:
        return 0, 0.0
    ag, bg, cg, dg = _cells(rows_got[bad], ksize, upsample)
    ar, br, cr, dr = _cells(rows_ref[bad], ksize, upsample)
    same_cell = (ag == ar) & (bg == br) & (cg == cr) & (dg == dr)
    worst_reloc = 0.0
    if bool(same_cell.any()):
        assert feats is not None, "a relocalisation argmax differs; pass feats=(featA, featB) to adjudicate it"
        na = feats[0].double() / (feats[0].double().pow(2).sum(0, keepdim=True) + 1e-6).sqrt()      # modules.py:6
        nb_ = feats[1].double() / (feats[1].double().pow(2).sum(0, keepdim=True) + 1e-6).sqrt()
        for r in bad[same_cell].tolist():
            pg, pr = (rows_got[r] - upsample // 2) // upsample, (rows_ref[r] - upsample // 2) // upsample   # (jA,iA,jB,iB)
            vg = float((na[:, pg[1], pg[0]] * nb_[:, pg[3], pg[2]]).sum())
            vr = float((na[:, pr[1], pr[0]] * nb_[:, pr[3], pr[2]]).sum())
            worst_reloc = max(worst_reloc, abs(vg - vr))
        # two correct fp32 evaluations of a 256-term dot product of unit vectors differ by ~sqrt(256) * 2^-24 = 1e-6
        # (any summation order; bound 256 * 2^-24 = 1.5e-5): a gap below 2e-6 in fp64 cannot be decided in fp32
        assert worst_reloc < 2e-6, f"relocalisation differs and the two positions are {worst_reloc:.2e} apart in fp64"
        keep = ~same_cell
        bad, ag, bg, cg, dg, ar, br, cr, dr = bad[keep], ag[keep], bg[keep], cg[keep], dg[keep], ar[keep], br[keep], cr[keep], dr[keep]
        if bad.numel() == 0:
            return int(same_cell.sum()), worst_reloc
    nB = corr64.shape[2] * corr64.shape[3]
    first = bad < nB                                   # B->A rows: B cell fixed, A cell chosen; else the converse
    assert bool(((cg == cr) & (dg == dr))[first].all()) and bool(((ag == ar) & (bg == br))[~first].all()), \
        "a differing row does not even belong to the same query cell"
    vg, vr = corr64[ag, bg, cg, dg], corr64[ar, br, cr, dr]
    gap = ((vg - vr).abs() / torch.maximum(vg.abs(), vr.abs()).clamp_min(1e-300))
    worst = float(gap.max())
    assert worst <= tol, f"{int((gap > tol).sum())} differing rows are not near-ties (worst relative gap {worst:.2e} > {tol})"
    return int(bad.numel()) + int(same_cell.sum()), max(worst, worst_reloc)
