"""TEST INFRASTRUCTURE ONLY -- forward error model of an fp32 evaluation of the coarse stage.

Only tests/, __graft_entry__.smoke() and bench.py's parity leg may import oracle/.

Two correct fp32 evaluations of the coarse stage (the oracle, the unmodified reference, the HIP kernels) sum in
different orders and therefore order the top two candidates of an argmax differently when those are closer than the
rounding error of the evaluations.  Such a row is *undecidable in fp32*: the reference's own answer for it is an
artefact of its summation order.  Which rows those are is decided from an ERROR MODEL computed from the operands --
not from a hand-picked tolerance:

  * the whole coarse stage is evaluated in fp64 (`Z`), and next to it a first-order forward error bound `E` of an fp32
    evaluation is propagated through the same stages (modules.py:6,41-53,11-34; ncn/model.py:145-176; conv4d.py:12-74):
        dot product of n terms:  |d| <= eps_n * sum_i |a_i b_i|,   eps_n = lam * sqrt(n) * 2^-24 + 2 * 2^-24
    (the probabilistic rounding model: errors of n roundings add like a random walk, lam = 1; plus the truncation of the
    bf16x3 products; measured fp32 errors stay below 10 % of the resulting bound, see `check`),
    errors of the inputs are carried through |W|-convolutions, the ratios of MutualMatching through their
    derivatives, max() through the maximum of the operand bounds;
  * `ErrorModel.check(volume32)` asserts that an fp32 volume (the oracle's, the kernel's) lies within `CHECK_LIMIT` x `E`
    of `Z` -- run on every use, so the model is validated against real data, not assumed.  Measured |error| / bound: 0.04-0.09
    (oracle), <= 0.12 (HIP kernels); the limit is 0.25, so a kernel whose error doubles fails here;
  * a row is DECIDABLE when the fp64 winner beats every competitor by more than the sum of their bounds: any fp32
    evaluation within the bounds must then return that winner.  `assert_decidable_rows` requires exactly that of a
    match list; `differing_rows_are_near_ties` accepts a difference between two lists only where the two candidates are
    closer in fp64 than `NEAR_TIE_TOL` (0.25) x the sum of their bounds: with both evaluations inside CHECK_LIMIT x E no
    larger gap can flip an argmax (the worst accepted gap on the fixtures is 0.06).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import p2p_oracle as orc

U = 2.0 ** -24
CHECK_LIMIT = 0.25        # an fp32 evaluation must stay within this fraction of the forward bound (measured: <= 0.12)
NEAR_TIE_TOL = 0.25       # a differing argmax is accepted when the fp64 gap is below this fraction of the two bounds


def _eps(n, lam):
    return lam * math.sqrt(n) * U + 2 * U


def _mm_err(p, ep):
    """MutualMatching (ncn/model.py:157-176) value and error bound: x = p^3 / ((r + eps)(c + eps))."""
    r = p.amax(dim=(2, 3), keepdim=True) + 1e-5
    c = p.amax(dim=(0, 1), keepdim=True) + 1e-5
    er = ep.amax(dim=(2, 3), keepdim=True)
    ec = ep.amax(dim=(0, 1), keepdim=True)
    x = p * ((p / r) * (p / c))
    ex = 3 * p * p / (r * c) * ep + x.abs() * (er / r + ec / c + 6 * U)
    return x, ex


def _net_err(x, ex, ncn, lam):
    """conv4d -> ReLU -> conv4d -> ReLU (ncn/model.py:132-141) with the bound carried through |W|."""
    w1, b1, w2, b2 = ncn["w1"], ncn["b1"], ncn["w2"], ncn["b2"]
    z16, z1 = torch.zeros_like(b1), torch.zeros_like(b2)
    h = F.relu(orc.conv4d(x[None], w1, b1))
    # |W| * e + eps (|W| * |x| + |b|) = |W| * (e + eps |x|) + eps |b|: one convolution per layer for the bound
    e1 = orc.conv4d((ex + _eps(82, lam) * x.abs())[None], w1.abs(), z16) + _eps(82, lam) * b1.abs().view(-1, 1, 1, 1, 1)
    y = F.relu(orc.conv4d(h, w2, b2))
    eps2 = _eps(16 * 81 + 1, lam)
    e2 = orc.conv4d(e1 + eps2 * h, w2.abs(), z1) + eps2 * b2.abs().view(-1, 1, 1, 1, 1)
    return y[0], e2[0]


class ErrorModel:
    """fp64 evaluation of forward_coarse_match (patch2pix.py:120-136) + forward error bound of an fp32 evaluation."""

    def __init__(self, feat_a, feat_b, state_dict, ksize=2, lam=1.0, keep_full=True):
        ncn, _, _ = orc.split_params(state_dict, torch.float64)
        fa, fb = feat_a.double(), feat_b.double()
        na, nb = orc.l2_normalize(fa, 0), orc.l2_normalize(fb, 0)
        c = orc.correlation(na, nb)
        # normalisation: a 256-term sum of squares under a root (half its relative error) and a division, per operand
        eps_c = _eps(fa.shape[0], lam) + 2 * (0.5 * _eps(fa.shape[0], lam) + 3 * U)
        ec = eps_c * orc.correlation(na.abs(), nb.abs())
        self.ksize = ksize
        self.C, self.EC = (c, ec) if keep_full else (None, None)
        if ksize > 1:
            k = ksize
            ha, wa, hb, wb = c.shape
            win = lambda t: t.reshape(ha // k, k, wa // k, k, hb // k, k, wb // k, k).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(
                ha // k, wa // k, hb // k, wb // k, k ** 4)
            cw, ew = win(c), win(ec)
            p, ep = cw.amax(-1), ew.amax(-1)
            top2 = torch.topk(cw, 2, dim=-1).values
            self.reloc_decidable = (top2[..., 0] - top2[..., 1]) > 2 * ep          # the pooling argmax cannot flip
            self.reloc_code = cw.argmax(-1)
        else:
            p, ep = c, ec
            self.reloc_decidable, self.reloc_code = None, None
        x, ex = _mm_err(p, ep)
        y1, e1 = _net_err(x, ex, ncn, lam)
        y2, e2 = _net_err(x.permute(2, 3, 0, 1).contiguous(), ex.permute(2, 3, 0, 1).contiguous(), ncn, lam)
        y = y1 + y2.permute(2, 3, 0, 1)
        ey = e1 + e2.permute(2, 3, 0, 1) + U * y.abs()
        self.Z, self.E = _mm_err(y, ey)

    def check(self, volume32, what="fp32 volume", limit=CHECK_LIMIT):
        """An fp32 evaluation must lie well inside the bound (`limit` x E); returns the worst |error| / bound ratio."""
        err = (volume32.double().reshape(self.Z.shape) - self.Z).abs()
        ratio = float((err / self.E.clamp_min(1e-300)).max())
        assert ratio <= limit, (f"{what}: |error| reaches {ratio:.2f} of the fp32 error model (limit {limit}) -- the volume is less "
                                f"accurate than an fp32 evaluation should be (or the model is wrong)")
        return ratio

    # ---- decidability of the two argmaxes per cell (extract_ncmatches.py:27-54)
    def _rows(self):
        ha, wa, hb, wb = self.Z.shape
        m, e = self.Z.reshape(ha * wa, hb * wb), self.E.reshape(ha * wa, hb * wb)
        out = []
        for dim in (0, 1):                                # B->A rows (argmax over A cells), then A->B rows
            lo, hi = m - e, m + e
            best = m.argmax(dim=dim)
            hi2 = hi.clone()
            if dim == 0:
                cols = torch.arange(hb * wb)
                win_lo = lo[best, cols]
                hi2[best, cols] = -math.inf
            else:
                rows = torch.arange(ha * wa)
                win_lo = lo[rows, best]
                hi2[rows, best] = -math.inf
            out.append((best, win_lo > hi2.amax(dim=dim)))
        return out

    def decidable(self):
        """(winner cell index, decidable?) for the nB B->A rows followed by the nA A->B rows."""
        (ba, da), (bb, db) = self._rows()
        return torch.cat((ba, bb)), torch.cat((da, db))


def _cells(rows, ksize, upsample=8):
    """pixel rows (xA,yA,xB,yB) = upsample*(ksize*cell + delta) + upsample//2  ->  pooled cells (a,b,c,d)."""
    idx = (rows - upsample // 2) // upsample // ksize
    return idx[:, 1], idx[:, 0], idx[:, 3], idx[:, 2]


def assert_decidable_rows(rows_got, model, upsample=8):
    """Every row whose argmax is decidable in fp32 (see the module header) must hold the fp64 winner; the relocalisation
    inside the winning cell must be the fp64 one where that is decidable too.  Returns (decidable rows, all rows)."""
    rows_got = torch.as_tensor(rows_got)
    ha, wa, hb, wb = model.Z.shape
    nB = hb * wb
    winner, ok = model.decidable()
    a, b, c, d = _cells(rows_got, model.ksize, upsample)
    chosen = torch.where(torch.arange(rows_got.shape[0]) < nB, a * wa + b, c * wb + d)
    wrong = ok & (chosen != winner)
    assert not bool(wrong.any()), (f"{int(wrong.sum())} rows are decidable in fp32 (fp64 margin above the error bound of both "
                                   f"candidates) but do not hold the fp64 winner, e.g. row {int(torch.nonzero(wrong)[0])}")
    if model.reloc_decidable is not None:
        k = model.ksize
        pos = (rows_got - upsample // 2) // upsample                       # (jA, iA, jB, iB) at full resolution
        code = (((pos[:, 1] % k) * k + pos[:, 0] % k) * k + pos[:, 3] % k) * k + pos[:, 2] % k
        dec = ok & model.reloc_decidable[a, b, c, d]
        bad = dec & (code != model.reloc_code[a, b, c, d])
        assert not bool(bad.any()), f"{int(bad.sum())} rows hold the right cell but a wrong (decidable) relocalisation"
        ok = dec
    return int(ok.sum()), int(rows_got.shape[0])


def differing_rows_are_near_ties(rows_got, rows_ref, model, upsample=8, tol=NEAR_TIE_TOL):
    """rows_*: [nB+nA,4] int64 (B->A rows first, then A->B, networks/patch2pix.py:351-355).  Returns (number of differing
    rows, worst fp64 gap / bound ratio); raises AssertionError on a difference the error model does not allow.
    A row can differ in two ways: another pooled cell won the softmax argmax (both candidates' final values must be
    closer in fp64 than `tol` x the sum of their bounds), or the same cell won but its relocalisation (the 4-D max-pool
    argmax, modules.py:11-34) picked another of the k^4 positions (the two full-resolution correlations must be)."""
    rows_got, rows_ref = torch.as_tensor(rows_got), torch.as_tensor(rows_ref)
    bad = torch.nonzero((rows_got != rows_ref).any(dim=1)).flatten()
    if bad.numel() == 0:
        return 0, 0.0
    k = model.ksize
    ag, bg, cg, dg = _cells(rows_got[bad], k, upsample)
    ar, br, cr, dr = _cells(rows_ref[bad], k, upsample)
    same_cell = (ag == ar) & (bg == br) & (cg == cr) & (dg == dr)
    worst = 0.0
    if bool(same_cell.any()):
        assert model.C is not None, "a relocalisation argmax differs; build the ErrorModel with keep_full=True"
        for r in bad[same_cell].tolist():
            pg, pr = (rows_got[r] - upsample // 2) // upsample, (rows_ref[r] - upsample // 2) // upsample   # (jA,iA,jB,iB)
            ig, ir = (pg[1], pg[0], pg[3], pg[2]), (pr[1], pr[0], pr[3], pr[2])
            gap, bound = float((model.C[ig] - model.C[ir]).abs()), float(model.EC[ig] + model.EC[ir])
            assert gap <= tol * bound, (f"row {r}: relocalisation differs and the two positions are {gap:.2e} apart in fp64 "
                                        f"({tol} x bound = {tol * bound:.2e})")
            worst = max(worst, gap / bound)
    if bool((~same_cell).any()):
        keep = ~same_cell
        rows = bad[keep]
        ag, bg, cg, dg, ar, br, cr, dr = ag[keep], bg[keep], cg[keep], dg[keep], ar[keep], br[keep], cr[keep], dr[keep]
        nB = model.Z.shape[2] * model.Z.shape[3]
        first = rows < nB                                  # B->A rows: B cell fixed, A cell chosen; else the converse
        assert bool(((cg == cr) & (dg == dr))[first].all()) and bool(((ag == ar) & (bg == br))[~first].all()), \
            "a differing row does not even belong to the same query cell"
        gap = (model.Z[ag, bg, cg, dg] - model.Z[ar, br, cr, dr]).abs()
        bound = model.E[ag, bg, cg, dg] + model.E[ar, br, cr, dr]
        ratio = gap / bound.clamp_min(1e-300)
        assert bool((ratio <= tol).all()), (f"{int((ratio > tol).sum())} differing rows are not near-ties: fp64 gap up to "
                                            f"{float(ratio.max()):.2f}x the fp32 error bound of the two candidates (accepted: {tol})")
        worst = max(worst, float(ratio.max()))
    return int(bad.numel()), worst


class LocalErrorModel:
    """The same fp64 evaluation + fp32 error bound, evaluated only where a differing row needs it.  At 960x1280 the pooled
    volume has 23 M cells and the full model costs minutes of fp64 4-D convolutions; a row's two candidates need the
    consensus output only on one A row / B column of the volume each (MutualMatching, ncn/model.py:157-176, divides by
    the maxima of exactly those), and a cell of the consensus output depends on a 5^4 neighbourhood of its input
    (conv4d.py:12-74 twice).  The pooled correlation and the first MutualMatching are evaluated in full (chunked GEMMs and
    elementwise work), the two consensus layers on crops [a-2:a+3, b-2:b+3, :, :] / [:, :, c-2:c+3, d-2:d+3]."""

    def __init__(self, feat_a, feat_b, state_dict, ksize=2, lam=1.0, chunk=8):
        assert ksize > 1, "the local model is for the pooled (ksize 2) volumes of large images"
        self.ncn, _, _ = orc.split_params(state_dict, torch.float64)
        self.lam, self.ksize = lam, ksize
        fa, fb = feat_a.double(), feat_b.double()
        self.na, self.nb = orc.l2_normalize(fa, 0), orc.l2_normalize(fb, 0)
        self.eps_c = _eps(fa.shape[0], lam) + 2 * (0.5 * _eps(fa.shape[0], lam) + 3 * U)
        k = ksize
        c_, ha, wa = fa.shape
        _, hb, wb = fb.shape
        nb_flat, nb_abs = self.nb.reshape(c_, -1), self.nb.abs().reshape(c_, -1)
        p = torch.empty(ha // k, wa // k, hb // k, wb // k, dtype=torch.float64)
        ep = torch.empty_like(p)
        for r0 in range(0, ha // k, chunk):           # pooled rows r0 .. r1 of A: k * (r1 - r0) feature rows
            r1 = min(r0 + chunk, ha // k)
            xa = self.na[:, k * r0:k * r1].reshape(c_, -1)
            for src, dst, xb in ((xa, p, nb_flat), (xa.abs(), ep, nb_abs)):
                cc = (src.t() @ xb).reshape(r1 - r0, k, wa // k, k, hb // k, k, wb // k, k)
                dst[r0:r1] = cc.permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(r1 - r0, wa // k, hb // k, wb // k, k ** 4).amax(-1)
        ep *= self.eps_c
        self.x, self.ex = _mm_err(p, ep)
        self.shape = tuple(p.shape)
        self._rows, self._cols = {}, {}

    def corr_at(self, ia, ib):
        """fp64 correlation and fp32 bound of one full-resolution position pair (relocalisation near-ties)."""
        va, vb = self.na[:, ia[0], ia[1]], self.nb[:, ib[0], ib[1]]
        return float((va * vb).sum()), self.eps_c * float((va * vb).abs().sum())

    def _consensus_crop(self, sl):
        """NeighConsensus.forward (both branches) and its bound on a crop of the MutualMatching output."""
        x, ex = self.x[sl].contiguous(), self.ex[sl].contiguous()
        y1, e1 = _net_err(x, ex, self.ncn, self.lam)
        y2, e2 = _net_err(x.permute(2, 3, 0, 1).contiguous(), ex.permute(2, 3, 0, 1).contiguous(), self.ncn, self.lam)
        y = y1 + y2.permute(2, 3, 0, 1)
        return y, e1 + e2.permute(2, 3, 0, 1) + U * y.abs()

    def _row(self, a, b):                       # y[a, b, :, :] and its bound
        if (a, b) not in self._rows:
            a0, b0 = max(a - 2, 0), max(b - 2, 0)
            y, e = self._consensus_crop((slice(a0, a + 3), slice(b0, b + 3)))
            self._rows[(a, b)] = (y[a - a0, b - b0], e[a - a0, b - b0])
        return self._rows[(a, b)]

    def _col(self, c, d):                       # y[:, :, c, d] and its bound
        if (c, d) not in self._cols:
            c0, d0 = max(c - 2, 0), max(d - 2, 0)
            y, e = self._consensus_crop((slice(None), slice(None), slice(c0, c + 3), slice(d0, d + 3)))
            self._cols[(c, d)] = (y[:, :, c - c0, d - d0], e[:, :, c - c0, d - d0])
        return self._cols[(c, d)]

    def cell(self, a, b, c, d):
        """(Z, E) of the final volume at one cell: the second MutualMatching on the consensus output."""
        yr, er = self._row(a, b)
        yc, ec = self._col(c, d)
        y, ey = yr[c, d], er[c, d]
        r, cm = yr.max() + 1e-5, yc.max() + 1e-5
        z = y * ((y / r) * (y / cm))
        e = 3 * y * y / (r * cm) * ey + z.abs() * (er.max() / r + ec.max() / cm + 6 * U)
        return float(z), float(e)


def differing_rows_are_near_ties_local(rows_got, rows_ref, feat_a, feat_b, state_dict, ksize=2, upsample=8, tol=NEAR_TIE_TOL,
                                       volume_got=None, check_limit=CHECK_LIMIT):
    """`differing_rows_are_near_ties` through the LocalErrorModel (built only when a row differs): the two candidates of a
    differing row must be closer in fp64 than `tol` x their fp32 bounds; with `volume_got` (the kernel's final volume) the
    kernel's values at the visited cells must also lie within `check_limit` x the bound of the fp64 values.
    Returns (number of differing rows, worst gap / bound ratio)."""
    rows_got, rows_ref = torch.as_tensor(rows_got), torch.as_tensor(rows_ref)
    bad = torch.nonzero((rows_got != rows_ref).any(dim=1)).flatten()
    if bad.numel() == 0:
        return 0, 0.0
    model = LocalErrorModel(feat_a, feat_b, state_dict, ksize)
    k, worst = ksize, 0.0
    nB = model.shape[2] * model.shape[3]
    for r in bad.tolist():
        g, f = _cells(rows_got[r:r + 1], k, upsample), _cells(rows_ref[r:r + 1], k, upsample)
        cg, cr = tuple(int(v[0]) for v in g), tuple(int(v[0]) for v in f)
        if cg == cr:                           # same pooled cell, another relocalisation: the two full-resolution correlations
            pg, pr = (rows_got[r] - upsample // 2) // upsample, (rows_ref[r] - upsample // 2) // upsample   # (jA, iA, jB, iB)
            (vg, eg), (vr, er) = (model.corr_at((int(q[1]), int(q[0])), (int(q[3]), int(q[2]))) for q in (pg, pr))
            gap, bound = abs(vg - vr), eg + er
        else:
            same_query = (cg[2:] == cr[2:]) if r < nB else (cg[:2] == cr[:2])
            assert same_query, f"row {r}: the differing row does not even belong to the same query cell"
            (zg, eg), (zr, er) = model.cell(*cg), model.cell(*cr)
            gap, bound = abs(zg - zr), eg + er
            if volume_got is not None:
                for cell, z, e in ((cg, zg, eg), (cr, zr, er)):
                    err = abs(float(volume_got[cell]) - z)
                    assert err <= check_limit * e, (f"row {r}: the kernel's volume at {cell} is {err:.2e} from the fp64 value "
                                                    f"({err / e:.2f} of the fp32 bound, limit {check_limit})")
        assert gap <= tol * bound, (f"row {r} differs and is no near-tie: fp64 gap {gap:.3e} = {gap / max(bound, 1e-300):.2f} of the "
                                    f"fp32 error bound of the two candidates (accepted: {tol})")
        worst = max(worst, gap / max(bound, 1e-300))
    return int(bad.numel()), worst
