"""Host-side proposal filtering -- role of reference networks/utils.py:38-72 (`filter_coarse`).

The reference does this step on the host too (np.unique on a [2*cells, 4] int64 array, a few
tens of KB), including its data-dependent fall-backs; moving it to the device is a "next" row of
SURVEY.md section 8(f).  Semantics kept: rows come back in lexicographic order, the score is the one
of the first occurrence, `mutual` keeps rows that occur more than once, empty selections fall back
to "keep everything", and `ptmax` shuffles with the *global* numpy RNG and tiles to exactly ptmax.

Differences are mechanical only: the whole batch crosses PCIe once in each direction, and the
row-wise unique runs on one packed 64-bit key per row (pixel coordinates are non-negative and
< 2^15, so the key order is the lexicographic row order) instead of numpy's structured-view sort.
"""
import numpy as np
import torch


def _unique_rows(rows):
    """(first_index, counts) of the lexicographically sorted distinct rows of an int64 [n,4] array."""
    if rows.size and rows.min() >= 0 and rows.max() < (1 << 15):
        key = ((rows[:, 0] << 48) | (rows[:, 1] << 32) | (rows[:, 2] << 16) | rows[:, 3])
        _, first, counts = np.unique(key, return_index=True, return_counts=True)
    else:
        _, first, counts = np.unique(rows, axis=0, return_index=True, return_counts=True)
    return first, counts


def filter_coarse(coarse_matches, match_scores, ncn_thres=0.0, mutual=True, ptmax=None, host_copy=None):
    """`host_copy=(rows_np, scores_np)` supplies an already transferred copy of the two arrays
    (see Patch2Pix.coarse_async), skipping the synchronous device-to-host copy."""
    if host_copy is not None:
        device = coarse_matches.device
        host_rows, host_scores = host_copy
    elif isinstance(coarse_matches, torch.Tensor):
        device = coarse_matches.device
        host_rows = coarse_matches.detach().cpu().numpy()
        host_scores = match_scores.detach().cpu().numpy()
    else:
        device = coarse_matches[0].device if len(coarse_matches) else torch.device("cpu")
        host_rows = [m.detach().cpu().numpy() for m in coarse_matches]
        host_scores = [s.detach().cpu().numpy() for s in match_scores]
    out_rows, out_scores = [], []
    for rows, scores in zip(host_rows, host_scores):
        scores = scores.reshape(-1)
        first, counts = _unique_rows(rows)
        sel = first[counts > 1] if mutual else first
        if len(sel) > 0:
            rows, scores = rows[sel], scores[sel]
        passed = np.nonzero(scores > ncn_thres)[0]
        if ptmax:
            if len(passed) == 0:
                passed = np.zeros(4, dtype=np.int64)
            order = np.arange(len(passed))
            np.random.shuffle(order)
            order = np.tile(order, ptmax // len(passed) + 1)[:ptmax]
            passed = passed[order]
        if len(passed) > 0:
            rows, scores = rows[passed], scores[passed]
        out_rows.append(rows)
        out_scores.append(scores)
    # one upload for the whole batch, then per-item views
    counts = [r.shape[0] for r in out_rows]
    all_rows = torch.from_numpy(np.ascontiguousarray(np.concatenate(out_rows))).to(device)
    all_scores = torch.from_numpy(np.ascontiguousarray(np.concatenate(out_scores))).to(device)
    return list(torch.split(all_rows, counts)), list(torch.split(all_scores, counts))
