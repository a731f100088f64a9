"""Host-side proposal filtering -- role of reference networks/utils.py:38-72 (`filter_coarse`).

The reference does this step on the host too (np.unique on a [2*cells, 4] int64 array, a few
tens of KB), including its data-dependent fall-backs; moving it to the device is a "next" row of
SURVEY.md section 8(f).  Semantics kept: rows come back in lexicographic order, the score is the one
of the first occurrence, `mutual` keeps rows that occur more than once, empty selections fall back
to "keep everything", and `ptmax` shuffles with the *global* numpy RNG and tiles to exactly ptmax.
"""
import numpy as np
import torch


def filter_coarse(coarse_matches, match_scores, ncn_thres=0.0, mutual=True, ptmax=None):
    kept_matches, kept_scores = [], []
    for rows, scores in zip(coarse_matches, match_scores):
        host = rows.detach().cpu().numpy()
        _, first, counts = np.unique(host, axis=0, return_index=True, return_counts=True)
        sel = first[counts > 1] if mutual else first
        if len(sel) > 0:
            sel_t = torch.from_numpy(np.ascontiguousarray(sel)).to(rows.device)
            rows, scores = rows[sel_t], scores[sel_t]
        passed = torch.nonzero(scores.flatten() > ncn_thres, as_tuple=False).flatten()
        if ptmax:
            if len(passed) == 0:
                passed = torch.zeros(4, dtype=torch.long, device=rows.device)
            order = np.arange(len(passed))
            np.random.shuffle(order)
            order = np.tile(order, ptmax // len(passed) + 1)[:ptmax]
            passed = passed[torch.from_numpy(order).to(passed.device)]
        if len(passed) > 0:
            rows, scores = rows[passed], scores[passed]
        kept_matches.append(rows)
        kept_scores.append(scores)
    return kept_matches, kept_scores
