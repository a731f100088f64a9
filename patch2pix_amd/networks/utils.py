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
    rows_np, scores_np = np.concatenate(out_rows), np.concatenate(out_scores)
    if device.type == "cuda":
        # through recycled pinned buffers: a copy from pageable memory would block the host until the stream
        # has drained, i.e. idle the GPU while the next launch is being prepared
        pin_r, pin_s, done = _pinned(rows_np.shape[0], device)
        pin_r[:rows_np.shape[0]].copy_(torch.from_numpy(rows_np))
        pin_s[:rows_np.shape[0]].copy_(torch.from_numpy(scores_np))
        all_rows = pin_r[:rows_np.shape[0]].to(device, non_blocking=True)
        all_scores = pin_s[:rows_np.shape[0]].to(device, non_blocking=True)
        done.record(torch.cuda.current_stream(device))      # the slot is reusable once this upload has run
    else:
        all_rows, all_scores = torch.from_numpy(rows_np), torch.from_numpy(scores_np)
    return list(torch.split(all_rows, counts)), list(torch.split(all_scores, counts))


_pin_rings = {}      # device index -> [ring of 8 slots, turn]


def _pinned(n, device):
    """(rows int64 [cap,4], scores f32 [cap], event) -- pinned staging buffers from a ring of 8 PER DEVICE (an event
    belongs to the device of the stream it was recorded on; one process may drive several GPUs).  The event is recorded
    by the caller after its asynchronous upload; a slot is handed out again only after that upload has completed, so a
    host that runs more than 8 uploads ahead of the stream waits here instead of overwriting a buffer still being read."""
    ring = _pin_rings.setdefault(device.index if device.index is not None else torch.cuda.current_device(), [[None] * 8, 0])
    slot = ring[1] % 8
    ring[1] += 1
    buf = ring[0][slot]
    if buf is not None:
        buf[2].synchronize()
    if buf is None or buf[0].shape[0] < n:
        cap = max(1024, 1 << (max(n, 1) - 1).bit_length())
        with torch.cuda.device(device):
            buf = (torch.empty((cap, 4), dtype=torch.int64).pin_memory(), torch.empty((cap,), dtype=torch.float32).pin_memory(),
                   torch.cuda.Event(blocking=True))
        ring[0][slot] = buf
    return buf
