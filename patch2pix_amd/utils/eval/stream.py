"""Streaming form of `estimate_matches` for long pair lists (the "next" rows of SURVEY.md section 8f: image loading
and the backbone as producers of the hot path).

The reference matches one pair per call (utils/eval/model_helper.py:64-109): load and resize both images with PIL,
run the backbone twice, match, copy back.  For a stream of pairs (MegaDepth / HPatches style evaluation) the same
results can be produced much faster by
  * decoding / resizing images in a thread pool while the GPU works (PIL releases the GIL in its decoders),
  * running the backbone once on the 2*B images of B same-sized pairs,
  * sharing one fine-stage launch between the B pairs and software-pipelining the host-side filter
    (Patch2Pix.coarse_async / fine_from_ticket).
`estimate_matches_stream` yields exactly the triples `estimate_matches` would return, in input order.
"""
from collections import deque
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from ..datasets.preprocess import load_im_pixels, normalise_pixels


def _load(job):
    idx, im1, im2, ksize, upsample, imsize = job
    t1, s1 = load_im_pixels(im1, ksize, upsample, imsize=imsize)          # uint8 [H,W,3]: normalised on the device
    t2, s2 = load_im_pixels(im2, ksize, upsample, imsize=imsize)
    return idx, t1, t2, np.array([tuple(s1) + tuple(s2)])


_pin_pool = {}      # (shape) -> [[pinned tensor, event-or-None], ...] ring of staging buffers for the image batches


def _upload(tensors, device):
    """Stack a list of equally shaped uint8 [H,W,3] images into a recycled PINNED batch buffer, start one asynchronous
    copy to the device and normalise there -> float32 [B,3,H,W].  (Stacking into pageable memory and copying from there cost 100-190 ms per batch of 8 pairs --
    more than the whole GPU work of the batch.)"""
    if torch.device(device).type != "cuda":            # host-logic tests drive the generator with a CPU stand-in of the net
        return normalise_pixels(torch.stack(tensors).to(device))
    shape = (len(tensors),) + tuple(tensors[0].shape)
    ring = _pin_pool.setdefault(shape, {"slots": [], "turn": 0})
    if len(ring["slots"]) < 4:
        ring["slots"].append([torch.empty(shape, dtype=tensors[0].dtype).pin_memory(), None])
    slot = ring["slots"][ring["turn"] % len(ring["slots"])]
    ring["turn"] += 1
    if slot[1] is not None:
        slot[1].synchronize()              # the copy that last read this buffer has finished
    torch.stack(tensors, out=slot[0])
    dev = slot[0].to(device, non_blocking=True)
    slot[1] = torch.cuda.Event()
    slot[1].record(torch.cuda.current_stream(device))
    return normalise_pixels(dev)


def _finish(net, ticket, metas, ncn_thres, mutual, io_thres):
    fine, conf, coarse = net.fine_from_ticket(ticket, ncn_thres=ncn_thres, mutual=mutual)
    # one device-to-host copy for the whole batch: [fine x1,y1,x2,y2 | confidence | coarse x1,y1,x2,y2]
    counts = [f.shape[0] for f in fine]
    packed = torch.cat([torch.cat(fine), torch.cat(conf)[:, None], torch.cat(coarse).float()], dim=1).cpu().numpy()
    out, start = [], 0
    for n, to_original in zip(counts, metas):
        rows = packed[start:start + n]
        start += n
        refined, confidence, proposals = rows[:, 0:4], rows[:, 4], rows[:, 5:9]
        keep = np.flatnonzero(confidence > io_thres)
        if keep.size:
            refined, confidence, proposals = refined[keep], confidence[keep], proposals[keep]
        out.append((to_original * refined, np.ascontiguousarray(confidence), to_original * proposals))
    return out


def estimate_matches_stream(net, pairs, ksize=2, ncn_thres=0.0, mutual=True, io_thres=0.25, imsize=None,
                            batch=8, workers=8):
    """Generator over `pairs` (iterable of (im1, im2) paths / file objects): yields
    (matches float64 [M,4], scores float32 [M], coarse_matches float64 [M,4]) per pair, in order."""
    jobs = [(i, a, b, ksize, net.upsample, imsize) for i, (a, b) in enumerate(pairs)]
    pending = deque()          # (ticket, metas) whose fine stage has not been issued yet
    with ThreadPoolExecutor(max_workers=max(1, workers)) as pool, torch.no_grad():
        loaded = pool.map(_load, jobs)
        group = []

        def flush():
            """Backbone on the 2*B images of the current group, coarse stage enqueued, ticket queued."""
            if not group:
                return
            im1 = _upload([g[1] for g in group], net.device)
            im2 = _upload([g[2] for g in group], net.device)
            if im1.shape == im2.shape:
                feats = net.extract.pyramid(torch.cat([im1, im2]))
                n = im1.shape[0]
                f1, f2 = [f[:n] for f in feats], [f[n:] for f in feats]
            else:
                f1, f2 = net.extract.pyramid(im1), net.extract.pyramid(im2)
            pending.append((net.coarse_async(f1, f2, ksize=ksize), [g[3] for g in group]))
            group.clear()

        for item in loaded:
            if group and (item[1].shape != group[0][1].shape or item[2].shape != group[0][2].shape or len(group) >= batch):
                flush()
                # keep one batch of coarse work enqueued ahead of the batch being filtered on the host
                while len(pending) > 1:
                    ticket, metas = pending.popleft()
                    yield from _finish(net, ticket, metas, ncn_thres, mutual, io_thres)
            group.append(item)
        flush()
        while pending:
            ticket, metas = pending.popleft()
            yield from _finish(net, ticket, metas, ncn_thres, mutual, io_thres)
