"""Streaming form of `estimate_matches` for long pair lists (the "next" rows of SURVEY.md section 8f: image loading
and the backbone as producers of the hot path).

The reference matches one pair per call (utils/eval/model_helper.py:64-109): load and resize both images with PIL,
run the backbone twice, match, copy back.  For a stream of pairs (MegaDepth / HPatches style evaluation) the same
results can be produced much faster by
  * decoding / resizing images in a thread pool while the GPU works (PIL releases the GIL in its decoders),
  * running the backbone once on the 2*B images of B same-sized pairs,
  * keeping the whole batch on the device between the backbone and the result (Patch2Pix.predict_fine_device: coarse
    stage, filter_coarse, both regressors; ops.match_tail_batch: io_thres and scaling), one asynchronous copy of the
    match arrays per batch, three batches in flight -- the main thread only issues work;
  * (device_filter=False, or a network the device path does not cover) sharing one fine-stage launch between the B
    pairs and software-pipelining the host-side filter (Patch2Pix.coarse_async / fine_from_ticket).
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
    return idx, t1, t2, np.array([tuple(s1) + tuple(s2)]), job


_pin_pool = {}      # (device, shape) -> ring of [pinned tensor, event-or-None] staging buffers for the image batches


def _upload(tensors, device):
    """Stack a list of equally shaped uint8 [H,W,3] images into a recycled PINNED batch buffer, start one asynchronous
    copy to the device and normalise there -> float32 [B,3,H,W].  (Stacking into pageable memory and copying from there cost 100-190 ms per batch of 8 pairs --
    more than the whole GPU work of the batch.)"""
    if torch.device(device).type != "cuda":            # host-logic tests drive the generator with a CPU stand-in of the net
        return normalise_pixels(torch.stack(tensors).to(device))
    shape = (len(tensors),) + tuple(tensors[0].shape)
    ring = _pin_pool.setdefault((str(torch.device(device)), shape), {"slots": [], "turn": 0})     # an event belongs to its device
    if len(ring["slots"]) < 4:
        ring["slots"].append([torch.empty(shape, dtype=tensors[0].dtype).pin_memory(), None])
    slot = ring["slots"][ring["turn"] % len(ring["slots"])]
    ring["turn"] += 1
    if slot[1] is not None:
        slot[1].synchronize()              # the copy that last read this buffer has finished
    # plain memcpys: torch.stack would fan this copy out over an OpenMP team as wide as the machine, whose spin-waiting
    # threads burn a container's CPU quota within milliseconds (measured: 80 ms stalls of every thread of the process)
    dst = slot[0].numpy()
    for i, t in enumerate(tensors):
        np.copyto(dst[i], t.numpy())
    dev = slot[0].to(device, non_blocking=True)
    slot[1] = torch.cuda.Event(blocking=True)
    slot[1].record(torch.cuda.current_stream(device))
    return normalise_pixels(dev)


_out_pool = {}      # device -> {"slots": [[pinned [cap, 9] tensor, event-or-None], ...], "turn": int}: staging of the match arrays


def _issue_fine(net, ticket, metas, ncn_thres, mutual):
    """filter_coarse on the host, the fine stage of the batch enqueued, its match arrays on their way to a pinned buffer:
    nothing here waits for the fine stage."""
    fine, conf, coarse = net.fine_from_ticket(ticket, ncn_thres=ncn_thres, mutual=mutual)
    counts = [f.shape[0] for f in fine]
    # one device-to-host copy for the whole batch: [fine x1,y1,x2,y2 | confidence | coarse x1,y1,x2,y2]
    packed = torch.cat([torch.cat(fine), torch.cat(conf)[:, None], torch.cat(coarse).float()], dim=1)
    if packed.device.type != "cuda":
        return dict(host=packed, n=packed.shape[0], counts=counts, metas=metas, event=None)
    n = packed.shape[0]
    ring = _out_pool.setdefault(str(packed.device), {"slots": [], "turn": 0})
    if len(ring["slots"]) < 4:
        ring["slots"].append([None, None])
    slot = ring["slots"][ring["turn"] % len(ring["slots"])]
    ring["turn"] += 1
    if slot[1] is not None:
        slot[1].synchronize()              # the copy that last wrote this buffer has finished (and was collected: FIFO)
    if slot[0] is None or slot[0].shape[0] < n:
        slot[0] = torch.empty((max(n + n // 2, 4096), 9), dtype=torch.float32).pin_memory()
    slot[0][:n].copy_(packed, non_blocking=True)
    slot[1] = torch.cuda.Event(blocking=True)
    slot[1].record(torch.cuda.current_stream(packed.device))
    return dict(host=slot[0], n=n, counts=counts, metas=metas, event=slot[1], keep=packed)


def _collect(rec, io_thres):
    if rec["event"] is not None:
        rec["event"].synchronize()
    packed = rec["host"][:rec["n"]].numpy()
    out, start = [], 0
    for n, to_original in zip(rec["counts"], rec["metas"]):
        rows = packed[start:start + n]
        start += n
        refined, confidence, proposals = rows[:, 0:4], rows[:, 4], rows[:, 5:9]
        keep = np.flatnonzero(confidence > io_thres)
        if keep.size:
            refined, confidence, proposals = refined[keep], confidence[keep], proposals[keep]
        out.append((to_original * refined, np.array(confidence, dtype=np.float32), to_original * proposals))
    return out


_dev_pool = {}      # (device, shapes) -> ring of pinned staging buffers for the padded outputs of the device path


def _issue_device(net, f1, f2, group, ksize, ncn_thres, mutual, io_thres):
    """Coarse stage, device-side filter_coarse, both regressors and the io_thres / scaling tail of the batch enqueued,
    the padded results on their way to pinned memory.  Nothing here waits for the GPU."""
    from ... import ops
    fine, scores, coarse, counts = net.predict_fine_device(f1, f2, ksize=ksize, ncn_thres=ncn_thres, mutual=mutual)
    scale = np.concatenate([g[3] for g in group]).astype(np.float64)
    outs = ops.match_tail_batch(fine, scores, coarse, counts, scale, io_thres)
    key = (str(fine.device),) + tuple(tuple(o.shape) for o in outs)
    ring = _dev_pool.setdefault(key, {"slots": [], "turn": 0})
    if len(ring["slots"]) < 4:
        ring["slots"].append([[torch.empty(o.shape, dtype=o.dtype).pin_memory() for o in outs], None])
    slot = ring["slots"][ring["turn"] % len(ring["slots"])]
    ring["turn"] += 1
    if slot[1] is not None:
        slot[1].synchronize()
    for h, o in zip(slot[0], outs):
        h.copy_(o, non_blocking=True)
    slot[1] = torch.cuda.Event(blocking=True)
    slot[1].record(torch.cuda.current_stream(fine.device))
    return dict(host=slot[0], event=slot[1], keep=outs, jobs=[g[4] for g in group])


def _collect_device(net, rec, ncn_thres, mutual, io_thres):
    rec["event"].synchronize()
    m, s, c, n = [h.numpy() for h in rec["host"]]
    out = []
    for b, job in enumerate(rec["jobs"]):
        k = int(n[b])
        if k < 0:       # a coordinate outside the device filter's packed key (an image side >= 2^15): the per-pair host path
            from .model_helper import estimate_matches
            for f in (job[1], job[2]):      # file objects were read by the loader thread: rewind them for the second decode
                if hasattr(f, "seek"):
                    f.seek(0)
            out.append(estimate_matches(net, job[1], job[2], ksize=job[3], ncn_thres=ncn_thres, mutual=mutual, io_thres=io_thres,
                                        eval_type="fine", imsize=job[5]))
        else:
            out.append((m[b, :k].copy(), s[b, :k].copy(), c[b, :k].copy()))
    return out


def _finish(net, ticket, metas, ncn_thres, mutual, io_thres):
    return _collect(_issue_fine(net, ticket, metas, ncn_thres, mutual), io_thres)


def _bounded_map(pool, fn, jobs, ahead):
    """pool.map with at most `ahead` items decoded but not yet consumed.  (Executor.map submits everything at once: the
    loader threads then run flat out, far ahead of the GPU, and every one of their Python-level steps competes with the
    main thread -- which issues a thousand launches per batch -- for the interpreter lock: measured 170 pairs/s with 16
    free-running loaders against 315 for the same stages run one after the other.)"""
    futures = deque()
    jobs = iter(jobs)
    for job in jobs:
        futures.append(pool.submit(fn, job))
        if len(futures) >= ahead:
            break
    while futures:
        item = futures.popleft().result()
        for job in jobs:
            futures.append(pool.submit(fn, job))
            break
        yield item


def estimate_matches_stream(net, pairs, ksize=2, ncn_thres=0.0, mutual=True, io_thres=0.25, imsize=None,
                            batch=8, workers=4, lookahead=None, device_filter=True):
    """Generator over `pairs` (iterable of (im1, im2) paths / file objects): yields
    (matches float64 [M,4], scores float32 [M], coarse_matches float64 [M,4]) per pair, in order.
    workers: loader threads (a 480x640 JPEG pair decodes in 2 ms: a few threads feed the GPU); lookahead: pairs decoded
    ahead of the batch being matched (default 3 batches); device_filter=False keeps filter_coarse on the host (the
    reference's numpy semantics literally; same results, tests/test_gpu_parity.py)."""
    on_device = (device_filter and torch.device(net.device).type == "cuda" and getattr(net, "panc", 1) == 1
                 and hasattr(net, "predict_fine_device"))
    jobs = ((i, a, b, ksize, net.upsample, imsize) for i, (a, b) in enumerate(pairs))
    pending = deque()          # (ticket, metas) whose fine stage has not been issued yet
    issued = deque()           # batches whose fine stage is enqueued and whose results are being copied to the host
    with ThreadPoolExecutor(max_workers=max(1, workers)) as pool, torch.no_grad():
        loaded = _bounded_map(pool, _load, jobs, lookahead or 3 * batch)
        group = []

        def _take(rec):
            return _collect_device(net, rec, ncn_thres, mutual, io_thres) if "jobs" in rec else _collect(rec, io_thres)

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
            if on_device:
                issued.append(_issue_device(net, f1, f2, group, ksize, ncn_thres, mutual, io_thres))
            else:
                pending.append((net.coarse_async(f1, f2, ksize=ksize), [g[3] for g in group]))
            group.clear()

        for item in loaded:
            if group and (item[1].shape != group[0][1].shape or item[2].shape != group[0][2].shape or len(group) >= batch):
                flush()
                # three batches in flight: coarse stage of the newest enqueued before the previous one is filtered on the
                # host and its fine stage enqueued, before the results of the one before that are unpacked -- the main
                # thread waits for the GPU only when the GPU is what limits the stream
                while len(pending) > 1:
                    ticket, metas = pending.popleft()
                    issued.append(_issue_fine(net, ticket, metas, ncn_thres, mutual))
                while len(issued) > (2 if on_device else 1):
                    yield from _take(issued.popleft())
            group.append(item)
        flush()
        while pending:
            ticket, metas = pending.popleft()
            issued.append(_issue_fine(net, ticket, metas, ncn_thres, mutual))
        while issued:
            yield from _take(issued.popleft())
