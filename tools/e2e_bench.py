"""End-to-end estimate_matches throughput (image files -> match arrays), informational:
PIL load + resize, backbone on PyTorch-ROCm (MIOpen), HIP matching path, D2H.  `measure()` is what bench.py
reports under "e2e"; run as a script it prints the same dict."""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from PIL import Image  # noqa: E402

from patch2pix_amd.utils import synthetic  # noqa: E402
from patch2pix_amd.utils.eval import model_helper  # noqa: E402


def producer_flop(net, H, W):
    """2 x MACs of every Conv2d the pyramid producer runs for one H x W image (forward hooks on the torch module, CPU meta pass)."""
    import torch.nn as nn
    total = [0.0]

    def hook(m, inp, outp):
        total[0] += 2.0 * outp.numel() / outp.shape[0] * (m.in_channels // m.groups) * m.kernel_size[0] * m.kernel_size[1]
    mods = [m for m in net.extract.modules() if isinstance(m, nn.Conv2d)]
    hs = [m.register_forward_hook(hook) for m in mods]
    prev = os.environ.get("P2P_BACKBONE")
    try:
        os.environ["P2P_BACKBONE"] = "miopen"            # the torch path of the same module: only shapes matter here
        with torch.no_grad():
            net.extract.pyramid(torch.zeros(1, 3, H, W, device=net.device))
    finally:
        for h in hs:
            h.remove()
        if prev is None:
            os.environ.pop("P2P_BACKBONE", None)
        else:
            os.environ["P2P_BACKBONE"] = prev
    return total[0]


def measure(net, H=480, W=640, pairs=4, reps=3, stream_pairs=160):
    """-> {per_pair_pairs_per_s, stream_pairs_per_s, backbone_ms_per_image, ...}; jpeg inputs (quality 95)."""
    from patch2pix_amd.utils.eval.stream import estimate_matches_stream
    out = {"input": f"{H}x{W} JPEG files, estimate_matches(ksize=2, io_thres=0.25); random-init weights, so the number of "
                    "proposals per pair is the model's own mutual matches, not ptmax"}
    with tempfile.TemporaryDirectory() as td:
        paths = []
        for i in range(max(pairs, 8)):
            a, b = synthetic.make_image_pair(100 + i, H, W)
            pa, pb = os.path.join(td, f"{i}a.jpg"), os.path.join(td, f"{i}b.jpg")
            Image.fromarray(a).save(pa, quality=95)
            Image.fromarray(b).save(pb, quality=95)
            paths.append((pa, pb))
        for pa, pb in paths[:2]:
            model_helper.estimate_matches(net, pa, pb, ksize=2, io_thres=0.25)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        for _ in range(reps):
            for pa, pb in paths[:pairs]:
                m, s, c = model_helper.estimate_matches(net, pa, pb, ksize=2, io_thres=0.25)
                n += 1
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out["per_pair_pairs_per_s"] = n / dt
        out["per_pair_matches_last"] = int(m.shape[0])
        # streaming form: threaded loading, batched backbone, shared fine launch
        work = (paths[:8] * (stream_pairs // 8 + 1))[:stream_pairs]
        list(estimate_matches_stream(net, (paths[:8] * 5), batch=8, workers=4))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nres = sum(1 for _ in estimate_matches_stream(net, work, batch=8, workers=4))
        torch.cuda.synchronize()
        out["stream_pairs_per_s"] = nres / (time.perf_counter() - t0)
        out["stream"] = f"estimate_matches_stream(batch=8, workers=4) over {nres} pairs: loader threads, HIP pyramid producer, device-side filter / fine stage / tail, three batches in flight"
    # single-pair latency, images already on the device: eager (host-side filter_coarse, ~700 launches) against the
    # whole path -- backbone, coarse stage, device-side filter, both regressors -- replayed as one hipGraph
    try:
        from patch2pix_amd.utils.eval.graphed import GraphedMatcher
        a, b = synthetic.make_image_pair(7, H, W)
        norm = lambda x: ((torch.from_numpy(x).permute(2, 0, 1).float() / 255.0 - torch.tensor([0.485, 0.456, 0.406])[:, None, None])
                          / torch.tensor([0.229, 0.224, 0.225])[:, None, None])[None].to(net.device)
        ia, ib = norm(a), norm(b)
        with torch.no_grad():
            for _ in range(3):
                net.predict_fine(ia, ib, ksize=2)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                net.predict_fine(ia, ib, ksize=2)
            torch.cuda.synchronize()
            out["latency_ms_pair_eager"] = (time.perf_counter() - t0) / 10 * 1e3
            g = GraphedMatcher(net, H, W).capture()
            g.load(ia, ib)
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                g.replay()
            torch.cuda.synchronize()
            out["latency_ms_pair_hipgraph"] = (time.perf_counter() - t0) / 20 * 1e3
            out["hipgraph_proposals"] = int(g.out[3][0])
    except Exception as e:      # informational
        out["latency_error"] = repr(e)
    try:
        from tools import stream_breakdown
        st = stream_breakdown.main(H, W, 8, streams=False)
        out["stages_of_a_batch_of_8_pairs_ms"] = {k: round(v, 3) for k, v in st.items() if k.endswith("_ms") or k.startswith("load_ms")}
        out["stages_note"] = ("each stage timed alone and synchronised; load_ms_per_pair_one_thread = PIL decode + resize of both "
                              f"images on one loader thread; {st['proposals_per_pair']:.0f} proposals per pair (the model's own mutual matches)")
    except Exception as e:      # informational
        out["stages_error"] = repr(e)
    def producer_ms(nb, reps=10):
        im = torch.randn(nb, 3, H, W, device=net.device)
        with torch.no_grad():
            for _ in range(3):
                net.extract.pyramid(im)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                net.extract.pyramid(im)
            torch.cuda.synchronize()
        return (time.perf_counter() - t0) / (reps * nb) * 1e3
    out["backbone_ms_per_image"] = producer_ms(2)
    out["backbone_ms_per_image_batch16"] = producer_ms(16)
    # algorithmic work of the producer (conv1 ... layer3 of ResNet34 with the stride patch, networks/resnet.py:125-173): counted
    # from the module's convolutions, 2 x output pixels x output channels x input channels x taps
    flop = producer_flop(net, H, W)
    ach = flop / (out["backbone_ms_per_image_batch16"] * 1e-3) / 1e12
    out["producer_roofline"] = {"kernel": "conv_kernel (+ stem_kernel, maxpool_nhwc_kernel, nhwc_to_nchw_kernel)", "bound": "mfma",
                                "algorithmic_flop_per_image": flop, "avg_ms_per_image_batch16": out["backbone_ms_per_image_batch16"],
                                "achieved": ach, "peak": 2500.0 / 3.0, "unit": "TFLOP/s", "frac": ach / (2500.0 / 3.0),
                                "note": "all 36 launches of one image's pyramid, batch of 16 images, wall clock / 16; fp16x2: three "
                                        "MFMA products per fp32 product, peak = 2500 / 3"}
    prev = os.environ.get("P2P_BACKBONE")
    try:
        os.environ["P2P_BACKBONE"] = "miopen"
        out["backbone_ms_per_image_batch16_miopen"] = producer_ms(16, reps=5)
    finally:
        if prev is None:
            os.environ.pop("P2P_BACKBONE", None)
        else:
            os.environ["P2P_BACKBONE"] = prev
    try:
        from tools import hpatches_substitute
        out["configs2"] = hpatches_substitute.measure(net)
    except Exception as e:      # informational
        out["configs2_error"] = repr(e)
    out["backbone"] = ("pyramid producer = HIP convolutions of csrc/backbone.hip (fp32-equivalent fp16x2); *_miopen = the same "
                       "torch module through PyTorch-ROCm / MIOpen fp32")
    return out


if __name__ == "__main__":
    import json
    from patch2pix_amd.utils.host import pin_process_to_gpu
    pin_process_to_gpu(0)
    torch.backends.cudnn.benchmark = True
    print(json.dumps(measure(model_helper.load_model(synthetic.make_checkpoint(0), lprint=lambda *a: None)), indent=1))
