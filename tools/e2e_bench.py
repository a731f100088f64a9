"""End-to-end estimate_matches throughput (image files -> match arrays), informational:
PIL load + resize, backbone on PyTorch-ROCm, HIP matching path, D2H."""
import os, sys, time, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from PIL import Image
from patch2pix_amd.utils import synthetic
from patch2pix_amd.utils.eval import model_helper

torch.backends.cudnn.benchmark = True
net = model_helper.load_model(synthetic.make_checkpoint(0), lprint=lambda *a: None)
with tempfile.TemporaryDirectory() as td:
    paths = []
    for i in range(4):
        a, b = synthetic.make_image_pair(100 + i, 480, 640)
        pa, pb = os.path.join(td, f"{i}a.png"), os.path.join(td, f"{i}b.png")
        Image.fromarray(a).save(pa); Image.fromarray(b).save(pb); paths.append((pa, pb))
    for pa, pb in paths: model_helper.estimate_matches(net, pa, pb, ksize=2, io_thres=0.25)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
    for rep in range(5):
        for pa, pb in paths:
            m, s, c = model_helper.estimate_matches(net, pa, pb, ksize=2, io_thres=0.25); n += 1
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"end-to-end estimate_matches: {n/dt:.1f} pairs/s ({dt/n*1e3:.1f} ms/pair), {m.shape[0]} matches in the last pair")
    # streaming form: threaded loading, batched backbone, shared fine launch
    from patch2pix_amd.utils.eval.stream import estimate_matches_stream
    for ext, q in (("jpg", {"quality": 95}), ("png", {})):
        files = []
        for i in range(8):
            a, b = synthetic.make_image_pair(200 + i, 480, 640)
            pa, pb = os.path.join(td, f"s{i}a.{ext}"), os.path.join(td, f"s{i}b.{ext}")
            Image.fromarray(a).save(pa, **q); Image.fromarray(b).save(pb, **q); files.append((pa, pb))
        work = files * 8
        list(estimate_matches_stream(net, files, batch=8, workers=16))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        nres = sum(1 for _ in estimate_matches_stream(net, work, batch=8, workers=16))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"end-to-end estimate_matches_stream ({ext}): {nres/dt:.1f} pairs/s ({dt/nres*1e3:.1f} ms/pair)")
    # backbone only
    im = torch.randn(1, 3, 480, 640, device="cuda")
    with torch.no_grad():
        for _ in range(3): net.extract.pyramid(im)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): net.extract.pyramid(im)
        torch.cuda.synchronize(); print(f"backbone forward_all: {(time.perf_counter()-t0)/20*1e3:.2f} ms per image")
