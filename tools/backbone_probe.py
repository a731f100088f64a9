"""Backbone (ResNet-34 to layer 3, fp32, PyTorch-ROCm/MIOpen) time per 480x640 image: memory format x find mode x batch."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from patch2pix_amd.utils import synthetic
from patch2pix_amd.utils.eval import model_helper
net = model_helper.load_model(synthetic.make_checkpoint(0), lprint=lambda *a: None)
dev = net.device
def run(tag, im, extract, n=10):
    with torch.no_grad():
        for _ in range(3): extract.pyramid(im)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): out = extract.pyramid(im)
        torch.cuda.synchronize()
    print(f"{tag}: {(time.perf_counter() - t0) / n / im.shape[0] * 1e3:.3f} ms per image", flush=True)
    return out
for bench in (False, True):
    torch.backends.cudnn.benchmark = bench
    for B in (2, 16):
        im = torch.randn(B, 3, 480, 640, device=dev)
        ref = run(f"NCHW  benchmark={bench} batch {B}", im, net.extract)
        import copy
        ext_cl = copy.deepcopy(net.extract).to(memory_format=torch.channels_last)
        out = run(f"NHWC  benchmark={bench} batch {B}", im.contiguous(memory_format=torch.channels_last), ext_cl)
        print("   max |diff| layer3:", (out[4] - ref[4]).abs().max().item(), " output contiguous NCHW:", out[4].is_contiguous())
        try:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                run(f"NCHW bf16 autocast (NOT parity-safe, reference only) batch {B}", im, net.extract)
        except Exception as e:
            print("autocast failed", e)
