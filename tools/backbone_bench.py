"""Pyramid producer on the GPU: the HIP convolutions of layer1..layer3 against MIOpen (same torch module, P2P_BACKBONE)
-- accuracy of both against an fp64 evaluation of one block chain, time per image by batch size, time per layer shape."""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from patch2pix_amd import ops  # noqa: E402
from patch2pix_amd.networks import resnet  # noqa: E402
from patch2pix_amd.utils import synthetic  # noqa: E402

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
H, W = int(os.environ.get("H", 480)), int(os.environ.get("W", 640))
net = resnet.ResNet34()
net.change_stride("layer3")
sd = synthetic.make_state_dict(0)
net.load_state_dict({k[len("extract."):]: v for k, v in sd.items() if k.startswith("extract.") and "layer4" not in k and "fc." not in k}, strict=False)
net = net.to(dev).eval()


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


with torch.no_grad():
    # accuracy: both paths against the module evaluated in fp64 on the CPU (small image: fp64 convolutions are slow)
    a, _ = synthetic.make_image_pair(3, 96, 128)
    im = (torch.from_numpy(a).permute(2, 0, 1).float() / 255.0)[None].to(dev)
    ref = [t.float() for t in net.double().cpu().pyramid(im.double().cpu())]
    net = net.float().to(dev)
    for mode in ("hip", "miopen"):
        os.environ["P2P_BACKBONE"] = mode
        got = net.pyramid(im)
        print(mode, "max |err| / max |ref| per level:", " ".join(f"{((g.cpu() - r).abs().max() / r.abs().max()).item():.2e}" for g, r in zip(got[1:], ref[1:])))
    for nb in (2, 8, 16):
        x = torch.randn(nb, 3, H, W, device=dev)
        for mode in ("hip", "miopen"):
            os.environ["P2P_BACKBONE"] = mode
            print(f"{mode:7s} batch {nb:2d} {H}x{W}: {timed(lambda: net.pyramid(x)) / nb:.3f} ms per image")
    # per layer shape (batch 16)
    os.environ["P2P_BACKBONE"] = "hip"
    nb = 16
    for ci, co, ks, st, h, w in ((64, 64, 3, 1, H // 4, W // 4), (64, 128, 3, 2, H // 4, W // 4), (128, 128, 3, 1, H // 8, W // 8),
                                 (128, 256, 3, 1, H // 8, W // 8), (256, 256, 3, 1, H // 8, W // 8), (128, 256, 1, 1, H // 8, W // 8)):
        conv = torch.nn.Conv2d(ci, co, ks, st, ks // 2, bias=False).to(dev)
        bn = torch.nn.BatchNorm2d(co).to(dev).eval()
        cv = ops.ConvBN(conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, st, dev)
        x = torch.relu(torch.randn(nb, h, w, ci, device=dev))
        xm = ops.absmax_batch(x)
        t = timed(lambda: cv.forward(x, xm), 20)
        xc = x.permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last)
        convc = conv.to(memory_format=torch.channels_last)
        tm = timed(lambda: torch.relu_(bn(convc(xc))), 20)
        fl = 2.0 * nb * (h // st) * (w // st) * co * ci * ks * ks
        print(f"conv {ci:3d}->{co:3d} k{ks} s{st} {h}x{w} x{nb}: hip {t:.3f} ms = {fl / t / 1e9:.0f} TFLOP/s ({fl / t / 1e9 / 833.3:.2f} of 833)   miopen NHWC {tm:.3f} ms")
