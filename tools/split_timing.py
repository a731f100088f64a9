"""Phase cycle counts of the split kernel (needs the -DP2P_SPLIT_TIMING build, P2P_LIB_PATH=tools/exp/lib_timing.so)."""
import sys, os; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ctypes, torch
from patch2pix_amd import ops, _lib
from patch2pix_amd.utils import synthetic
dev = torch.device("cuda:0")
sd = synthetic.make_state_dict(0, backbone=False)
sub = lambda p: {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}
mid = ops.RegressorWeights(sub("regress_mid."), dev); fine = ops.RegressorWeights(sub("regress_fine."), dev)
H, W, n = 480, 640, int(os.environ.get("NPROP", "2000"))
p1 = synthetic.make_pyramid(7, H, W); p2 = synthetic.make_pyramid(8, H, W)
g = torch.Generator().manual_seed(9)
props = torch.stack([torch.randint(0, W + 1, (n,), generator=g), torch.randint(0, H + 1, (n,), generator=g),
                     torch.randint(0, W + 1, (n,), generator=g), torch.randint(0, H + 1, (n,), generator=g)], 1).to(dev)
g1 = [t.to(dev) for t in p1[:4]]; g2 = [t.to(dev) for t in p2[:4]]
pa, ka = ops._pyramid(g1); pb, kb = ops._pyramid(g2)
m1 = torch.empty((n, 4), device=dev); q1 = torch.empty((n,), device=dev); m2 = torch.empty((n, 4), device=dev); q2 = torch.empty((n,), device=dev)
raw = torch.zeros((5 * n + 64 * 8 * 8,), device=dev)
for _ in range(3):
    _lib.check(_lib.p2p_regress(mid.handle, fine.handle, ctypes.byref(pa), ctypes.byref(pb), props.data_ptr(), 0, n,
                                m1.data_ptr(), q1.data_ptr(), raw.data_ptr(), m2.data_ptr(), q2.data_ptr(), None,
                                ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "regress")
torch.cuda.synchronize()
d = raw[5 * n:].view(64, 8, 8).cpu()
names = ["gather", "scale", "im2col", "conv1", "Hwrite", "conv2", "epilogue", "fc+parse"]
med = [d[..., i].median().item() for i in range(8)]
print("ticks per wave, median over 64 workgroups x 8 waves (level 0):")
for nme, v in zip(names, med):
    print(f"  {nme:9s} {v:9.0f}  ({100 * v / sum(med):4.1f} %)")
print(f"  total     {sum(med):9.0f};  ideal MFMA ticks with two waves per SIMD: conv1 {2 * 584 * 6 * 32}, conv2 {2 * 576 * 6 * 32}")
