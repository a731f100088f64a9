"""Phase cycle counts of the fp16x2 regress kernel (needs the -DP2P_X3_TIMING build: P2P_ALLOW_EXPERIMENT=1 P2P_LIB_PATH=tools/exp/lib_timing.so; MODE=fp16x2w for the conv1 launch of the Winograd path)."""
import sys, os; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ctypes, torch
from patch2pix_amd import ops, _lib
from patch2pix_amd.utils import synthetic
dev = torch.device("cuda:0")
sd = synthetic.make_state_dict(0, backbone=False)
if os.environ.get("ZERO_W"):      # power experiment: all-zero convolution weights (same instruction stream, no operand toggling)
    for k in list(sd):
        if ".conv.0." in k or ".conv.2." in k: sd[k] = sd[k] * 0
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
raw = torch.zeros((5 * n + 64 * 8 * 16,), device=dev)
MODE = os.environ.get('MODE', 'fp16x2')
mid.set_mode(MODE); fine.set_mode(MODE)
ws = torch.empty(_lib.p2p_regress_workspace_bytes(n), dtype=torch.uint8, device=dev)
for _ in range(3):
    _lib.check(_lib.p2p_regress(mid.handle, fine.handle, ctypes.byref(pa), ctypes.byref(pb), props.data_ptr(), 0, n,
                                m1.data_ptr(), q1.data_ptr(), raw.data_ptr(), m2.data_ptr(), q2.data_ptr(), None,
                                ws.data_ptr(), ws.numel(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "regress")
torch.cuda.synchronize()
d = raw[5 * n:].view(64, 8, 16).cpu()
names = ["gather", "scale+tab", "im2col", "level0+sync", "P (level 1)", "C (levels 2+3)", "fold", "conv1 end sync", "BN1+H write",
         "conv2 entry sync", "conv2 MFMA", "epilogue+sync", "-"]
if MODE == "fp16x2w":
    names[8], names[9], names[10] = "BN1 + H -> LDS", "transform + stores issued", "-"
for grp, sl in (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8))):
    med = [d[:, sl, i].median().item() for i in range(13)]
    print(f"ticks per wave, median over 64 workgroups, {grp} (level 0, the middle proposal of the work-group's share):")
    for nme, v in zip(names, med):
        print(f"  {nme:20s} {v:9.0f}  ({100 * v / sum(med):4.1f} %)")
    print(f"  total                {sum(med):9.0f}")
print("MFMA issue slots x 32 cycles x 2 waves per SIMD: level0 3072, P 55296, C 82944, conv2 221184")
