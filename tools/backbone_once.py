import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from patch2pix_amd.networks import resnet
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
net = resnet.ResNet34(); net.change_stride("layer3"); net = net.to(dev).eval()
x = torch.randn(int(os.environ.get("NB", 16)), 3, 480, 640, device=dev)
with torch.no_grad():
    for _ in range(6):
        net.pyramid(x)
torch.cuda.synchronize()
