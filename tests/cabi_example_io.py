"""File format of examples/cabi_coarse.c (shared by the GPU test and the emulated CPU test)."""
import numpy as np


def write_input(path, sd, fa, fb, ksize, upsample=8):
    """fa/fb: [B,C,h,w] fp32 CPU tensors; sd: state_dict holding the NeighConsensus filters."""
    nb, c, ha, wa = fa.shape
    _, _, hb, wb = fb.shape
    with open(path, "wb") as f:
        f.write(np.array([nb, c, ha, wa, hb, wb, ksize, upsample], dtype=np.int32).tobytes())
        for key in ("ncn.conv.0.weight", "ncn.conv.0.bias", "ncn.conv.2.weight", "ncn.conv.2.bias"):
            f.write(sd[key].detach().cpu().numpy().astype(np.float32).tobytes())
        f.write(fa.contiguous().numpy().tobytes())
        f.write(fb.contiguous().numpy().tobytes())


def read_output(path, ncell, nmatch, ksize):
    """-> (corr fp32 [ncell], delta uint8 [ncell] or None, matches int64 [nmatch*4], scores fp32 [nmatch])."""
    raw = open(path, "rb").read()
    off = 0
    corr = np.frombuffer(raw, np.float32, ncell, off); off += 4 * ncell
    delta = None
    if ksize > 1:
        delta = np.frombuffer(raw, np.uint8, ncell, off); off += ncell
    matches = np.frombuffer(raw, np.int64, nmatch * 4, off); off += 8 * nmatch * 4
    scores = np.frombuffer(raw, np.float32, nmatch, off); off += 4 * nmatch
    assert off == len(raw), "output file has trailing or missing bytes"
    return corr, delta, matches, scores
