"""Torch-tensor front end of the C ABI: owns nothing but the two kinds of weight handle.
Every function launches asynchronously on torch's current HIP stream of the tensors' device."""
import ctypes
import math
import os

import torch

from . import _lib


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32c(t, name):
    if t.dtype != torch.float32 or not t.is_cuda:
        raise TypeError(f"{name} must be a float32 tensor on the GPU (got {t.dtype}, {t.device})")
    return t.contiguous()


def _host(t):
    return t.detach().to("cpu", torch.float32).contiguous()


class NcnWeights:
    """Device-resident NeighConsensus filters (reference networks/ncn/model.py:124-143)."""

    def __init__(self, w1, b1, w2, b2, device):
        keep = [_host(w1), _host(b1), _host(w2), _host(b2)]
        if tuple(keep[0].shape) != (3, 16, 1, 3, 3, 3) or tuple(keep[2].shape) != (3, 1, 16, 3, 3, 3):
            raise NotImplementedError("only NeighConsensus(kernel_sizes=[3,3], channels=[16,1]) is implemented")
        self.handle = ctypes.c_void_p()
        with torch.cuda.device(device):
            _lib.check(_lib.p2p_ncn_create(*[t.data_ptr() for t in keep], ctypes.byref(self.handle)), "p2p_ncn_create")
        self.device = torch.device(device)

    def set_tile(self, ta=0, tb=0, tc=0):
        """Force the work-group tile of the consensus kernel (tests, sweeps); (0, 0, 0) = automatic.  Results do not depend on it."""
        _lib.check(_lib.p2p_ncn_set_tile(self.handle, int(ta), int(tb), int(tc)), "p2p_ncn_set_tile")

    def __del__(self):
        if getattr(self, "handle", None) and _lib is not None:      # _lib is None during interpreter shutdown
            _lib.p2p_ncn_destroy(self.handle)
            self.handle = None


# Experiments and the tile-independence tests: (mt, nt, wn) forced on every ConvBN launch (p2p_conv_set_tile); None = the
# library picks by launch size.  The environment variable P2P_CONV_TILE="mt,nt,wn" sets it at import (tools).
def _parse_conv_tile(text):
    try:
        t = tuple(int(v) for v in text.split(","))
    except ValueError:
        t = ()
    if len(t) != 3 or min(t) < 0:
        raise ValueError(f"P2P_CONV_TILE={text!r}: expected three non-negative integers 'mt,nt,wn'")
    return t


FORCED_CONV_TILE = _parse_conv_tile(os.environ["P2P_CONV_TILE"]) if os.environ.get("P2P_CONV_TILE") else None


class ConvBN:
    """Device-resident packed Conv2d(bias=False) + BatchNorm2d (eval) of the pyramid producer
    (reference networks/resnet.py:26-60); `forward` works on fp32 NHWC activations."""

    def __init__(self, conv_weight, bn_weight, bn_bias, bn_mean, bn_var, stride, device):
        keep = [_host(t) for t in (conv_weight, bn_weight, bn_bias, bn_mean, bn_var)]
        co, ci, ks, ks2 = keep[0].shape
        if ks != ks2:
            raise NotImplementedError("square kernels only")
        bn = _lib.BnParams(*[t.data_ptr() for t in keep[1:]])
        self.handle = ctypes.c_void_p()
        with torch.cuda.device(device):
            _lib.check(_lib.p2p_conv_create(keep[0].data_ptr(), ctypes.byref(bn), ci, co, ks, int(stride), ctypes.byref(self.handle)),
                       "p2p_conv_create")
        self.ci, self.co, self.ks, self.stride = ci, co, ks, int(stride)
        self.device = torch.device(device)
        self._tile = None

    def __del__(self):
        if getattr(self, "handle", None) and _lib is not None:
            _lib.p2p_conv_destroy(self.handle)
            self.handle = None

    def forward(self, x, xmax, residual=None, relu=True, ymax=None):
        """x [n,h,w,ci] fp32 NHWC, xmax [n] int32 (float bits of max |x| per image) -> y [n,ho,wo,co]; ymax (optional
        [n] int32, ZERO on entry) receives the float bits of max |y| per image."""
        n, h, w, ci = x.shape
        if ci != self.ci or x.dtype != torch.float32 or not x.is_cuda or not x.is_contiguous():
            raise TypeError(f"ConvBN.forward: expected a contiguous float32 NHWC tensor with {self.ci} channels on the GPU")
        if FORCED_CONV_TILE != self._tile:
            _lib.check(_lib.p2p_conv_set_tile(self.handle, *(FORCED_CONV_TILE or (0, 0, 0))), "p2p_conv_set_tile")
            self._tile = FORCED_CONV_TILE
        pad = self.ks // 2
        ho, wo = (h + 2 * pad - self.ks) // self.stride + 1, (w + 2 * pad - self.ks) // self.stride + 1
        y = torch.empty((n, ho, wo, self.co), device=x.device, dtype=torch.float32)
        if residual is not None and (residual.shape != y.shape or not residual.is_contiguous()):
            raise TypeError("ConvBN.forward: residual must be a contiguous NHWC tensor of the output's shape")
        _lib.check(_lib.p2p_conv_forward(self.handle, x.data_ptr(), xmax.data_ptr(), n, h, w,
                                         residual.data_ptr() if residual is not None else None, int(relu), y.data_ptr(),
                                         ymax.data_ptr() if ymax is not None else None, _stream()), "p2p_conv_forward")
        return y


class Stem:
    """Device-resident packed conv1 7x7/2 + bn1 (+ ReLU) of the pyramid producer (reference networks/resnet.py:101-103)."""

    def __init__(self, conv_weight, bn_weight, bn_bias, bn_mean, bn_var, device):
        keep = [_host(t) for t in (conv_weight, bn_weight, bn_bias, bn_mean, bn_var)]
        if tuple(keep[0].shape) != (64, 3, 7, 7):
            raise NotImplementedError("only the ResNet stem Conv2d(3, 64, 7, 2, 3) is implemented")
        bn = _lib.BnParams(*[t.data_ptr() for t in keep[1:]])
        self.handle = ctypes.c_void_p()
        with torch.cuda.device(device):
            _lib.check(_lib.p2p_stem_create(keep[0].data_ptr(), ctypes.byref(bn), ctypes.byref(self.handle)), "p2p_stem_create")
        self.device = torch.device(device)

    def __del__(self):
        if getattr(self, "handle", None) and _lib is not None:
            _lib.p2p_stem_destroy(self.handle)
            self.handle = None

    def forward(self, image):
        """image [n,3,h,w] fp32 NCHW -> relu(bn1(conv1(image))) [n,64,ho,wo] NCHW."""
        image = _f32c(image, "image")
        n, c, h, w = image.shape
        if c != 3:
            raise TypeError("Stem.forward: three-channel images only")
        imax = absmax_batch(image)
        y = torch.empty((n, 64, (h - 1) // 2 + 1, (w - 1) // 2 + 1), device=image.device, dtype=torch.float32)
        _lib.check(_lib.p2p_stem_forward(self.handle, image.data_ptr(), imax.data_ptr(), n, h, w, y.data_ptr(), _stream()), "p2p_stem_forward")
        return y


def maxpool_nhwc(x, ymax=None):
    """MaxPool2d(3, 2, 1) of x [n,c,h,w] NCHW -> y [n,hp,wp,c] NHWC; ymax: optional [n] int32, ZERO on entry."""
    x = _f32c(x, "x")
    n, c, h, w = x.shape
    y = torch.empty((n, (h - 1) // 2 + 1, (w - 1) // 2 + 1, c), device=x.device, dtype=torch.float32)
    _lib.check(_lib.p2p_maxpool_nhwc(x.data_ptr(), n, c, h, w, y.data_ptr(), ymax.data_ptr() if ymax is not None else None, _stream()),
               "p2p_maxpool_nhwc")
    return y


def nhwc_to_nchw(x):
    """x [n,h,w,c] -> [n,c,h,w] contiguous."""
    x = _f32c(x, "x")
    n, h, w, c = x.shape
    y = torch.empty((n, c, h, w), device=x.device, dtype=torch.float32)
    _lib.check(_lib.p2p_nhwc_to_nchw(x.data_ptr(), n, h, w, c, y.data_ptr(), _stream()), "p2p_nhwc_to_nchw")
    return y


def absmax_batch(x):
    """x [n, ...] contiguous fp32 on the GPU -> [n] int32: float bits of max |x| per item."""
    x = _f32c(x, "x")
    out = torch.empty((x.shape[0],), device=x.device, dtype=torch.int32)
    _lib.check(_lib.p2p_absmax_batch(x.data_ptr(), x[0].numel(), x.shape[0], out.data_ptr(), _stream()), "p2p_absmax_batch")
    return out


class RegressorWeights:
    """Device-resident packed FeatRegressNet (reference networks/modules.py:56-112).
    `sd` maps the sub-state_dict keys ('conv.0.weight', 'fc.6.bias', ...) to tensors."""

    def __init__(self, sd, device):
        exp = {"conv.0.weight": (512, 518, 3, 3), "conv.2.weight": (512, 512, 3, 3), "fc.0.weight": (512, 512),
               "fc.3.weight": (256, 512), "fc.6.weight": (5, 256)}
        for k, shp in exp.items():
            if tuple(sd[k].shape) != shp:
                raise NotImplementedError(f"regressor {k} has shape {tuple(sd[k].shape)}; only the released "
                                          f"configuration {shp} is implemented")
        keep = {k: _host(v) for k, v in sd.items() if v.is_floating_point()}
        p = _lib.RegressorParams()

        def bn(prefix):
            return _lib.BnParams(keep[prefix + ".weight"].data_ptr(), keep[prefix + ".bias"].data_ptr(),
                                 keep[prefix + ".running_mean"].data_ptr(), keep[prefix + ".running_var"].data_ptr())

        p.conv1_w = keep["conv.0.weight"].data_ptr(); p.bn1 = bn("conv.1")
        p.conv2_w = keep["conv.2.weight"].data_ptr(); p.bn2 = bn("conv.3")
        p.fc1_w = keep["fc.0.weight"].data_ptr(); p.fc1_b = keep["fc.0.bias"].data_ptr(); p.bnf1 = bn("fc.1")
        p.fc2_w = keep["fc.3.weight"].data_ptr(); p.fc2_b = keep["fc.3.bias"].data_ptr(); p.bnf2 = bn("fc.4")
        p.fc3_w = keep["fc.6.weight"].data_ptr(); p.fc3_b = keep["fc.6.bias"].data_ptr()
        self.handle = ctypes.c_void_p()
        with torch.cuda.device(device):
            _lib.check(_lib.p2p_regressor_create(ctypes.byref(p), ctypes.byref(self.handle)), "p2p_regressor_create")
        self.device = torch.device(device)
        env = os.environ.get("P2P_REGRESS_MODE")       # tools: the mode new handles start in (read here, not in the library)
        if env:
            self.set_mode(env)

    def set_mode(self, mode):
        """'fp16x2w' (default: fp32-equivalent, two fp16 planes under exact power-of-two scales, 3 MFMA products, the second
        convolution as Winograd F(2x2,3x3) GEMMs), 'fp16x2' (the same arithmetic, both convolutions direct, one launch) or 'f32'
        (exact fp32 MFMA).  The first selection of a
        non-default mode packs and uploads that mode's weight stream (host work, ~1 s)."""
        if mode not in _lib.REGRESS_MODES:
            removed = {"bf16x2": "round 5", "bf16x3": "round 4"}
            why = f" ({mode!r} was removed in {removed[mode]})" if mode in removed else ""
            raise ValueError(f"unknown regressor mode {mode!r}{why}: one of {sorted(_lib.REGRESS_MODES)}")
        # packing another mode's weight stream allocates and copies on the CURRENT device: make that the handle's device
        with torch.cuda.device(self.device):
            _lib.check(_lib.p2p_regressor_set_mode(self.handle, _lib.REGRESS_MODES[mode]), "p2p_regressor_set_mode")

    @property
    def mode(self):
        code = _lib.p2p_regressor_get_mode(self.handle)
        return {v: k for k, v in _lib.REGRESS_MODES.items()}[code]

    def __del__(self):
        if getattr(self, "handle", None) and _lib is not None:
            _lib.p2p_regressor_destroy(self.handle)
            self.handle = None


_workspaces = {}

# Optional launch timing (bench.py): when a list is installed here, regress() brackets its launch
# with HIP events recorded on the stream the kernel is launched on and appends (start, end, n).
regress_events = None


def _workspace(device, nbytes):
    """One growing scratch buffer per (device, stream)."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


# Upper bound on the coarse-stage scratch of one call: a batch whose pairs need more is processed in groups
# (the hidden volume of the consensus net is 184 MB per 480x640 pair, 2.9 GB per 960x1280 pair).
COARSE_WORKSPACE_LIMIT = 16 << 30


def coarse_forward_batch(feat_a, feat_b, ksize, ncn, want_delta=True, out_corr=None, out_delta=None):
    """forward_coarse_match for the B pairs of feat_a [B,C,hA,wA] / feat_b [B,C,hB,wB] (fp32 GPU) in one launch
    per kernel.  Returns (corr [B,hA',wA',hB',wB'], packed delta uint8 of the same shape or None); `out_*` let
    the caller provide (contiguous) outputs."""
    feat_a, feat_b = _f32c(feat_a, "feat_a"), _f32c(feat_b, "feat_b")
    if feat_a.dim() != 4 or feat_b.dim() != 4 or feat_a.shape[0] != feat_b.shape[0]:
        raise ValueError("coarse_forward_batch expects [B,C,h,w] feature maps with equal B")
    nb, c, ha, wa = feat_a.shape
    _, c2, hb, wb = feat_b.shape
    if c != c2:
        raise ValueError("channel mismatch between the two feature maps")
    dev = feat_a.device
    k = max(ksize, 1)
    shape = (nb, ha // k, wa // k, hb // k, wb // k)
    corr = out_corr if out_corr is not None else torch.empty(shape, dtype=torch.float32, device=dev)
    delta = None
    if ksize > 1 and want_delta:
        delta = out_delta if out_delta is not None else torch.empty(shape, dtype=torch.uint8, device=dev)
    assert corr.is_contiguous() and corr.numel() == math.prod(shape) and corr.dtype == torch.float32
    assert delta is None or (delta.is_contiguous() and delta.numel() == math.prod(shape) and delta.dtype == torch.uint8)
    if nb == 0:
        return corr, delta
    with torch.cuda.device(dev):
        per_pair = _lib.p2p_coarse_workspace_bytes(c, ha, wa, hb, wb, ksize)
        if per_pair == 0:
            raise ValueError("coarse_forward: bad sizes")
        pairs = max(1, min(nb, COARSE_WORKSPACE_LIMIT // per_pair))
        ws = _workspace(dev, pairs * per_pair)
        _lib.check(_lib.p2p_coarse_forward_batch(feat_a.data_ptr(), feat_b.data_ptr(), nb, c, ha, wa, hb, wb, ksize,
                                                 ncn.handle, corr.data_ptr(),
                                                 delta.data_ptr() if delta is not None else None,
                                                 ws.data_ptr(), ws.numel(), _stream()), "p2p_coarse_forward_batch")
    return corr, delta


def coarse_forward(feat_a, feat_b, ksize, ncn, want_delta=True, out_corr=None, out_delta=None):
    """forward_coarse_match for one pair.  feat_*: [C,h,w] fp32 GPU.  Returns (corr [hA',wA',hB',wB'],
    packed delta uint8 of the same shape or None)."""
    corr, delta = coarse_forward_batch(feat_a[None], feat_b[None], ksize, ncn, want_delta,
                                       out_corr[None] if out_corr is not None else None,
                                       out_delta[None] if out_delta is not None else None)
    return corr[0], (delta[0] if delta is not None else None)


def neigh_consensus_batch(x, ncn):
    """NeighConsensus.forward (reference networks/ncn/model.py:145-155) on a batch of volumes x [B,hA,wA,hB,wB] fp32 GPU
    -> the same shape: both consensus layers and both symmetric branches in one kernel (csrc/consensus.hip)."""
    x = _f32c(x, "x")
    if x.dim() != 5:
        raise ValueError("neigh_consensus_batch expects [B,hA,wA,hB,wB]")
    nb, ha, wa, hb, wb = x.shape
    y = torch.empty_like(x)
    if nb == 0 or x.numel() == 0:
        return y
    with torch.cuda.device(x.device):
        ws = torch.empty(nb, dtype=torch.int32, device=x.device)
        _lib.check(_lib.p2p_neigh_consensus_batch(x.data_ptr(), nb, ha, wa, hb, wb, ncn.handle, y.data_ptr(), ws.data_ptr(), nb * 4,
                                                  _stream()), "p2p_neigh_consensus_batch")
    return y


def delta_unpack(delta, ksize):
    """Packed argmax byte -> the reference's (max_i, max_j, max_k, max_l) int64 tensors."""
    out = torch.empty((4,) + tuple(delta.shape), dtype=torch.int64, device=delta.device)
    with torch.cuda.device(delta.device):
        _lib.check(_lib.p2p_delta_unpack(delta.data_ptr(), delta.numel(), ksize, out.data_ptr(), _stream()),
                   "p2p_delta_unpack")
    return out[0], out[1], out[2], out[3]


def coarse_matches_batch(corr, delta, ksize, upsample, center=True, out_matches=None, out_scores=None):
    """cal_coarse_matches for B pairs: corr [B,hA',wA',hB',wB'] (+ packed delta of the same shape or None) ->
    ([B,nB+nA,4] int64 pixel matches, [B,nB+nA] fp32 scores)."""
    corr = _f32c(corr, "corr4d")
    if corr.dim() != 5:
        raise ValueError("coarse_matches_batch expects corr4d of shape [B,hA,wA,hB,wB]")
    nb, ha, wa, hb, wb = corr.shape
    n = ha * wa + hb * wb
    dev = corr.device
    matches = out_matches if out_matches is not None else torch.empty((nb, n, 4), dtype=torch.int64, device=dev)
    scores = out_scores if out_scores is not None else torch.empty((nb, n), dtype=torch.float32, device=dev)
    assert matches.is_contiguous() and matches.numel() == nb * n * 4 and matches.dtype == torch.int64
    assert scores.is_contiguous() and scores.numel() == nb * n and scores.dtype == torch.float32
    if delta is not None:
        delta = delta.contiguous()
        if delta.dtype != torch.uint8 or delta.numel() != corr.numel():
            raise ValueError("delta must be the packed uint8 volume with the shape of corr4d")
    if nb == 0:
        return matches, scores
    with torch.cuda.device(dev):
        _lib.check(_lib.p2p_coarse_matches_batch(corr.data_ptr(), delta.data_ptr() if delta is not None else None, nb,
                                                 ha, wa, hb, wb, ksize, upsample, int(bool(center)),
                                                 matches.data_ptr(), scores.data_ptr(), _stream()),
                   "p2p_coarse_matches_batch")
    return matches, scores


def coarse_matches(corr, delta, ksize, upsample, center=True, out_matches=None, out_scores=None):
    """cal_coarse_matches for one pair: ([nB+nA,4] int64 pixel matches, [nB+nA] fp32 scores)."""
    m, sc = coarse_matches_batch(corr[None], delta[None] if delta is not None else None, ksize, upsample, center,
                                 out_matches[None] if out_matches is not None else None,
                                 out_scores[None] if out_scores is not None else None)
    return m[0], sc[0]


def filter_coarse_batch(matches, scores, ncn_thres=0.0, mutual=True):
    """filter_coarse (networks/utils.py:38-72, no ptmax) on the device for a batch: matches [B,n,4] int64, scores [B,n]
    fp32 -> (rows [B,n,4], scores [B,n], counts int32 [B]); the first counts[b] rows of item b are valid, in the
    reference's order.  counts[b] == -1 asks for the host path (a coordinate outside [0, 2^15)).  Any n up to 2^20:
    lists longer than 8192 rows are sorted through a scratch buffer (16 bytes per padded row)."""
    if matches.dtype != torch.int64 or matches.dim() != 3 or matches.shape[-1] != 4 or not matches.is_cuda:
        raise TypeError("matches must be an int64 [B,n,4] tensor on the GPU")
    scores = _f32c(scores, "scores")
    matches = matches.contiguous()
    nb, n, _ = matches.shape
    dev = matches.device
    out_m, out_s = torch.empty_like(matches), torch.empty_like(scores)
    counts = torch.empty((nb,), dtype=torch.int32, device=dev)
    if nb and n:
        with torch.cuda.device(dev):
            need = _lib.p2p_filter_coarse_workspace_bytes(nb, n)
            ws = torch.empty(need, dtype=torch.uint8, device=dev) if need else None      # stream-ordered by the allocator
            _lib.check(_lib.p2p_filter_coarse_batch(matches.data_ptr(), scores.data_ptr(), nb, n, float(ncn_thres),
                                                    int(bool(mutual)), out_m.data_ptr(), out_s.data_ptr(),
                                                    counts.data_ptr(), ws.data_ptr() if need else None, need, _stream()),
                       "p2p_filter_coarse_batch")
    else:
        counts.zero_()
    return out_m, out_s, counts


_small_rings = {}      # device index -> [ring of 8 [pinned byte buffer, event], turn]


def small_to_device(array, dtype, device):
    """A small host array -> device tensor through a ring of pinned staging buffers (per device), asynchronously.  A copy
    from pageable memory makes the host wait for everything queued on the stream before it; a caller that pipelines
    batches must never do that."""
    t = torch.as_tensor(array, dtype=dtype).contiguous()
    device = torch.device(device)
    nbytes = t.numel() * t.element_size()
    ring = _small_rings.setdefault(device.index if device.index is not None else torch.cuda.current_device(), [[None] * 8, 0])
    slot = ring[1] % 8
    ring[1] += 1
    buf = ring[0][slot]
    if buf is not None:
        buf[1].synchronize()               # the upload that last read this buffer has completed
    if buf is None or buf[0].numel() < nbytes:
        with torch.cuda.device(device):
            buf = [torch.empty((max(4096, nbytes),), dtype=torch.uint8).pin_memory(), torch.cuda.Event(blocking=True)]
        ring[0][slot] = buf
    host = buf[0][:nbytes].view(dtype).view(t.shape)
    host.copy_(t)
    out = host.to(device, non_blocking=True)
    buf[1].record(torch.cuda.current_stream(device))
    return out


def match_tail_batch(fine, scores, coarse, counts, scale, io_thres):
    """The tail of estimate_matches (utils/eval/model_helper.py:92-109) on the device for padded batch outputs:
    fine [B,n,4] fp32, scores [B,n] fp32, coarse [B,n,4] int64, counts int32 [B] (device), scale [B,4] float64 ->
    (matches [B,n,4] float64, scores [B,n] fp32, coarse [B,n,4] float64, counts int32 [B]); rows with score > io_thres
    are kept in order (all rows if none passes) and scaled to original-image pixels."""
    if fine.dtype != torch.float32 or coarse.dtype != torch.int64 or counts.dtype != torch.int32 or not fine.is_cuda:
        raise TypeError("match_tail_batch: fine fp32, coarse int64, counts int32 on the GPU expected")
    fine, scores, coarse = fine.contiguous(), _f32c(scores, "scores"), coarse.contiguous()
    nb, n, _ = fine.shape
    dev = fine.device
    scale = scale.to(torch.float64).reshape(nb, 4) if torch.is_tensor(scale) and scale.is_cuda else \
        small_to_device(torch.as_tensor(scale, dtype=torch.float64).reshape(nb, 4), torch.float64, dev)
    out_m = torch.empty((nb, n, 4), dtype=torch.float64, device=dev)
    out_c = torch.empty((nb, n, 4), dtype=torch.float64, device=dev)
    out_s = torch.empty((nb, n), dtype=torch.float32, device=dev)
    out_n = torch.empty((nb,), dtype=torch.int32, device=dev)
    if nb and n:
        with torch.cuda.device(dev):
            _lib.check(_lib.p2p_match_tail_batch(fine.data_ptr(), scores.data_ptr(), coarse.data_ptr(), counts.data_ptr(),
                                                 scale.data_ptr(), nb, n, float(io_thres), out_m.data_ptr(), out_s.data_ptr(),
                                                 out_c.data_ptr(), out_n.data_ptr(), _stream()), "p2p_match_tail_batch")
    else:
        out_n.zero_()
    return out_m, out_s, out_c, out_n


def regress_batch_dev(reg1, reg2, pyrs1, pyrs2, proposals, counts, want_raw=False):
    """regress_batch with the proposal counts in device memory: proposals [B,stride,4] (int64 or float32), counts int32
    [B] on the GPU; every output is padded to [B,stride,...], rows beyond counts[b] are left uninitialised."""
    nb, stride, _ = proposals.shape
    dev = proposals.device
    if proposals.dtype not in (torch.int64, torch.float32):
        raise TypeError("proposals must be int64 or float32")
    if counts.dtype != torch.int32 or counts.numel() != nb or not counts.is_cuda:
        raise TypeError("counts must be an int32 [B] tensor on the GPU")
    proposals = proposals.contiguous()
    pyr_a, pyr_b, keep = (_lib.Pyramid * nb)(), (_lib.Pyramid * nb)(), []
    for i in range(nb):
        pa, ka = _pyramid(pyrs1[i])
        pb, kb = _pyramid(pyrs2[i])
        pyr_a[i], pyr_b[i] = pa, pb
        keep.append((ka, kb))
    two = reg2 is not None
    out = {"matches1": torch.empty((nb, stride, 4), device=dev), "probs1": torch.empty((nb, stride), device=dev)}
    if two:
        out["matches2"], out["probs2"] = torch.empty((nb, stride, 4), device=dev), torch.empty((nb, stride), device=dev)
    if want_raw:
        out["raw1"] = torch.empty((nb, stride, 5), device=dev)
        if two:
            out["raw2"] = torch.empty((nb, stride, 5), device=dev)
    g = lambda k: out[k].data_ptr() if k in out else None
    if nb and stride:
        with torch.cuda.device(dev):
            ws = _regress_scratch(dev, nb * stride, reg1)
            _lib.check(_lib.p2p_regress_batch_dev(reg1.handle, reg2.handle if two else None, nb, pyr_a, pyr_b,
                                                  counts.data_ptr(), stride, proposals.data_ptr(),
                                                  int(proposals.is_floating_point()), g("matches1"), g("probs1"), g("raw1"),
                                                  g("matches2"), g("probs2"), g("raw2"), ws.data_ptr(), ws.numel(), _stream()),
                       "p2p_regress_batch_dev")
    del keep
    return out


def _regress_scratch(dev, n, reg=None):
    """Scratch of one regress call (the pooled convolution features wait there for the batched FC tail; in the default mode
    also the Winograd-transformed input of the second convolution, chunk by chunk): a fresh stream-ordered allocation per
    call, so that calls on different streams never share it; sized for the regressor's arithmetic mode."""
    if reg is None:
        nbytes = _lib.p2p_regress_workspace_bytes(int(n))
    else:
        nbytes = _lib.p2p_regress_workspace_bytes_mode(int(n), _lib.p2p_regressor_get_mode(reg.handle))
    return torch.empty(max(int(nbytes), 128), dtype=torch.uint8, device=dev)


def _pyramid(levels):
    if len(levels) != 4:
        raise ValueError("a pyramid is the 4 maps of feat_idx [0,1,2,3]")
    lv = [_f32c(t, "pyramid level") for t in levels]
    h, w = lv[0].shape[-2:]
    if h < 8 or w < 8:
        raise ValueError(f"images must be at least 8x8 pixels (got {h}x{w})")
    up = lambda d, j: (d + (1 << j) - 1) >> j          # the backbone's strided layers round up (resnet.py)
    exp = [(3, h, w), (64, up(h, 1), up(w, 1)), (64, up(h, 2), up(w, 2)), (128, up(h, 3), up(w, 3))]
    for t, e in zip(lv, exp):
        if tuple(t.shape) != e:
            raise ValueError(f"pyramid level has shape {tuple(t.shape)}, expected {e}")
    p = _lib.Pyramid()
    for j in range(4):
        p.level[j] = lv[j].data_ptr()
    p.height, p.width = h, w
    return p, lv


def regress_batch(reg1, reg2, pyrs1, pyrs2, proposals, want_mid=True, want_raw=False):
    """forward_fine_match for a list of pairs in ONE launch; with reg2 the mid->fine chain runs inside it.

    pyrs1/pyrs2: per pair, the 4 maps of feat_idx [0,1,2,3]; proposals: per pair [n_i,4] int64 or
    float32 (same dtype for all).  Returns a list of dicts 'matches1','probs1' (+ 'matches2','probs2'
    when reg2 is given; 'raw*' on request), views into shared buffers."""
    nitems = len(proposals)
    if nitems == 0:
        return []
    dev = proposals[0].device
    dtype = proposals[0].dtype
    if dtype == torch.int64:
        is_float = 0
    elif dtype == torch.float32:
        is_float = 1
    else:
        raise TypeError("proposals must be int64 or float32")
    if any(p.dtype != dtype for p in proposals):
        raise TypeError("all proposal arrays of a batch must share one dtype")
    counts = [int(p.shape[0]) for p in proposals]
    n = sum(counts)
    allp = proposals[0].contiguous() if nitems == 1 else torch.cat([p.reshape(-1, 4) for p in proposals])
    pyr_a = (_lib.Pyramid * nitems)()
    pyr_b = (_lib.Pyramid * nitems)()
    keep = []
    for i in range(nitems):
        pa, ka = _pyramid(pyrs1[i])
        pb, kb = _pyramid(pyrs2[i])
        pyr_a[i], pyr_b[i] = pa, pb
        keep.append((ka, kb))
    cnt = (ctypes.c_int * nitems)(*counts)
    bufs = {}

    def buf(key, cols, cond=True):
        if not cond:
            return None
        bufs[key] = torch.empty((n, cols) if cols > 1 else (n,), dtype=torch.float32, device=dev)
        return bufs[key].data_ptr()

    two = reg2 is not None
    m1 = buf("matches1", 4, want_mid or not two)
    q1 = buf("probs1", 1, want_mid or not two)
    r1 = buf("raw1", 5, want_raw)
    m2 = buf("matches2", 4, two)
    q2 = buf("probs2", 1, two)
    r2 = buf("raw2", 5, two and want_raw)
    if n:
        with torch.cuda.device(dev):
            ws = _regress_scratch(dev, n, reg1)
            ev = None
            if regress_events is not None:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            _lib.check(_lib.p2p_regress_batch(reg1.handle, reg2.handle if two else None, nitems, pyr_a, pyr_b, cnt,
                                              allp.data_ptr(), is_float, m1, q1, r1, m2, q2, r2, ws.data_ptr(), ws.numel(),
                                              _stream()), "p2p_regress_batch")
            if ev is not None:
                ev[1].record()
                regress_events.append((ev[0], ev[1], n, 2 if two else 1))
    del keep
    outs, start = [], 0
    for c in counts:
        outs.append({k: v[start:start + c] for k, v in bufs.items()})
        start += c
    return outs


def regress(reg1, reg2, pyr1, pyr2, proposals, want_mid=True, want_raw=False):
    """Single-pair form of regress_batch (returns one dict)."""
    return regress_batch(reg1, reg2, [pyr1], [pyr2], [proposals], want_mid, want_raw)[0]
