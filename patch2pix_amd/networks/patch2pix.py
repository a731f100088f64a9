"""`Patch2Pix` with the reference's public surface (networks/patch2pix.py) on top of libp2p_hip.

Same constructor config, attributes (`device`, `upsample`, `psize`, `panc`, ...), method names,
argument meaning and return layouts as the reference class, so callers such as
utils/eval/model_helper.py (and image-matching-toolbox through it) work unchanged.  The ResNet
backbone (networks/resnet.py) keeps its torch parameters and runs its convolutions through the same
library; everything after the feature pyramid is HIP as well.  Inference only:
`config.training=True` raises (training is out of scope).
"""
import numpy as np
import torch
import torch.nn as nn

from . import resnet
from .utils import filter_coarse
from .. import ops


class _Holder(nn.Module):
    """Parameter container that reproduces a sub-tree of the reference state_dict."""

    def __init__(self, spec):
        super().__init__()
        children = {}
        for name, (shape, kind) in spec.items():
            head, _, rest = name.partition(".")
            if rest:
                children.setdefault(head, {})[rest] = (shape, kind)
            elif kind == "param":
                self.register_parameter(head, nn.Parameter(torch.zeros(shape), requires_grad=False))
            else:
                dtype = torch.int64 if kind == "long" else torch.float32
                self.register_buffer(head, torch.zeros(shape, dtype=dtype))
        for head, sub in children.items():
            self.add_module(head, _Holder(sub))


def _bn_spec(prefix, n):
    return {f"{prefix}.weight": ((n,), "param"), f"{prefix}.bias": ((n,), "param"),
            f"{prefix}.running_mean": ((n,), "buffer"), f"{prefix}.running_var": ((n,), "buffer"),
            f"{prefix}.num_batches_tracked": ((), "long")}


def _regressor_spec(feat_dim):
    spec = {"conv.0.weight": ((512, 2 * feat_dim, 3, 3), "param"), "conv.2.weight": ((512, 512, 3, 3), "param"),
            "fc.0.weight": ((512, 512), "param"), "fc.0.bias": ((512,), "param"),
            "fc.3.weight": ((256, 512), "param"), "fc.3.bias": ((256,), "param"),
            "fc.6.weight": ((5, 256), "param"), "fc.6.bias": ((5,), "param")}
    for prefix, n in (("conv.1", 512), ("conv.3", 512), ("fc.1", 512), ("fc.4", 256)):
        spec.update(_bn_spec(prefix, n))
    return spec


_NCN_SPEC = {"conv.0.weight": ((3, 16, 1, 3, 3, 3), "param"), "conv.0.bias": ((16,), "param"),
             "conv.2.weight": ((3, 1, 16, 3, 3, 3), "param"), "conv.2.bias": ((1,), "param")}


class Delta4d:
    """The reference's `delta4d` tuple (max_i, max_j, max_k, max_l), each int64 [B,1,h1,w1,h2,w2]
    (networks/modules.py:24-34), kept packed (one byte per cell) until somebody indexes it."""

    def __init__(self, packed, ksize):
        self.packed = packed            # uint8 [B,h1,w1,h2,w2]
        self.ksize = ksize
        self._planes = None

    def _materialise(self):
        if self._planes is None:
            planes = ops.delta_unpack(self.packed, self.ksize)          # four int64 [B,h1,w1,h2,w2]
            self._planes = tuple(p.unsqueeze(1) for p in planes)
        return self._planes

    def __iter__(self):
        return iter(self._materialise())

    def __len__(self):
        return 4

    def __getitem__(self, i):
        return self._materialise()[i]


class Patch2Pix(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.device = torch.device(config.device)
        if self.device.type != "cuda":
            raise RuntimeError("patch2pix_amd runs the matching path on an MI355X; no CPU fallback exists "
                               f"(config.device={config.device})")
        if getattr(config, "training", False):
            raise NotImplementedError("training is out of scope of the MI355X matching path")
        self.backbone = config.backbone
        if self.backbone != "ResNet34":
            raise NotImplementedError(f"backbone {self.backbone}: only ResNet34 (the released model) is provided")
        self.change_stride = config.change_stride
        self.upsample = 16
        self.feats_downsample = [1, 2, 2, 2, 2]
        feat_dims = [3, 64, 64, 128, 256]
        self.extract = resnet.ResNet34()
        if self.change_stride:
            self.extract.change_stride(target="layer3")
            self.upsample //= 2
            self.feats_downsample[-1] = 1
        else:
            raise NotImplementedError("change_stride=False (upsample 16) is not implemented by the HIP fine stage")
        self.ncn = _Holder(_NCN_SPEC)

        self.regressor_config = config.regressor_config
        self.regress_mid = None
        self.regress_fine = None
        if self.regressor_config:
            rc = self.regressor_config
            self.regr_batch = config.regr_batch
            self.feat_idx = list(config.feat_idx)
            if self.feat_idx != [0, 1, 2, 3]:
                raise NotImplementedError(f"feat_idx {self.feat_idx}: only [0,1,2,3] is implemented")
            rc.feat_dim = sum(feat_dims[i] for i in self.feat_idx)
            if (list(rc.conv_dims), list(rc.conv_kers), list(getattr(rc, "conv_strs", [2, 2])), list(rc.fc_dims),
                    rc.feat_comb) != ([512, 512], [3, 3], [2, 1], [512, 256], "pre"):
                raise NotImplementedError("only the released regressor configuration is implemented")
            self.ptype = ["center", "center"]
            self.psize = rc.psize
            if list(self.psize) != [16, 16]:
                raise NotImplementedError("psize must be [16,16]")
            self.pshift = rc.pshift
            self.panc = rc.panc
            self.shared = rc.shared
            self.regress_mid = _Holder(_regressor_spec(rc.feat_dim))
            self.regress_fine = self.regress_mid if self.shared else _Holder(_regressor_spec(rc.feat_dim))
        self.to(self.device)
        self._packed = None
        self._pinned = {}        # up to 4 tickets in flight per shape
        self._pin_turn = 0
        self.init_weights_(weights_dict=config.weights_dict)
        self.eval()

    # ------------------------------------------------------------------ weights
    def init_weights_(self, weights_dict=None, pretrained=True):
        """Load a reference state_dict (networks/patch2pix.py:98-109).  `extract.layer4.*` and any other
        key this inference-only model does not own is ignored, like the reference's strict=False branch."""
        if weights_dict:
            own = self.state_dict()
            picked = {k: v for k, v in weights_dict.items() if k in own}
            missing = [k for k in own if k not in picked and not k.endswith("num_batches_tracked")]
            if missing and any(not k.startswith("extract.") for k in missing) and any(
                    k.startswith(("ncn.", "regress_")) for k in weights_dict):
                raise KeyError(f"checkpoint lacks hot-path weights: {missing[:5]} ...")
            self.load_state_dict(picked, strict=False)
            if not any(k.startswith(("ncn.", "regress_")) for k in picked):
                import warnings
                warnings.warn("Patch2Pix: the checkpoint holds none of the ncn.* / regress_* weights; the matching path keeps "
                              "its initial (zero) parameters and its matches are meaningless", RuntimeWarning)
        self._packed = None

    def load_state_dict(self, *args, **kwargs):
        self._packed = None
        return super().load_state_dict(*args, **kwargs)

    def _weights(self):
        """Pack (BN folding, MFMA fragment order, transposed-branch filters) once per weight load."""
        if self._packed is None:
            sd = self.state_dict()
            ncn = ops.NcnWeights(sd["ncn.conv.0.weight"], sd["ncn.conv.0.bias"], sd["ncn.conv.2.weight"],
                                 sd["ncn.conv.2.bias"], self.device)
            mid = fine = None
            if self.regress_mid is not None:
                sub = lambda p: {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}
                mid = ops.RegressorWeights(sub("regress_mid."), self.device)
                fine = mid if self.shared else ops.RegressorWeights(sub("regress_fine."), self.device)
            self._packed = (ncn, mid, fine)
        return self._packed

    # ------------------------------------------------------------------ coarse stage
    def forward_coarse_match(self, feat1, feat2, ksize=1):
        ncn = self._weights()[0]
        b, _, h1, w1 = feat1.shape
        _, _, h2, w2 = feat2.shape
        k = max(ksize, 1)
        shape = (b, 1, h1 // k, w1 // k, h2 // k, w2 // k)
        corr4d = torch.empty(shape, dtype=torch.float32, device=feat1.device)
        packed = torch.empty(shape, dtype=torch.uint8, device=feat1.device) if ksize > 1 else None
        ops.coarse_forward_batch(feat1, feat2, ksize, ncn, out_corr=corr4d[:, 0],
                                 out_delta=packed[:, 0] if packed is not None else None)
        delta4d = Delta4d(packed[:, 0], ksize) if ksize > 1 else None
        return corr4d, delta4d

    def forward(self, im1, im2, ksize=1, return_feats=False):
        feat1s, feat2s = self._pyramids(im1, im2)
        feat1, feat2 = feat1s[-1], feat2s[-1]
        corr4d, delta4d = self.forward_coarse_match(feat1, feat2, ksize=ksize)
        if return_feats:
            return corr4d, delta4d, feat1s, feat2s
        return corr4d, delta4d

    def cal_coarse_matches(self, corr4d, delta4d, ksize=1, do_softmax=True, upsample=16, sort=False,
                           center=True, pshift=0):
        if not do_softmax:
            raise NotImplementedError("cal_coarse_matches(do_softmax=False) is not implemented")
        if delta4d is not None and not isinstance(delta4d, Delta4d):
            di, dj, dk, dl = delta4d            # reference-format int64 planes -> packed byte
            s = ((di * ksize + dj) * ksize + dk) * ksize + dl
            delta4d = Delta4d(s[:, 0].to(torch.uint8).contiguous(), ksize)
        nb, _, h1, w1, h2, w2 = corr4d.shape
        matches_, score_ = ops.coarse_matches_batch(corr4d[:, 0], delta4d.packed if delta4d is not None else None,
                                                    ksize, upsample, center)
        if sort:
            order = torch.sort(-score_)[1]
            score_ = torch.gather(score_, 1, order)
            matches_ = torch.gather(matches_, 1, order.unsqueeze(-1).expand(-1, -1, 4))
        return matches_, score_

    def shift_to_anchors(self, matches):
        """patch2pix.py:377-402: panc == 8 replaces each match by its 8 corner-shifted anchors."""
        if self.panc == 1:
            return matches
        s = self.pshift
        tmpl = torch.tensor([[-s, -s, 0, 0], [s, -s, 0, 0], [-s, s, 0, 0], [s, s, 0, 0],
                             [0, 0, -s, -s], [0, 0, s, -s], [0, 0, -s, s], [0, 0, s, s]], device=self.device)
        return [(m.unsqueeze(1) + tmpl).reshape(-1, 4) for m in matches]

    # ------------------------------------------------------------------ fine stage
    def forward_fine_match(self, feats1, feats2, coarse_matches, psize=16, ptype="center", regressor=None):
        """One regressor over every batch item (reference patch2pix.py:186-218).  `regressor` is
        `self.regress_mid` / `self.regress_fine` like in the reference call sites."""
        _, mid_w, fine_w = self._weights()
        w = fine_w if (regressor is self.regress_fine and regressor is not self.regress_mid) else mid_w
        nb = len(coarse_matches)
        outs = ops.regress_batch(w, None, [[f[b] for f in feats1[:4]] for b in range(nb)],
                                 [[f[b] for f in feats2[:4]] for b in range(nb)],
                                 [self._as_proposals(p) for p in coarse_matches])
        return [o["matches1"] for o in outs], [o["probs1"] for o in outs]

    def _as_proposals(self, props):
        props = props.to(self.device)
        if props.dtype not in (torch.int64, torch.float32):
            props = props.float() if props.is_floating_point() else props.long()
        return props

    def _fine_chain(self, feats1, feats2, coarse_matches):
        """mid -> fine for every batch item in one launch each (patch2pix.py:259-272)."""
        _, mid_w, fine_w = self._weights()
        nb = len(coarse_matches)
        outs = ops.regress_batch(mid_w, fine_w, [[f[b] for f in feats1[:4]] for b in range(nb)],
                                 [[f[b] for f in feats2[:4]] for b in range(nb)],
                                 [self._as_proposals(p) for p in coarse_matches])
        return ([o["matches2"] for o in outs], [o["probs2"] for o in outs],
                [o["matches1"] for o in outs], [o["probs1"] for o in outs])

    # ------------------------------------------------------------------ public prediction API
    def predict_coarse(self, im1, im2, ksize=2, ncn_thres=0.0, mutual=False, center=True):
        corr4d, delta4d = self.forward(im1, im2, ksize)
        coarse_matches, match_scores = self.cal_coarse_matches(corr4d, delta4d, ksize=ksize,
                                                               upsample=self.upsample, center=center)
        return filter_coarse(coarse_matches, match_scores, ncn_thres, mutual)

    def coarse_async(self, feats1, feats2, ksize=2):
        """Enqueue the coarse stage of a batch and an asynchronous device-to-host copy of its
        (small) match arrays behind it; returns a ticket for `fine_from_ticket`.  Lets a
        caller enqueue the next batch's coarse stage before it filters the current one on the host,
        so that the host-side filter_coarse (reference networks/utils.py:38-72) never idles the GPU."""
        corr4d, delta4d = self.forward_coarse_match(feats1[-1], feats2[-1], ksize=ksize)
        matches_, score_ = self.cal_coarse_matches(corr4d, delta4d, ksize=ksize, upsample=self.upsample, center=True)
        main = torch.cuda.current_stream(self.device)
        # pinned staging buffers are recycled (allocating pinned memory synchronises with the device)
        key = (tuple(matches_.shape), self._pin_turn)
        self._pin_turn = (self._pin_turn + 1) % 4
        if key not in self._pinned:
            self._pinned[key] = [torch.empty(matches_.shape, dtype=matches_.dtype, pin_memory=True),
                                 torch.empty(score_.shape, dtype=score_.dtype, pin_memory=True), None]
        slot = self._pinned[key]
        if slot[2] is not None and not slot[2].get("consumed", False):
            # a fifth ticket of this shape while the first is still pending: its staging buffers are taken over, the
            # old ticket falls back to its own device-to-host copy when (if) it is consumed
            slot[2]["done"].synchronize()
            slot[2]["stale"] = True
        host_m, host_s = slot[0], slot[1]
        # The copies go to the stream that produced the arrays, in front of whatever the caller enqueues next: the
        # regress launch is one persistent work-group per compute unit, so a copy kernel on a side stream that becomes
        # ready when that launch has started would wait for its end (17 ms at 6400 proposals) -- and the host with it.
        host_m.copy_(matches_, non_blocking=True)
        host_s.copy_(score_, non_blocking=True)
        done = torch.cuda.Event(blocking=True)
        done.record(main)
        ticket = dict(feats1=feats1, feats2=feats2, matches=matches_, scores=score_, host=(host_m, host_s), done=done)
        slot[2] = ticket
        return ticket

    def fine_from_ticket(self, ticket, ncn_thres=0.0, mutual=True, return_all=False, ptmax=None):
        ticket["done"].synchronize()
        host_m, host_s = ticket["host"]
        coarse_matches, match_scores = filter_coarse(ticket["matches"], ticket["scores"], ncn_thres, mutual, ptmax=ptmax,
                                                     host_copy=None if ticket.get("stale") else (host_m.numpy(), host_s.numpy()))
        ticket["consumed"] = True          # its pinned staging slot may be reused
        coarse_matches = self.shift_to_anchors(coarse_matches)
        fine, fine_scores, mid, mid_scores = self._fine_chain(ticket["feats1"], ticket["feats2"], coarse_matches)
        if return_all:
            return fine, fine_scores, mid, mid_scores, coarse_matches
        return fine, fine_scores, coarse_matches

    def predict_fine_device(self, feats1, feats2, ksize=2, ncn_thres=0.0, mutual=True):
        """The whole path without a host round trip: coarse stage -> device-side filter_coarse -> both
        regressors reading the proposal counts from device memory.  Returns padded device tensors
        (fine [B,n,4], fine_scores [B,n], coarse [B,n,4] int64, counts int32 [B]); `unpad` turns them into the lists
        predict_fine returns.  Not available for the training-time options (ptmax, panc > 1) and for images whose pixel
        coordinates do not fit 15 bits -- use predict_fine_from_feats there.  Any number of coarse rows up to 2^20 per
        pair (lists beyond the 8192 rows that fit LDS are sorted through a scratch buffer, csrc/filter.hip)."""
        if self.panc != 1:
            raise NotImplementedError("predict_fine_device: panc > 1 goes through predict_fine_from_feats")
        if max(feats1[0].shape[-2:] + feats2[0].shape[-2:]) >= (1 << 15):
            raise NotImplementedError("predict_fine_device: image sides must be below 32768 pixels")
        _, mid_w, fine_w = self._weights()
        corr4d, delta4d = self.forward_coarse_match(feats1[4], feats2[4], ksize=ksize)
        matches_, score_ = self.cal_coarse_matches(corr4d, delta4d, ksize=ksize, upsample=self.upsample, center=True)
        coarse, _, counts = ops.filter_coarse_batch(matches_, score_, ncn_thres, mutual)
        nb = coarse.shape[0]
        out = ops.regress_batch_dev(mid_w, fine_w, [[f[b] for f in feats1[:4]] for b in range(nb)],
                                    [[f[b] for f in feats2[:4]] for b in range(nb)], coarse, counts)
        return out["matches2"], out["probs2"], coarse, counts

    @staticmethod
    def unpad(fine, fine_scores, coarse, counts):
        """One device-to-host copy of the counts, then per-item views: the (fine, scores, coarse) lists of predict_fine.
        A count of -1 (the device filter met a coordinate outside its packed key) raises: the padded rows of that item
        hold nothing, use predict_fine_from_feats for such inputs."""
        n = counts.cpu().tolist()
        if any(c < 0 for c in n):
            raise RuntimeError("Patch2Pix.unpad: the device-side filter_coarse asked for the host path (count -1: a "
                               "coordinate outside [0, 2^15)); call predict_fine_from_feats / estimate_matches instead")
        return ([fine[b, :c] for b, c in enumerate(n)], [fine_scores[b, :c] for b, c in enumerate(n)],
                [coarse[b, :c] for b, c in enumerate(n)])

    def predict_fine_from_feats(self, feats1, feats2, ksize=2, ncn_thres=0.0, mutual=True, return_all=False,
                                ptmax=None):
        """predict_fine after the backbone: the part of the path that is HIP end to end.
        `ptmax` (opt-in, training semantics of utils.py:55-63) caps/tiles the proposals."""
        ticket = self.coarse_async(feats1, feats2, ksize)
        return self.fine_from_ticket(ticket, ncn_thres, mutual, return_all, ptmax)

    def _pyramids(self, im1, im2):
        """The two pyramids; equally sized image batches go through the backbone as one batch (an image's pyramid does not
        depend on its batch mates: tests/test_gpu_parity.py::test_backbone_batch_and_tile_independence)."""
        if im1.shape == im2.shape and im1.is_cuda:
            feats = self.extract.pyramid(torch.cat([im1, im2]))
            n = im1.shape[0]
            return [f[:n] for f in feats], [f[n:] for f in feats]
        return self.extract.pyramid(im1), self.extract.pyramid(im2)

    def predict_fine(self, im1, im2, ksize=2, ncn_thres=0.0, mutual=True, return_all=False):
        feats1, feats2 = self._pyramids(im1, im2)
        return self.predict_fine_from_feats(feats1, feats2, ksize, ncn_thres, mutual, return_all)

    def refine_matches(self, im1, im2, coarse_matches, io_thres):
        """patch2pix.py:278-318: refine caller-supplied coarse matches (numpy or tensor [N,4])."""
        if len(coarse_matches) == 0:
            return np.empty((0, 4)), np.empty((0,)), np.empty((0, 4))
        if isinstance(coarse_matches, np.ndarray):
            coarse_t = torch.from_numpy(coarse_matches).to(self.device)
        else:
            coarse_t = coarse_matches.to(self.device)
            coarse_matches = coarse_matches.cpu().data.numpy()
        feats1, feats2 = self._pyramids(im1, im2)
        fine, fine_scores, _, _ = self._fine_chain(feats1, feats2, [coarse_t])
        refined = fine[0].cpu().data.numpy()
        scores = fine_scores[0].cpu().data.numpy()
        if io_thres > 0:
            pos = np.where(scores > io_thres)[0]
            if len(pos) > 0:
                coarse_matches, refined, scores = coarse_matches[pos], refined[pos], scores[pos]
        return refined, scores, coarse_matches
