"""One image pair (or a batch of equally sized pairs) as ONE hipGraph launch: pyramid producer (csrc/backbone.hip) ->
coarse stage -> device-side filter_coarse -> both regressors, no host round trip in between.

The reference runs a pair as ~700 kernel launches with two host synchronisations (filter_coarse on the host,
networks/utils.py:38-72; the numpy tail of estimate_matches, utils/eval/model_helper.py:92-109).  Here the device path
of `Patch2Pix.predict_fine_device` is captured once per input shape with `torch.cuda.CUDAGraph` (hipGraph on ROCm:
the C-ABI kernels are launched on torch's current stream, so the capture records them like torch's own) and replayed
per pair: the launch cost of the whole path becomes one hipGraphLaunch, which is what bounds the latency of a single
pair.

    g = GraphedMatcher(net, height=480, width=640)           # captures on first use
    fine, scores, coarse = g(im1, im2)                        # normalised [1,3,H,W] tensors on net.device

Outputs are the lists `predict_fine` returns (batch item -> [n,4] fp32, [n] fp32, [n,4] int64).  Not available for the
training-time options (ptmax, panc > 1): those draw from the host's numpy RNG (networks/utils.py:55-63).
"""
import torch


class GraphedMatcher:
    def __init__(self, net, height, width, batch=1, ksize=2, ncn_thres=0.0, mutual=True, with_backbone=True, height2=None,
                 width2=None):
        if net.panc != 1:
            raise NotImplementedError("GraphedMatcher: panc > 1 (training-time proposals) is not captured")
        self.net, self.ksize, self.ncn_thres, self.mutual = net, ksize, ncn_thres, mutual
        self.with_backbone = with_backbone
        dev = net.device
        h2, w2 = height2 or height, width2 or width
        if with_backbone:
            self.in1 = torch.zeros((batch, 3, height, width), device=dev)
            self.in2 = torch.zeros((batch, 3, h2, w2), device=dev)
        else:       # feature pyramids are the static inputs: shapes from one probe run of the backbone
            with torch.no_grad():
                self.in1 = [torch.zeros_like(f) for f in net.extract.pyramid(torch.zeros((batch, 3, height, width), device=dev))]
                self.in2 = [torch.zeros_like(f) for f in net.extract.pyramid(torch.zeros((batch, 3, h2, w2), device=dev))]
        self.graph = None
        self.out = None

    def _run(self):
        net = self.net
        if self.with_backbone and self.in1.shape == self.in2.shape:
            # both images through the backbone as one batch: half the launches, twice the work-groups per launch
            feats = net.extract.pyramid(torch.cat([self.in1, self.in2]))
            n = self.in1.shape[0]
            f1, f2 = [f[:n] for f in feats], [f[n:] for f in feats]
        elif self.with_backbone:
            f1, f2 = net.extract.pyramid(self.in1), net.extract.pyramid(self.in2)
        else:
            f1, f2 = self.in1, self.in2
        return net.predict_fine_device(f1, f2, ksize=self.ksize, ncn_thres=self.ncn_thres, mutual=self.mutual)

    def capture(self):
        """Warm up on a side stream (kernel attributes, packed weights, workspaces), then record."""
        side = torch.cuda.Stream(device=self.net.device)
        side.wait_stream(torch.cuda.current_stream(self.net.device))
        with torch.no_grad(), torch.cuda.stream(side):
            for _ in range(3):
                self._run()
        torch.cuda.current_stream(self.net.device).wait_stream(side)
        torch.cuda.synchronize(self.net.device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.out = self._run()
        return self

    def replay(self):
        """Replay on the data currently in the static input buffers; returns the padded device tensors
        (fine [B,n,4], scores [B,n], coarse [B,n,4] int64, counts int32 [B]) -- overwritten by the next replay."""
        if self.graph is None:
            self.capture()
        self.graph.replay()
        return self.out

    def load(self, a, b):
        if self.with_backbone:
            self.in1.copy_(a, non_blocking=True)
            self.in2.copy_(b, non_blocking=True)
        else:
            for dst, src in zip(self.in1, a):
                dst.copy_(src, non_blocking=True)
            for dst, src in zip(self.in2, b):
                dst.copy_(src, non_blocking=True)

    def __call__(self, a, b):
        """a, b: normalised image batches [B,3,H,W] (with_backbone) or the two lists of five pyramid levels."""
        self.load(a, b)
        fine, scores, coarse, counts = self.replay()
        if bool((counts < 0).any()):
            raise RuntimeError("GraphedMatcher: a coordinate does not fit the device filter's packed key")
        fine, scores, coarse = self.net.unpad(fine, scores, coarse, counts)
        # the static output buffers are overwritten by the next replay: hand out copies
        return [t.clone() for t in fine], [t.clone() for t in scores], [t.clone() for t in coarse]
