"""TEST INFRASTRUCTURE: load tests/hipemu/_build/libp2p_emu.so (the kernel sources compiled for the host, see
build_emu.py) with the prototypes of the real binding, plus small helpers that call it on CPU tensors."""
import ctypes
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import build_emu  # noqa: E402

from patch2pix_amd import _lib as real  # noqa: E402  (prototypes only; nothing of the real library is called)


def load():
    emu = ctypes.CDLL(build_emu.build())
    for name in real.EXPORTS:
        src, dst = getattr(real.lib, name), getattr(emu, name)
        dst.argtypes, dst.restype = src.argtypes, src.restype
    return emu


def check(emu, status, what):
    if status != 0:
        raise RuntimeError(f"{what}: {emu.p2p_last_error().decode()}")


def ptr(t):
    assert t is None or (t.device.type == "cpu" and t.is_contiguous())
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def ncn_create(emu, sd):
    keep = [sd[k].detach().float().contiguous() for k in
            ("ncn.conv.0.weight", "ncn.conv.0.bias", "ncn.conv.2.weight", "ncn.conv.2.bias")]
    h = ctypes.c_void_p()
    check(emu, emu.p2p_ncn_create(*[t.data_ptr() for t in keep], ctypes.byref(h)), "p2p_ncn_create")
    return h


def coarse_forward_batch(emu, ncn, fa, fb, ksize, ws_pairs=None):
    """fa, fb: [B,C,h,w] fp32 CPU -> (corr [B,hA',wA',hB',wB'], delta uint8 or None)."""
    fa, fb = fa.contiguous(), fb.contiguous()
    nb, c, ha, wa = fa.shape
    _, _, hb, wb = fb.shape
    k = max(ksize, 1)
    shape = (nb, ha // k, wa // k, hb // k, wb // k)
    corr = torch.empty(shape, dtype=torch.float32)
    delta = torch.empty(shape, dtype=torch.uint8) if ksize > 1 else None
    per_pair = emu.p2p_coarse_workspace_bytes(c, ha, wa, hb, wb, ksize)
    ws = torch.empty((ws_pairs or nb) * per_pair + 256, dtype=torch.uint8)
    base = (ws.data_ptr() + 255) & ~255
    check(emu, emu.p2p_coarse_forward_batch(ptr(fa), ptr(fb), nb, c, ha, wa, hb, wb, ksize, ncn, ptr(corr), ptr(delta),
                                            ctypes.c_void_p(base), (ws_pairs or nb) * per_pair, None),
          "p2p_coarse_forward_batch")
    return corr, delta


def neigh_consensus_batch(emu, ncn, x):
    """x [B,hA,wA,hB,wB] fp32 CPU -> NeighConsensus.forward (the fused kernel)."""
    x = x.contiguous()
    y = torch.empty_like(x)
    nb, ha, wa, hb, wb = x.shape
    ws = torch.zeros(nb, dtype=torch.int32)
    check(emu, emu.p2p_neigh_consensus_batch(ptr(x), nb, ha, wa, hb, wb, ncn, ptr(y), ptr(ws), nb * 4, None), "p2p_neigh_consensus_batch")
    return y


def coarse_matches_batch(emu, corr, delta, ksize, upsample, center=True):
    nb, ha, wa, hb, wb = corr.shape
    n = ha * wa + hb * wb
    m = torch.empty((nb, n, 4), dtype=torch.int64)
    s = torch.empty((nb, n), dtype=torch.float32)
    check(emu, emu.p2p_coarse_matches_batch(ptr(corr), ptr(delta), nb, ha, wa, hb, wb, ksize, upsample, int(center),
                                            ptr(m), ptr(s), None), "p2p_coarse_matches_batch")
    return m, s


def regressor_create(emu, sd, mode):
    """sd: sub-state_dict of one FeatRegressNet ('conv.0.weight', ...); mode 'f32' | 'fp16x2'."""
    keep = {k: v.detach().float().contiguous() for k, v in sd.items() if v.is_floating_point()}
    p = real.RegressorParams()

    def bn(prefix):
        return real.BnParams(keep[prefix + ".weight"].data_ptr(), keep[prefix + ".bias"].data_ptr(),
                             keep[prefix + ".running_mean"].data_ptr(), keep[prefix + ".running_var"].data_ptr())

    p.conv1_w = keep["conv.0.weight"].data_ptr(); p.bn1 = bn("conv.1")
    p.conv2_w = keep["conv.2.weight"].data_ptr(); p.bn2 = bn("conv.3")
    p.fc1_w = keep["fc.0.weight"].data_ptr(); p.fc1_b = keep["fc.0.bias"].data_ptr(); p.bnf1 = bn("fc.1")
    p.fc2_w = keep["fc.3.weight"].data_ptr(); p.fc2_b = keep["fc.3.bias"].data_ptr(); p.bnf2 = bn("fc.4")
    p.fc3_w = keep["fc.6.weight"].data_ptr(); p.fc3_b = keep["fc.6.bias"].data_ptr()
    h = ctypes.c_void_p()
    check(emu, emu.p2p_regressor_create(ctypes.byref(p), ctypes.byref(h)), "p2p_regressor_create")
    check(emu, emu.p2p_regressor_set_mode(h, real.REGRESS_MODES[mode]), "p2p_regressor_set_mode")
    return h


def regress_scratch(emu, n):
    """(keep-alive tensor, 128-byte aligned address, bytes) of the scratch p2p_regress* needs for n proposal slots."""
    need = emu.p2p_regress_workspace_bytes(int(n))
    ws = torch.empty(need + 128, dtype=torch.uint8)
    return ws, ctypes.c_void_p((ws.data_ptr() + 127) & ~127), need


def regress(emu, reg1, reg2, pyr1, pyr2, proposals):
    """One pair through p2p_regress_batch: pyr*: the 4 maps of feat_idx [0,1,2,3] (CPU fp32), proposals [n,4]
    int64 or float32.  Returns dict matches1/probs1/raw1 (+ *2 with reg2)."""
    def pyramid(levels):
        lv = [t.contiguous() for t in levels]
        q = real.Pyramid()
        for j in range(4):
            q.level[j] = lv[j].data_ptr()
        q.height, q.width = lv[0].shape[-2:]
        return q, lv
    pa, ka = pyramid(pyr1)
    pb, kb = pyramid(pyr2)
    n = proposals.shape[0]
    proposals = proposals.contiguous()
    out = {k: torch.empty((n, c) if c > 1 else (n,), dtype=torch.float32)
           for k, c in (("matches1", 4), ("probs1", 1), ("raw1", 5), ("matches2", 4), ("probs2", 1), ("raw2", 5))}
    two = reg2 is not None
    arr_a, arr_b = (real.Pyramid * 1)(pa), (real.Pyramid * 1)(pb)
    cnt = (ctypes.c_int * 1)(n)
    ws, wsp, wsn = regress_scratch(emu, n)
    check(emu, emu.p2p_regress_batch(reg1, reg2 if two else None, 1, arr_a, arr_b, cnt, proposals.data_ptr(),
                                     int(proposals.is_floating_point()), out["matches1"].data_ptr(),
                                     out["probs1"].data_ptr(), out["raw1"].data_ptr(),
                                     out["matches2"].data_ptr() if two else None, out["probs2"].data_ptr() if two else None,
                                     out["raw2"].data_ptr() if two else None, wsp, wsn, None), "p2p_regress_batch")
    del ka, kb
    return out


def filter_coarse_batch(emu, matches, scores, thres, mutual):
    """matches [B,n,4] int64, scores [B,n] fp32 (CPU) -> list of (rows, scores) per item, or None where the kernel
    asked for the host fallback."""
    matches, scores = matches.contiguous(), scores.contiguous()
    nb, n, _ = matches.shape
    om, osc = torch.empty_like(matches), torch.empty_like(scores)
    cnt = torch.empty(nb, dtype=torch.int32)
    emu.p2p_filter_coarse_workspace_bytes.restype = ctypes.c_size_t
    need = emu.p2p_filter_coarse_workspace_bytes(nb, n)
    ws = torch.empty(max(need, 1), dtype=torch.uint8)
    check(emu, emu.p2p_filter_coarse_batch(ptr(matches), ptr(scores), nb, n, ctypes.c_float(thres), int(mutual), ptr(om), ptr(osc),
                                           ptr(cnt), ptr(ws) if need else None, ctypes.c_size_t(need), None),
          "p2p_filter_coarse_batch")
    out = []
    for b in range(nb):
        c = int(cnt[b])
        out.append(None if c < 0 else (om[b, :c].clone(), osc[b, :c].clone()))
    return out


def match_tail_batch(emu, fine, scores, coarse, counts, scale, io_thres):
    """p2p_match_tail_batch on CPU tensors -> list of (matches f64, scores f32, coarse f64) per item (None for count -1)."""
    fine, scores, coarse = fine.contiguous(), scores.contiguous(), coarse.contiguous()
    counts, scale = counts.to(torch.int32).contiguous(), scale.to(torch.float64).contiguous()
    nb, n, _ = fine.shape
    om, oc = torch.empty((nb, n, 4), dtype=torch.float64), torch.empty((nb, n, 4), dtype=torch.float64)
    osc, on = torch.empty((nb, n), dtype=torch.float32), torch.empty((nb,), dtype=torch.int32)
    check(emu, emu.p2p_match_tail_batch(ptr(fine), ptr(scores), ptr(coarse), ptr(counts), ptr(scale), nb, n, float(io_thres),
                                        ptr(om), ptr(osc), ptr(oc), ptr(on), None), "p2p_match_tail_batch")
    return [None if int(on[b]) < 0 else (om[b, :int(on[b])].clone(), osc[b, :int(on[b])].clone(), oc[b, :int(on[b])].clone())
            for b in range(nb)]


def _bn(bn):
    return real.BnParams(*[t.data_ptr() for t in bn])


def conv_bn(emu, weight, bn, stride, x_nchw, residual_nchw=None, relu=True, tile=None):
    """p2p_conv_create + p2p_absmax_batch + p2p_conv_forward on CPU tensors (NCHW in and out; the kernel works on NHWC).
    -> (y [n,co,ho,wo], float max |y| per image as the kernel reports it)."""
    co, ci, ks, _ = weight.shape
    keep = [weight.contiguous()] + [t.contiguous() for t in bn]
    b = _bn(keep[1:])
    h = ctypes.c_void_p()
    check(emu, emu.p2p_conv_create(ptr(keep[0]), ctypes.byref(b), ci, co, ks, stride, ctypes.byref(h)), "p2p_conv_create")
    if tile:
        check(emu, emu.p2p_conv_set_tile(h, *tile), "p2p_conv_set_tile")
    n, _, hh, ww = x_nchw.shape
    x = x_nchw.permute(0, 2, 3, 1).contiguous()
    xmax = torch.zeros(n, dtype=torch.int32)
    check(emu, emu.p2p_absmax_batch(ptr(x), hh * ww * ci, n, ptr(xmax), None), "p2p_absmax_batch")
    pad = ks // 2
    ho, wo = (hh + 2 * pad - ks) // stride + 1, (ww + 2 * pad - ks) // stride + 1
    y = torch.empty(n, ho, wo, co)
    ymax = torch.zeros(n, dtype=torch.int32)
    res = residual_nchw.permute(0, 2, 3, 1).contiguous() if residual_nchw is not None else None
    check(emu, emu.p2p_conv_forward(h, ptr(x), ptr(xmax), n, hh, ww, ptr(res), int(relu), ptr(y), ptr(ymax), None), "p2p_conv_forward")
    emu.p2p_conv_destroy(h)
    return y.permute(0, 3, 1, 2).contiguous(), ymax.view(torch.float32)


def stem_pool(emu, weight, bn, image):
    """p2p_stem_forward, p2p_maxpool_nhwc, p2p_nhwc_to_nchw -> (level 1 NCHW, pooled NHWC, pooled NCHW, pooled max)."""
    keep = [weight.contiguous()] + [t.contiguous() for t in bn]
    b = _bn(keep[1:])
    h = ctypes.c_void_p()
    check(emu, emu.p2p_stem_create(ptr(keep[0]), ctypes.byref(b), ctypes.byref(h)), "p2p_stem_create")
    image = image.contiguous()
    n, _, hh, ww = image.shape
    imax = torch.zeros(n, dtype=torch.int32)
    check(emu, emu.p2p_absmax_batch(ptr(image), 3 * hh * ww, n, ptr(imax), None), "p2p_absmax_batch")
    ho, wo = (hh - 1) // 2 + 1, (ww - 1) // 2 + 1
    y = torch.empty(n, 64, ho, wo)
    check(emu, emu.p2p_stem_forward(h, ptr(image), ptr(imax), n, hh, ww, ptr(y), None), "p2p_stem_forward")
    emu.p2p_stem_destroy(h)
    hp, wp = (ho - 1) // 2 + 1, (wo - 1) // 2 + 1
    pooled = torch.empty(n, hp, wp, 64)
    pmax = torch.zeros(n, dtype=torch.int32)
    check(emu, emu.p2p_maxpool_nhwc(ptr(y), n, 64, ho, wo, ptr(pooled), ptr(pmax), None), "p2p_maxpool_nhwc")
    back = torch.empty(n, 64, hp, wp)
    check(emu, emu.p2p_nhwc_to_nchw(ptr(pooled), n, hp, wp, 64, ptr(back), None), "p2p_nhwc_to_nchw")
    return y, pooled, back, pmax.view(torch.float32)
