"""Where the time of estimate_matches_stream goes: loader pool throughput by worker count, the main thread's stages of one
batch timed one after the other (synchronised), and the stream itself by (batch, workers)."""
import os
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from PIL import Image  # noqa: E402

from patch2pix_amd.utils import synthetic  # noqa: E402
from patch2pix_amd.utils.eval import model_helper, stream  # noqa: E402


def main(H=480, W=640, B=8, streams=True):
    net = model_helper.load_model(synthetic.make_checkpoint(0), lprint=lambda *a: None)
    out = {}
    with tempfile.TemporaryDirectory() as td, torch.no_grad():
        paths = []
        for i in range(8):
            a, b = synthetic.make_image_pair(100 + i, H, W)
            pa, pb = os.path.join(td, f"{i}a.jpg"), os.path.join(td, f"{i}b.jpg")
            Image.fromarray(a).save(pa, quality=95)
            Image.fromarray(b).save(pb, quality=95)
            paths.append((pa, pb))
        jobs = [(i, a, b, 2, net.upsample, None) for i, (a, b) in enumerate(paths * 16)]
        t0 = time.perf_counter()
        for j in jobs[:8]:
            stream._load(j)
        out["load_ms_per_pair_one_thread"] = (time.perf_counter() - t0) / 8 * 1e3
        for w in (2, 4, 8, 16):
            with ThreadPoolExecutor(max_workers=w) as pool:
                t0 = time.perf_counter()
                n = sum(1 for _ in pool.map(stream._load, jobs))
                out[f"loader_pairs_per_s_{w}_threads"] = n / (time.perf_counter() - t0)
        group = [stream._load(j) for j in jobs[:B]]

        def stage(fn, reps=5):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                r = fn()
                torch.cuda.synchronize()
            return (time.perf_counter() - t0) / reps * 1e3, r
        t_up, ims = stage(lambda: torch.cat([stream._upload([g[1] for g in group], net.device), stream._upload([g[2] for g in group], net.device)]))
        t_bb, feats = stage(lambda: net.extract.pyramid(ims))
        f1, f2 = [f[:B] for f in feats], [f[B:] for f in feats]
        from patch2pix_amd import ops
        import numpy as np
        t_co, _ = stage(lambda: net.cal_coarse_matches(*net.forward_coarse_match(f1[4], f2[4], ksize=2), ksize=2, upsample=net.upsample, center=True))
        t_dev, res = stage(lambda: net.predict_fine_device(f1, f2, ksize=2))
        scale = np.concatenate([g[3] for g in group])
        t_tail, _ = stage(lambda: [o.cpu() for o in ops.match_tail_batch(*res, scale, 0.25)])
        out.update({"batch_pairs": B, "proposals_per_pair": float(res[3].float().mean().item()),
                    "upload_normalise_ms": t_up, "backbone_ms": t_bb, "coarse_stage_ms": t_co,
                    "filter_and_fine_stage_ms": t_dev - t_co, "tail_and_copy_back_ms": t_tail,
                    "sum_ms": t_up + t_bb + t_dev + t_tail, "serial_pairs_per_s": B / (t_up + t_bb + t_dev + t_tail) * 1e3})
        work = paths * 20
        for b, w in () if not streams else ((8, 2), (8, 4), (16, 4), (32, 4)):
            list(stream.estimate_matches_stream(net, work[:5 * b], batch=b, workers=w))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = sum(1 for _ in stream.estimate_matches_stream(net, work, batch=b, workers=w))
            torch.cuda.synchronize()
            out[f"stream_pairs_per_s_batch{b}_workers{w}"] = n / (time.perf_counter() - t0)
    return out


if __name__ == "__main__":
    import json
    print(json.dumps(main(), indent=1))
