"""BASELINE configs[2] substitute (the HPatches sequences are not in the container).

A mixed-size batch of real photographs through `estimate_matches_stream`:
  * the reference's three example pairs (tests/golden/images/pair_{1,2,3}) at imsize = 1024 (README.md:33-34 of the
    reference; 1024 x 768 / 768 x 1024 / 1024 x 576 after utils/datasets/preprocess.py:32-60), and
  * seeded homography warps of the six example photographs (image, warp(image)) -- the HPatches protocol: the ground-truth
    homography is known, MMA@3px = the fraction of the returned matches whose first point, mapped by it, lands within
    3 px of the second (image-matching-toolbox's HPatches metric; it is not part of the reference).

`measure(net)` reports pairs/s of the stream over that list, MMA@3px of the warp pairs and, for the photograph pairs, the
agreement with the unmodified reference's recorded output (tests/golden/real_pair_*.npz when the checkpoint matches).
bench.py puts it under `e2e.configs2`; tests/test_gpu_parity.py::test_homography_warp_pairs checks HIP against the CPU oracle.
"""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from PIL import Image  # noqa: E402

IMAGES = os.path.join(ROOT, "tests", "golden", "images")
PHOTO_PAIRS = [("pair_1", None), ("pair_2", 640), ("pair_3", 1024)]        # imsize of the committed reference goldens


def random_homography(rng, w, h, jitter=0.12):
    """Homography that moves the four corners by up to `jitter` of the image extent (seeded): H maps source pixels to
    warped pixels."""
    src = np.array([[0, 0], [w, 0], [w, h], [0, h]], dtype=np.float64)
    dst = src + rng.uniform(-jitter, jitter, (4, 2)) * np.array([w, h])
    A = []
    for (x, y), (u, v) in zip(src, dst):
        A.append([x, y, 1, 0, 0, 0, -u * x, -u * y, -u])
        A.append([0, 0, 0, x, y, 1, -v * x, -v * y, -v])
    _, _, vt = np.linalg.svd(np.array(A))
    H = vt[-1].reshape(3, 3)
    return H / H[2, 2]


def warp_image(img, H):
    """PIL perspective transform: its coefficients map OUTPUT pixels to INPUT pixels, i.e. they are H^-1."""
    Hi = np.linalg.inv(H)
    Hi = Hi / Hi[2, 2]
    return img.transform(img.size, Image.PERSPECTIVE, tuple(Hi.reshape(-1)[:8]), Image.BICUBIC)


def make_warp_pairs(out_dir, seed=0, per_image=1):
    """(path, warped path, H) for every example photograph."""
    rng = np.random.RandomState(seed)
    pairs = []
    for pair in ("pair_1", "pair_2", "pair_3"):
        for name in ("1.jpg", "2.jpg"):
            img = Image.open(os.path.join(IMAGES, pair, name)).convert("RGB")
            for k in range(per_image):
                H = random_homography(rng, *img.size)
                src = os.path.join(out_dir, f"{pair}_{name[0]}_src.png")
                dst = os.path.join(out_dir, f"{pair}_{name[0]}_warp{k}.png")
                img.save(src)
                warp_image(img, H).save(dst)
                pairs.append((src, dst, H))
    return pairs


def mma(matches, H, thr=3.0):
    """Mean matching accuracy: fraction of matches (x1, y1, x2, y2) with |H (x1, y1) - (x2, y2)| < thr pixels."""
    if len(matches) == 0:
        return 0.0
    p = np.concatenate([matches[:, :2], np.ones((len(matches), 1))], axis=1) @ H.T
    err = np.linalg.norm(p[:, :2] / p[:, 2:3] - matches[:, 2:4], axis=1)
    return float((err < thr).mean())


def measure(net, warp_imsize=1024, reps=3, batch=4, workers=4):
    from patch2pix_amd.utils.eval.stream import estimate_matches_stream
    out = {"workload": "BASELINE configs[2] substitute: the reference's 3 example photograph pairs (imsize None / 640 / 1024 as in its "
                       f"goldens) + 6 seeded homography warps of the example photographs (imsize {warp_imsize}), mixed sizes, through "
                       "estimate_matches_stream (io_thres 0.25, ksize 2)"}
    with tempfile.TemporaryDirectory() as td:
        warps = make_warp_pairs(td)
        jobs = []       # (im1, im2, imsize): the stream takes one imsize per call, so the list is streamed per imsize group
        for name, imsize in PHOTO_PAIRS:
            jobs.append((os.path.join(IMAGES, name, "1.jpg"), os.path.join(IMAGES, name, "2.jpg"), imsize))
        for a, b, _ in warps:
            jobs.append((a, b, warp_imsize))
        groups = {}
        for j in jobs:
            groups.setdefault(j[2], []).append(j[:2])

        def run():
            res = {}
            for imsize, plist in groups.items():
                for pr, r in zip(plist, estimate_matches_stream(net, plist, imsize=imsize, batch=batch, workers=workers)):
                    res[pr] = r
            return res

        import torch
        run()                                   # warm-up (backbone packs, workspaces, pinned rings)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            res = run()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out["pairs"] = len(jobs)
        out["pairs_per_s"] = reps * len(jobs) / dt
        out["matches_per_pair"] = [int(res[j[:2]][0].shape[0]) for j in jobs]
        out["mma3_warps"] = [round(mma(res[(a, b)][0], H), 4) for a, b, H in warps]
        out["mma3_warps_mean"] = float(np.mean(out["mma3_warps"]))
        out["mma3_note"] = ("MMA@3px against the ground-truth homography; the checkpoint is a seeded random initialisation (the pretrained "
                            "weights are not available offline), so the value says nothing about matching quality -- parity of the metric "
                            "with the CPU oracle is tests/test_gpu_parity.py::test_homography_warp_pairs")
        agree = {}
        for name, imsize in PHOTO_PAIRS:
            f = os.path.join(ROOT, "tests", "golden", f"real_{name}.npz")
            if not os.path.exists(f):
                continue
            g = np.load(f, allow_pickle=True)
            m, s, c = res[(os.path.join(IMAGES, name, "1.jpg"), os.path.join(IMAGES, name, "2.jpg"))]
            refc = {tuple(np.round(r, 4)): i for i, r in enumerate(g["fine_coarse"])}
            hits = [(i, refc[tuple(np.round(r, 4))]) for i, r in enumerate(c) if tuple(np.round(r, 4)) in refc]
            frac = len(hits) / max(len(g["fine_coarse"]), 1)
            within3 = 0.0
            if hits:
                gi, ri = np.array([h[0] for h in hits]), np.array([h[1] for h in hits])
                within3 = float((np.abs(m[gi] - g["fine_matches"][ri]).max(axis=1) < 3.0).mean())
            agree[name] = {"reference_matches": int(len(g["fine_coarse"])), "reproduced": round(frac, 4),
                           "mma3_style_agreement": round(frac * within3, 4)}
        out["agreement_with_reference_goldens"] = agree
        out["agreement_note"] = ("plain seeded checkpoint (seed 0) = the checkpoint of tests/golden/real_pair_*.npz: near-flat volumes, the "
                                 "near-tie stress case (tests/test_gpu_parity.py::test_real_image_pairs has the contrast checkpoint)")
    return out


if __name__ == "__main__":
    import json
    from patch2pix_amd.utils import synthetic
    from patch2pix_amd.utils.eval import model_helper
    from patch2pix_amd.utils.host import pin_process_to_gpu
    pin_process_to_gpu(0)
    print(json.dumps(measure(model_helper.load_model(synthetic.make_checkpoint(0), lprint=lambda *a: None)), indent=1))
