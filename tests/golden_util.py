"""Helpers shared by the golden-vector tests: regenerate the seeded inputs a fixture was made
from (oracle/make_golden.py) and verify the recorded input checksum."""
import os

import numpy as np

from patch2pix_amd.utils import synthetic

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_sd_cache = {}


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def checksum(tensors):
    return float(sum(t.double().abs().sum().item() for t in tensors))


def state_dict(seed=0):
    if seed not in _sd_cache:
        sd = synthetic.make_state_dict(seed)
        g = load("weights_checksum")
        if int(g["sd_seed"]) == seed:
            got = checksum([v for v in sd.values() if v.is_floating_point()])
            assert abs(got - float(g["checksum"])) <= 1e-9 * float(g["checksum"]), "synthetic weights drifted"
        _sd_cache[seed] = sd
    return _sd_cache[seed]


def coarse_inputs(g):
    p1, p2 = synthetic.make_correlated_pyramids(int(g["seed"]), int(g["H"]), int(g["W"]))
    assert abs(checksum([p1[4], p2[4]]) - float(g["input_checksum"])) < 1e-6 * float(g["input_checksum"])
    return p1, p2


def fine_inputs(g):
    p1 = synthetic.make_pyramid(int(g["seed"]), int(g["H"]), int(g["W"]))
    p2 = synthetic.make_pyramid(int(g["seed"]) + 1, int(g["H"]), int(g["W"]))
    assert abs(checksum(p1 + p2) - float(g["input_checksum"])) < 1e-6 * float(g["input_checksum"])
    return p1, p2


def pair_inputs(g):
    p1, p2 = synthetic.make_correlated_pyramids(int(g["seed"]), int(g["H"]), int(g["W"]))
    assert abs(checksum(p1 + p2) - float(g["input_checksum"])) < 1e-6 * float(g["input_checksum"])
    return p1, p2


COARSE_CASES = ["coarse_64x96_k2", "coarse_96x64_k2", "coarse_48x64_k1", "coarse_128x160_k2"]
FINE_CASES = ["fine_48x64", "fine_96x128"]
PAIR_CASES = ["predict_fine_128x160", "predict_fine_192x256"]
