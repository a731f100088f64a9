"""ctypes binding of libp2p_hip.so (the C ABI declared in include/p2p_hip.h).

The product path has no CPU fallback: if the shared library is missing, importing this module
raises, and every wrapper raises RuntimeError carrying p2p_last_error() on a non-zero status.
"""
import ctypes
import os

import torch  # noqa: F401  (must be loaded first so libamdhip64.so.7 resolves to torch's copy)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libp2p_hip.so")
# Kernel experiments (tools/ab_variants.sh) load another build through P2P_LIB_PATH -- only together with
# P2P_ALLOW_EXPERIMENT=1, so that a stray variable in a user's environment can never swap the library.
_ALLOW_EXPERIMENT = os.environ.get("P2P_ALLOW_EXPERIMENT") == "1"
if os.environ.get("P2P_LIB_PATH"):
    if not _ALLOW_EXPERIMENT:
        raise ImportError("P2P_LIB_PATH is set but P2P_ALLOW_EXPERIMENT=1 is not: refusing to load a library other than "
                          f"{LIB_PATH}")
    LIB_PATH = os.environ["P2P_LIB_PATH"]

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build the HIP extension first (python -m patch2pix_amd.build, "
        "or __graft_entry__.build()). There is no CPU fallback for the matching path.")

lib = ctypes.CDLL(LIB_PATH)

c_float_p = ctypes.c_void_p   # raw addresses (torch .data_ptr()) are passed as void*
c_stream = ctypes.c_void_p


class BnParams(ctypes.Structure):
    _fields_ = [("weight", ctypes.c_void_p), ("bias", ctypes.c_void_p),
                ("running_mean", ctypes.c_void_p), ("running_var", ctypes.c_void_p)]


class RegressorParams(ctypes.Structure):
    _fields_ = [("conv1_w", ctypes.c_void_p), ("bn1", BnParams),
                ("conv2_w", ctypes.c_void_p), ("bn2", BnParams),
                ("fc1_w", ctypes.c_void_p), ("fc1_b", ctypes.c_void_p), ("bnf1", BnParams),
                ("fc2_w", ctypes.c_void_p), ("fc2_b", ctypes.c_void_p), ("bnf2", BnParams),
                ("fc3_w", ctypes.c_void_p), ("fc3_b", ctypes.c_void_p)]


class Pyramid(ctypes.Structure):
    _fields_ = [("level", ctypes.c_void_p * 4), ("height", ctypes.c_int), ("width", ctypes.c_int)]


def _sig(name, restype, argtypes):
    fn = getattr(lib, name)
    fn.restype = restype
    fn.argtypes = argtypes
    return fn


p2p_version = _sig("p2p_version", ctypes.c_int, [])
VERSION_EXPERIMENT = 0x40000000     # csrc/p2p_common.h: a build with timing-experiment switches (wrong results by design)
if p2p_version() & VERSION_EXPERIMENT and not _ALLOW_EXPERIMENT:
    raise ImportError(f"{LIB_PATH} is an experiment build (-DP2P_EXPERIMENT); set P2P_ALLOW_EXPERIMENT=1 to load it")
p2p_last_error = _sig("p2p_last_error", ctypes.c_char_p, [])
p2p_ncn_create = _sig("p2p_ncn_create", ctypes.c_int,
                      [ctypes.c_void_p] * 4 + [ctypes.POINTER(ctypes.c_void_p)])
p2p_ncn_destroy = _sig("p2p_ncn_destroy", None, [ctypes.c_void_p])
p2p_ncn_set_tile = _sig("p2p_ncn_set_tile", ctypes.c_int, [ctypes.c_void_p] + [ctypes.c_int] * 3)
p2p_regressor_create = _sig("p2p_regressor_create", ctypes.c_int,
                            [ctypes.POINTER(RegressorParams), ctypes.POINTER(ctypes.c_void_p)])
p2p_regressor_destroy = _sig("p2p_regressor_destroy", None, [ctypes.c_void_p])
p2p_coarse_workspace_bytes = _sig("p2p_coarse_workspace_bytes", ctypes.c_size_t, [ctypes.c_int] * 6)
p2p_coarse_forward = _sig("p2p_coarse_forward", ctypes.c_int,
                          [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 6 +
                          [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, c_stream])
p2p_coarse_forward_batch = _sig("p2p_coarse_forward_batch", ctypes.c_int,
                                [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 7 +
                                [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, c_stream])
p2p_neigh_consensus_batch = _sig("p2p_neigh_consensus_batch", ctypes.c_int,
                                 [ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                                           ctypes.c_size_t, c_stream])
p2p_delta_unpack = _sig("p2p_delta_unpack", ctypes.c_int,
                        [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, c_stream])
p2p_coarse_matches = _sig("p2p_coarse_matches", ctypes.c_int,
                          [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 7 +
                          [ctypes.c_void_p, ctypes.c_void_p, c_stream])
p2p_coarse_matches_batch = _sig("p2p_coarse_matches_batch", ctypes.c_int,
                                [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 8 +
                                [ctypes.c_void_p, ctypes.c_void_p, c_stream])
p2p_filter_coarse_workspace_bytes = _sig("p2p_filter_coarse_workspace_bytes", ctypes.c_size_t, [ctypes.c_int, ctypes.c_int])
p2p_filter_coarse_batch = _sig("p2p_filter_coarse_batch", ctypes.c_int,
                               [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int,
                                ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, c_stream])
p2p_match_tail_batch = _sig("p2p_match_tail_batch", ctypes.c_int,
                            [ctypes.c_void_p] * 5 + [ctypes.c_int, ctypes.c_int, ctypes.c_float] + [ctypes.c_void_p] * 4 + [c_stream])
p2p_regress = _sig("p2p_regress", ctypes.c_int,
                   [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(Pyramid), ctypes.POINTER(Pyramid),
                    ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 6 + [ctypes.c_void_p, ctypes.c_size_t, c_stream])
p2p_regress_workspace_bytes = _sig("p2p_regress_workspace_bytes", ctypes.c_size_t, [ctypes.c_int])
p2p_regress_workspace_bytes_mode = _sig("p2p_regress_workspace_bytes_mode", ctypes.c_size_t, [ctypes.c_int, ctypes.c_int])

p2p_regress_batch = _sig("p2p_regress_batch", ctypes.c_int,
                         [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(Pyramid), ctypes.POINTER(Pyramid),
                          ctypes.POINTER(ctypes.c_int), ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 6 +
                         [ctypes.c_void_p, ctypes.c_size_t, c_stream])

p2p_regress_batch_dev = _sig("p2p_regress_batch_dev", ctypes.c_int,
                             [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(Pyramid), ctypes.POINTER(Pyramid),
                              ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 6 +
                             [ctypes.c_void_p, ctypes.c_size_t, c_stream])

p2p_conv_create = _sig("p2p_conv_create", ctypes.c_int,
                       [ctypes.c_void_p, ctypes.POINTER(BnParams)] + [ctypes.c_int] * 4 + [ctypes.POINTER(ctypes.c_void_p)])
p2p_conv_destroy = _sig("p2p_conv_destroy", None, [ctypes.c_void_p])
p2p_conv_set_tile = _sig("p2p_conv_set_tile", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int])
p2p_conv_forward = _sig("p2p_conv_forward", ctypes.c_int,
                        [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 3 +
                        [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, c_stream])
p2p_stem_create = _sig("p2p_stem_create", ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(BnParams), ctypes.POINTER(ctypes.c_void_p)])
p2p_stem_destroy = _sig("p2p_stem_destroy", None, [ctypes.c_void_p])
p2p_stem_forward = _sig("p2p_stem_forward", ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p, c_stream])
p2p_maxpool_nhwc = _sig("p2p_maxpool_nhwc", ctypes.c_int, [ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_void_p, c_stream])
p2p_nhwc_to_nchw = _sig("p2p_nhwc_to_nchw", ctypes.c_int, [ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p, c_stream])
p2p_absmax_batch = _sig("p2p_absmax_batch", ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, c_stream])

p2p_regressor_set_mode = _sig("p2p_regressor_set_mode", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int])
p2p_regressor_get_mode = _sig("p2p_regressor_get_mode", ctypes.c_int, [ctypes.c_void_p])
REGRESS_MODES = {"f32": 0, "fp16x2": 3, "fp16x2w": 4}

EXPORTS = ["p2p_version", "p2p_last_error", "p2p_ncn_create", "p2p_ncn_destroy", "p2p_ncn_set_tile", "p2p_regressor_create",
           "p2p_regressor_destroy", "p2p_coarse_workspace_bytes", "p2p_coarse_forward", "p2p_coarse_forward_batch",
           "p2p_neigh_consensus_batch", "p2p_delta_unpack", "p2p_coarse_matches", "p2p_coarse_matches_batch", "p2p_filter_coarse_workspace_bytes", "p2p_filter_coarse_batch", "p2p_match_tail_batch", "p2p_regress", "p2p_regress_workspace_bytes", "p2p_regress_workspace_bytes_mode", "p2p_regress_batch", "p2p_regress_batch_dev", "p2p_regressor_set_mode",
           "p2p_regressor_get_mode", "p2p_conv_create", "p2p_conv_destroy", "p2p_conv_set_tile", "p2p_conv_forward", "p2p_absmax_batch", "p2p_stem_create", "p2p_stem_destroy", "p2p_stem_forward",
           "p2p_maxpool_nhwc", "p2p_nhwc_to_nchw"]


def check(status, what):
    if status != 0:
        msg = p2p_last_error().decode("utf-8", "replace")
        if status == -3:
            raise NotImplementedError(f"{what}: {msg}")
        raise RuntimeError(f"{what} failed with status {status}: {msg}")
