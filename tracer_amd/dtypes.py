"""numpy structured dtypes of trc_ray / trc_hit (include/tracer_abi.h)."""
import ctypes as C

import numpy as np

from . import abi

RAY_DTYPE = np.dtype([("origin", np.float32, 3), ("tmax", np.float32), ("direction", np.float32, 3),
                      ("_pad", np.uint32)])
HIT_DTYPE = np.dtype([("hit", np.int32), ("pType", np.int32), ("pIndex", np.uint32), ("t", np.float32),
                      ("p", np.float32, 3), ("gn", np.float32, 3), ("sn", np.float32, 3), ("uv", np.float32, 2),
                      ("material", np.uint32), ("PDF", np.float32),
                      ("n_descend", np.uint32), ("n_return", np.uint32), ("n_leaf", np.uint32)])
assert RAY_DTYPE.itemsize == C.sizeof(abi.Ray) and HIT_DTYPE.itemsize == C.sizeof(abi.Hit)


def make_rays(origins, directions, tmax=None):
    rays = np.zeros(len(origins), dtype=RAY_DTYPE)
    rays["origin"] = origins
    rays["direction"] = directions
    rays["tmax"] = np.float32(np.finfo(np.float32).max) if tmax is None else tmax
    return rays


CAMREC_DTYPE = np.dtype([("ratio", np.float32, 4), ("position", np.float32, 4), ("direction", np.float32, 4),
                         ("valid", np.uint8), ("_pad0", np.uint8, 15), ("alternative", np.float32, 4),
                         ("flux", np.float32, 4), ("radius", np.float32), ("photonCount", np.uint32),
                         ("_pad1", np.uint32, 2)])
PHOTON_DTYPE = np.dtype([("flux", np.float32, 4), ("normal", np.float32, 4), ("position", np.float32, 4),
                         ("direction", np.float32, 4), ("step", np.uint8), ("active", np.uint8), ("_pad", np.uint8, 14)])
assert CAMREC_DTYPE.itemsize == C.sizeof(abi.CameraRecord) and PHOTON_DTYPE.itemsize == C.sizeof(abi.PhotonRecord)
