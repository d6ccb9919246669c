"""tracer_amd -- MI355X-native path-tracing hot path of iaomw/Tracer's RT_Metal.

  abi     ctypes mirror of include/tracer_abi.h (PODs in the reference's layouts)
  host    scene assembly + SAH BVH builder (libtrc_host.so, CPU)
  device  the HIP path tracer (libtracer_amd.so, gfx950); no CPU fallback
"""
from . import abi  # noqa: F401
