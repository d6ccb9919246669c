"""The drop-in boundary: header compiles as C and C++, both libraries export every declared symbol,
and the ctypes mirror agrees with the header's layout."""
import ctypes as C
import os
import re
import subprocess

from conftest import ROOT
from tracer_amd import abi, device, host

HEADER = os.path.join(ROOT, "include", "tracer_abi.h")
HOOKS_HEADER = os.path.join(ROOT, "include", "tracer_test_hooks.h")


def test_header_compiles_as_c_and_cxx(tmp_path):
    # the static asserts inside the header lock SURVEY.md Appendix A's sizes/offsets
    for cc, std, name in (("gcc", "-std=c11", "t.c"), ("g++", "-std=c++17", "t.cpp")):
        src = tmp_path / name
        src.write_text('#include "tracer_abi.h"\n#include "tracer_test_hooks.h"\nint main(void){return (int)sizeof(trc_scene) * 0;}\n')
        subprocess.check_call([cc, std, "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"),
                               str(src)])


def _declared(prefix_re, header=HEADER):
    text = open(header).read()
    return sorted(set(re.findall(r"\b(" + prefix_re + r"\w+)\s*\(", text)))


def _exported(path):
    out = subprocess.check_output(["nm", "-D", "--defined-only", path], text=True)
    return {line.split()[-1] for line in out.splitlines() if line.strip()}


def test_device_library_exports_every_declared_symbol():
    declared = [s for s in _declared(r"trc_(?!host_)") if s not in ("trc_tile_owner",)]
    assert set(declared) == set(abi.DEVICE_SYMBOLS), set(declared) ^ set(abi.DEVICE_SYMBOLS)
    exported = _exported(device.lib_path())
    missing = [s for s in declared if s not in exported]
    assert not missing, missing
    # the library LOADS without a GPU; only trc_create reports the missing device (no silent fallback)
    L = device.lib()
    assert L.trc_abi_version() == abi.TRC_ABI_VERSION and L.trc_build_flavor() == b"exact"
    # the fast-math build of the same sources exports the same ABI and says what it is
    fast = _exported(device.fast_lib_path())
    assert not [s for s in declared if s not in fast]
    assert device.lib(fast_math=True).trc_build_flavor() == b"fast-math"


def test_test_hooks_are_exported_by_the_hooks_build_only():
    """Product and laboratory apart: the entry points of include/tracer_test_hooks.h -- exhaustive arithmetic
    checks, the SPPM hash, the per-site cycle profile -- exist in libtracer_amd_hooks.so (the same sources + -DTRC_TEST_HOOKS)
    and in NEITHER product library; the hooks build still exports the whole product ABI, so the tests that need a hook run the
    product's own kernels."""
    hooks = _declared(r"trc_", HOOKS_HEADER)
    assert set(hooks) == set(abi.HOOK_SYMBOLS), set(hooks) ^ set(abi.HOOK_SYMBOLS)
    assert not set(hooks) & set(_declared(r"trc_"))                       # declared in one header only
    product, fast, lab = _exported(device.lib_path()), _exported(device.fast_lib_path()), _exported(device.hooks_lib_path())
    assert not [s for s in hooks if s in product or s in fast]
    assert not [s for s in hooks + abi.DEVICE_SYMBOLS if s not in lab]
    # nothing but the declared ABI leaves the product library under the trc_ prefix
    assert {s for s in product if s.startswith("trc_")} == set(abi.DEVICE_SYMBOLS)
    assert device.lib().trc_has_test_hooks() == 0 and device.lib(fast_math=True).trc_has_test_hooks() == 0
    assert device.lib(hooks=True).trc_has_test_hooks() == 1 and device.lib(hooks=True).trc_build_flavor() == b"exact"


def test_host_library_exports_every_declared_symbol():
    declared = _declared(r"trc_host_")
    assert set(declared) == set(abi.HOST_SYMBOLS), set(declared) ^ set(abi.HOST_SYMBOLS)
    exported = _exported(host.lib_path())
    assert not [s for s in declared if s not in exported]


def test_device_library_has_no_oracle_or_cpu_fallback():
    """The product must not link or embed the checker."""
    out = subprocess.check_output(["ldd", device.lib_path()], text=True)
    assert "oracle" not in out
    syms = _exported(device.lib_path())
    assert not [s for s in syms if s.startswith("orc_")]


def test_ctypes_layout_matches_header_offsets():
    off = lambda t, f: getattr(t, f).offset
    assert off(abi.BVH, "pType") == 16 and off(abi.BVH, "pIndex") == 20 and off(abi.BVH, "bBOX") == 32
    assert off(abi.Sphere, "center") == 16 and off(abi.Sphere, "material") == 224 and off(abi.Sphere, "boundingBOX") == 240
    assert off(abi.Square, "range_i") == 8 and off(abi.Square, "axis_k") == 24 and off(abi.Square, "value_k") == 28
    assert off(abi.Square, "model_matrix") == 32 and off(abi.Square, "material") == 224
    assert off(abi.Cube, "box") == 192 and off(abi.Cube, "material") == 224
    assert off(abi.Material, "eta") == 12 and off(abi.Material, "roughness") == 16 and off(abi.Material, "textureInfo") == 32
    assert off(abi.TextureInfo, "albedo") == 16
    assert off(abi.Camera, "vfov") == 48 and off(abi.Camera, "focus_dist") == 64 and off(abi.Camera, "u") == 80
    assert off(abi.Camera, "cornerLowLeft") == 160
    # trc_hit / trc_params / trc_stats: the header static-asserts the same numbers
    assert C.sizeof(abi.Hit) == 80 and off(abi.Hit, "p") == 16 and off(abi.Hit, "uv") == 52 and off(abi.Hit, "n_descend") == 68
    assert C.sizeof(abi.Params) == 32 and C.sizeof(abi.Stats) == 112


def test_trc_create_fails_loudly_without_gpu_when_no_device():
    L = device.lib()
    h = C.c_void_p()
    st = L.trc_create(9999, C.byref(h))      # no such device anywhere
    assert st == abi.ERR_NO_DEVICE and not h.value
