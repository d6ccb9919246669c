"""trc_host_load_density_pbrt against the REFERENCE's own parser: RT_Metal/Tracer/minipbrt.cpp compiled where it lies
into oracle/_ref/libminipbrt_ref.so (oracle/Makefile, oracle/ref_minipbrt_shim.cpp), doing what AAPLRenderer.mm:629-636
does.  On the reference's cloud (cloud/cloud.pbrt -> Include geometry/density_render.70.pbrt, 100 x 100 x 40) when the
reference tree is mounted, and on generated files everywhere the library exists."""
import ctypes as C
import os

import numpy as np
import pytest

from tracer_amd import host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_LIB = os.path.join(ROOT, "oracle", "_ref", "libminipbrt_ref.so")
REF_CLOUD = "/root/reference/RT_Metal/cloud"


def ref_load(path):
    L = C.CDLL(REF_LIB)
    L.ref_minipbrt_load_density.argtypes = [C.c_char_p] + [C.POINTER(C.c_int)] * 3 + [C.POINTER(C.POINTER(C.c_float))]
    L.ref_minipbrt_free.argtypes = [C.c_void_p]
    nx, ny, nz, p = C.c_int(), C.c_int(), C.c_int(), C.POINTER(C.c_float)()
    rc = L.ref_minipbrt_load_density(os.fsencode(path), C.byref(nx), C.byref(ny), C.byref(nz), C.byref(p))
    if rc != 0:
        return None
    try:
        return np.ctypeslib.as_array(p, shape=(nz.value, ny.value, nx.value)).copy()
    finally:
        L.ref_minipbrt_free(p)


needs_ref = pytest.mark.skipif(not os.path.exists(REF_LIB), reason="oracle/_ref/libminipbrt_ref.so not built (reference absent)")


@needs_ref
@pytest.mark.skipif(not os.path.isdir(REF_CLOUD), reason="reference tree not mounted")
@pytest.mark.parametrize("name", ["cloud.pbrt", "geometry/density_render.70.pbrt"])
def test_reference_cloud_reads_like_minipbrt(name):
    path = os.path.join(REF_CLOUD, name)
    ref = ref_load(path)
    mine = host.load_density_pbrt(path)
    assert ref is not None and ref.shape == (40, 100, 100) == mine.shape
    assert np.array_equal(mine.view(np.uint32), ref.view(np.uint32))          # every float, bit for bit
    assert ref.max() == 1.0 and 0.04 < ref.mean() < 0.05


MEDIUM = '''MakeNamedMedium "smoke" "string type" "heterogeneous" "integer nx" {nx} "integer ny" [ {ny} ] "integer nz" {nz}
\t"point p0" [ 0.01 0.01 0.01 ] "point p1" [ 1.99 1.99 0.79 ]   # trailing comment "integer nx" 99
\t"float density" [
{values} ]
'''


def write_scene(tmp_path, values, nx, ny, nz, include=False):
    body = MEDIUM.format(nx=nx, ny=ny, nz=nz, values=values)
    if not include:
        p = tmp_path / "medium.pbrt"
        p.write_text("# a comment first\nWorldBegin\n" + body + "WorldEnd\n")
        return str(p)
    (tmp_path / "geometry").mkdir()
    (tmp_path / "geometry" / "grid.pbrt").write_text(body)
    p = tmp_path / "top.pbrt"
    p.write_text('LookAt 0 0 5  0 0 0  0 1 0\nCamera "perspective" "float fov" [15]\nWorldBegin\n'
                 '#Include "geometry/missing.pbrt"\nTransformBegin\n\tInclude "geometry/grid.pbrt"\n'
                 '\t  "color sigma_a" [10 10 10] "color sigma_s" [90 90 90]\nTransformEnd\nWorldEnd\n')
    return str(p)


@pytest.mark.parametrize("include", [False, True])
def test_generated_files(tmp_path, include):
    nx, ny, nz = 5, 4, 3
    tokens = ["0", "1", ".5", "-0", "+2.5e+1", "1e-3", "3.", "0.1", "7.0E2", "1e-45", "0.30000001192092896", "16777217"]
    vals = [tokens[i % len(tokens)] for i in range(nx * ny * nz)]
    text = "\n".join(" ".join(vals[r * nx:(r + 1) * nx]) for r in range(ny * nz))
    path = write_scene(tmp_path, text, nx, ny, nz, include)
    mine = host.load_density_pbrt(path)
    assert mine.shape == (nz, ny, nx)
    want = np.array([np.float32(float(t)) for t in vals], dtype=np.float32).reshape(nz, ny, nx)
    assert np.array_equal(mine.view(np.uint32), want.view(np.uint32))
    if os.path.exists(REF_LIB):
        ref = ref_load(path)
        assert ref is not None and np.array_equal(mine.view(np.uint32), ref.view(np.uint32))


def test_reader_errors(tmp_path):
    p = tmp_path / "short.pbrt"
    p.write_text(MEDIUM.format(nx=2, ny=2, nz=2, values="1 2 3"))                  # 3 of 8 values
    with pytest.raises(Exception):
        host.load_density_pbrt(str(p))
    q = tmp_path / "loop.pbrt"
    q.write_text('Include "loop.pbrt"\n')                                          # include cycle
    with pytest.raises(Exception):
        host.load_density_pbrt(str(q))
    with pytest.raises(Exception):
        host.load_density_pbrt(str(tmp_path / "absent.pbrt"))
