"""The two file readers scene ingestion gained in round 3.

PLY (`Shape "plymesh"`): trc_host_mesh_load_ply against the REFERENCE's own parser -- minipbrt's PLYMesh::triangle_mesh()
(RT_Metal/Tracer/minipbrt.cpp:4380-4450, compiled where it lies into oracle/_ref/libminipbrt_ref.so): positions, normals,
uv and triangle indices field for field, for ascii / binary_little_endian / binary_big_endian files with extra
properties, quads (minipbrt's split: (0 1 3) (2 3 1)) and a second list property.

Radiance RGBE (.hdr): trc_host_load_hdr against the published decoding (mantissa * 2^(e - 136), e = 0 -> 0) on files this
test encodes itself, flat and run-length encoded, rows bottom-up like the reference's vertically flipped backdrop texture
(AAPLRenderer.mm:352-383)."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

from tracer_amd import abi, host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_LIB = os.path.join(ROOT, "oracle", "_ref", "libminipbrt_ref.so")
needs_ref = pytest.mark.skipif(not os.path.exists(REF_LIB), reason="oracle/_ref/libminipbrt_ref.so not built (reference absent)")


def make_mesh(seed, n_side=5):
    """a bumpy grid: vertices with normals + uv, faces = quads and triangles mixed"""
    rs = np.random.RandomState(seed)
    xs, ys = np.meshgrid(np.arange(n_side, dtype=np.float32), np.arange(n_side, dtype=np.float32))
    P = np.stack([xs.ravel(), ys.ravel(), rs.uniform(-0.3, 0.3, n_side * n_side).astype(np.float32)], 1)
    N = rs.normal(size=P.shape).astype(np.float32)
    N /= np.linalg.norm(N, axis=1, keepdims=True)
    UV = np.stack([xs.ravel() / (n_side - 1), ys.ravel() / (n_side - 1)], 1).astype(np.float32)
    faces = []
    for j in range(n_side - 1):
        for i in range(n_side - 1):
            a, b, c, d = j * n_side + i, j * n_side + i + 1, (j + 1) * n_side + i + 1, (j + 1) * n_side + i
            if (i + j) % 3 == 0:
                faces += [[a, b, c], [a, c, d]]
            else:
                faces.append([a, b, c, d])
    return P, N.astype(np.float32), UV, faces


def write_ply(path, P, N, UV, faces, fmt):
    """vertex: x y z, a uchar `red` in between, nx ny nz, s t (the other spelling of uv); face: a uchar flag, the index list
    (ushort counts for the binary formats exercise a wider count type), then a second list nobody wants"""
    head = ["ply", f"format {fmt} 1.0", "comment written by tests/test_ply_hdr_readers.py", f"element vertex {len(P)}",
            "property float x", "property float y", "property uchar red", "property float z",
            "property float nx", "property float ny", "property float nz", "property double s", "property float t",
            f"element face {len(faces)}", "property uchar flag",
            "property list uchar int vertex_indices" if fmt == "ascii" else "property list ushort uint vertex_indices",
            "property list uchar float weights", "end_header"]
    with open(path, "wb") as f:
        f.write(("\n".join(head) + "\n").encode())
        if fmt == "ascii":
            for p, n, t in zip(P, N, UV):
                f.write(f"{p[0]!r} {p[1]!r} 200 {p[2]!r} {n[0]!r} {n[1]!r} {n[2]!r} {float(t[0])!r} {t[1]!r}\n".replace("np.float32(", "").replace(")", "").encode())
            for k, fc in enumerate(faces):
                f.write((f"{k % 7} {len(fc)} " + " ".join(map(str, fc)) + " 2 0.5 0.25\n").encode())
        else:
            e = "<" if fmt == "binary_little_endian" else ">"
            for p, n, t in zip(P, N, UV):
                f.write(struct.pack(e + "ffBffffdf", p[0], p[1], 200, p[2], n[0], n[1], n[2], float(t[0]), t[1]))
            for k, fc in enumerate(faces):
                f.write(struct.pack(e + "BH" + "I" * len(fc) + "Bff", k % 7, len(fc), *fc, 2, 0.5, 0.25))


def ref_meshes(pbrt_path):
    L = C.CDLL(REF_LIB)
    L.ref_minipbrt_triangle_meshes.argtypes = [C.c_char_p, C.POINTER(C.c_uint), C.POINTER(C.c_uint)] + [C.POINTER(C.POINTER(C.c_float))] * 4 + \
                                              [C.POINTER(C.POINTER(C.c_uint))]
    L.ref_minipbrt_free.argtypes = [C.c_void_p]
    nv, ni = C.c_uint(), C.c_uint()
    P, N, uv, M = (C.POINTER(C.c_float)() for _ in range(4))
    I = C.POINTER(C.c_uint)()
    assert L.ref_minipbrt_triangle_meshes(os.fsencode(pbrt_path), C.byref(nv), C.byref(ni), C.byref(P), C.byref(N), C.byref(uv), C.byref(M), C.byref(I)) == 0
    try:
        return (np.ctypeslib.as_array(P, shape=(nv.value, 3)).copy(), np.ctypeslib.as_array(N, shape=(nv.value, 3)).copy(),
                np.ctypeslib.as_array(uv, shape=(nv.value, 2)).copy(), np.ctypeslib.as_array(I, shape=(ni.value,)).copy())
    finally:
        for p in (P, N, uv, M, I):
            L.ref_minipbrt_free(p)


def mesh_arrays(m):
    v = np.ctypeslib.as_array(C.cast(m.vertices_ptr, C.POINTER(C.c_float)), shape=(m.n_vertices, 8)).copy()
    i = np.ctypeslib.as_array(C.cast(m.indices_ptr, C.POINTER(C.c_uint32)), shape=(m.n_indices,)).copy()
    return v, i


@pytest.mark.parametrize("fmt", ["ascii", "binary_little_endian", "binary_big_endian"])
def test_ply_reader_equals_what_the_file_says_and_minipbrt(tmp_path, fmt):
    P, N, UV, faces = make_mesh(3)
    ply = tmp_path / "grid.ply"
    write_ply(ply, P, N, UV, faces, fmt)
    v, idx = mesh_arrays(host.Mesh.load_ply(str(ply)))
    assert np.array_equal(v[:, :3].view(np.uint32), P.view(np.uint32)) and np.array_equal(v[:, 3:6].view(np.uint32), N.view(np.uint32))
    assert np.array_equal(v[:, 6:8].view(np.uint32), UV.view(np.uint32))
    want = []
    for fc in faces:                                                   # triangles as they are, quads (0 1 3) (2 3 1)
        want += fc if len(fc) == 3 else [fc[0], fc[1], fc[3], fc[2], fc[3], fc[1]]
    assert np.array_equal(idx, np.array(want, np.uint32))
    if os.path.exists(REF_LIB):
        scene = tmp_path / "scene.pbrt"
        scene.write_text('WorldBegin\nShape "plymesh" "string filename" "grid.ply"\nWorldEnd\n')
        rP, rN, ruv, rI = ref_meshes(str(scene))
        assert np.array_equal(rP.view(np.uint32), v[:, :3].view(np.uint32)) and np.array_equal(rN.view(np.uint32), v[:, 3:6].view(np.uint32))
        assert np.array_equal(ruv.view(np.uint32), v[:, 6:8].view(np.uint32)) and np.array_equal(rI, idx)


def test_ply_reader_rejects(tmp_path):
    P, N, UV, faces = make_mesh(4, 3)
    good = tmp_path / "good.ply"
    write_ply(good, P, N, UV, faces, "binary_little_endian")
    data = good.read_bytes()
    cases = {"truncated": data[:-9], "no_magic": b"plx" + data[3:], "bad_index": data.replace(struct.pack("<I", 8), struct.pack("<I", 99), 1),
             "huge_count": data.replace(b"element vertex 9", b"element vertex 999999999"),
             "no_faces": data.replace(b"vertex_indices", b"something_else")}
    for name, blob in cases.items():
        p = tmp_path / (name + ".ply")
        p.write_bytes(blob)
        with pytest.raises(RuntimeError):
            host.Mesh.load_ply(str(p))
    with pytest.raises(RuntimeError):
        host.Mesh.load_ply(str(tmp_path / "missing.ply"))
    # a file without normals gets smooth ones
    p = tmp_path / "flat.ply"
    p.write_text("ply\nformat ascii 1.0\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\nelement face 1\n"
                 "property list uchar int vertex_index\nend_header\n0 0 0\n1 0 0\n0 1 0\n3 0 1 2\n")
    v, idx = mesh_arrays(host.Mesh.load_ply(str(p)))
    assert list(idx) == [0, 1, 2] and np.allclose(v[:, 3:6], [0, 0, 1])


# ---------------------------------------------------------------- Radiance RGBE
def to_rgbe(img):
    """float (h, w, 3) -> uint8 (h, w, 4), Ward's float2rgbe"""
    m = img.max(axis=2)
    out = np.zeros(img.shape[:2] + (4,), np.uint8)
    mant, expo = np.frexp(m)
    ok = m > 1e-32
    scale = np.where(ok, mant * 256.0 / np.where(ok, m, 1), 0)
    out[..., :3] = np.clip(img * scale[..., None], 0, 255).astype(np.uint8)
    out[..., 3] = np.where(ok, expo + 128, 0).astype(np.uint8)
    out[~ok] = 0
    return out


def rle_channel(row):
    out, i, n = bytearray(), 0, len(row)
    while i < n:
        run = 1
        while i + run < n and run < 127 and row[i + run] == row[i]:
            run += 1
        if run >= 4:
            out += bytes([128 + run, row[i]]); i += run
        else:
            j = i
            while j < n and j - i < 128:
                r = 1
                while j + r < n and r < 4 and row[j + r] == row[j]:
                    r += 1
                if r >= 4:
                    break
                j += 1
            out += bytes([j - i]) + bytes(row[i:j]); i = j
    return bytes(out)


def write_hdr(path, rgbe, rle):
    h, w = rgbe.shape[:2]
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\n# made by a test\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=1.0\n\n" + f"-Y {h} +X {w}\n".encode())
        for y in range(h):
            if rle:
                f.write(bytes([2, 2, w >> 8, w & 255]))
                for c in range(4):
                    f.write(rle_channel(rgbe[y, :, c]))
            else:
                f.write(rgbe[y].tobytes())


def decode(rgbe):
    e = rgbe[..., 3].astype(np.int32)
    scale = np.where(e > 0, np.ldexp(np.float32(1.0), e - 136), np.float32(0)).astype(np.float32)
    return (rgbe[..., :3].astype(np.float32) * scale[..., None]).astype(np.float32)


@pytest.mark.parametrize("rle", [False, True])
def test_hdr_reader_decodes_the_published_format(tmp_path, rle):
    rs = np.random.RandomState(12)
    h, w = 19, 40
    img = (rs.uniform(0, 1, (h, w, 3)) ** 3 * 50).astype(np.float32)
    img[3:6, 5:30] = img[3, 5]                         # runs for the encoder
    img[10, :] = 0.0                                   # black: e = 0
    img[12, 7] = [1e-3, 2e4, 0.5]
    rgbe = to_rgbe(img)
    p = tmp_path / ("rle.hdr" if rle else "flat.hdr")
    write_hdr(p, rgbe, rle)
    got = host.load_hdr(str(p))
    want = decode(rgbe)[::-1]                           # rows bottom-up
    assert got.shape == (h, w, 3) and np.array_equal(got.view(np.uint32), np.ascontiguousarray(want).view(np.uint32))
    assert np.abs(got[::-1] - img).max() / img.max() < 0.01          # and it IS the image, to the format's 8-bit mantissa


def test_hdr_reader_rejects(tmp_path):
    rgbe = to_rgbe(np.ones((4, 16, 3), np.float32))
    p = tmp_path / "a.hdr"
    write_hdr(p, rgbe, True)
    data = p.read_bytes()
    for name, blob in {"truncated": data[:-5], "magic": b"#?NOPE" + data[6:], "format": data.replace(b"32-bit_rle_rgbe", b"32-bit_rle_xyze"),
                       "orientation": data.replace(b"-Y 4 +X 16", b"+Y 4 +X 16"), "width": data.replace(bytes([2, 2, 0, 16]), bytes([2, 2, 0, 17]), 1)}.items():
        q = tmp_path / (name + ".hdr")
        q.write_bytes(blob)
        with pytest.raises(RuntimeError):
            host.load_hdr(str(q))


@pytest.mark.gpu
def test_frame_under_an_hdr_file_backdrop(gpu, cornell_spheres, tmp_path):
    """an .hdr file -> trc_host_load_hdr -> trc_set_environment_map -> the frame the oracle renders under the same map"""
    from oracle import pyoracle as po
    rs = np.random.RandomState(5)
    img = (rs.uniform(0, 1, (24, 48, 3)) ** 2 * 6).astype(np.float32)
    p = tmp_path / "sky.hdr"
    write_hdr(p, to_rgbe(img), True)
    env = host.load_hdr(str(p))
    W, H = 96, 64
    cam = host.prepare_camera(W, H)
    gpu.upload_scene(cornell_spheres.view); gpu.set_camera(cam); gpu.resize(W, H)
    gpu.set_environment_map(env); po.set_environment_map(env)
    try:
        for integ in (abi.INTEGRATOR_PATH, abi.INTEGRATOR_MIS):
            rng = host.fill_rng(4, W, H)
            gpu.upload_rng(rng); gpu.clear_accum(); gpu.render(spp=6, integrator=integ)
            ref, _ = po.render(cornell_spheres.view, cam, W, H, rng, spp=6, integrator=integ)
            got = gpu.download_accum()
            assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)) and got[0, 0, :3].max() > 0   # the corner sees the sky
    finally:
        gpu.set_environment_map(None); po.set_environment_map(None)
