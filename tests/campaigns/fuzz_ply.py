#!/usr/bin/env python3
"""PLY reader campaign (CPU; needs oracle/_ref/libminipbrt_ref.so): random files -- ascii / little / big endian; vertex properties in random
order with random extra scalar and list properties of every PLY type; positions as float or double; normals and uv present or not, uv
spelled u v / s t / texture_u texture_v; faces of 3..4 vertices under count types uchar / ushort / uint and index types of every integer
width, named vertex_indices or vertex_index, with other list properties before and after -- through trc_host_mesh_load_ply and the
reference's minipbrt (as `Shape "plymesh"`): positions, normals, uv and the triangle list, bit for bit.   python3 tests/campaigns/fuzz_ply.py <a> <b>"""
import os, struct, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from tracer_amd import host
from test_ply_hdr_readers import ref_meshes, mesh_arrays

TYPES = {"char": "b", "uchar": "B", "short": "h", "ushort": "H", "int": "i", "uint": "I", "float": "f", "double": "d"}
ALIASES = {"char": "int8", "uchar": "uint8", "short": "int16", "ushort": "uint16", "int": "int32", "uint": "uint32", "float": "float32", "double": "float64"}

def val(rs, t):
    if t in ("float", "double"): return float(np.float32(rs.uniform(-5, 5)))
    lo, hi = {"char": (-100, 100), "uchar": (0, 200), "short": (-3000, 3000), "ushort": (0, 6000), "int": (-10**6, 10**6), "uint": (0, 10**6)}[t]
    return int(rs.randint(lo, hi))

a, b = int(sys.argv[1]), int(sys.argv[2])
bad = skipped = 0
with tempfile.TemporaryDirectory() as d:
    for seed in range(a, b):
        rs = np.random.RandomState(seed)
        nv = int(rs.randint(4, 40)); nf = int(rs.randint(1, 30))
        fmt = ["ascii", "binary_little_endian", "binary_big_endian"][seed % 3]
        P = rs.uniform(-10, 10, (nv, 3)).astype(np.float32); N = rs.normal(size=(nv, 3)).astype(np.float32); UV = rs.rand(nv, 2).astype(np.float32)
        has_n, has_uv = rs.rand() < 0.6, rs.rand() < 0.6
        uvn = [("u", "v"), ("s", "t"), ("texture_u", "texture_v"), ("texture_s", "texture_t")][int(rs.randint(4))]
        ptype = "double" if rs.rand() < 0.2 else "float"
        alias = rs.rand() < 0.3
        # vertex properties: the wanted ones in random order, with extras in between
        props = [("x", ptype, 0), ("y", ptype, 1), ("z", ptype, 2)]
        if has_n: props += [("nx", "float", 3), ("ny", "float", 4), ("nz", "float", 5)]
        if has_uv: props += [(uvn[0], "float", 6), (uvn[1], "float", 7)]
        if rs.rand() < 0.5: rs.shuffle(props)
        full = []
        for p in props:
            while rs.rand() < 0.25: full.append((f"extra{len(full)}", list(TYPES)[int(rs.randint(8))], None))
            full.append(p)
        vlist = rs.rand() < 0.2                        # a list property inside the vertex element
        faces = [[int(x) for x in rs.choice(nv, int(rs.choice([3, 3, 4])), replace=False)] for _ in range(nf)]
        ct, it = ["uchar", "ushort", "uint"][int(rs.randint(3))], ["int", "uint", "short", "ushort", "uchar", "char"][int(rs.randint(6))]
        if it in ("char",) and nv > 100: it = "int"
        fname = "vertex_indices" if rs.rand() < 0.7 else "vertex_index"
        pre, post = rs.rand() < 0.4, rs.rand() < 0.4
        tn = (lambda t: ALIASES[t]) if alias else (lambda t: t)
        head = ["ply", f"format {fmt} 1.0", "comment tests/campaigns/fuzz_ply.py", f"element vertex {nv}"] + [f"property {tn(t)} {n}" for n, t, _ in full]
        if vlist: head.append(f"property list uchar {tn('short')} vl")
        head.append(f"element face {nf}")
        if pre: head.append(f"property {tn('uchar')} flag")
        head.append(f"property list {tn(ct)} {tn(it)} {fname}")
        if post: head.append(f"property list uchar {tn('float')} weights")
        head.append("end_header")
        rows_v = []
        for k in range(nv):
            src = [P[k, 0], P[k, 1], P[k, 2], N[k, 0], N[k, 1], N[k, 2], UV[k, 0], UV[k, 1]]
            rows_v.append([(t, float(src[i]) if i is not None else val(rs, t)) for _, t, i in full] + ([("L", [int(x) for x in rs.randint(-5, 5, int(rs.randint(0, 4)))])] if vlist else []))
        path = os.path.join(d, f"m{seed}.ply")
        with open(path, "wb") as f:
            f.write(("\n".join(head) + "\n").encode())
            e = "<" if fmt == "binary_little_endian" else ">"
            for row in rows_v:
                if fmt == "ascii":
                    out = []
                    for t, v in row:
                        if t == "L": out += [str(len(v))] + [str(x) for x in v]
                        else: out.append(repr(v))
                    f.write((" ".join(out) + "\n").encode())
                else:
                    for t, v in row:
                        if t == "L": f.write(struct.pack(e + "B" + "h" * len(v), len(v), *v))
                        else: f.write(struct.pack(e + TYPES[t], v))
            for k, fc in enumerate(faces):
                if fmt == "ascii":
                    f.write(((f"{k % 7} " if pre else "") + f"{len(fc)} " + " ".join(map(str, fc)) + (" 2 0.5 0.25" if post else "") + "\n").encode())
                else:
                    if pre: f.write(struct.pack(e + "B", k % 7))
                    f.write(struct.pack(e + TYPES[ct] + TYPES[it] * len(fc), len(fc), *fc))
                    if post: f.write(struct.pack(e + "Bff", 2, 0.5, 0.25))
        scene = os.path.join(d, f"s{seed}.pbrt"); open(scene, "w").write(f'WorldBegin\nShape "plymesh" "string filename" "m{seed}.ply"\nWorldEnd\n')
        why = None
        try: rP, rN, ruv, rI = ref_meshes(scene)
        except AssertionError: skipped += 1; continue                    # minipbrt itself refuses the file
        if len(rP) == 0: skipped += 1; continue                          # ... or loads nothing from it (index types / names it does not take)
        try:
            v, idx = mesh_arrays(host.Mesh.load_ply(path))
        except Exception as ex:
            why = f"loader refuses what minipbrt reads: {ex}"
        if not why:
            if len(rP) != len(v) or not np.array_equal(rP.view(np.uint32), v[:, :3].view(np.uint32)): why = "positions"
            elif has_n and not np.array_equal(rN.view(np.uint32), v[:, 3:6].view(np.uint32)): why = "normals"
            elif has_uv and not np.array_equal(ruv.view(np.uint32), v[:, 6:8].view(np.uint32)): why = "uv"
            elif not np.array_equal(rI, idx): why = f"indices {len(rI)} / {len(idx)}"
        if why:
            bad += 1
            keep = os.path.join(ROOT, "gpurun_out", f"fuzz_ply_{seed}.ply"); os.makedirs(os.path.dirname(keep), exist_ok=True); open(keep, "wb").write(open(path, "rb").read())
            print(f"MISMATCH seed {seed}: {why}; {fmt} ptype {ptype} alias {alias} n {has_n} uv {has_uv} {uvn} count {ct} index {it} name {fname} pre {pre} post {post} vlist {vlist}  ({keep})", flush=True)
print(f"seeds {a}..{b - 1}: {b - a - bad - skipped} passed, {skipped} refused by minipbrt itself, {bad} FAILED")
