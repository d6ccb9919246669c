#!/usr/bin/env python3
"""Pins for include/trc_sobol.h, taken from the reference's own tables (run in the build container, where
/root/reference exists): CRC-32 of every dimension of SobolMatrices32 that the generator provides and of every row of
VdCSobolMatrices / VdCSobolMatricesInv (RT_Metal/Metal/Sobolmatrices.metal:69,26701,26827), plus a few values of
SobolSampleFloat / SobolIntervalToIndex (SobolSampler.hh:126-160) evaluated on the reference's tables.

Only checksums and sample points are written -- the tables themselves stay in the reference."""
import json, os, re, struct, sys, zlib

REF = "/root/reference/RT_Metal/Metal/Sobolmatrices.metal"
DIMS, SIZE = 40, 52


def body(src, name):
    i = src.index("{", src.index(name))
    depth, j = 0, i
    while True:
        depth += {"{": 1, "}": -1}.get(src[j], 0)
        if depth == 0:
            return src[i:j + 1]
        j += 1


def words(text):
    return [int(x, 16) for x in re.findall(r"0x([0-9a-fA-F]+)", text)]


def main():
    src = open(REF).read()
    m32 = words(body(src, "SobolMatrices32[NumSobolDimensions * SobolMatrixSize] ="))
    assert len(m32) == 1024 * SIZE
    rows = lambda name: [words(r) for r in re.findall(r"\{([^{}]*)\}", body(src, name))]
    vdc, inv = rows("VdCSobolMatrices[][SobolMatrixSize] ="), rows("VdCSobolMatricesInv[][SobolMatrixSize] =")
    pad = lambda r: r + [0] * (SIZE - len(r))
    out = {
        "source": "RT_Metal/Metal/Sobolmatrices.metal (pbrt-v3 core/sobolmatrices.cpp)",
        "matrices32_crc32": [zlib.crc32(struct.pack("<52I", *m32[d * SIZE:(d + 1) * SIZE])) for d in range(DIMS)],
        "vdc_crc32": [zlib.crc32(struct.pack("<52Q", *pad(r))) for r in vdc],
        "inv_crc32": [zlib.crc32(struct.pack("<52Q", *pad(r))) for r in inv],
    }

    def sample_u32(index, dim):                       # SobolSampleFloat before the float conversion, :150-160
        v, i = 0, dim * SIZE
        while index:
            if index & 1:
                v ^= m32[i]
            index >>= 1; i += 1
        return v

    def interval_to_index(m, sample_index, px, py):   # SobolIntervalToIndex, :126-148
        if m == 0:
            return 0
        index, delta, c, si = sample_index << (2 * m), 0, 0, sample_index
        while si:
            if si & 1:
                delta ^= pad(vdc[m - 1])[c]
            si >>= 1; c += 1
        b, c = ((px << m) | py) ^ delta, 0
        while b:
            if b & 1:
                index ^= pad(inv[m - 1])[c]
            b >>= 1; c += 1
        return index

    out["sample_u32"] = [[i, d, sample_u32(i, d)] for i in (1, 2, 3, 7, 1000, 123456789, (5 << 22) | 0x2F3A1) for d in (0, 1, 2, 5, 15, 39)]
    out["interval_to_index"] = [[m, s, x, y, interval_to_index(m, s, x, y)]
                                for m, s, x, y in [(1, 0, 1, 0), (4, 3, 5, 9), (7, 0, 100, 27), (11, 0, 0, 0), (11, 1, 1919, 1079),
                                                   (11, 63, 960, 540), (11, 1000, 7, 2047), (13, 5, 8000, 4000)]]
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sobol_tables.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
