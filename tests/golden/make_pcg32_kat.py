#!/usr/bin/env python3
"""Generates tests/golden/pcg32_kat.json from the REFERENCE's own PCG32.

oracle/_ref/libpcg_ref.so is compiled by oracle/Makefile from
/root/reference/RT_Metal/Tracer/pcg_basic.c where it lies (never copied).  This is the only part of
the hot path the reference lets us execute off macOS, so it is the only executable pin of the oracle.
Run in the build container:  make -C oracle && python tests/golden/make_pcg32_kat.py
"""
import ctypes as C
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
ref = C.CDLL(os.path.join(HERE, "..", "..", "oracle", "_ref", "libpcg_ref.so"))


class Pcg(C.Structure):
    _fields_ = [("state", C.c_uint64), ("inc", C.c_uint64)]


ref.pcg32_srandom_r.argtypes = [C.POINTER(Pcg), C.c_uint64, C.c_uint64]
ref.pcg32_random_r.argtypes = [C.POINTER(Pcg)]
ref.pcg32_random_r.restype = C.c_uint32

cases = []
for initstate, initseq in [(42, 54), (0, 0), (0x5EED0000, 0), (0x5EED0000, 1919 + 1079 * 1920),
                           (0xFFFFFFFFFFFFFFFF, 0xFFFFFFFFFFFFFFFF), (1234567, 131 * 77 - 1)]:
    r = Pcg()
    ref.pcg32_srandom_r(C.byref(r), initstate, initseq)
    state0, inc = r.state, r.inc
    outs = [ref.pcg32_random_r(C.byref(r)) for _ in range(8)]
    cases.append({"initstate": initstate, "initseq": initseq, "state_after_seed": state0, "inc": inc,
                  "outputs": outs, "state_after_outputs": r.state})
# raw stepping from arbitrary (state, inc) words, incl. an EVEN inc as the kernel's word swap produces (B-1)
raw = []
for state, inc in [(0x853c49e6748fea9b, 0xda3e39cb94b95bdb), (0x0123456789abcdef, 0x00000000deadbeee), (0, 0)]:
    r = Pcg(state, inc)
    outs = [ref.pcg32_random_r(C.byref(r)) for _ in range(6)]
    raw.append({"state": state, "inc": inc, "outputs": outs, "state_after": r.state})
json.dump({"source": "RT_Metal/Tracer/pcg_basic.c (reference, compiled in place)", "srandom": cases, "raw": raw},
          open(os.path.join(HERE, "pcg32_kat.json"), "w"), indent=1)
print("wrote pcg32_kat.json; first case:", [hex(x) for x in cases[0]["outputs"][:6]])
