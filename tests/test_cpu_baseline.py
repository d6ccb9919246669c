"""oracle/cpu_baseline.cpp -- the reference's RT_Weekend / RT_Nextweek CPU tracer restated in C++ (BASELINE config 1, the
`cpu_rt_weekend` leg of bench.py; RT_Weekend/Tracer/main.swift:3-17,59-135, RT_Nextweek/Tracer/Render.swift:171-192).
Checked here: it is deterministic, independent of the worker count, matches its committed fingerprints
(tests/golden/cpu_baseline.json, made by tests/golden/make_cpu_baseline_golden.py), and its images obey what the scenes
dictate analytically (the emitter saturates, the walls carry their albedo's hue, a miss returns the sky gradient)."""
import json
import math
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "cpu_baseline")
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "cpu_baseline.json")))

pytestmark = pytest.mark.skipif(not os.path.exists(EXE), reason="oracle/cpu_baseline not built (make -C oracle)")


def run(scene, w, h, spp, threads, ppm=None):
    cmd = [EXE, "--scene", scene, "--width", str(w), "--height", str(h), "--spp", str(spp), "--threads", str(threads)]
    if ppm:
        cmd += ["--ppm", str(ppm)]
    return json.loads(subprocess.run(cmd, capture_output=True, text=True, check=True, timeout=120).stdout)


def read_ppm(path):
    raw = open(path, "rb").read()
    parts = raw.split(b"\n", 3)
    assert parts[0] == b"P6" and parts[2] == b"255"
    w, h = map(int, parts[1].split())
    return np.frombuffer(parts[3], np.uint8).reshape(h, w, 3)


@pytest.mark.parametrize("gold", GOLD, ids=[g["scene"] for g in GOLD])
def test_fingerprints_and_thread_count_independence(gold):
    seen = set()
    for threads in (1, 3, 8):
        d = run(gold["scene"], gold["width"], gold["height"], gold["spp"], threads)
        assert d["threads"] == threads
        seen.add((d["rays"], d["image_fnv1a"], d["mean_pixel"]))
    assert len(seen) == 1, f"image or ray count depends on the worker count: {seen}"
    rays, fnv, mean = seen.pop()
    assert rays == gold["rays"] and fnv == gold["image_fnv1a"] and mean == gold["mean_pixel"]


def test_ray_unit_is_one_hit_test_per_path_vertex():
    # a "ray" is one world.hitTest call: at least one per sample, at most 51 (depth < 50, main.swift:3-17)
    g = GOLD[0]
    n_paths = g["width"] * g["height"] * g["spp"]
    assert n_paths <= g["rays"] <= 51 * n_paths


def test_cornell_image_obeys_the_scene(tmp_path):
    """RT_Nextweek Render.swift:171-192 + camera1 :59-81: camera (278, 278, -800) looking down +z with a 40 degree vertical
    field of view; emitter rect x in [213, 343], z in [227, 332] at y = 554 with emission 15; the x = 555 wall is green, the x = 0
    wall red, everything else white 0.73.  hitTest has no left/right flip, so world +x is image right... the camera's u axis is
    cross(vup, w) with w = lookFrom - lookAt, which points to world -x: world x = 555 (green) appears on the image's LEFT."""
    W = H = 128
    d = run("cornell", W, H, 64, 8, tmp_path / "c.ppm")
    img = read_ppm(tmp_path / "c.ppm").astype(np.float64)
    half = math.tan(math.radians(20.0))

    def pixel_of(x, y, z):                       # projection through camera1 (aspect 1): image column, row from the top
        dz = z + 800.0
        u = 0.5 + 0.5 * ((278.0 - x) / dz) / half
        v = 0.5 + 0.5 * ((y - 278.0) / dz) / half
        return int(u * W), int((1.0 - v) * H)

    # the emitter seen directly: 15 -> sqrt -> clamps to 255 in every channel
    cx, cy = pixel_of(278.0, 554.0, 279.5)
    assert (img[cy, cx] == 255).all(), img[cy, cx]
    # side walls at mid height, mid depth: hue of their albedo (green 0.12 .45 .15 on x = 555, red .65 .05 .05 on x = 0)
    gx, gy = pixel_of(555.0, 278.0, 278.0)
    rx, ry = pixel_of(0.0, 278.0, 278.0)
    green = img[gy - 3:gy + 4, max(gx - 1, 0):gx + 6].mean(axis=(0, 1))
    red = img[ry - 3:ry + 4, rx - 5:rx + 2].mean(axis=(0, 1))
    assert gx < W // 2 < rx, (gx, rx)
    assert green[1] > 1.5 * green[0] and green[1] > 1.5 * green[2], green
    assert red[0] > 2.0 * red[1] and red[0] > 2.0 * red[2], red
    # back wall (white 0.73): neutral within the colour bleeding of the side walls
    bx, by = pixel_of(278.0, 400.0, 555.0)
    back = img[by - 3:by + 4, bx - 3:bx + 4].mean(axis=(0, 1))
    assert back.min() > 0.7 * back.max() and back.mean() > 40, back
    # energy sanity: a closed box lit by one small emitter -- neither black nor blown out
    assert 25.0 < d["mean_pixel"] < 120.0, d["mean_pixel"]
    assert d["rays"] >= W * H * 64


def test_random_scene_sky_gradient(tmp_path):
    """main.swift:3-17: a ray that hits nothing returns (1 - t) * white + t * (0.5, 0.7, 1.0) with t = 0.5 * (dir.y + 1).
    The top rows of the `random` scene (camera (13, 2, 3) -> origin, 20 degree fov) see only sky above the horizon spheres."""
    W, H = 200, 100
    run("random", W, H, 16, 8, tmp_path / "r.ppm")
    img = read_ppm(tmp_path / "r.ppm").astype(np.float64) / 255.0
    top = img[0:4].mean(axis=(0, 1)) ** 2          # undo the gamma-2 of main.swift:100-105
    # unit direction towards the top of the image: camera w = normalize(13, 2, 3), pitch = half the 20 degree fov
    w = np.array([13.0, 2.0, 3.0]); w /= np.linalg.norm(w)
    up = np.array([0.0, 1.0, 0.0]); u = np.cross(up, w); u /= np.linalg.norm(u); v = np.cross(w, u)
    d = -w + math.tan(math.radians(10.0)) * v * (1.0 - 4.0 / H)
    d /= np.linalg.norm(d)
    t = 0.5 * (d[1] + 1.0)
    expect = (1.0 - t) * np.ones(3) + t * np.array([0.5, 0.7, 1.0])
    assert np.allclose(top, expect, atol=0.02), (top, expect)
