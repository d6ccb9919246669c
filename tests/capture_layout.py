"""Layout facts of a tone-mapped Cornell frame (RT_Metal's default scene, Tracer.mm:127-411) -- shared by
tests/golden/make_capture_layout.py, which extracts them from the reference's own screenshot
(/root/reference/Captures/capture_t.jpg, README.md:11) into tests/golden/capture_layout.json, and by
tests/test_capture_layout.py, which extracts them from frames rendered here (oracle on the CPU, HIP on the GPU).

Nothing numerical about radiance is compared (the screenshot is a JPEG of a window, lit by an HDR file that is missing from the
reference's repository, with a Stanford bunny that is not part of the default scene): the facts are where things ARE -- which wall
is red, where the box, its back wall and the light sit in the picture, how many checker squares span the back wall and the
ceiling, where the tall block stands.  Every position is normalised to the painted box (outer edge of the red wall to outer
edge of the green wall), so window chrome, drop shadows and the picture's resolution drop out.  They catch a flipped axis, a
swapped wall, a wrong camera or field of view, a mirrored or upside-down output stage."""
import numpy as np


def _runs(mask_1d, min_len):
    """[start, end) of the longest run of True at least min_len long, or None."""
    best, start = None, None
    for i, v in enumerate(list(mask_1d) + [False]):
        if v and start is None:
            start = i
        elif not v and start is not None:
            if i - start >= min_len and (best is None or i - start > best[1] - best[0]):
                best = (start, i)
            start = None
    return best


def _square_count(profile):
    """Number of checker squares across a luminance profile: the edges between squares are the steps of the smoothed profile
    (a wall's shading drifts slowly, a checker edge jumps), and the median distance between neighbouring steps divides the
    profile's length."""
    p = np.asarray(profile, np.float64)
    k = max(1, len(p) // 80)
    p = np.convolve(np.pad(p, k, mode="edge"), np.ones(2 * k + 1) / (2 * k + 1), mode="valid")   # against Monte-Carlo noise / JPEG
    d = np.abs(p[2 * k:] - p[:-2 * k])                                        # step height over the smoothing width
    strong = d > 0.4 * np.percentile(d, 99)
    edges, i = [], 0
    while i < len(strong):
        if strong[i]:
            j = i
            while j < len(strong) and strong[j]:
                j += 1
            edges.append(i + int(np.argmax(d[i:j])))
            i = j
        else:
            i += 1
    if len(edges) < 2:
        return 0.0
    gaps = np.diff(edges)
    gaps = gaps[gaps > len(p) / 40]
    return len(p) / float(np.median(gaps)) if len(gaps) else 0.0


def extract(rgb):
    """rgb: (H, W, 3) uint8, top row first.  Returns the facts as a dict of floats (positions in units of the painted box)."""
    img = np.asarray(rgb, np.float64)
    H, W = img.shape[:2]
    # hue decisions are taken on a blurred copy (a 64-sample frame is speckled; the screenshot is a JPEG)
    kb = max(1, W // 240)
    def blur(c):
        ker = np.ones(2 * kb + 1) / (2 * kb + 1)
        c = np.apply_along_axis(lambda v: np.convolve(np.pad(v, kb, mode="edge"), ker, mode="valid"), 0, c)
        return np.apply_along_axis(lambda v: np.convolve(np.pad(v, kb, mode="edge"), ker, mode="valid"), 1, c)
    R, G, B = blur(img[..., 0]), blur(img[..., 1]), blur(img[..., 2])
    red = (R > 10) & (R > 3.0 * np.maximum(G, B))
    green = (G > 8) & (G > 2.2 * np.maximum(R, B))
    # the walls: columns in which a long vertical stretch is wall-coloured
    def wall_columns(mask):
        col = mask.sum(axis=0)
        return _runs(col > 0.2 * H, W // 50)
    rc, gc = wall_columns(red), wall_columns(green)
    assert rc and gc, "no red or green wall found"
    facts = {"red_wall_is_left_of_green": float(rc[1] <= gc[0])}
    left_wall, right_wall = (rc, gc) if rc[1] <= gc[0] else (gc, rc)
    x0, x1 = left_wall[0], right_wall[1]                 # painted box, outer edges
    bx0, bx1 = left_wall[1], right_wall[0]               # back wall, inner edges of the side walls
    lmask = red if left_wall is rc else green
    outer = np.flatnonzero(lmask[:, x0 + 1:x0 + 2 + W // 100].any(axis=1))
    inner = np.flatnonzero(lmask[:, bx0 - 2 - W // 100:bx0 - 1].any(axis=1))
    y0, y1 = outer.min(), outer.max() + 1
    by0, by1 = inner.min(), inner.max() + 1
    fw, fh = float(x1 - x0), float(y1 - y0)
    facts.update({
        "box_aspect": fw / fh,
        "back_wall_left": (bx0 - x0) / fw, "back_wall_right": (bx1 - x0) / fw,
        "back_wall_top": (by0 - y0) / fh, "back_wall_bottom": (by1 - y0) / fh,
    })
    # the light: saturated white in the ceiling (upper part of the box, between the side walls)
    top = img[y0:y0 + int(0.3 * fh), x0:x1]
    mn = top.min(axis=2)
    white = mn >= max(200.0, 0.95 * np.percentile(mn, 99))      # ACES tops out just below 255; the percentile ignores fireflies
    cols = _runs(white.sum(axis=0) > 0.02 * fh, max(2, int(0.03 * fw)))
    rows = _runs(white.sum(axis=1) > 0.04 * fw, max(2, int(0.01 * fh)))
    assert cols and rows, "no light found in the ceiling"
    facts.update({"light_left": cols[0] / fw, "light_right": cols[1] / fw, "light_top": rows[0] / fh, "light_bottom": rows[1] / fh})
    # checker squares: across the back wall (first row of squares, left 60 %: the screenshot has a bunny on the right) and across
    # the front edge of the ceiling
    luma = 0.2126 * R + 0.7152 * G + 0.0722 * B
    bw = bx1 - bx0
    yb = by0 + int(0.05 * (by1 - by0))                  # inside the first of the four rows of squares
    band = luma[yb:yb + max(2, int(0.15 * (by1 - by0))), bx0 + 2:bx0 + int(0.6 * bw)].mean(axis=0)
    facts["back_wall_squares_across"] = _square_count(band) / 0.6 * (len(band) / (0.6 * bw))
    yc = y0 + max(1, int(0.015 * fh))
    band = luma[yc:yc + max(2, int(0.02 * fh)), x0 + int(0.12 * fw):x0 + int(0.88 * fw)].mean(axis=0)
    facts["ceiling_squares_across_front"] = _square_count(band) / 0.76
    # the tall block in front of the back wall: yellowish (gold Metal, MicrofacetBXDF.h:445-449) columns at 60 % of the box height
    yt = y0 + int(0.60 * fh)
    strip = img[yt - max(1, int(0.02 * fh)):yt + max(1, int(0.02 * fh)), bx0:bx1]
    r, g, b = strip[..., 0].mean(axis=0), strip[..., 1].mean(axis=0), strip[..., 2].mean(axis=0)
    gold = (r > 60) & (g > 0.7 * r) & (b < 0.72 * g) & (g < 1.25 * r)
    tb = _runs(gold, max(2, int(0.03 * fw)))
    assert tb, "no gold block found"
    facts.update({"tall_block_left": (tb[0] + bx0 - x0) / fw, "tall_block_right": (tb[1] + bx0 - x0) / fw})
    return {k: round(float(v), 4) for k, v in facts.items()}


# ---------------------------------------------------------------- the twelve spheres (Captures/capture_o.jpg, README.md:21)
# capture_o is the reference's BVH view of the scene with prepareSphereList's spheres inserted (Tracer.mm:306-369): the walls are
# not drawn, the boxes of the tree are -- the outermost rectangle is the scene's front face, i.e. the painted box of capture_t --
# and the spheres are: a row of five near the ceiling, a row of six on the floor (a bunny that the repository does not ship stands in
# the middle and hides parts of them).  Facts: the silhouettes' centres, normalised to that rectangle.

def sphere_rows_from_wireframe(rgb):
    """rgb: (H, W, 3) uint8 of the BVH view.  Returns {"top": [(x, y, w, h), ...], "bottom": [...]} in units of the outer box;
    silhouettes that the bunny's wireframe cuts come out narrower than the whole ones (the caller keeps both apart by width)."""
    img = np.asarray(rgb, np.float64)
    H, W = img.shape[:2]
    R, G, B = img[..., 0], img[..., 1], img[..., 2]
    white = np.minimum(np.minimum(R, G), B) > 200
    # the outer rectangle: the first and last long white lines inside the window (rows / columns more than half white)
    inner = white[int(0.06 * H):int(0.94 * H), int(0.04 * W):int(0.96 * W)]
    rows = np.flatnonzero(inner.sum(axis=1) > 0.5 * inner.shape[1]) + int(0.06 * H)
    cols = np.flatnonzero(inner.sum(axis=0) > 0.5 * inner.shape[0]) + int(0.04 * W)
    assert len(rows) >= 2 and len(cols) >= 2, "no wireframe rectangle found"
    y0, y1, x0, x1 = rows.min() + 1.5, rows.max() - 0.5, cols.min() + 1.5, cols.max() - 0.5
    fw, fh = x1 - x0, y1 - y0
    green = (G > 35) & (G > 1.25 * R) & (G > 1.25 * B)
    red = (R > 45) & (R > 2.2 * G) & (R > 2.2 * B)

    def blobs(mask, ylo, yhi, minw):
        ya = int(y0 + ylo * fh)
        sub = mask[ya:int(y0 + yhi * fh), int(x0):int(x1)]
        k = int(0.004 * fw)                                   # close the gaps of the wireframe lines drawn over the spheres
        col = np.convolve(sub.sum(axis=0) > 0.01 * fh, np.ones(2 * k + 1), mode="same") > 0
        out, start = [], None
        for i, v in enumerate(list(col) + [False]):
            if v and start is None:
                start = i
            elif not v and start is not None:
                a, b = start + k, i - k
                start = None
                if b - a < minw * fw:
                    continue
                r = np.flatnonzero(sub[:, a:b].sum(axis=1) > 0.2 * (b - a))
                out.append(((a + b) / 2 / fw, ((r.min() + r.max() + 1) / 2 + ya - y0) / fh, (b - a) / fw, (r.max() + 1 - r.min()) / fh))
        return [tuple(round(float(v), 4) for v in t) for t in out]
    return {"box_aspect": round(float(fw / fh), 4), "top": blobs(green, 0.12, 0.36, 0.02), "bottom": blobs(red, 0.78, 0.99, 0.03)}


def sphere_rows_from_hits(ptype, pindex, hit, sphere_type=0):
    """The same facts from a scene: (H, W) arrays of the primary rays' Scene::hit results, row 0 = v 0 = the BOTTOM of the picture.
    Returns {"box_aspect", "spheres": [(x, y, w, h), ...]} with y measured downwards like a picture's, normalised to the extent of
    everything the camera sees (the front edges of the box)."""
    any_hit = hit > 0
    cols, rows = np.flatnonzero(any_hit.any(axis=0)), np.flatnonzero(any_hit.any(axis=1))
    x0, x1, y0, y1 = cols.min(), cols.max() + 1, rows.min(), rows.max() + 1
    fw, fh = float(x1 - x0), float(y1 - y0)
    out = []
    for s in np.unique(pindex[ptype == sphere_type]):
        m = (ptype == sphere_type) & (pindex == s)
        c, r = np.flatnonzero(m.any(axis=0)), np.flatnonzero(m.any(axis=1))
        out.append(((c.min() + c.max() + 1) / 2 - x0, y1 - (r.min() + r.max() + 1) / 2, c.max() + 1 - c.min(), r.max() + 1 - r.min()))
    return {"box_aspect": fw / fh, "spheres": [(a / fw, b / fh, c / fw, d / fh) for a, b, c, d in out]}

