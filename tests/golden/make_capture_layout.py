#!/usr/bin/env python3
"""Extracts layout facts from the reference's own screenshot of RT_Metal's default scene into
tests/golden/capture_layout.json.  Run in the build container (reads /root/reference/Captures/capture_t.jpg, the picture at
the top of the reference's README.md:11); the JSON is data -- a dozen numbers -- and travels, the picture does not.

The extraction itself is tests/capture_layout.py::extract, the same function the tests apply to frames rendered here."""
import json, os, sys
import numpy as np
from PIL import Image
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import capture_layout as cl

SRC = "/root/reference/Captures/capture_t.jpg"
img = np.asarray(Image.open(SRC).convert("RGB"))
facts = cl.extract(img)
# tolerance per fact, in the fact's own unit (fractions of the painted box; squares): what a JPEG of a retina window, a
# 64-sample frame and two different light set-ups (the screenshot is lit by an HDR backdrop that is missing from the repository)
# leave between two pictures of the same geometry; a flipped axis, a swapped wall or a wrong field of view moves these facts
# by 0.1-0.6
tol = {k: 0.02 for k in facts}
tol.update({"red_wall_is_left_of_green": 0.0, "box_aspect": 0.04, "back_wall_squares_across": 0.7, "ceiling_squares_across_front": 0.7})
out = {"source": "Captures/capture_t.jpg (README.md:11)", "picture_size": [int(img.shape[1]), int(img.shape[0])],
       "facts": facts, "tolerance": tol}
json.dump(out, open(os.path.join(HERE, "capture_layout.json"), "w"), indent=1)
print(json.dumps(out, indent=1))

# the twelve spheres of prepareSphereList, from the BVH view of the scene (Captures/capture_o.jpg, README.md:21)
SRC_O = "/root/reference/Captures/capture_o.jpg"
img = np.asarray(Image.open(SRC_O).convert("RGB"))
rows = cl.sphere_rows_from_wireframe(img)
out = {"source": "Captures/capture_o.jpg (README.md:21): the BVH view, walls not drawn, a bunny (not in the repository) in the middle",
       "picture_size": [int(img.shape[1]), int(img.shape[0])], "box_aspect": rows["box_aspect"],
       "top": rows["top"], "bottom": rows["bottom"],
       "tolerance": {"x": 0.015, "whole_silhouette": 0.006, "height": 0.015, "box_aspect": 0.03},   # the dark underside of a sphere is not "red"
       "what": "x, y, width, height of the green (top row) and red (bottom row) silhouettes in units of the outer wireframe rectangle; "
               "silhouettes cut by the bunny's wireframe are narrower than the whole ones and only their x counts"}
json.dump(out, open(os.path.join(HERE, "capture_spheres.json"), "w"), indent=1)
print(json.dumps(out, indent=1))

