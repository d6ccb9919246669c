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
