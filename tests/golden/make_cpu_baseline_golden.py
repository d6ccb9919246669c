#!/usr/bin/env python3
"""Regenerates tests/golden/cpu_baseline.json: ray count, image hash and mean pixel of oracle/cpu_baseline (the C++
restatement of RT_Weekend / RT_Nextweek's CPU tracer, BASELINE config 1) on two small deterministic runs.  The reference
itself is Swift with arc4random() and cannot run here, so these values pin the restatement against ITSELF (determinism,
thread-count independence, accidental edits), not against the reference; the physical checks live in the test."""
import json, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
EXE = os.path.join(ROOT, "oracle", "cpu_baseline")
CASES = [("cornell", 64, 64, 4), ("random", 200, 100, 8)]
out = []
for scene, w, h, spp in CASES:
    line = subprocess.run([EXE, "--scene", scene, "--width", str(w), "--height", str(h), "--spp", str(spp), "--threads", "4"],
                          capture_output=True, text=True, check=True).stdout
    d = json.loads(line)
    out.append({k: d[k] for k in ("scene", "width", "height", "spp", "rays", "mean_pixel", "image_fnv1a")})
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "cpu_baseline.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
