#!/usr/bin/env python3
"""8x8 against 4x4 pixel blocks (TRC_FLAG_SMALL_BLOCKS) on frames with few blocks: kernel ms per launch, config 2's scene."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracer_amd import abi, host
from tracer_amd.device import Tracer
sc = host.HostScene(abi.SCENE_CORNELL_SPHERES)
t = Tracer(0); t.upload_scene(sc.view); t.set_environment((0, 0, 0))
for W, H in ((320, 180), (480, 270), (640, 360), (800, 450), (960, 540), (1280, 720)):
    t.set_camera(host.prepare_camera(W, H)); t.resize(W, H)
    for spp in (64, 16):
        row = []
        for small in (False, True):
            for _ in range(2):                      # the second launch has the adaptive order
                t.seed(1); t.render(spp=spp, small_blocks=small); t.synchronize()
            t.seed(1); t.reset_stats(); t.render(spp=spp, small_blocks=small); t.synchronize()
            row.append(t.stats().kernel_ms)
        print(f"{W}x{H} ({(W + 7) // 8 * ((H + 7) // 8)} blocks) {spp} spp: 8x8 {row[0]:.3f} ms, 4x4 {row[1]:.3f} ms")
