"""Every transfer between device memory and caller memory goes through the context's pinned staging buffer in 4 MB chunks, two in flight
(trc_copy_to_host / trc_copy_to_device, trc_ctx.hpp; DESIGN.md section 6 says why).  Round trips at the chunk boundaries."""
import numpy as np
import pytest

from tracer_amd import abi, host

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("W,H", [(1, 1), (3, 5), (512, 511), (512, 512), (512, 513), (1024, 1024), (1024, 1537), (1920, 1080)])
def test_frame_round_trips_at_the_chunk_boundaries(gpu, cornell, W, H):
    """RGBA32F / RGBA32Uint frames of exactly one chunk (512 x 512 x 16 B = 4 MiB), one texel row less and more, two chunks, an odd number
    of chunks and a bit: what is uploaded is what comes back, byte for byte, into fresh and into reused destination memory"""
    rs = np.random.RandomState(W * 7919 + H)
    gpu.upload_scene(cornell.view); gpu.resize(W, H)
    acc = rs.standard_normal((H, W, 4)).astype(np.float32)
    rng = rs.randint(0, 2**32, size=(H, W, 4), dtype=np.uint64).astype(np.uint32)
    gpu.upload_accum(acc); gpu.upload_rng(rng)
    for _ in range(2):
        assert np.array_equal(gpu.download_accum().view(np.uint32), acc.view(np.uint32))
        assert np.array_equal(gpu.download_rng(), rng)
    # ... and the other way round after the device changed them: seed + clear, then the known texture of trc_seed
    gpu.seed(123); gpu.clear_accum()
    assert np.array_equal(gpu.download_rng(), host.fill_rng(123, W, H)) and not gpu.download_accum().any()


def test_large_ray_batches_cross_several_chunks(gpu, cornell_spheres):
    """trc_trace_rays: 400 000 rays = 12.8 MB up, 35 MB of hit records down -- against the same batch in pieces"""
    from conftest import camera_rays
    gpu.upload_scene(cornell_spheres.view)
    rays = camera_rays(host.prepare_camera(800, 500), 800, 500, step=1)
    assert len(rays) == 400000
    whole = gpu.trace_rays(rays)
    parts = np.concatenate([gpu.trace_rays(rays[i:i + 70001]) for i in range(0, len(rays), 70001)])
    assert whole.tobytes() == parts.tobytes() and int(whole["hit"].sum()) > 100000
