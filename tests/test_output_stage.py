"""Output stage: exposure + ACES of fragmentShader (Render.metal:29-75) in the oracle, the PNG
writer of the host library, and the device path against the oracle."""
import numpy as np
import pytest

from oracle import pyoracle
from tracer_amd import abi, host


def aces(c, e):
    c = np.float32(c) * np.float32(e)
    return (c * (np.float32(2.51) * c + np.float32(0.03))) / (c * (np.float32(2.43) * c + np.float32(0.59)) + np.float32(0.14))


def test_tonemap_known_answers():
    img = np.zeros((4, 6, 4), np.float32)
    img[..., 0], img[..., 1], img[..., 2], img[..., 3] = 0.5, 0.25, 1.0, 1.0
    out, e = pyoracle.tonemap(img)
    luma = np.float32(0.5) * np.float32(0.2126) + np.float32(0.25) * np.float32(0.7152) + np.float32(1.0) * np.float32(0.0722)
    assert abs(e - np.exp(-luma)) < 1e-6                       # expose = 1 - (1 - exp(-luma))
    want = [int(np.clip(aces(v, e), 0, 1) * 255 + 0.5) for v in (0.5, 0.25, 1.0)]
    assert (out[..., :3] == np.array(want, np.uint8)).all() and (out[..., 3] == 255).all()
    # black frame: exposure 1, black out; NaN / negative / huge pixels are tolerated
    out, e = pyoracle.tonemap(np.zeros((2, 2, 4), np.float32))
    assert e == 1.0 and (out[..., :3] == 0).all()
    bad = np.zeros((2, 2, 4), np.float32); bad[0, 0, 0] = np.nan; bad[0, 1, 1] = -3; bad[1, 0, 2] = 1e30
    out, e = pyoracle.tonemap(bad)
    assert 0.0 < e <= 1.0 and out[1, 0, 0] == 0                # rows are flipped: frame row 0 is the last PNG row
    # vertical flip
    ramp = np.zeros((3, 1, 4), np.float32); ramp[:, 0, 0] = [0.0, 0.1, 5.0]
    out, _ = pyoracle.tonemap(ramp)
    assert out[0, 0, 0] > out[1, 0, 0] > out[2, 0, 0] == 0


def test_png_writer_round_trip(tmp_path):
    from PIL import Image
    rs = np.random.RandomState(1)
    for shape in ((1, 1, 4), (7, 13, 4), (300, 257, 4)):       # the last one spans several 65535-byte stored blocks
        img = rs.randint(0, 256, size=shape).astype(np.uint8)
        f = tmp_path / f"t{shape[0]}.png"
        host.write_png(str(f), img)
        back = np.asarray(Image.open(f).convert("RGBA"))
        assert back.shape == img.shape and (back == img).all()


@pytest.mark.gpu
def test_device_tonemap_equals_oracle(gpu, cornell_spheres):
    W, H = 200, 120
    gpu.upload_scene(cornell_spheres.view); gpu.set_camera(host.prepare_camera(W, H)); gpu.resize(W, H)
    gpu.seed(3); gpu.clear_accum(); gpu.render(spp=8)
    acc = gpu.download_accum()
    got, e = gpu.tonemap()
    want, e_ref = pyoracle.tonemap(acc)
    assert e == e_ref and (got == want).all()
    assert got[..., :3].max() > 100                                # the light is in the frame
