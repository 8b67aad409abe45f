"""Canvas::save on the device (rt_ppm_encode through the C ABI) against the oracle's writer, which
follows src/canvas.rs:47-94 statement by statement: the files must be identical byte for byte."""
import numpy as np
import pytest
import torch

import oracle as orc

pytestmark = pytest.mark.gpu


def reference_file(img):
    """The oracle's Canvas::save for colours that quantise to img (ceil(c * 255) == value)."""
    rgb64 = np.where(img > 0, (img.astype(np.float64) - 0.5) / 255.0, 0.0)
    assert np.array_equal(orc.quantise(rgb64), img)
    return orc.ppm_encode(rgb64)


def encode_on_device(pkg, img):
    h, w, _ = img.shape
    dev = torch.from_numpy(np.ascontiguousarray(img)).to("cuda:0") if img.size else torch.zeros(1, dtype=torch.uint8, device="cuda:0")
    cap = pkg.bindings.ppm_max_bytes(w, h)
    out = torch.full((cap,), 0x7E, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    n = pkg.bindings.ppm_encode_device(dev.data_ptr(), w, h, out.data_ptr(), cap)
    return out[:n].cpu().numpy().tobytes()


def images():
    rng = np.random.default_rng(11)
    yield "zeros", np.zeros((9, 37, 3), np.uint8)                       # one digit everywhere: lines of 12 pixels
    yield "white", np.full((5, 41, 3), 255, np.uint8)                   # three digits everywhere
    yield "one pixel", np.array([[[7, 77, 177]]], np.uint8)
    yield "one row", rng.integers(0, 256, (1, 200, 3), dtype=np.uint8)
    yield "one column", rng.integers(0, 256, (130, 1, 3), dtype=np.uint8)
    yield "63 pixels", rng.integers(0, 256, (7, 9, 3), dtype=np.uint8)  # less than a chunk
    yield "64 pixels", rng.integers(0, 256, (8, 8, 3), dtype=np.uint8)
    yield "65 pixels", rng.integers(0, 256, (5, 13, 3), dtype=np.uint8)
    yield "random", rng.integers(0, 256, (67, 129, 3), dtype=np.uint8)
    # digit counts drawn independently so that every branch of the writer and every column occurs
    pick = rng.integers(0, 3, (150, 211, 3))
    yield "mixed widths", np.choose(pick, [rng.integers(0, 10, pick.shape), rng.integers(10, 100, pick.shape),
                                           rng.integers(100, 256, pick.shape)]).astype(np.uint8)
    yield "more than one span", rng.integers(0, 256, (180, 200, 3), dtype=np.uint8)  # 36000 pixels > 16384
    dark = rng.integers(0, 12, (300, 301, 3), dtype=np.uint8)  # mostly one digit, some two
    yield "dark", dark


@pytest.mark.parametrize("name,img", list(images()), ids=[n for n, _ in images()])
def test_device_ppm_is_the_reference_file(pkg, name, img):
    want = reference_file(img)
    got = encode_on_device(pkg, img)
    assert len(got) == len(want)
    assert got == want
    assert pkg.bindings.ppm_encode(img) == want  # the host encoder agrees as well
    assert pkg.bindings.ppm_encode_via_device(img) == want  # host buffers staged through the GPU


def test_device_ppm_of_a_rendered_4k_frame(pkg, root):
    import os
    scene = pkg.Scene.from_file(os.path.join(root, "scenes", "glass.json"), base_dir=root, device=0)
    info = scene.info()
    w, h = info["width"], info["height"]
    frame = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda:0")
    scene.render_device(rgb8_ptr=frame.data_ptr())
    cap = pkg.bindings.ppm_max_bytes(w, h)
    out = torch.empty((cap,), dtype=torch.uint8, device="cuda:0")
    n = pkg.bindings.ppm_encode_device(frame.data_ptr(), w, h, out.data_ptr(), cap)
    got = out[:n].cpu().numpy().tobytes()
    want = pkg.bindings.ppm_encode(frame.cpu().numpy())
    assert got == want
    assert got.startswith(b"P3\n3840 2160\n255\n") and got.endswith(b"\n")
    assert max(len(line) for line in got[:200000].split(b"\n")) <= 73


def test_device_ppm_refuses_a_small_buffer(pkg):
    frame = torch.zeros((4, 4, 3), dtype=torch.uint8, device="cuda:0")
    out = torch.empty((16,), dtype=torch.uint8, device="cuda:0")
    with pytest.raises(pkg.bindings.RtError):
        pkg.bindings.ppm_encode_device(frame.data_ptr(), 4, 4, out.data_ptr(), 16)


def test_canvas_read_write_and_save(pkg, root, tmp_path):
    """The reference's Canvas doctest (src/canvas.rs:104-113) and its out-of-bounds rules, then a
    rendered canvas: read_pixel returns the f64 colours of the render, save writes the reference file."""
    import os
    c = pkg.Canvas(8, 8)
    assert c.read_pixel(4, 2) == (0.0, 0.0, 0.0)
    c.write_pixel(4, 2, (1.0, 0.0, 1.0))
    assert c.read_pixel(4, 2) == (1.0, 0.0, 1.0)
    c.write_pixel(8, 2, (1.0, 1.0, 1.0))  # silently ignored (src/canvas.rs:115-118)
    c.write_pixel(2, 8, (1.0, 1.0, 1.0))
    assert c.read_pixel(8, 2) is None and c.read_pixel(0, 8) is None
    assert c.pixels().sum() == 2.0

    path = os.path.join(root, "scenes", "glass.json")
    gpu = pkg.Scene.from_file(path, base_dir=root, device=0)
    gpu.set_canvas(97, 61)
    cpu = orc.OracleScene.from_file(path, base_dir=root)
    cpu.set_canvas(97, 61)
    canvas = gpu.render_canvas(jobs=1)
    assert (canvas.width, canvas.height) == (97, 61)
    out, _ = gpu.render()
    assert np.array_equal(canvas.pixels(), out["rgb64"])
    assert canvas.read_pixel(96, 60) == tuple(out["rgb64"][60, 96])
    assert canvas.read_pixel(97, 0) is None
    want = cpu.render(jobs=8)
    assert np.abs(canvas.pixels() - want["rgb64"]).max() <= 1e-9
    file = tmp_path / "canvas.ppm"
    canvas.save(file)
    assert file.read_bytes() == orc.ppm_encode(canvas.pixels())
    canvas.write_pixel(0, 0, (float("nan"), -1.0, 2.0))  # NaN -> 0, clamp (src/canvas.rs:60-62)
    canvas.save(file)
    assert file.read_bytes() == orc.ppm_encode(canvas.pixels())
    assert file.read_bytes().split(b"\n")[3].startswith(b"0 0 255 ")
