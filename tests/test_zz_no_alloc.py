"""rt_ppm_encode_static (the encoder that allocates nothing: its workspace is a fixed array in the
library's device image) against rt_ppm_encode, through the C ABI: the files must be equal byte for
byte, on calls in a row that share the workspace, the later ones with smaller frames."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def encode(pkg, frame, static):
    h, w, _ = frame.shape
    cap = pkg.bindings.ppm_max_bytes(w, h)
    out = torch.full((cap,), 0x7E, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    n = pkg.bindings.ppm_encode_device(frame.data_ptr(), w, h, out.data_ptr(), cap, static=static)
    return out[:n].cpu().numpy().tobytes()


def test_static_workspace_encoder_equals_the_allocating_one(pkg):
    rng = np.random.default_rng(5)
    # two calls in a row, the second with a smaller frame (the workspace keeps the first call's contents),
    # then a frame of less than one chunk and a larger one again
    for h, w in ((1080, 1920), (270, 481), (3, 5), (540, 960)):
        pick = rng.integers(0, 3, (h, w, 3))
        img = np.choose(pick, [rng.integers(0, 10, pick.shape), rng.integers(10, 100, pick.shape),
                               rng.integers(100, 256, pick.shape)]).astype(np.uint8)
        frame = torch.from_numpy(img).to("cuda:0")
        want = encode(pkg, frame, static=False)
        got = encode(pkg, frame, static=True)
        assert len(got) == len(want)
        assert got == want
        assert got == pkg.bindings.ppm_encode(img)  # and both are the host writer's file


def test_static_workspace_encoder_refuses_what_does_not_fit(pkg):
    amd, _ = pkg.bindings.libraries()
    limit = int(amd.rt_ppm_static_max_pixels())
    assert limit == 3840 * 2160
    w, h = 3840, 2161
    frame = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda:0")
    cap = pkg.bindings.ppm_max_bytes(w, h)
    out = torch.empty((cap,), dtype=torch.uint8, device="cuda:0")
    with pytest.raises(pkg.bindings.RtError) as e:
        pkg.bindings.ppm_encode_device(frame.data_ptr(), w, h, out.data_ptr(), cap, static=True)
    assert e.value.status == -2 and "exceed" in e.value.message
    # the largest frame that fits
    frame = torch.full((2160, 3840, 3), 200, dtype=torch.uint8, device="cuda:0")
    cap = pkg.bindings.ppm_max_bytes(3840, 2160)
    n = pkg.bindings.ppm_encode_device(frame.data_ptr(), 3840, 2160, out.data_ptr(), cap, static=True)
    m = pkg.bindings.ppm_encode_device(frame.data_ptr(), 3840, 2160, out.data_ptr(), cap, static=False)
    assert n == m == len(b"P3\n3840 2160\n255\n") + 3840 * 2160 * 12 + 0
