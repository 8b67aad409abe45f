"""rt_ppm_encode_static (the encoder that allocates nothing: its workspace is a fixed array in the
library's device image) against rt_ppm_encode, through the C ABI: the files must be equal byte for
byte, on calls in a row that share the workspace, the later ones with smaller frames."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def encode(pkg, frame, static):
    """static=False: rt_ppm_encode made to allocate its workspace (RT_AMD_PPM_ALLOCATE=1, read per call), as it does
    for frames beyond 4K or when the fixed array is taken: the path that does not touch the fixed array."""
    import os
    h, w, _ = frame.shape
    cap = pkg.bindings.ppm_max_bytes(w, h)
    out = torch.full((cap,), 0x7E, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    if not static:
        os.environ["RT_AMD_PPM_ALLOCATE"] = "1"
    try:
        n = pkg.bindings.ppm_encode_device(frame.data_ptr(), w, h, out.data_ptr(), cap, static=static)
    finally:
        os.environ.pop("RT_AMD_PPM_ALLOCATE", None)
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


def test_frames_beyond_the_fixed_array_allocate_their_workspace(pkg):
    """rt_ppm_encode on a frame one row larger than the fixed array holds takes the hipMalloc path: byte for byte
    the host writer's file."""
    rng = np.random.default_rng(9)
    img = rng.integers(0, 256, (2161, 3840, 3), dtype=np.uint8)
    frame = torch.from_numpy(img).to("cuda:0")
    got = encode(pkg, frame, static=False)
    assert got == pkg.bindings.ppm_encode(img)


def test_calls_that_meet_in_the_fixed_array_take_turns_or_allocate(pkg):
    """One thread keeps rt_ppm_encode_static busy in the GPU's array with 4K frames while two others call
    rt_ppm_encode (which never waits: it allocates when the array is taken) and rt_ppm_encode_static (which
    waits for its turn): every file must be the host writer's, whichever path a call took."""
    import threading
    rng = np.random.default_rng(13)
    big = rng.integers(0, 256, (2160, 3840, 3), dtype=np.uint8)
    small = [rng.integers(0, 256, (90 + 7 * k, 160 + 3 * k, 3), dtype=np.uint8) for k in range(4)]
    want_big = pkg.bindings.ppm_encode(big)
    want_small = [pkg.bindings.ppm_encode(im) for im in small]
    d_big = torch.from_numpy(big).to("cuda:0")
    d_small = [torch.from_numpy(im).to("cuda:0") for im in small]
    torch.cuda.synchronize()
    errors = []

    def worker(frame, want, static, rounds):
        try:
            h, w, _ = frame.shape
            cap = pkg.bindings.ppm_max_bytes(w, h)
            out = torch.empty((cap,), dtype=torch.uint8, device="cuda:0")
            for _ in range(rounds):
                n = pkg.bindings.ppm_encode_device(frame.data_ptr(), w, h, out.data_ptr(), cap, static=static)
                if out[:n].cpu().numpy().tobytes() != want:
                    errors.append("wrong file from %s" % ("rt_ppm_encode_static" if static else "rt_ppm_encode"))
                    return
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(d_big, want_big, True, 6))]
    for k in range(4):
        threads.append(threading.Thread(target=worker, args=(d_small[k], want_small[k], k % 2 == 1, 25)))
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
