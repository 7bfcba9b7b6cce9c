"""Front end of the path: TEST-phase frame sampling, the GPU input stage
kernel (VideoData output contract) and the online-recognition working memory."""
import numpy as np
import pytest

import eco_oracle as orc
from eco_amd import hip, online, video


def test_segment_offsets():
    # video_data_layer.cpp:147,179-184: avg = frames // segments; (avg - new_length + 1)//2 + i*avg
    assert video.test_segment_offsets(160, 16) == [5 + 10 * i for i in range(16)]
    assert video.test_segment_offsets(100, 4) == [12, 37, 62, 87]
    assert video.test_segment_offsets(100, 4, new_length=5) == [10, 35, 60, 85]
    assert video.test_segment_offsets(10, 16) == [0] * 16          # fewer frames than segments: avg = 0
    assert video.test_segment_offsets(16, 16) == list(range(16))
    with pytest.raises(ValueError):
        video.test_segment_offsets(0, 16)
    assert video.center_crop_offsets(256, 340, 224) == (16, 58)     # data_transformer.cpp:239-240
    with pytest.raises(ValueError):
        video.center_crop_offsets(200, 340, 224)


@pytest.mark.parametrize("H,W,ch,cw,ho,wo,mirror", [(12, 17, 8, 9, 2, 4, False), (12, 17, 8, 9, 2, 4, True),
                                                      (16, 16, 16, 16, 0, 0, False), (9, 11, 5, 6, 4, 5, True)])
def test_video_input_kernel(backend, H, W, ch, cw, ho, wo, mirror):
    rng = np.random.default_rng(0)
    F = 3
    frames = rng.integers(0, 256, size=(F, H, W, 3), dtype=np.uint8)
    mean = (104.0, 117.0, 123.0)
    ref = orc.video_transform(frames, ch, cw, ho, wo, mean, 0.5, mirror)
    if backend.kind == "emu":
        buf = np.ascontiguousarray(frames)
        backend._keep.append(buf)
        backend.alloc._dll.emu_register_buffer(buf.ctypes.data, buf.nbytes)
        fptr = buf.ctypes.data
    else:
        import torch
        t = torch.from_numpy(frames).cuda()
        backend._keep.append(t)
        fptr = t.data_ptr()
    y = backend.empty(ref.shape)
    backend.lib.video_input_forward(fptr, backend.ptr(y), F, H, W, ch, cw, ho, wo, mean, 0.5, mirror)
    assert np.array_equal(backend.host(y, ref.shape), ref)          # integer -> float arithmetic: bit-exact
    with pytest.raises(hip.EcoError, match="does not fit"):
        backend.lib.video_input_forward(fptr, backend.ptr(y), F, H, W, ch, cw, H - ch + 1, wo, mean, 1.0, False)


def test_working_memory_scheme():
    """Replays online_recognition.py:60-82 literally and compares slot contents."""
    mem = online.WorkingMemory()
    ref_slots = []
    counter = 0
    for win in range(8):
        frames = list(range(counter, counter + 16))
        counter += 16
        got = mem.add_window(frames)
        ref_slots.append(list(frames))
        n = len(ref_slots)
        if n > 5:
            del ref_slots[0]
            algo = online.SAMPLING_SCHEME[4]
        else:
            algo = online.SAMPLING_SCHEME[n - 1]
        for y in range(len(algo)):
            idx = np.rint(np.linspace(0, len(ref_slots[y]) - 1, algo[y])).astype(np.int16)
            ref_slots[y] = [ref_slots[y][i] for i in idx]
        flat = [f for s in ref_slots for f in s]
        assert got == flat and len(got) == 16
        assert got == sorted(got)                       # oldest first
    assert [len(s) for s in mem.slots] == [1, 1, 2, 4, 8]
    with pytest.raises(ValueError):
        mem.add_window(list(range(5)))


def test_running_average_modes():
    a, b = np.array([1.0, 5.0, 2.0]), np.array([4.0, 0.0, 2.0])
    r = online.update_running(None, a, "paper")
    assert np.array_equal(r, a)
    assert np.allclose(online.update_running(r, b, "paper"), [2.5, 2.5, 2.0])
    s1 = online.update_running(None, a, "script")
    assert np.allclose(s1, a)
    s2 = online.update_running(s1, b, "script")       # (C,1) + (C,) broadcast -> b[i] + mean(s1)
    assert np.allclose(s2, b + a.mean())


@pytest.mark.gpu
def test_online_recognizer_end_to_end():
    """3 windows of 256x340 frames through VideoInput + ECO-Lite on the GPU == oracle on the same clips."""
    from eco_amd import fillers, models
    from eco_amd.net import Net
    from eco_amd.netspec import NetSpec
    proto = models.eco_lite_deploy(num_segments=16, num_clips=1)
    spec = NetSpec.from_prototxt(proto)
    params = fillers.synthetic_params(spec)
    net = Net(proto, params=params)
    vin = video.VideoInput(net)
    rec = online.OnlineRecognizer(net, vin)
    rng = np.random.default_rng(1)
    mem = online.WorkingMemory()
    running = None
    for win in range(3):
        frames = [rng.integers(0, 256, size=(256, 340, 3), dtype=np.uint8) for _ in range(16)]
        pred = None
        for f in frames:
            pred = rec.push(f)
        clip = mem.add_window(frames)
        x = orc.video_transform(np.stack(clip, 0), 224, 224, 16, 60, video.ECO_MEAN_BGR)
        ref = orc.forward(spec, params, {"data": x})["fc8"][0].astype(np.float64)
        running = online.update_running(running, ref)
        assert np.abs(rec.last_logits - ref).max() < 1e-3 * np.abs(ref).max()
        assert pred == int(np.argmax(running))
