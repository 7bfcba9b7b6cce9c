"""VideoData front end of the path: TEST-phase frame sampling + the GPU input stage.

Restates the parts of the reference's ``VideoDataLayer`` that define *what the net sees*
(caffe_3d/src/caffe/layers/video_data_layer.cpp:134-238, util/io.cpp:368-421,
data_transformer.cpp:147-330); JPEG decoding / resizing (OpenCV) is outside the path and stays
with the caller, which hands over decoded uint8 frames.

* ``test_segment_offsets`` -- which frames of a video a TEST-phase clip uses.
* ``VideoInput`` -- uint8 ``[F, H, W, 3]`` (OpenCV BGR, interleaved) -> the net's ``data`` blob
  (planar fp32, centre-cropped, mean-subtracted) with one HIP kernel, so only 1 byte per sample
  crosses PCIe instead of 4.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np

from . import hip

ECO_MEAN_BGR = (104.0, 117.0, 123.0)  # mean_value entries of models_ECO_*/kinetics/ECO_*.prototxt


def test_segment_offsets(num_frames: int, num_segments: int, new_length: int = 1) -> List[int]:
    """0-based offsets of the first frame of every segment in TEST phase
    (video_data_layer.cpp:147-149,179-184): average_duration = num_frames / num_segments (integer
    division); offset_i = (average_duration - new_length + 1)/2 + i*average_duration, or 0 when the
    segment is shorter than new_length.  The reference then reads image files ``offset + 1 ...
    offset + new_length`` (1-based names, util/io.cpp:379-381)."""
    if num_frames <= 0 or num_segments <= 0 or new_length <= 0:
        raise ValueError("num_frames, num_segments and new_length must be positive")
    avg = int(num_frames) // int(num_segments)
    out = []
    for i in range(num_segments):
        out.append((avg - new_length + 1) // 2 + i * avg if avg >= new_length else 0)
    return out


def center_crop_offsets(height: int, width: int, crop: int):
    """TEST-phase crop origin (data_transformer.cpp:236-241)."""
    if height < crop or width < crop:
        raise ValueError(f"frame {height}x{width} is smaller than the {crop}x{crop} crop")  # CHECK_GE in the reference
    return (height - crop) // 2, (width - crop) // 2


class VideoInput:
    """Fills a net's ``data`` blob from decoded uint8 frames on the GPU."""

    def __init__(self, net, blob: str = "data", mean: Sequence[float] = ECO_MEAN_BGR, scale: float = 1.0,
                 mirror: bool = False) -> None:
        self.net = net
        self.blob = blob
        self.mean = tuple(float(m) for m in mean)
        self.scale = float(scale)
        self.mirror = bool(mirror)
        self._staging = None

    def load(self, frames, h_off: Optional[int] = None, w_off: Optional[int] = None) -> None:
        """``frames``: uint8 ``[F, H, W, 3]`` ndarray (host) or CUDA torch tensor; F, crop size and
        channel count must match the blob ``[F, 3, crop, crop]``."""
        import torch
        b = self.net.blobs[self.blob]
        F, C, ch, cw = b.shape
        if C != 3:
            raise ValueError("VideoInput expects a [F,3,H,W] data blob")
        if isinstance(frames, np.ndarray):
            if frames.dtype != np.uint8:
                raise TypeError("frames must be uint8")
            frames = torch.from_numpy(np.ascontiguousarray(frames))
        if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[3] != 3 or frames.shape[0] != F:
            raise ValueError(f"frames must be uint8 [{F}, H, W, 3], got {tuple(frames.shape)} {frames.dtype}")
        H, W = int(frames.shape[1]), int(frames.shape[2])
        dst = b.tensor
        if frames.device != dst.device:
            if self._staging is None or self._staging.shape != frames.shape:
                self._staging = torch.empty(frames.shape, dtype=torch.uint8, device=dst.device)
            self._staging.copy_(frames, non_blocking=True)
            frames = self._staging
        frames = frames.contiguous()
        if h_off is None or w_off is None:
            ho, wo = center_crop_offsets(H, W, ch) if ch == cw else ((H - ch) // 2, (W - cw) // 2)
            h_off = ho if h_off is None else h_off
            w_off = wo if w_off is None else w_off
        stream = torch.cuda.current_stream(dst.device).cuda_stream
        hip.load().video_input_forward(frames.data_ptr(), dst.data_ptr(), F, H, W, ch, cw, h_off, w_off, self.mean,
                                       self.scale, self.mirror, stream)
