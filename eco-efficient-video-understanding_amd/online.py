"""Online (streaming) recognition driver -- the reference's webcam demo without the webcam.

Restates the frame-buffer logic of scripts/online_recognition/online_recognition.py:23,48-98:
every ``window`` (16) incoming frames form a new slot of the working memory; the memory keeps at
most 5 slots which are thinned to ``[16] / [8,8] / [4,4,8] / [2,2,4,8] / [1,1,2,4,8]`` frames
(oldest first) by ``rint(linspace(0, len-1, k))`` so that the net always sees 16 frames that cover
an exponentially longer past; the clip is cropped to rows 16:240, cols 60:284 of the 256x340 frame
(:86), mean-subtracted, run through the net, and the logits are averaged with the running
prediction.  cv2 capture / display is out of scope; feed decoded frames with ``push``.

Averaging note: the script computes ``np.mean(prediction + initial_predictions, axis=1)`` on a
``(C,1)`` array; from the second window on ``initial_predictions`` is ``(C,)`` and broadcasting
turns that into ``prediction[i] + mean(initial_predictions)`` -- a constant shift that leaves the
arg-max equal to the current window's.  ``averaging="paper"`` (default) implements what the ECO
paper describes, P_A <- (P_N + P_A) / 2; ``averaging="script"`` reproduces the script literally.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np

SAMPLING_SCHEME = [[16], [8, 8], [4, 4, 8], [2, 2, 4, 8], [1, 1, 2, 4, 8]]  # online_recognition.py:23


def thin(frames: list, k: int) -> list:
    """Keep k of the frames: indices rint(linspace(0, len-1, k)) (online_recognition.py:76-78)."""
    idx = np.rint(np.linspace(0, len(frames) - 1, k)).astype(np.int16)
    return [frames[i] for i in idx]


class WorkingMemory:
    """The slot buffer; ``add_window`` returns the 16 frames to classify (oldest first)."""

    def __init__(self, window: int = 16) -> None:
        if window != 16:
            raise ValueError("the reference scheme is defined for 16-frame windows")
        self.window = window
        self.slots: List[list] = []

    def add_window(self, frames: list) -> list:
        if len(frames) != self.window:
            raise ValueError(f"a window has {self.window} frames")
        self.slots.append(list(frames))
        n = len(self.slots)
        if n > 5:
            del self.slots[0]
            scheme = SAMPLING_SCHEME[4]
        else:
            scheme = SAMPLING_SCHEME[n - 1]
        for y, k in enumerate(scheme):
            self.slots[y] = thin(self.slots[y], k)
        return [f for slot in self.slots for f in slot]


class OnlineRecognizer:
    def __init__(self, net, video_input, crop_origin=(16, 60), averaging: str = "paper") -> None:
        if averaging not in ("paper", "script"):
            raise ValueError("averaging must be 'paper' or 'script'")
        self.net = net
        self.video_input = video_input
        self.crop_origin = crop_origin
        self.averaging = averaging
        self.memory = WorkingMemory(16)
        self._pending: list = []
        self.running: Optional[np.ndarray] = None
        self.last_logits: Optional[np.ndarray] = None

    def push(self, frame: np.ndarray) -> Optional[int]:
        """Feed one decoded uint8 HxWx3 frame; returns the predicted class index whenever a
        16-frame window completes, else None."""
        self._pending.append(frame)
        if len(self._pending) < self.memory.window:
            return None
        clip = self.memory.add_window(self._pending)
        self._pending = []
        self.video_input.load(np.stack(clip, 0), h_off=self.crop_origin[0], w_off=self.crop_origin[1])
        self.net.forward_device()
        logits = self.net.blobs[self.net.outputs[0]].data[0].astype(np.float64).copy()
        self.last_logits = logits
        self.running = update_running(self.running, logits, self.averaging)
        return int(np.argmax(self.running))


def update_running(running: Optional[np.ndarray], logits: np.ndarray, averaging: str = "paper") -> np.ndarray:
    if averaging == "paper":
        return logits.copy() if running is None else 0.5 * (logits + running)
    # literal script: prediction (C,1) + initial (C,1) zeros first, then (C,) -> broadcast (C,C), mean over axis 1
    pred = logits.reshape(-1, 1)
    init = np.zeros_like(pred) if running is None else running
    return np.mean(pred + init, axis=1)
