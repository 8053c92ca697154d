"""Clip-level analysis loop over the C ABI: the data flow of GenerateOpticalFlowDatabase
(reference cpp/opticalflow.cc:209-321) without the database -- records are handed to a sink.

For every frame1: make frame1 and its valid +-{1,2,4,8} neighbours resident (gray + pyramid, once per
frame instead of once per pair), detect keypoints (or take the ones supplied on resume), track into
all targets in one launch, emit  keypoints(frame1)  and  flow(frame1 -> frame2)  records.
"""
from __future__ import annotations

from typing import Callable, Iterable

import numpy as np

from . import hip

IMAGE_SKIPS = (-8, -4, -2, -1, 1, 2, 4, 8)  # reference cpp/opticalflow.cc:76-77
RING = 17                                    # reference SequentialWrapper<17>, opticalflow_thread.h:34-79


class ClipAnalyzer:
    """frame_source(frame_id) -> H x W x 3 uint8 (numpy, or torch tensor already on the GPU)."""

    def __init__(self, ctx: hip.Context, width: int, height: int, first_frame: int, num_frames: int,
                 frame_source: Callable[[int], object], gftt: hip.GfttOptions | None = None,
                 flow: hip.FlowOptions | None = None):
        self.ctx, self.w, self.h = ctx, width, height
        self.first, self.end = first_frame, first_frame + num_frames
        self.source = frame_source
        self.gftt = gftt or hip.gftt_options()
        self.flow = flow or hip.flow_options()
        self.slots = [hip.Frame(ctx, width, height, self.flow.window_size, self.flow.max_level) for _ in range(RING)]
        self.slot_id = [None] * RING

    def close(self):
        for f in self.slots:
            f.close()
        self.slots = []

    def _resident(self, frame_id: int) -> hip.Frame:
        s = frame_id % RING
        if self.slot_id[s] != frame_id:
            self.slots[s].set_rgb(self.source(frame_id))
            self.slot_id[s] = frame_id
        return self.slots[s]

    def targets_of(self, frame1: int) -> list[int]:
        return [frame1 + s for s in IMAGE_SKIPS if self.first <= frame1 + s < self.end]

    def process(self, frame1: int, known_keypoints: np.ndarray | None = None,
                targets: Iterable[int] | None = None):
        """Returns (keypoints [N,2], detected: bool, {frame2: (src_idx, tgt_xy, err)})."""
        f1 = self._resident(frame1)
        tg_ids = list(self.targets_of(frame1) if targets is None else targets)
        tg = [self._resident(t) for t in tg_ids]
        detected = False
        if known_keypoints is not None and len(known_keypoints) > 0:
            f1.set_keypoints(known_keypoints)
        elif self.slot_kps_valid(frame1):
            pass
        else:
            f1.detect(self.gftt)
            detected = True
            self._kps_frame = frame1
        flows = {}
        if tg:
            res = hip.lk_track_filtered(self.ctx, f1, tg, self.flow)
            flows = dict(zip(tg_ids, res))
        return f1.keypoints(), detected, flows

    def slot_kps_valid(self, frame1: int) -> bool:
        return False
