"""Clip-level analysis loop over the C ABI: the data flow of GenerateOpticalFlowDatabase
(reference cpp/opticalflow.cc:209-321) without the database -- records are handed to a sink.

For every frame1: make frame1 and its valid +-{1,2,4,8} neighbours resident (gray + pyramid, once per
frame instead of once per pair), detect keypoints (or take the ones supplied on resume), track into
all targets in one launch, emit  keypoints(frame1)  and  flow(frame1 -> frame2)  records.  Jobs are
pipelined through pc_analyzer: frame f+1 is submitted before frame f is collected.
"""
from __future__ import annotations

from typing import Callable, Iterable

from . import hip

IMAGE_SKIPS = (-8, -4, -2, -1, 1, 2, 4, 8)  # reference cpp/opticalflow.cc:76-77
RING = 17                                    # reference SequentialWrapper<17>, opticalflow_thread.h:34-79
LOOKAHEAD = 1                                # PC_ANALYZER_LOOKAHEAD: frame1 + 9 becomes resident one step early


class ClipAnalyzer:
    """frame_source(frame_id) -> H x W x 3 uint8 (numpy, or torch tensor already on the GPU)."""

    def __init__(self, ctx: hip.Context, width: int, height: int, first_frame: int, num_frames: int,
                 frame_source: Callable[[int], object], gftt: hip.GfttOptions | None = None,
                 flow: hip.FlowOptions | None = None, max_jobs: int = 3):
        self.first, self.end = first_frame, first_frame + num_frames
        self.source = frame_source
        self.an = hip.Analyzer(ctx, width, height, gftt, flow, RING, max_jobs)
        self.max_jobs = max_jobs
        self.highest = None  # highest frame id made resident so far

    def close(self):
        self.an.close()

    def targets_of(self, frame1: int) -> list[int]:
        return [frame1 + s for s in IMAGE_SKIPS if self.first <= frame1 + s < self.end]

    def _ensure_resident(self, upto: int, frame1: int):
        """Frames are requested once each, in increasing order (like SequentialWrapper)."""
        lo = max(self.first, frame1 - 8) if self.highest is None else self.highest + 1
        for fid in range(lo, min(upto, self.end - 1) + 1):
            self.an.put_frame(fid, self.source(fid), will_detect=True)
            self.highest = fid

    def submit(self, frame1: int, targets: Iterable[int] | None = None):
        self._ensure_resident(frame1 + 8 + LOOKAHEAD, frame1)
        tg = list(self.targets_of(frame1) if targets is None else targets)
        self.an.submit(frame1, tg)

    def run(self, frame_ids: Iterable[int], sink: Callable | None = None, copy: bool = True):
        """Process frame1 ids in order, keeping the job pipeline full; sink(frame1, kps, detected, flows)."""
        for f in frame_ids:
            if self.an.pending == self.max_jobs:
                r = self.an.collect(copy)
                if sink:
                    sink(*r)
            self.submit(f)
        while self.an.pending:
            r = self.an.collect(copy)
            if sink:
                sink(*r)
