"""Frame sharding + flow-database stitch for multi-GPU analysis.

Detection + LK are independent per frame1 (the only coupling is read-only access to frames within
+-8), so each rank (one process per GPU) owns a contiguous range of frame1 ids and additionally
ingests an 8-frame halo on each side as targets only.  No collective on the data path.  The flow
database is stitched once at the end with an all-gather of the packed records
(torch.distributed: backend "nccl" == RCCL over xGMI on the GPU box, "gloo" in the CPU tests), so every
rank holds the complete record set (rank 0 writes SQLite; tracking can start anywhere).

Record order and bytes are independent of the number of ranks.
"""
from __future__ import annotations

import os

import numpy as np

HALO = 8  # largest |skip| of the reference (cpp/opticalflow.cc:76-77)


def shard_range(first_frame: int, num_frames: int, world: int, rank: int) -> tuple[int, int]:
    """frame1 ids [begin, end) owned by `rank`: first + r*n/R ... (SURVEY 8(e) partitioning)."""
    begin = first_frame + (num_frames * rank) // world
    end = first_frame + (num_frames * (rank + 1)) // world
    return begin, end


def resident_range(begin: int, end: int, first_frame: int, num_frames: int) -> tuple[int, int]:
    """frames a rank must ingest: its shard plus the halo, clipped to the clip."""
    return max(first_frame, begin - HALO), min(first_frame + num_frames, end + HALO)


def pack_records(records) -> tuple[np.ndarray, np.ndarray]:
    """records: iterable of (frame1, kps [N,2] f32, {frame2: (idx u32 [M], xy f32 [M,2], err f32 [M])}).
    -> (header int64 [n, 3 + 16], payload uint8)."""
    hdr, blobs = [], []
    for frame1, kps, flows in records:
        row = [int(frame1), len(kps), len(flows)]
        blobs.append(np.ascontiguousarray(kps, np.float32).view(np.uint8).ravel())
        items = sorted(flows.items())
        for f2, (idx, xy, err) in items:
            row += [int(f2), len(idx)]
            blobs += [np.ascontiguousarray(idx, np.uint32).view(np.uint8).ravel(),
                      np.ascontiguousarray(xy, np.float32).view(np.uint8).ravel(),
                      np.ascontiguousarray(err, np.float32).view(np.uint8).ravel()]
        row += [0, 0] * (8 - len(items))
        hdr.append(row)
    header = np.array(hdr, np.int64).reshape(-1, 19)
    payload = np.concatenate(blobs) if blobs else np.zeros(0, np.uint8)
    return header, payload


def unpack_records(header: np.ndarray, payload: np.ndarray):
    out, o = [], 0
    for row in header:
        frame1, n_kps, n_t = int(row[0]), int(row[1]), int(row[2])
        kps = payload[o:o + n_kps * 8].view(np.float32).reshape(n_kps, 2).copy()
        o += n_kps * 8
        flows = {}
        for t in range(n_t):
            f2, m = int(row[3 + 2 * t]), int(row[4 + 2 * t])
            idx = payload[o:o + 4 * m].view(np.uint32).copy()
            o += 4 * m
            xy = payload[o:o + 8 * m].view(np.float32).reshape(m, 2).copy()
            o += 8 * m
            err = payload[o:o + 4 * m].view(np.float32).copy()
            o += 4 * m
            flows[f2] = (idx, xy, err)
        out.append((frame1, kps, flows))
    assert o == len(payload)
    return out


def all_gather_packed(header: np.ndarray, payload: np.ndarray, device=None, group=None):
    """All-gather of one rank's packed records. Returns (headers, payloads) lists indexed by rank.
    Two collectives: sizes (2 x int64 per rank), then the padded byte payloads."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    dev = device if device is not None else torch.device("cpu")
    local = torch.from_numpy(np.concatenate([header.view(np.uint8).ravel(), payload])).to(dev)
    sizes = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([header.shape[0], payload.shape[0]], dtype=torch.int64, device=dev), group=group)
    totals = [int(s[0]) * 19 * 8 + int(s[1]) for s in sizes]
    mx = max(totals) if totals else 0
    padded = torch.zeros(max(mx, 1), dtype=torch.uint8, device=dev)
    padded[:local.numel()] = local
    gathered = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(gathered, padded, group=group)
    headers, payloads = [], []
    for r in range(world):
        buf = gathered[r].cpu().numpy()
        nh = int(sizes[r][0]) * 19 * 8
        headers.append(buf[:nh].view(np.int64).reshape(-1, 19).copy())
        payloads.append(buf[nh:nh + int(sizes[r][1])].copy())
    return headers, payloads


def all_gather_records(records, device=None, group=None):
    """-> the records of all ranks, in frame1 order (ranks own increasing frame ranges)."""
    h, p = pack_records(records)
    hs, ps = all_gather_packed(h, p, device, group)
    out = []
    for hh, pp in zip(hs, ps):
        out += unpack_records(hh, pp)
    out.sort(key=lambda r: r[0])
    return out


def write_records(db, records):
    """Store stitched records through a polychase_core.Database (keypoints row first: FK)."""
    for frame1, kps, flows in records:
        if not db.keypoints_exist(frame1):
            db.write_keypoints(frame1, kps)
        for f2, (idx, xy, err) in sorted(flows.items()):
            if not db.image_pair_flow_exists(frame1, f2):
                db.write_image_pair_flow(frame1, f2, idx, xy, err)


# ----------------------------------------------------------------------------------------------
# Device-resident path (bench / production multi-GPU): the analyzer appends every job's records to
# a device log (pc_analyzer_set_device_log); the stitch is ONE size exchange + ONE all-gather of
# the raw log bytes over RCCL, with no host copy of the payload.
# ----------------------------------------------------------------------------------------------
LOG_MAGIC = 0x50434C4F47303031


def log_capacity_bytes(n_frames: int, max_keypoints: int, n_targets: int = 8) -> int:
    per = 256 + max_keypoints * 8 + max_keypoints * n_targets * 16 + 64
    return n_frames * per


def all_gather_device_log(log, used: int, group=None):
    """log: uint8 tensor (CUDA under RCCL; CPU under gloo in the tests), first `used` bytes valid.  Returns
    (gathered [world, max_used] uint8 on the same device, sizes list).  Collectives: all_gather of one int64, then
    the payload -- all_gather_into_tensor straight into the result under NCCL/RCCL."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    dev = log.device
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([used], dtype=torch.int64, device=dev), group=group)
    sizes = [int(s.item()) for s in sizes]
    mx = max(16, (max(sizes) + 15) // 16 * 16)
    if mx > log.numel():     # another rank's shard is longer than this rank's whole buffer: pad a copy
        src = torch.zeros(mx, dtype=torch.uint8, device=dev)
        src[:used] = log[:used]
    else:
        src = log[:mx]       # bytes past `used` are padding, never parsed
    out = torch.empty((world, mx), dtype=torch.uint8, device=dev)
    if dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(out.view(-1), src, group=group)
    else:
        dist.all_gather(list(out.unbind(0)), src.contiguous(), group=group)
    return out, sizes


def pack_device_log(records) -> np.ndarray:
    """records (frame1, kps [N,2] f32, {frame2: (idx, xy, err)}) -> the bytes pc_analyzer_set_device_log would have
    produced for them (include/polychase_hip.h), one record after the other."""
    up16 = lambda v: (v + 15) & ~15
    chunks = []
    for frame1, kps, flows in records:
        items = sorted(flows.items())
        kps = np.ascontiguousarray(kps, np.float32).reshape(-1, 2)
        rows = len(kps) * len(items)
        hdr = np.zeros(16, np.int64)
        hdr[0], hdr[1], hdr[2], hdr[3], hdr[12] = LOG_MAGIC, frame1, len(kps), len(items), rows
        off = np.zeros(16, np.int64)
        idx, xy, err = np.zeros(rows, np.uint32), np.zeros((rows, 2), np.float32), np.zeros(rows, np.float32)
        o = 0
        for t, (f2, (i_, x_, e_)) in enumerate(items):
            hdr[4 + t] = f2
            m = len(i_)
            idx[o:o + m], xy[o:o + m], err[o:o + m] = i_, np.asarray(x_, np.float32).reshape(-1, 2), e_
            o += m
            off[t + 1] = o
        rec = bytearray(hdr.tobytes() + off.tobytes() + kps.tobytes())
        for part in (idx.tobytes(), xy.tobytes(), err.tobytes()):
            rec += b"\0" * (up16(len(rec)) - len(rec)) + part
        rec += b"\0" * (up16(len(rec)) - len(rec))
        chunks.append(bytes(rec))
    return np.frombuffer(b"".join(chunks), np.uint8).copy()


class ChunkedLogStitch:
    """All-gathers a device log piece by piece WHILE the analysis keeps running, so that only the last piece
    is exposed at the end (a single all-gather of a whole shard's log costs 10-40 ms per 100 frames at 8 ranks,
    a sizeable part of the step time; in pieces it hides behind the LK launches of the following frames).

    gather(start, end): bytes [start, end) of the local log are complete (the job that wrote the last of them has
    been collected).  The piece sizes are agreed on through `side_group` -- a gloo group, so that the exchange does not
    queue behind the previous piece's payload on the NCCL stream -- then the payload all-gather is enqueued and the
    call returns.  Without a process group the pieces are views of the local log."""

    name = "rccl all_gather"

    def finish(self):
        """Nothing to do here: the payload all-gathers are stream-ordered, the caller's device synchronisation ends them."""

    def close(self):
        pass

    def __init__(self, log, group=None, side_group=None):
        import torch.distributed as dist

        self.log, self.group, self.side = log, group, side_group
        self.dist = dist if (dist.is_available() and dist.is_initialized()) else None
        self.world = self.dist.get_world_size(group) if self.dist else 1
        # no process group: the pieces are views of the local log.  A ONE-rank group still runs every collective (the
        # RCCL calls of the N-rank job on a one-GPU test box, tests/test_rccl_gpu.py).
        self.local_only = self.dist is None
        self.pieces = []   # (gathered [world, padded] uint8, sizes [world])
        self._pool = []    # receive buffers allocated ahead of the timed region (reserve)

    def reserve(self, n_pieces: int, max_piece_bytes: int):
        """Allocates the receive buffers of the next `n_pieces` gathers now (world x max_piece_bytes each): a fresh
        allocation of half a gigabyte inside the timed region is a hipMalloc that stalls the device."""
        import torch

        if self.local_only:
            return
        mx = max(16, (int(max_piece_bytes) + 15) // 16 * 16)
        self._reserved = [torch.empty(self.world * mx, dtype=torch.uint8, device=self.log.device) for _ in range(n_pieces)]
        self._pool = list(self._reserved)

    def reset(self):
        """Forget the pieces gathered so far (the log is about to be reused from offset 0) and take their receive
        buffers back into the pool."""
        self.pieces = []
        self._pool = list(getattr(self, "_reserved", []))

    def warm_up(self):
        """One tiny exchange per group before anything is timed: the first collective of a communicator sets up its
        connections (hundreds of milliseconds for RCCL)."""
        import torch

        if self.local_only:
            return
        dist = self.dist
        if self.side is not None:
            dist.all_gather([torch.zeros(1, dtype=torch.int64) for _ in range(self.world)], torch.zeros(1, dtype=torch.int64),
                            group=self.side)
        dev = self.log.device
        tiny = torch.zeros(16, dtype=torch.uint8, device=dev)
        out = torch.empty((self.world, 16), dtype=torch.uint8, device=dev)
        if dist.get_backend(self.group) == "nccl":
            dist.all_gather_into_tensor(out.view(-1), tiny, group=self.group)
        else:
            dist.all_gather(list(out.unbind(0)), tiny, group=self.group)
        dist.all_gather([torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(self.world)],
                        torch.zeros(1, dtype=torch.int64, device=dev), group=self.group)

    def gather(self, start: int, end: int):
        import torch

        n = end - start
        if self.local_only:
            self.pieces.append((self.log[start:end][None], [n]))
            return
        dist = self.dist
        if self.side is not None:   # CPU tensors over gloo: independent of whatever the NCCL stream is doing
            sizes_t = [torch.zeros(1, dtype=torch.int64) for _ in range(self.world)]
            dist.all_gather(sizes_t, torch.tensor([n], dtype=torch.int64), group=self.side)
        else:                       # no side group: same exchange on the main group (waits for the previous piece)
            dev = self.log.device
            sizes_t = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(self.world)]
            dist.all_gather(sizes_t, torch.tensor([n], dtype=torch.int64, device=dev), group=self.group)
        sizes = [int(t.item()) for t in sizes_t]
        mx = max(16, (max(sizes) + 15) // 16 * 16)
        if start + mx > self.log.numel():
            raise RuntimeError("device log has no slack for the padded piece")
        if self._pool and self._pool[-1].numel() >= self.world * mx:
            out = self._pool.pop()[: self.world * mx].view(self.world, mx)
        else:
            out = torch.empty((self.world, mx), dtype=torch.uint8, device=self.log.device)
        # bytes past `end` may still be written by the next frames: they are padding, never parsed
        piece = self.log[start:start + mx]
        if dist.get_backend(self.group) == "nccl":
            dist.all_gather_into_tensor(out.view(-1), piece, group=self.group)   # straight into `out`, no staging copy
        else:
            dist.all_gather(list(out.unbind(0)), piece, group=self.group)        # gloo (CPU tests)
        self.pieces.append((out, sizes))

    def rank_logs(self):
        """-> per rank: (numpy uint8 log, used bytes), the pieces concatenated in order."""
        out = []
        for r in range(self.world):
            parts = [g[r, :sz[r]].cpu().numpy() for g, sz in self.pieces]
            buf = np.concatenate(parts) if parts else np.zeros(0, np.uint8)
            out.append((buf, len(buf)))
        return out


class PeerLogStitch:
    """The same all-gather as ChunkedLogStitch -- every rank ends up with every rank's record log, piece by piece, while
    the analysis keeps running -- moved by the COPY ENGINES instead of a collective kernel: every rank allocates a receive
    buffer and exports it (HIP IPC: include/polychase_hip.h, pc_peer_buffer_*), maps the buffers of all other ranks into
    ITS OWN device's address space, and pushes each complete piece of its log straight into slot [rank] of every peer with
    device-to-device copies issued on streams of its own device -- seven independent point-to-point writes over xGMI, no
    CU involved, and nothing of this process ever runs on a peer's GPU (no context, no queue there: with torch's
    cross-device tensors every process would hold queues on all eight GPUs, and a GPU whose hardware queue slots are
    oversubscribed by several processes time-slices them -- two ranks sharing one GPU with 16 queues each: 10 x slower).

    Why not the RCCL all-gather for this: its kernel (rcclGenericKernel, gfx950 build of this image's librccl.so)
    allocates 261-280 registers per lane and 19.7 KB of LDS per 256-lane workgroup.  The LK launches keep three wavefronts
    of 136 VGPRs resident on every SIMD, 104 registers per lane are free: a collective workgroup gets
    a CU only by keeping LK off it.  Measured with a kernel of that shape beside the running pipeline
    (tools/coresidency_probe.py): an 11 MB copy takes 0.54 ms instead of 0.03 and the step grows by 12 %; the same bytes as
    a device-to-device copy take 0.02 ms and cost nothing.  Copy engines need no wavefront slot.

    Layout: recv[r] is a copy of rank r's log (same offsets), so a piece [start, end) of the local log goes to
    peer.recv[rank][start:end]; finish() waits for this rank's pushes; the barrier that ends the caller's timed region
    then means every slot is complete; rank_logs() exchanges the used sizes (outside the timed region).  Same interface as
    ChunkedLogStitch (reserve / reset / warm_up / gather / finish / rank_logs / close)."""

    name = "xgmi peer copies (copy engines, HIP IPC)"

    def __init__(self, log, recv_ptr: int, peer_ptrs, slot_bytes: int, group=None, side_group=None):
        """Built by make_log_stitch (which exchanges the IPC handles): recv_ptr = this rank's receive buffer
        [world x slot_bytes] (slot_bytes >= every rank's log), peer_ptrs[r] = rank r's receive buffer mapped for this device
        (None for r == rank)."""
        import torch
        import torch.distributed as dist

        from . import hip

        self.L = hip.load()
        self.log, self.group, self.side = log, group, side_group
        self.dist = dist
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.local_only = False
        self.dev = log.device.index or 0
        self.slot = int(slot_bytes)
        self.recv_ptr, self.peer_ptrs = recv_ptr, peer_ptrs
        # Copies on ONE stream run one after the other -- one link at a time, 40-60 GB/s -- and eight ranks at 1080p need
        # 7 x 5.5 MB per 0.31 ms = 124 GB/s out of every GPU (four ranks: 53 GB/s).  So the peers are dealt round-robin onto
        # one stream for two ranks, two up to four ranks, four beyond; their transfers run side by side on different SDMA
        # engines / links.  The first is the NULL stream: it owns a hardware queue already, is otherwise idle while the
        # analysis runs (the analyzer's streams are non-blocking: no implicit synchronisation with it), and copies on it cost
        # the pipeline nothing (measured, tools/coresidency_probe.py; further copy-only streams cost up to 10 % there with
        # device-to-host copies -- an upper bound, section 6 of DESIGN.md).  bench.py raises GPU_MAX_HW_QUEUES so that none of
        # this shares a hardware queue with a job lane (shared, a copy on the null stream costs the step 35-60 %).
        n_streams = 1 if self.world <= 2 else (2 if self.world <= 4 else 4)
        if os.environ.get("POLYCHASE_PEER_PUSH_STREAMS"):
            n_streams = max(1, min(8, int(os.environ["POLYCHASE_PEER_PUSH_STREAMS"])))
        n_streams = min(n_streams, max(1, self.world - 1))
        self.streams = [torch.cuda.default_stream(log.device)] + [torch.cuda.Stream(device=log.device) for _ in range(n_streams - 1)]
        self.used = 0
        self.sizes = None
        # push order: rank+1, rank+2, ... so that at any moment the ranks write to different peers / links
        self.order = [(self.rank + k) % self.world for k in range(1, self.world)]
        self.stream_of = {r: self.streams[i % n_streams] for i, r in enumerate(self.order)}
        # (The pushes are issued by the thread that drives the analyzer.  A pusher thread was tried: every hand-over of the
        # interpreter lock between the two threads costs up to Python's 5 ms switch interval.)

    def _push(self, r: int, dst_off: int, src_ptr: int, nbytes: int):
        from . import hip

        hip._check(self.L.pc_peer_copy_async(self.dev, self.peer_ptrs[r] + self.rank * self.slot + dst_off, src_ptr, nbytes,
                                            self.stream_of[r].cuda_stream or None))

    def _sync(self):
        for st in self.streams:
            st.synchronize()

    def _download(self, r: int, nbytes: int) -> np.ndarray:
        """slot r of this rank's receive buffer"""
        from . import hip

        out = np.empty(nbytes, np.uint8)
        if nbytes:
            hip._check(self.L.pc_peer_buffer_download(self.dev, out.ctypes.data, self.recv_ptr + r * self.slot, nbytes))
        return out

    def reserve(self, n_pieces: int, max_piece_bytes: int):
        pass

    def reset(self):
        self.used = 0
        self.sizes = None

    def warm_up(self):
        pass    # make_log_stitch has probed every link already

    def probe(self) -> bool:
        """One small push to every peer, a meeting, and a look at what arrived: True when slot r of this rank's buffer
        holds rank r's pattern for every r (the first copy to a peer also sets up the mapping).  Collective; every rank
        goes to both meetings whatever happens to it."""
        import torch

        ok = True
        self.probe_error = None
        pattern = None
        try:
            pattern = torch.full((64,), self.rank + 1, dtype=torch.uint8, device=self.log.device)
            zeros = torch.zeros(64, dtype=torch.uint8, device=self.log.device)
            torch.cuda.synchronize(self.log.device)
            for r in range(self.world):      # clear the first bytes of every slot of the own buffer
                from . import hip
                hip._check(self.L.pc_peer_copy_async(self.dev, self.recv_ptr + r * self.slot, zeros.data_ptr(), 64, None))
            self._sync()
        except Exception as e:
            ok, self.probe_error = False, e
        self._meet()
        try:
            for r in self.order:
                self._push(r, 0, pattern.data_ptr(), 64)
            self._sync()
        except Exception as e:
            ok, self.probe_error = False, e
        self._meet()
        try:
            for r in range(self.world):
                if r != self.rank:
                    got = self._download(r, 64)
                    if not bool((got == r + 1).all()):
                        ok = False
                        self.probe_error = RuntimeError(f"rank {self.rank}: slot {r} holds {got[:4].tolist()} instead of {r + 1}")
        except Exception as e:
            ok, self.probe_error = False, e
        return ok

    def _meet(self):
        self.dist.barrier(group=self.side if self.side is not None else self.group)

    def gather(self, start: int, end: int):
        if end <= start:
            return
        src = self.log.data_ptr() + start
        for r in self.order:
            self._push(r, start, src, end - start)
        self.used = max(self.used, end)

    def finish(self):
        """Inside the timed region, after the last gather: this rank's pushes have landed in the peers' memory.  The
        barrier that ends the region (every rank has passed this point) then means every slot is complete."""
        self._sync()

    def rank_logs(self):
        """-> per rank: (numpy uint8 log, used bytes).  Collective (the used sizes are exchanged here, outside the timed
        region); call after finish() and a barrier."""
        import torch

        g = self.side if self.side is not None else self.group
        on_host = self.side is not None or self.dist.get_backend(self.group) != "nccl"
        dev = "cpu" if on_host else self.log.device
        sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(self.world)]
        self.dist.all_gather(sizes, torch.tensor([self.used], dtype=torch.int64, device=dev), group=g)
        self.sizes = [int(t.item()) for t in sizes]
        out = []
        for r in range(self.world):
            buf = self.log[:self.sizes[r]].cpu().numpy() if r == self.rank else self._download(r, self.sizes[r])
            out.append((buf, self.sizes[r]))
        return out

    def close(self):
        """Peers drop their mappings before the owner frees the buffer: meet, unmap, meet, free."""
        self._sync()
        self._meet()
        for ptr in self.peer_ptrs or []:
            if ptr:
                self.L.pc_peer_buffer_close(self.dev, ptr)
        self.peer_ptrs = None
        self._meet()
        if self.recv_ptr:
            self.L.pc_peer_buffer_free(self.dev, self.recv_ptr)
        self.recv_ptr = 0


def make_log_stitch(log, group=None, side_group=None, prefer: str = "rccl"):
    """The stitch of the benchmark's N > 1 path: `prefer` = "rccl" (ChunkedLogStitch) or "peer" (PeerLogStitch when EVERY
    rank can set it up, else the all-gather).  Every rank takes part in every exchange below whatever happened to it
    locally, so a failure on one rank cannot leave the others waiting."""
    import ctypes as C
    import sys

    import torch
    import torch.distributed as dist

    if prefer != "peer" or not log.is_cuda or not (dist.is_available() and dist.is_initialized()):
        return ChunkedLogStitch(log, group=group, side_group=side_group)
    from . import hip

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    g = side_group if side_group is not None else group
    on_host = side_group is not None or dist.get_backend(group) != "nccl"
    dev = log.device.index or 0
    L = None
    recv, handle, err = C.c_void_p(), None, None
    # one slot size for everybody: the ranks size their logs from their own frames' keypoint counts
    slot = torch.tensor([int(log.numel())], dtype=torch.int64, device="cpu" if on_host else log.device)
    dist.all_reduce(slot, op=dist.ReduceOp.MAX, group=g)
    slot = (int(slot.item()) + 255) // 256 * 256
    try:
        L = hip.load()
        if os.environ.get("POLYCHASE_TEST_BREAK_PEER_EXPORT") == str(rank):     # tests: one rank cannot export
            raise RuntimeError("test: buffer export disabled on this rank")
        hip._check(L.pc_peer_buffer_alloc(dev, world * slot, C.byref(recv)))
        hb = C.create_string_buffer(64)
        hip._check(L.pc_peer_buffer_export(dev, recv, hb))
        handle = hb.raw
    except Exception as e:
        err = e
    everyone = [None] * world
    dist.all_gather_object(everyone, handle, group=g)
    peers = [None] * world
    if err is None and all(h is not None for h in everyone):
        try:
            for r in range(world):
                if r != rank:
                    ptr = C.c_void_p()
                    hip._check(L.pc_peer_buffer_open(dev, everyone[r], C.byref(ptr)))
                    peers[r] = ptr.value
        except Exception as e:
            err = e

    def agreed(flag: bool) -> bool:
        ok = torch.tensor([1 if flag else 0], dtype=torch.int64, device="cpu" if on_host else log.device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=g)
        return int(ok.item()) == 1

    def release():
        for ptr in peers:
            if ptr:
                L.pc_peer_buffer_close(dev, ptr)
        dist.barrier(group=g)      # mappings dropped everywhere before the buffers go
        if recv.value:
            L.pc_peer_buffer_free(dev, recv)

    if agreed(err is None and all(peers[r] for r in range(world) if r != rank)):
        st = PeerLogStitch(log, recv.value, peers, slot, group=group, side_group=side_group)
        if agreed(st.probe()):
            return st
        err = st.probe_error or RuntimeError("a probe push did not arrive in a peer's buffer (here or on another rank)")
    if err is not None:
        print(f"[polychase_amd.distributed] peer copies unavailable ({type(err).__name__}: {err}); stitching with the RCCL all-gather",
              file=sys.stderr)
    if L is not None:
        release()
    else:
        dist.barrier(group=g)
    return ChunkedLogStitch(log, group=group, side_group=side_group)


# (The ordered, credit-controlled hand-over of a rank's record log to rank 0 -- the PRODUCT's stitch -- lives in ONE place since round
# 6: csrc/host/multi_gpu.cc, reached through polychase_core.generate_optical_flow_database_multi_gpu; rounds 3-5 also carried a
# Python implementation of the same protocol here, OrderedPieceGather.  What stays in this module is the benchmark's side: the
# all-gather stitches of bench.py --gpus N, and the record format's pack / parse helpers.)
def parse_device_log(buf: np.ndarray, used: int):
    """buf: uint8 array of one rank's log.  -> records like pack_records' input."""
    out, o = [], 0
    up16 = lambda v: (v + 15) & ~15
    while o < used:
        hdr = buf[o:o + 128].view(np.int64)
        assert int(hdr[0]) == LOG_MAGIC, "corrupt device log"
        frame1, n, nt, rows = int(hdr[1]), int(hdr[2]), int(hdr[3]), int(hdr[12])
        off = buf[o + 128:o + 256].view(np.int64)
        o_kps = o + 256
        o_idx = up16(o_kps + n * 8)
        o_xy = up16(o_idx + rows * 4)
        o_err = up16(o_xy + rows * 8)
        kps = buf[o_kps:o_kps + n * 8].view(np.float32).reshape(n, 2).copy()
        idx = buf[o_idx:o_idx + rows * 4].view(np.uint32)
        xy = buf[o_xy:o_xy + rows * 8].view(np.float32).reshape(rows, 2)
        err = buf[o_err:o_err + rows * 4].view(np.float32)
        flows = {}
        for t in range(nt):
            a, b = int(off[t]), int(off[t + 1])
            flows[int(hdr[4 + t])] = (idx[a:b].copy(), xy[a:b].copy(), err[a:b].copy())
        out.append((frame1, kps, flows))
        o = up16(o_err + rows * 4)
    return out
