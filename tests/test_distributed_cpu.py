"""CPU, world_size 2, gloo: the N>1 path of the analysis -- frame sharding with halo and the
all-gather stitch of the flow records -- produces the same record set as a single rank."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from polychase_amd import distributed as D  # noqa: E402


def fake_record(frame1, first, n):
    """Deterministic stand-in for one frame1 result (shape/dtype of the real records)."""
    rng = np.random.default_rng(frame1)
    k = 20 + frame1 % 7
    kps = rng.integers(0, 500, (k, 2)).astype(np.float32)
    flows = {}
    for s in (-8, -4, -2, -1, 1, 2, 4, 8):
        f2 = frame1 + s
        if first <= f2 < first + n:
            m = int(rng.integers(0, k + 1))
            idx = np.sort(rng.choice(k, m, replace=False)).astype(np.uint32)
            flows[f2] = (idx, rng.normal(size=(m, 2)).astype(np.float32), rng.random(m).astype(np.float32))
    return frame1, kps, flows


def test_shard_ranges_partition_the_clip():
    for n in (1, 7, 30, 300, 2400):
        for world in (1, 2, 3, 8):
            got = []
            for r in range(world):
                b, e = D.shard_range(5, n, world, r)
                got += list(range(b, e))
                lo, hi = D.resident_range(b, e, 5, n)
                assert lo <= b and hi >= e and lo >= 5 and hi <= 5 + n
                assert (b - lo == min(8, b - 5)) and (hi - e == min(8, 5 + n - e))
            assert got == list(range(5, 5 + n))


def test_pack_roundtrip():
    recs = [fake_record(f, 1, 12) for f in range(1, 13)]
    h, p = D.pack_records(recs)
    back = D.unpack_records(h, p)
    for a, b in zip(recs, back):
        assert a[0] == b[0] and np.array_equal(a[1], b[1]) and sorted(a[2]) == sorted(b[2])
        for f2 in a[2]:
            assert all(np.array_equal(x, y) for x, y in zip(a[2][f2], b[2][f2]))
    h0, p0 = D.pack_records([])
    assert D.unpack_records(h0, p0) == []


def _worker(rank, world, port, first, n, out_dir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b, e = D.shard_range(first, n, world, rank)
    local = [fake_record(f, first, n) for f in range(b, e)]
    allr = D.all_gather_records(local)
    h, p = D.pack_records(allr)
    np.save(os.path.join(out_dir, f"h{rank}.npy"), h)
    np.save(os.path.join(out_dir, f"p{rank}.npy"), p)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [23, 3])     # 3 frames: rank shards of 1 and 2, empty flows at the clip ends
def test_two_ranks_stitch_to_the_single_rank_result(tmp_path, n):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    first = 4
    mp.spawn(_worker, args=(2, port, first, n, str(tmp_path)), nprocs=2, join=True)
    h_ref, p_ref = D.pack_records([fake_record(f, first, n) for f in range(first, first + n)])
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / f"h{r}.npy"), h_ref)
        assert np.array_equal(np.load(tmp_path / f"p{r}.npy"), p_ref)


def _fake_log(frames, first, n):
    """a device-log image (pc_analyzer_set_device_log layout) of fake records, plus the piece boundaries"""
    up16 = lambda v: (v + 15) & ~15
    chunks, bounds = [], [0]
    for f in frames:
        frame1, kps, flows = fake_record(f, first, n)
        items = sorted(flows.items())
        rows = len(kps) * len(items)
        hdr = np.zeros(16, np.int64)
        hdr[0], hdr[1], hdr[2], hdr[3], hdr[12] = D.LOG_MAGIC, frame1, len(kps), len(items), rows
        off = np.zeros(16, np.int64)
        idx, xy, err = np.zeros(rows, np.uint32), np.zeros((rows, 2), np.float32), np.zeros(rows, np.float32)
        o = 0
        for t, (f2, (i_, x_, e_)) in enumerate(items):
            hdr[4 + t] = f2
            idx[o:o + len(i_)], xy[o:o + len(i_)], err[o:o + len(i_)] = i_, x_, e_
            o += len(i_)
            off[t + 1] = o
        rec = bytearray()
        for part in (hdr.tobytes(), off.tobytes(), kps.tobytes()):
            rec += part
        for part in (idx.tobytes(), xy.tobytes(), err.tobytes()):
            rec += b"\0" * (up16(len(rec)) - len(rec)) + part
        rec += b"\0" * (up16(len(rec)) - len(rec))
        chunks.append(bytes(rec))
        bounds.append(bounds[-1] + len(rec))
    return np.frombuffer(b"".join(chunks), np.uint8).copy(), bounds


def _stitch_worker(rank, world, port, first, n, out_dir, use_side):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    side = dist.new_group(backend="gloo") if use_side else None
    b, e = D.shard_range(first, n, world, rank)
    log_np, bounds = _fake_log(range(b, e), first, n)
    log = torch.zeros(len(log_np) + 4096, dtype=torch.uint8)     # slack for the padded pieces
    log[:len(log_np)] = torch.from_numpy(log_np)
    st = D.make_log_stitch(log, side_group=side, prefer="peer")     # host tensors: the factory must hand out the all-gather stitch
    assert isinstance(st, D.ChunkedLogStitch)
    st.warm_up()
    st.reserve(2, 1 << 15)                                         # two pooled receive buffers, the rest allocated on demand
    per = 3                                                        # frames per piece (the last piece is ragged)
    n_pieces = (max(D.shard_range(first, n, world, r)[1] - D.shard_range(first, n, world, r)[0] for r in range(world)) + per - 1) // per
    for region in range(2):                                        # bench.py reuses the log and the stitch for every timed region
        st.reset()
        for c in range(n_pieces):                                  # every rank issues the same number of collectives
            lo, hi = min(c * per, e - b), min((c + 1) * per, e - b)
            st.gather(bounds[lo], bounds[hi])
        recs = []
        for buf, used in st.rank_logs():
            recs += D.parse_device_log(buf, used)
        assert len(recs) == n, f"region {region}: {len(recs)} records, expected {n}"
    h, p = D.pack_records(sorted(recs, key=lambda r: r[0]))
    np.save(os.path.join(out_dir, f"sh{rank}.npy"), h)
    np.save(os.path.join(out_dir, f"sp{rank}.npy"), p)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("use_side", [True, False])   # sizes over the gloo side group / over the main group
def test_chunked_log_stitch_two_ranks(tmp_path, use_side):
    """The overlapped stitch of the bench's N > 1 path: pieces of the device logs all-gathered one by one, sizes agreed
    on a gloo side group; every rank ends up with every record, identical to the single-rank record set."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    first, n = 4, 17                      # shards of 8 and 9 frames: ragged pieces, one rank with an empty last piece
    mp.spawn(_stitch_worker, args=(2, port, first, n, str(tmp_path), use_side), nprocs=2, join=True)
    ref = []
    for f in range(first, first + n):     # what the log carries: flows keep capacity rows only up to their offsets
        ref.append(fake_record(f, first, n))
    h_ref, p_ref = D.pack_records(ref)
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / f"sh{r}.npy"), h_ref)
        assert np.array_equal(np.load(tmp_path / f"sp{r}.npy"), p_ref)


def test_chunked_log_stitch_one_rank_group(tmp_path):
    """A ONE-rank process group still runs every collective of the stitch (what tests/test_rccl_gpu.py does with RCCL on
    a one-GPU box); only the absence of a process group makes the pieces views of the local log."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    first, n = 4, 7
    mp.spawn(_stitch_worker, args=(1, port, first, n, str(tmp_path), True), nprocs=1, join=True)
    h_ref, p_ref = D.pack_records([fake_record(f, first, n) for f in range(first, first + n)])
    assert np.array_equal(np.load(tmp_path / "sh0.npy"), h_ref) and np.array_equal(np.load(tmp_path / "sp0.npy"), p_ref)


# ----------------------------------------------------------------------------------------------
# records -> database: the product path of the multi-rank analysis (polychase_amd/analyze.py) after the GPU part.
# Rank r holds the record log of its frame range, the logs are all-gathered (gloo here, RCCL on the GPU box), rank 0
# stores them through polychase_core.write_optical_flow_records: the file must equal the single-rank one.
# ----------------------------------------------------------------------------------------------
def _core():
    sys.path.insert(0, os.path.join(ROOT, "polychase_amd", "core"))
    import polychase_core
    return polychase_core


def _db_dump(path):
    import sqlite3
    con = sqlite3.connect(path)
    k = list(con.execute("select rowid, image_id, rows, keypoints from keypoints order by rowid"))
    f = list(con.execute("select rowid, image_id_from, image_id_to, rows, src_keypoints_indices, tgt_keypoints, flow_errors "
                         "from optical_flow order by rowid"))
    con.close()
    return k, f


def _db_worker(rank, world, port, first, n, out_dir):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b, e = D.shard_range(first, n, world, rank)
    log_np = D.pack_device_log([fake_record(f, first, n) for f in range(b, e)])
    log = torch.from_numpy(log_np) if len(log_np) else torch.zeros(0, dtype=torch.uint8)
    gathered, sizes = D.all_gather_device_log(log, len(log_np))
    if rank == 0:
        core = _core()
        for r in range(world):
            core.write_optical_flow_records(os.path.join(out_dir, "sharded.db"), gathered[r, :sizes[r]].numpy().copy(), sizes[r])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [21, 2])
def test_two_ranks_write_the_single_rank_database(tmp_path, n):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    first = 3
    mp.spawn(_db_worker, args=(2, port, first, n, str(tmp_path)), nprocs=2, join=True)
    core = _core()
    records = [fake_record(f, first, n) for f in range(first, first + n)]
    # single rank, the C++ writer on one log
    one = D.pack_device_log(records)
    st = core.write_optical_flow_records(str(tmp_path / "single.db"), one, len(one))
    assert st.frames_processed == n and st.keypoint_rows_written == n
    # and the record-by-record writer over the Database class (what GenerateOpticalFlowDatabase's writer thread does)
    db = core.Database(str(tmp_path / "python.db"))
    D.write_records(db, records)
    db.close()
    ref = _db_dump(str(tmp_path / "single.db"))
    assert len(ref[0]) == n and len(ref[1]) == sum(len(r[2]) for r in records)
    assert _db_dump(str(tmp_path / "sharded.db")) == ref
    assert _db_dump(str(tmp_path / "python.db")) == ref
    # storing a log twice changes nothing (rows that exist are kept, like a resumed run)
    core.write_optical_flow_records(str(tmp_path / "single.db"), one, len(one))
    assert _db_dump(str(tmp_path / "single.db")) == ref


def test_record_log_roundtrip_and_corruption():
    records = [fake_record(f, 1, 12) for f in range(1, 13)]
    log = D.pack_device_log(records)
    back = D.parse_device_log(log, len(log))
    assert [r[0] for r in back] == list(range(1, 13))
    for (f, k, fl), (f2, k2, fl2) in zip(records, back):
        assert np.array_equal(k, k2) and sorted(fl) == sorted(fl2)
        for t in fl:
            assert all(np.array_equal(a, b) for a, b in zip(fl[t], fl2[t]))
    core = _core()
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        bad = log.copy()
        bad[0] ^= 0xFF
        with pytest.raises(RuntimeError):
            core.write_optical_flow_records(os.path.join(td, "x.db"), bad, len(bad))
        with pytest.raises(RuntimeError):
            core.write_optical_flow_records(os.path.join(td, "y.db"), log, len(log) - 40)


# ----------------------------------------------------------------------------------------------
# The streamed stitch of the PRODUCT (polychase_amd/analyze.py -> csrc/host/multi_gpu.cc: GenerateOpticalFlowDatabaseMultiGpu): pieces
# of the ranks' logs travel to rank 0 in frame order under credit flow control and are stored as they arrive.  ONE implementation
# since round 6 (the Python twin of rounds 3-5 is gone): these tests drive the C++ protocol itself on this GPU-less box through its
# testing aid -- a rank's shard given as ready-made record logs, payload over the control connection (transport "tcp");
# polychase_core._multi_gpu_protocol_selftest.  World sizes 2 and 3, one process per rank.
# ----------------------------------------------------------------------------------------------
def _protocol_worker(rank, world, port, first, n, out_dir, per, delay_ms, fail_rank, fail_after, q):
    core = _core()
    b, e = D.shard_range(first, n, world, rank)
    pieces = []
    for lo in range(b, e, per):
        hi = min(lo + per, e)
        pieces.append((bytes(D.pack_device_log([fake_record(f, first, n) for f in range(lo, hi)])), lo, hi - lo))
    try:
        r = core._multi_gpu_protocol_selftest(world, rank, port, os.path.join(out_dir, "streamed.db"), pieces,
                                              delay_ms=delay_ms if rank == 0 else 0, fail_after=fail_after if rank == fail_rank else -1)
        q.put((rank, "ok", int(r["pieces"]), int(r["bytes_moved"]), bool(r["cancelled"])))
    except RuntimeError as ex:
        q.put((rank, "error", str(ex), 0, False))


def _run_protocol(world, first, n, out_dir, per, delay_ms=0, fail_rank=-1, fail_after=-1):
    import multiprocessing as mp
    import time
    ctx = mp.get_context("spawn")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    q = ctx.Queue()
    procs = [ctx.Process(target=_protocol_worker, args=(r, world, port, first, n, out_dir, per, delay_ms, fail_rank, fail_after, q))
             for r in range(world)]
    t0 = time.time()
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        item = q.get(timeout=120)
        got[item[0]] = item
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return got, time.time() - t0


@pytest.mark.parametrize("world,n,per,delay_ms", [(2, 21, 3, 0), (3, 31, 2, 60), (3, 2, 4, 0)])
def test_multi_gpu_protocol_streams_the_single_rank_database(tmp_path, world, n, per, delay_ms):
    """every rank's pieces reach rank 0 in frame order (ranks ascending, a rank's pieces ascending) whatever the timing -- rank 0
    slow with its own shard (delay_ms per piece: the senders fill their two-deep queues and wait for credits meanwhile), a rank
    with no frames at all (n = 2 over 3 ranks) -- and the database is the single-rank one, row for row"""
    first = 3
    got, _ = _run_protocol(world, first, n, str(tmp_path), per, delay_ms=delay_ms)
    assert all(v[1] == "ok" for v in got.values()), got
    sent = sum(got[r][2] for r in range(1, world))
    assert got[0][2] == sent and got[0][3] == sum(got[r][3] for r in range(1, world))      # pieces / bytes received == sent
    expect_pieces = sum(len(range(*D.shard_range(first, n, world, r), per)) for r in range(1, world))
    assert sent == expect_pieces
    core = _core()
    one = D.pack_device_log([fake_record(f, first, n) for f in range(first, first + n)])
    core.write_optical_flow_records(str(tmp_path / "single.db"), one, len(one))
    assert _db_dump(str(tmp_path / "streamed.db")) == _db_dump(str(tmp_path / "single.db"))


@pytest.mark.parametrize("fail_rank,fail_after", [(0, 0), (0, 2), (2, 1), (1, 0)])
def test_multi_gpu_protocol_a_failing_rank_releases_every_rank(tmp_path, fail_rank, fail_after):
    """a rank's shard throws -- rank 0 before or while it receives, another rank between two pieces -- while the
    others wait for credits or sit on a full queue: EVERY rank comes back with an error, quickly, nobody hangs"""
    got, seconds = _run_protocol(3, 1, 40, str(tmp_path), 2, delay_ms=20, fail_rank=fail_rank, fail_after=fail_after)
    assert seconds < 60
    assert all(v[1] == "error" for v in got.values()), got
    assert "on purpose" in got[fail_rank][2]


def test_record_writer_refuses_other_keypoints(tmp_path):
    """A database that already holds ANOTHER analysis of a frame (other keypoints) must not receive this one's flows."""
    core = _core()
    a = D.pack_device_log([fake_record(5, 1, 12)])
    f, kps, flows = fake_record(5, 1, 12)
    b = D.pack_device_log([(f, kps[::-1].copy(), flows)])
    path = str(tmp_path / "x.db")
    core.write_optical_flow_records(path, a, len(a))
    before = _db_dump(path)
    with pytest.raises(RuntimeError, match="differ from the record's"):
        core.write_optical_flow_records(path, b, len(b))
    assert _db_dump(path) == before
    core.write_optical_flow_records(path, a, len(a))     # the same analysis again is fine (rows kept)
    assert _db_dump(path) == before
