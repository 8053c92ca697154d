"""GPU: the multi-rank analysis as a product (polychase_amd/analyze.py over polychase_core.generate_optical_flow_shard /
generate_optical_flow_records / write_optical_flow_records; SURVEY 8(e), BASELINE config C4).  One GPU is enough to check everything but the RCCL call
itself: a shard is the same C++ driver over a frame1 range with an 8-frame halo, its records go to a device log, the logs
are stored in frame order -- the database must not depend on how the clip is cut
(reference: cpp/opticalflow.cc:209-321, cpp/database.cc:183-214)."""
import os
import sqlite3
import sys

import numpy as np
import pytest

from polychase_amd import distributed as D
from polychase_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def core():
    import torch  # noqa: F401
    sys.path.insert(0, os.path.join(ROOT, "polychase_amd", "core"))
    import polychase_core
    return polychase_core


def _dump(path):
    con = sqlite3.connect(path)
    k = list(con.execute("select rowid, image_id, rows, keypoints from keypoints order by rowid"))
    f = list(con.execute("select rowid, image_id_from, image_id_to, rows, src_keypoints_indices, tgt_keypoints, flow_errors "
                         "from optical_flow order by rowid"))
    con.close()
    return k, f


def _shards_to_db(core, vi, frames_of, cuts, path, fopt, capacity):
    """what `world` ranks do, one after the other on this GPU: a log per shard, stored in rank order"""
    import torch
    for b, e in cuts:
        log = torch.empty(capacity, dtype=torch.uint8, device="cuda")
        used, st = core.generate_optical_flow_records(vi, frames_of, None, b, e, log.data_ptr(), log.numel(), core.GFTTOptions(), fopt)
        assert st.frames_processed == e - b
        host = log[:used].cpu().numpy()
        recs = D.parse_device_log(host, used)
        assert [r[0] for r in recs] == list(range(b, e))
        core.write_optical_flow_records(path, host, used)


def test_database_does_not_depend_on_the_number_of_ranks(core, tmp_path):
    w, h, n, first = 320, 240, 37, 5
    clip = synth.NoiseClip(w, h, n)
    frames = [clip.frame(t) for t in range(n)]
    acc = lambda fid: frames[fid - first]
    vi = core.VideoInfo(w, h, first, n)
    fo = core.OpticalFlowOptions()
    ref = str(tmp_path / "single.db")
    core.generate_optical_flow_database(vi, acc, None, ref, core.GFTTOptions(), fo)
    want = _dump(ref)
    assert len(want[0]) == n
    cap = D.log_capacity_bytes(n + 1, w * h // 20)
    for world in (1, 2, 3, 8):
        path = str(tmp_path / f"w{world}.db")
        _shards_to_db(core, vi, acc, [D.shard_range(first, n, world, r) for r in range(world)], path, fo, cap)
        assert _dump(path) == want, f"{world} ranks give another database"


def test_requested_frames_of_a_shard_are_its_range_plus_the_halo(core):
    import torch
    w, h, n, first = 160, 120, 40, 1
    clip = synth.NoiseClip(w, h, n)
    asked = []

    def acc(fid):
        asked.append(fid)
        return clip.frame(fid - first)

    log = torch.empty(D.log_capacity_bytes(16, w * h // 20), dtype=torch.uint8, device="cuda")
    used, st = core.generate_optical_flow_records(core.VideoInfo(w, h, first, n), acc, None, 14, 22, log.data_ptr(), log.numel())
    assert asked == list(range(6, 30))          # 14 - 8 .. 21 + 8, each once, increasing
    assert st.frames_processed == 8
    # a log that is too small is an error, not a truncated result
    with pytest.raises(RuntimeError, match="device log full"):
        core.generate_optical_flow_records(core.VideoInfo(w, h, first, n), lambda f: clip.frame(f - first), None, 14, 22, log.data_ptr(), 4096)


def test_shard_log_handed_over_in_pieces(core):
    """generate_optical_flow_shard with a two-part log: pieces of 3 frames (and pieces that end early because the next
    record does not fit the part) concatenate to the log of the one-piece run; a part is only reused after its piece
    has been handed over (the callback copies it out at that moment)."""
    import torch
    w, h, n, first = 320, 240, 40, 1
    clip = synth.NoiseClip(w, h, n)
    acc = lambda fid: clip.frame(fid - first)
    vi = core.VideoInfo(w, h, first, n)
    b, e = 9, 30
    whole = torch.empty(D.log_capacity_bytes(e - b + 1, w * h // 20), dtype=torch.uint8, device="cuda")
    used, st = core.generate_optical_flow_records(vi, acc, None, b, e, whole.data_ptr(), whole.numel())
    want = whole[:used].cpu().numpy()
    per_frame = used // (e - b)
    for piece_frames, part_bytes in ((3, 1 << 22), (1000, int(per_frame * 4.5) // 16 * 16), (4, int(per_frame * 2.5) // 16 * 16)):
        log = torch.empty(2 * part_bytes, dtype=torch.uint8, device="cuda")
        got, meta = [], []

        def on_piece(piece, offset, nbytes, first_frame1, n_frames):
            assert offset == (piece % 2) * part_bytes and nbytes <= part_bytes
            got.append(log[offset:offset + nbytes].cpu().numpy())
            meta.append((piece, first_frame1, n_frames))
            log[offset:offset + part_bytes].fill_(0xEE)        # whatever comes next must be written again
            torch.cuda.synchronize()

        res = core.generate_optical_flow_shard(vi, acc, None, "", b, e, log.data_ptr(), log.numel(), 2, piece_frames, on_piece, False)
        assert not res["cancelled"] and res["pieces"] == len(meta) and res["stats"].frames_processed == e - b
        assert [m[0] for m in meta] == list(range(len(meta)))
        assert [m[1] for m in meta] == [b + sum(x[2] for x in meta[:k]) for k in range(len(meta))] and sum(m[2] for m in meta) == e - b
        assert all(m[2] <= piece_frames for m in meta) and len(meta) >= (e - b + piece_frames - 1) // piece_frames
        # record by record (rows past a flow's row_offset are unspecified bytes of the log: not compared)
        cat = np.concatenate(got)
        assert len(cat) == len(want)
        for (f1, k1, fl1), (f2, k2, fl2) in zip(D.parse_device_log(cat, len(cat)), D.parse_device_log(want, len(want))):
            assert f1 == f2 and np.array_equal(k1, k2) and sorted(fl1) == sorted(fl2)
            assert all(np.array_equal(a, b) for t in fl1 for a, b in zip(fl1[t], fl2[t]))
    # a part that cannot hold one record is an error
    small = torch.empty(2 * 4096, dtype=torch.uint8, device="cuda")
    with pytest.raises(RuntimeError, match="device log full"):
        core.generate_optical_flow_shard(vi, acc, None, "", b, e, small.data_ptr(), small.numel(), 2, 3, lambda *a: None, False)


def test_cancelled_shard_reports_it_and_keeps_the_finished_records(core):
    import torch
    w, h, n, first = 320, 240, 40, 1
    clip = synth.NoiseClip(w, h, n)
    acc = lambda fid: clip.frame(fid - first)
    log = torch.empty(2 << 22, dtype=torch.uint8, device="cuda")
    seen = []

    def on_piece(piece, offset, nbytes, first_frame1, n_frames):
        seen.extend(r[0] for r in D.parse_device_log(log[offset:offset + nbytes].cpu().numpy(), nbytes))

    calls = []

    def cb(progress, msg):
        calls.append(msg)
        return len(calls) <= 7            # cancel when frame 9 + 7 is about to start

    res = core.generate_optical_flow_shard(core.VideoInfo(w, h, first, n), acc, cb, "", 9, 30, log.data_ptr(), log.numel(), 2, 3, on_piece, False)
    assert res["cancelled"] and calls[-1] == "Cancelled"
    assert seen == list(range(9, 16)) and res["stats"].frames_processed == 7
    # the one-piece entry point: used bytes cover the finished records too (round 2 returned 0 here)
    one = torch.empty(1 << 23, dtype=torch.uint8, device="cuda")
    calls.clear()
    used, st = core.generate_optical_flow_records(core.VideoInfo(w, h, first, n), acc, cb, 9, 30, one.data_ptr(), one.numel())
    assert [r[0] for r in D.parse_device_log(one[:used].cpu().numpy(), used)] == list(range(9, 16))


def test_shard_straight_into_the_database(core, tmp_path):
    """rank 0 of the multi-rank run: its shard goes through the writer thread of the single-GPU path; the other shards
    arrive as logs -- together the single-run database, halo frames are not detected"""
    import torch
    w, h, n, first = 320, 240, 30, 2
    clip = synth.NoiseClip(w, h, n)
    asked = []

    def acc(fid):
        asked.append(fid)
        return clip.frame(fid - first)

    vi = core.VideoInfo(w, h, first, n)
    ref = str(tmp_path / "single.db")
    core.generate_optical_flow_database(vi, acc, None, ref)
    path = str(tmp_path / "two.db")
    asked.clear()
    res = core.generate_optical_flow_shard(vi, acc, None, path, 2, 17)
    assert asked == list(range(2, 25)) and res["stats"].frames_processed == 15 and res["stats"].keypoint_rows_written == 15
    log = torch.empty(1 << 24, dtype=torch.uint8, device="cuda")
    used, _ = core.generate_optical_flow_records(vi, acc, None, 17, 32, log.data_ptr(), log.numel())
    w2 = core.OpticalFlowRecordWriter(path)
    w2.write(log[:used].cpu().numpy(), used)
    w2.close()
    assert _dump(path) == _dump(ref)


def test_analyze_entry_point_single_rank(core, tmp_path):
    """polychase_amd.analyze.analyze without a process group = one rank owning the whole clip"""
    from polychase_amd import analyze
    w, h, n, first = 320, 240, 20, 1
    clip = synth.NoiseClip(w, h, n, device="cuda")
    dev_frames = [clip.frame_torch(t) for t in range(n)]
    host_frames = [f.cpu().numpy() for f in dev_frames]
    out = analyze.analyze(w, h, first, n, lambda fid: dev_frames[fid - first], str(tmp_path / "a.db"))
    assert out["world"] == 1 and out["frames"] == n and out["written"]["keypoint_rows"] == n
    core.generate_optical_flow_database(core.VideoInfo(w, h, first, n), lambda fid: host_frames[fid - first], None, str(tmp_path / "b.db"))
    assert _dump(str(tmp_path / "a.db")) == _dump(str(tmp_path / "b.db"))


def test_c4_rank_workload_at_4k(core, tmp_path):
    """The per-rank workload of config C4 (3840x2160, max_level 4): a shard in the middle of a clip -- halo on both
    sides, device log, parse, store -- against the rows the unsharded run writes for the same frames."""
    import torch
    w, h, n, first = 3840, 2160, 26, 1
    clip = synth.NoiseClip(w, h, 300, device="cuda")
    frames = [clip.frame_torch(100 + t) for t in range(n)]
    acc = lambda fid: frames[fid - first]
    vi = core.VideoInfo(w, h, first, n)
    fo = core.OpticalFlowOptions()
    fo.max_level = 4
    ref = str(tmp_path / "single.db")
    core.generate_optical_flow_database(vi, acc, None, ref, core.GFTTOptions(), fo)
    b, e = 10, 16
    log = torch.empty(D.log_capacity_bytes(e - b + 1, w * h // 40), dtype=torch.uint8, device="cuda")
    used, st = core.generate_optical_flow_records(vi, acc, None, b, e, log.data_ptr(), log.numel(), core.GFTTOptions(), fo)
    host = log[:used].cpu().numpy()
    recs = D.parse_device_log(host, used)
    assert [r[0] for r in recs] == list(range(b, e))
    assert all(len(r[2]) == 8 for r in recs)              # interior frames: all eight pairs, into the halo too
    assert min(len(r[1]) for r in recs) > 100_000         # a 4K frame of this clip has ~160 k keypoints
    core.write_optical_flow_records(str(tmp_path / "shard.db"), host, used)
    con = sqlite3.connect(ref)
    want_k = list(con.execute("select image_id, rows, keypoints from keypoints where image_id >= ? and image_id < ? order by image_id", (b, e)))
    want_f = list(con.execute("select image_id_from, image_id_to, rows, src_keypoints_indices, tgt_keypoints, flow_errors from optical_flow "
                              "where image_id_from >= ? and image_id_from < ? order by image_id_from, rowid", (b, e)))
    con.close()
    con = sqlite3.connect(str(tmp_path / "shard.db"))
    got_k = list(con.execute("select image_id, rows, keypoints from keypoints order by image_id"))
    got_f = list(con.execute("select image_id_from, image_id_to, rows, src_keypoints_indices, tgt_keypoints, flow_errors from optical_flow "
                             "order by image_id_from, rowid"))
    con.close()
    assert got_k == want_k and got_f == want_f


def test_launcher_with_two_ranks_on_one_gpu(tmp_path):
    """polychase_amd.analyze as a user launches it (torch.distributed.run, two processes): both ranks on GPU 0 over gloo
    (POLYCHASE_ANALYZE_SHARE_GPU=1) -- shard ranges, rank 0's shard straight into the database, rank 1's log handed over
    in pieces under credit flow control, rank 0's store of what arrives -- against the same command with one process;
    and `--gpus 3`, which launches its ranks itself."""
    import json
    import subprocess

    def run(cmd, db, frames=41, extra=()):
        env = dict(os.environ, POLYCHASE_ANALYZE_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
            env.pop(k, None)
        cmd = cmd + ["--synthetic", "c1", "--frames", str(frames), "--database", db, "--piece-frames", "4", *extra]
        r = subprocess.run(cmd, text=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=env, cwd=ROOT)
        assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
        try:
            return [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
        except json.JSONDecodeError as e:
            raise AssertionError(f"unparsable result line ({e}):\n{r.stdout[-3000:]}\n--- stderr:\n{r.stderr[-2000:]}")

    def torchrun(nproc, port):
        return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
                "--master-port", str(port), "-m", "polychase_amd.analyze"]

    one, two, three = str(tmp_path / "one.db"), str(tmp_path / "two.db"), str(tmp_path / "three.db")
    run(torchrun(1, 29551), one)
    outs = run(torchrun(2, 29552), two)
    a, b = _dump(one), _dump(two)
    assert len(a[0]) == 41 and a == b
    assert sorted(o["rank"] for o in outs) == [0, 1] and not any(o["cancelled"] for o in outs)
    r1 = [o for o in outs if o["rank"] == 1][0]
    assert r1["pieces"] >= 5 and r1["log_bytes"] > 0                     # 20 or 21 frames in pieces of 4
    outs = run([sys.executable, "-m", "polychase_amd.analyze", "--gpus", "3"], three)
    assert sorted(o["rank"] for o in outs) == [0, 1, 2] and _dump(three) == a
