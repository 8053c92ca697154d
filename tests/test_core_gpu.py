"""GPU: the reference's Python surface (polychase_core) end to end.  BASELINE config C1 (640x480x30
translating checkerboard) through OpticalFlowThread's request/provide protocol, exactly as
blender_addon/operators/analysis.py drives it; DB contents compared with the reference-shaped CPU path."""
import os
import sqlite3
import sys
import time

import numpy as np
import pytest

import oracle
from polychase_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def core():
    import torch  # noqa: F401
    sys.path.insert(0, os.path.join(ROOT, "polychase_amd", "core"))
    import polychase_core
    return polychase_core


def _run_thread(core, frames, first, path, stop_after=None, **kw):
    """The caller loop of analysis.py:242-285. Returns (requested ids, progress messages, errors)."""
    h, w, _ = frames[0].shape
    th = core.OpticalFlowThread(core.VideoInfo(w, h, first, len(frames)), path, **kw)
    requested, progress, errors = [], [], []
    done = False
    t0 = time.time()
    while not done and time.time() - t0 < 120:
        msg = th.try_pop()
        if msg is None:
            time.sleep(0.0005)
            continue
        if isinstance(msg, core.OpticalFlowRequest):
            requested.append(msg.frame_id)
            th.provide_frame(msg.frame_id, frames[msg.frame_id - first])
        elif isinstance(msg, core.OpticalFlowProgress):
            progress.append((msg.progress, msg.progress_message))
            if stop_after is not None and len(progress) == stop_after:
                th.request_stop()
        elif isinstance(msg, core.CppException):
            errors.append(msg.what())
        elif msg is True:
            done = True
    th.join()
    assert done
    return requested, progress, errors


def _dump(path):
    con = sqlite3.connect(path)
    k = {r[0]: (r[1], r[2]) for r in con.execute("select image_id, rows, keypoints from keypoints")}
    f = {(r[0], r[1]): r[2:] for r in con.execute(
        "select image_id_from, image_id_to, rows, src_keypoints_indices, tgt_keypoints, flow_errors from optical_flow")}
    con.close()
    return k, f


def _journal_header(path):
    """file-format write/read version of the SQLite header: 2, 2 = WAL (what the reference's Open() leaves)"""
    return open(path, "rb").read(20)[18:20]


def _expect(frames, first, **kw):
    kps, flows = oracle.analyze_clip(frames, first_frame=first, threads=4, **kw)
    k = {f: (len(v), v.tobytes()) for f, v in kps.items()}
    fl = {key: (len(v[0]), v[0].tobytes(), v[1].tobytes(), v[2].tobytes()) for key, v in flows.items()}
    return k, fl


def test_c1_checkerboard_through_thread_protocol(core, tmp_path):
    frames = synth.checkerboard_clip(30)
    path = str(tmp_path / "c1.db")
    requested, progress, errors = _run_thread(core, frames, 1, path)
    assert not errors
    assert requested == list(range(1, 31))            # every frame exactly once, increasing (appendix B.7)
    assert progress[0] == (0.0, "Processing frame 1") and progress[-1] == (1.0, "Done")
    assert [m for _, m in progress[:-1]] == [f"Processing frame {i}" for i in range(1, 31)]
    k, f = _dump(path)
    ek, ef = _expect(frames, 1)
    assert k == ek
    assert f == ef
    assert len(f) == 8 * 30 - 30


@pytest.mark.parametrize("batch", ["1", "8", "5"])
def test_cancel_then_resume_gives_identical_database(core, tmp_path, monkeypatch, batch):
    """... for every size of the writer's transactions (POLYCHASE_DB_BATCH_FRAMES: 1 = a transaction per frame as in rounds
    1-4, 8 = the default, 5 = a size that does not divide the clip): a cancelled run commits what it finished, the resumed run
    recomputes the rest, the file is the uninterrupted run's."""
    monkeypatch.setenv("POLYCHASE_DB_BATCH_FRAMES", batch)
    clip = synth.NoiseClip(320, 240, 24)
    frames = [clip.frame(t) for t in range(24)]
    full, part = str(tmp_path / "full.db"), str(tmp_path / "part.db")
    _, _, e = _run_thread(core, frames, 1, full)
    assert not e
    _, progress, e = _run_thread(core, frames, 1, part, stop_after=9)
    assert not e and progress[-1] == (1.0, "Cancelled")
    # the driver loads under a rollback journal and hands the file back in WAL mode -- also when cancelled
    assert _journal_header(full) == b"\x02\x02" and _journal_header(part) == b"\x02\x02"
    k_part, f_part = _dump(part)
    assert 0 < len(f_part) < 8 * 24 - 30
    requested, progress, e = _run_thread(core, frames, 1, part)        # resume (opticalflow.cc:168-178, :286)
    assert not e and progress[-1] == (1.0, "Done")
    assert _dump(part) == _dump(full)
    # a third run finds everything present: nothing is recomputed or rewritten
    _, _, e = _run_thread(core, frames, 1, part)
    assert not e and _dump(part) == _dump(full)
    assert _journal_header(part) == b"\x02\x02"


def test_bulk_load_can_be_switched_off_and_gives_the_same_rows(core, tmp_path, monkeypatch):
    clip = synth.NoiseClip(256, 192, 12)
    frames = [clip.frame(t) for t in range(12)]
    a, b = str(tmp_path / "bulk.db"), str(tmp_path / "wal.db")
    _, _, e = _run_thread(core, frames, 1, a)
    assert not e
    monkeypatch.setenv("POLYCHASE_DB_BULK_LOAD", "0")
    _, _, e = _run_thread(core, frames, 1, b)
    assert not e
    assert _dump(a) == _dump(b)
    assert _journal_header(a) == _journal_header(b) == b"\x02\x02"
    assert open(a, "rb").read(100)[16:18] == open(b, "rb").read(100)[16:18]      # page size
    assert open(a, "rb").read(100)[52:56] == open(b, "rb").read(100)[52:56]      # auto-vacuum setting


def test_sync_binding_with_options_and_errors(core, tmp_path):
    clip = synth.NoiseClip(256, 192, 12)
    frames = [clip.frame(t) for t in range(12)]
    g = core.GFTTOptions()
    g.max_corners = 200
    g.min_distance = 7.0
    fo = core.OpticalFlowOptions()
    fo.max_level = 2
    fo.window_size = 9
    fo.term_max_iters = 10
    msgs = []
    path = str(tmp_path / "sync.db")
    stats = core.generate_optical_flow_database(core.VideoInfo(256, 192, 5, 12), lambda fid: frames[fid - 5],
                                                lambda p, m: (msgs.append(m) or True), path, g, fo)
    assert msgs[-1] == "Done" and stats.frames_processed == 12
    ek, ef = _expect(frames, 5, gopt=oracle.gftt_options(max_corners=200, min_distance=7.0),
                     fopt=oracle.flow_options(max_level=2, window_size=9, term_max_iters=10))
    k, f = _dump(path)
    assert k == ek and f == ef
    # missing frame -> runtime error with the reference's message (sic)
    with pytest.raises(RuntimeError, match="Rquested frame #5 was not provided"):
        core.generate_optical_flow_database(core.VideoInfo(256, 192, 5, 12), lambda fid: None, None, str(tmp_path / "m.db"))
    # wrong shape -> CHECK failure (opticalflow.cc:196-198)
    with pytest.raises(Exception, match="Assertion failed"):
        core.generate_optical_flow_database(core.VideoInfo(256, 192, 5, 12), lambda fid: frames[0][:100], None,
                                            str(tmp_path / "s.db"))


@pytest.mark.parametrize("win,arith", [(17, "opencv_x86"), (21, "opencv_x86"), (31, "opencv_x86"), (21, "canonical"), (31, "canonical"), (3, "opencv_x86")])
def test_large_windows_through_generate_optical_flow_database(core, tmp_path, monkeypatch, win, arith):
    """OpticalFlowOptions.window_size is a free attribute in the reference (cpp/opticalflow.h:27-33, polychase_pybind.cc:138-145)
    and OpenCV's own default is 21: windows above 16 (and 3) through the binding, database byte-equal to the reference-shaped
    CPU path in the default and the canonical arithmetic"""
    monkeypatch.setenv("POLYCHASE_ARITH", arith)
    clip = synth.NoiseClip(333, 211, 11)
    frames = [clip.frame(t) for t in range(11)]
    fo = core.OpticalFlowOptions()
    fo.window_size = win
    fo.max_level = 3
    path = str(tmp_path / f"w{win}.db")
    stats = core.generate_optical_flow_database(core.VideoInfo(333, 211, 1, 11), lambda fid: frames[fid - 1], None, path,
                                                core.GFTTOptions(), fo)
    assert stats.frames_processed == 11
    emu = (oracle.EMU_LK_SIMD | oracle.EMU_SOBEL_FMA) if arith == "opencv_x86" else 0
    with oracle.emulation(emu):
        ek, ef = _expect(frames, 1, fopt=oracle.flow_options(window_size=win, max_level=3))
    k, f = _dump(path)
    assert k == ek and f == ef


def test_device_resident_frames_through_the_binding(core, tmp_path):
    import torch
    clip = synth.NoiseClip(256, 192, 10)
    frames = [clip.frame(t) for t in range(10)]
    dev = [torch.from_numpy(f).cuda() for f in frames]
    a, b = str(tmp_path / "host.db"), str(tmp_path / "dev.db")
    core.generate_optical_flow_database(core.VideoInfo(256, 192, 1, 10), lambda fid: frames[fid - 1], None, a)
    core.generate_optical_flow_database(core.VideoInfo(256, 192, 1, 10), lambda fid: dev[fid - 1], None, b)
    assert _dump(a) == _dump(b)


def test_float_frames_through_thread_and_sync_binding(core, tmp_path):
    """provide_frame / the frame accessor accept Blender's float32 (H, W, 4) pixels directly: same database as
    when the addon converts them with `(image_data[:, :, :3] * 255).astype(np.uint8)` first."""
    rng = np.random.default_rng(3)
    clip = synth.NoiseClip(256, 192, 12)
    u8 = [clip.frame(t) for t in range(12)]
    f32 = []
    for f in u8:
        x = (f.astype(np.float32) + rng.uniform(0.05, 0.95, f.shape).astype(np.float32)) / np.float32(255.0)
        f32.append(np.ascontiguousarray(np.concatenate([x, np.ones(f.shape[:2] + (1,), np.float32)], axis=2)))
        assert np.array_equal((f32[-1][:, :, :3] * 255).astype(np.uint8), f)
    a, b, c = (str(tmp_path / n) for n in ("u8.db", "f32_thread.db", "f32_sync.db"))
    core.generate_optical_flow_database(core.VideoInfo(256, 192, 1, 12), lambda fid: u8[fid - 1], None, a)
    requested, _, errors = _run_thread(core, f32, 1, b)
    assert not errors and requested == list(range(1, 13))
    core.generate_optical_flow_database(core.VideoInfo(256, 192, 1, 12), lambda fid: f32[fid - 1], None, c)
    assert _dump(a) == _dump(b) == _dump(c)


def test_write_images_dumps_the_frames_and_their_keypoints(core, tmp_path):
    """write_images=True (reference cpp/opticalflow.cc:80-96, :228-232, :265-267): <database dir>/frames/%06d.png is
    the frame, keypoints_%06d.png differs from it exactly on the crosses of the frame's stored keypoints -- for host
    frames, device frames and Blender's float frames."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_debug_images_cpu import read_png

    w, h, n = 160, 120, 10
    clip = synth.NoiseClip(w, h, n)
    frames = [clip.frame(t) for t in range(n)]
    variants = {"host": lambda fid: frames[fid - 1],
                "device": lambda fid: torch.from_numpy(frames[fid - 1]).cuda(),
                "float": lambda fid: np.concatenate([frames[fid - 1].astype(np.float32) / 255.0, np.ones((h, w, 1), np.float32)], axis=2)}
    for name, acc in variants.items():
        d = tmp_path / name
        d.mkdir()
        db = str(d / "clip.db")
        core.generate_optical_flow_database(core.VideoInfo(w, h, 1, n), acc, None, db, write_images=True)
        con = core.Database(db)
        for fid in (1, 5, n):
            plain = read_png(d / "frames" / f"{fid:06d}.png")
            marked = read_png(d / "frames" / f"keypoints_{fid:06d}.png")
            if name == "float":
                # (x / 255 * 255).astype(uint8) truncates: within one grey level of the original
                assert np.abs(plain.astype(int) - frames[fid - 1].astype(int)).max() <= 1
            else:
                assert np.array_equal(plain, frames[fid - 1])
            kps = con.read_keypoints(fid).astype(int)
            mask = np.zeros((h, w), bool)
            for x, y in kps:
                mask[y, max(0, x - 5):x + 6] = True
                mask[max(0, y - 5):y + 6, x] = True
            changed = (marked != plain).any(axis=2)
            assert len(kps) > 20 and not (changed & ~mask).any() and changed.sum() > 0.5 * mask.sum()
        con.close()
        assert len(list((d / "frames").glob("*.png"))) == 2 * n


def test_parked_engine_is_reused_and_replaced(core, tmp_path):
    """A finished call parks its engine (pc_analyzer_reset) for the next one: same geometry -> reused, other geometry or
    options -> replaced; the databases equal those of a process that never parks (tests/test_env_variants_gpu.py covers
    POLYCHASE_ENGINE_CACHE=0 across processes) and of the oracle."""
    clip_a = synth.NoiseClip(320, 240, 20)
    clip_b = synth.NoiseClip(256, 192, 14)
    fa = [clip_a.frame(t) for t in range(20)]
    fb = [clip_b.frame(t) for t in range(14)]

    def run(frames, name, first=1, **opt):
        h, w, _ = frames[0].shape
        fo = core.OpticalFlowOptions()
        for k, val in opt.items():
            setattr(fo, k, val)
        p = str(tmp_path / name)
        st = core.generate_optical_flow_database(core.VideoInfo(w, h, first, len(frames)), lambda fid: frames[fid - first], None, p,
                                                 core.GFTTOptions(), fo)
        return p, st

    core.release_cached_engine()
    p1, s1 = run(fa, "a1.db")
    p2, s2 = run(fa, "a2.db", first=7)        # same geometry, other frame ids: a stale resident frame must not be found
    # the second call found the engine (the flag; the times -- 10-20 ms against < 1 -- only on a quiet box: round 6 saw 8.5 vs 9.3 ms
    # of database creation on a busy one)
    assert s2.engine_reused and not s1.engine_reused, (s1.seconds_setup, s2.seconds_setup)
    p3, _ = run(fb, "b.db")                   # other geometry: the parked engine is replaced
    p4, _ = run(fa, "a4.db", max_level=2)     # other options
    p5, _ = run(fa, "a5.db")
    core.release_cached_engine()
    p6, s6 = run(fa, "a6.db")
    assert not s6.engine_reused
    assert _dump(p1) == _dump(p5) == _dump(p6)
    k1, k2 = _dump(p1)[0], _dump(p2)[0]
    assert {f + 6: kv for f, kv in k1.items()} == k2                          # the same keypoints under shifted ids
    assert _dump(p1) == _expect(fa, 1)                                        # ... and the oracle's database
    grays = [oracle.rgb2gray(f) for f in fb]
    got = core.Database(p3).read_keypoints(5)
    assert np.array_equal(got, oracle.gftt(grays[4]))
    assert _dump(p4)[1] != _dump(p1)[1]


def test_parked_engine_is_not_reused_across_a_change_of_the_arithmetic_mode(core, tmp_path, monkeypatch):
    """settings read when the engine is created (POLYCHASE_ARITH, stream layout, ...) are part of the key of the
    parked engine.  On step-edge content the two arithmetic modes give different flow bits: a run under
    POLYCHASE_ARITH=canonical after a default (opencv_x86) run must produce the canonical database, not the parked engine's."""
    frames = synth.checkerboard_clip(12, w=320, h=240)

    def run(name):
        p = str(tmp_path / name)
        core.generate_optical_flow_database(core.VideoInfo(320, 240, 1, len(frames)), lambda fid: frames[fid - 1], None, p,
                                            core.GFTTOptions(), core.OpticalFlowOptions())
        return _dump(p)

    core.release_cached_engine()
    monkeypatch.delenv("POLYCHASE_ARITH", raising=False)
    x86 = run("x86.db")
    monkeypatch.setenv("POLYCHASE_ARITH", "canonical")
    can = run("can.db")
    monkeypatch.delenv("POLYCHASE_ARITH")
    again = run("x86b.db")
    core.release_cached_engine()
    assert x86 == again
    assert x86[1] != can[1], "the canonical run reused the engine parked by the x86 run"
    with oracle.emulation(oracle.EMU_CANONICAL):
        assert can == _expect(frames, 1)
    assert x86 == _expect(frames, 1)


def test_parked_engine_is_given_back_after_the_idle_time(tmp_path):
    """the parked engine must not sit on GBs of the host process's GPU memory for ever.  With
    POLYCHASE_ENGINE_CACHE_IDLE_S=1 (read once per process: a subprocess) a call 3 s after the previous one creates its engine
    again, a call right after it does not."""
    import subprocess
    code = (
        "import os, sys, time\n"
        f"sys.path.insert(0, {ROOT!r}); sys.path.insert(0, os.path.join({ROOT!r}, 'polychase_amd', 'core'))\n"
        "import torch, polychase_core as core\n"
        "from polychase_amd import synth\n"
        "clip = synth.NoiseClip(320, 240, 12); fr = [clip.frame(t) for t in range(12)]\n"
        "def run():\n"
        "    return int(core.generate_optical_flow_database(core.VideoInfo(320, 240, 1, 12), lambda f: fr[f - 1], None, '', core.GFTTOptions(), core.OpticalFlowOptions()).engine_reused)\n"
        "t0 = core._engine_cache_timer_running()\n"
        "a = run(); t1 = core._engine_cache_timer_running(); b = run(); time.sleep(3.0); t2 = core._engine_cache_timer_running()\n"
        "c = run(); d = run(); t3 = core._engine_cache_timer_running(); core.release_cached_engine(); t4 = core._engine_cache_timer_running()\n"
        "print('SETUP', a, b, c, d)\n"
        "print('TIMER', int(t0), int(t1), int(t2), int(t3), int(t4))\n")
    env = dict(os.environ, POLYCHASE_ENGINE_CACHE_IDLE_S="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, text=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    line = [l for l in r.stdout.splitlines() if l.startswith("SETUP")]
    assert r.returncode == 0 and line, r.stderr[-2000:]
    a, b, c, d = map(int, line[0].split()[1:])            # "the call took the parked engine"
    assert (a, b) == (0, 1) and d == 1, (a, b, c, d)      # taken from the slot
    assert c == 0, (a, b, c, d)                           # created again: the idle timer had destroyed the parked engine
    # the timer is a thread that exists only while an engine is parked: none before the first run, one while
    # parked, gone after the idle time fired, one again, joined by release_cached_engine()
    timer = [l for l in r.stdout.splitlines() if l.startswith("TIMER")][0].split()[1:]
    assert timer == ["0", "1", "0", "1", "0"], timer
