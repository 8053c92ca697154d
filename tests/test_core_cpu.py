"""CPU: the polychase_core module (reference pybind surface) imports, and its Database is
byte-compatible with the reference's SQLite format (cpp/database.cc:64-135, :350-400)."""
import os
import sqlite3
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def core():
    from polychase_amd import build
    build.build_all()
    import torch  # noqa: F401  (one HIP runtime per process: load torch's first, see polychase_amd/hip.py)
    sys.path.insert(0, os.path.join(ROOT, "polychase_amd", "core"))
    import polychase_core
    return polychase_core


ANALYSIS_NAMES = ["Database", "ImagePairFlow", "VideoInfo", "GFTTOptions", "OpticalFlowOptions", "OpticalFlowThread",
                  "OpticalFlowProgress", "OpticalFlowRequest", "CppException", "generate_optical_flow_database"]


def test_exports_reference_names(core):
    for n in ANALYSIS_NAMES:
        assert hasattr(core, n), n


def test_struct_defaults(core):
    g = core.GFTTOptions()
    assert (g.quality_level, g.min_distance, g.block_size, g.gradient_size, g.max_corners, g.use_harris,
            g.harris_k) == (0.01, 5.0, 3, 3, 0, False, 0.04)
    assert not hasattr(g, "grid_rows")  # not exposed by the reference either (polychase_pybind.cc:128-136)
    f = core.OpticalFlowOptions()
    assert (f.window_size, f.max_level, f.term_max_iters, f.term_epsilon, f.min_eigen_threshold) == (10, 3, 30, 0.01, 1e-4)
    v = core.VideoInfo(width=640, height=480, first_frame=1, num_frames=30)
    assert (v.width, v.height, v.first_frame, v.num_frames) == (640, 480, 1, 30)


def test_database_schema_and_blobs(core, tmp_path):
    path = str(tmp_path / "flow.db")
    db = core.Database(path)
    kps = np.array([[10, 20], [30.5, 40.25], [7, 8]], np.float32)
    db.write_keypoints(5, kps)
    idx = np.array([0, 2], np.uint32)
    tgt = np.array([[11.5, 21.25], [8, 9]], np.float32)
    err = np.array([0.5, 1.5], np.float32)
    db.write_image_pair_flow(5, 6, idx, tgt, err)
    assert db.keypoints_exist(5) and not db.keypoints_exist(6)
    assert db.image_pair_flow_exists(5, 6) and not db.image_pair_flow_exists(6, 5)
    assert np.array_equal(db.read_keypoints(5), kps)
    assert db.read_keypoints(99).shape == (0, 2)
    f = db.read_image_pair_flow(5, 6)
    assert (f.image_id_from, f.image_id_to) == (5, 6)
    assert np.array_equal(f.src_kps_indices, idx) and np.array_equal(f.tgt_kps, tgt) and np.array_equal(f.flow_errors, err)
    assert db.find_optical_flows_from_image(5) == [6] and db.find_optical_flows_to_image(6) == [5]
    assert db.get_min_image_id_with_keypoints() == 5 == db.get_max_image_id_with_keypoints()
    with pytest.raises(RuntimeError, match="SQLite error"):   # plain INSERT: duplicate key throws (database.cc:357)
        db.write_keypoints(5, kps)
    db.close()

    con = sqlite3.connect(path)
    tables = dict(con.execute("select name, sql from sqlite_master where type='table'").fetchall())
    norm = lambda s: " ".join(s.split())
    assert norm(tables["keypoints"]) == norm(
        "CREATE TABLE keypoints( image_id INTEGER PRIMARY KEY NOT NULL, rows INTEGER NOT NULL, keypoints BLOB NOT NULL )")
    assert norm(tables["optical_flow"]) == norm(
        "CREATE TABLE optical_flow( image_id_from INTEGER NOT NULL, image_id_to INTEGER NOT NULL, rows INTEGER NOT NULL, "
        "src_keypoints_indices BLOB NOT NULL, tgt_keypoints BLOB NOT NULL, flow_errors BLOB NOT NULL, "
        "PRIMARY KEY(image_id_from, image_id_to), FOREIGN KEY(image_id_from) REFERENCES keypoints(image_id) ON DELETE CASCADE )")
    rows, blob = con.execute("select rows, keypoints from keypoints where image_id=5").fetchone()
    assert rows == 3 and blob == kps.tobytes()          # raw little-endian memcpy (database.cc:137-158)
    r = con.execute("select rows, src_keypoints_indices, tgt_keypoints, flow_errors from optical_flow").fetchone()
    assert r == (2, idx.tobytes(), tgt.tobytes(), err.tobytes())
    assert con.execute("pragma journal_mode").fetchone()[0] == "wal"
    # the reference issues PRAGMA auto_vacuum=1 AFTER journal_mode=WAL has initialised the file
    # (database.cc:80,:89), where SQLite ignores it; same pragma order here => same on-disk header
    assert con.execute("pragma auto_vacuum").fetchone()[0] == 0
    con.close()


def test_reads_database_written_with_reference_statements(core, tmp_path):
    """A DB created by the reference's SQL (python sqlite3 here) is readable by our Database."""
    path = str(tmp_path / "ref.db")
    con = sqlite3.connect(path)
    con.execute("CREATE TABLE IF NOT EXISTS keypoints(image_id INTEGER PRIMARY KEY NOT NULL, rows INTEGER NOT NULL, keypoints BLOB NOT NULL);")
    con.execute("CREATE TABLE IF NOT EXISTS optical_flow(image_id_from INTEGER NOT NULL, image_id_to INTEGER NOT NULL, rows INTEGER NOT NULL, "
                "src_keypoints_indices BLOB NOT NULL, tgt_keypoints BLOB NOT NULL, flow_errors BLOB NOT NULL, PRIMARY KEY(image_id_from, image_id_to), "
                "FOREIGN KEY(image_id_from) REFERENCES keypoints(image_id) ON DELETE CASCADE);")
    kps = np.arange(10, dtype=np.float32).reshape(5, 2)
    con.execute("INSERT INTO keypoints(image_id, rows, keypoints) VALUES(?, ?, ?);", (3, 5, kps.tobytes()))
    con.commit()
    con.close()
    db = core.Database(path)
    assert np.array_equal(db.read_keypoints(3), kps)
    db.close()


def test_no_gpu_means_loud_failure(core, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    vi = core.VideoInfo(64, 48, 1, 2)
    with pytest.raises(RuntimeError, match="no HIP device|No HIP|HIP"):
        core.generate_optical_flow_database(vi, lambda fid: np.zeros((48, 64, 3), np.uint8), None, str(tmp_path / "x.db"))


def test_page_size_keeps_the_database_interchangeable(core, tmp_path, monkeypatch):
    """New databases get 64-KiB pages (2x faster inserts), POLYCHASE_DB_PAGE_SIZE chooses another size (4096 = the
    reference's): a plain SQLite property of NEW files; same schema and rows, an existing database keeps its page size,
    nonsense values leave SQLite's default."""
    kp = np.arange(20, dtype=np.float32).reshape(10, 2)
    idx = np.arange(5, dtype=np.uint32)

    def fill(path):
        db = core.Database(path)
        db.write_keypoints(3, kp)
        db.write_image_pair_flow(3, 4, idx, kp[:5], np.ones(5, np.float32))
        db.close()

    def page_size(path):
        con = sqlite3.connect(path)
        v = con.execute("PRAGMA page_size").fetchone()[0]
        schema = con.execute("select sql from sqlite_master order by name").fetchall()
        rows = con.execute("select * from optical_flow").fetchall() + con.execute("select * from keypoints").fetchall()
        con.close()
        return v, schema, rows

    monkeypatch.delenv("POLYCHASE_DB_PAGE_SIZE", raising=False)
    a, b, c, d = (str(tmp_path / n) for n in ("default.db", "reference.db", "bogus.db", "mid.db"))
    fill(a)
    monkeypatch.setenv("POLYCHASE_DB_PAGE_SIZE", "4096")
    fill(b)
    db = core.Database(a)                      # re-opening an existing file does not change it
    db.close()
    monkeypatch.setenv("POLYCHASE_DB_PAGE_SIZE", "12345")
    fill(c)
    monkeypatch.setenv("POLYCHASE_DB_PAGE_SIZE", "32768")
    fill(d)
    pa, sa, ra = page_size(a)
    pb, sb, rb = page_size(b)
    pc_, sc, rc = page_size(c)
    pd, sd, rd = page_size(d)
    assert pa == 65536 and pb == 4096 and pc_ == 4096 and pd == 32768
    assert sa == sb == sc == sd and ra == rb == rc == rd


def test_bulk_load_journal_mode_round_trip(core, tmp_path):
    """The analysis driver inserts under a rollback journal and restores WAL mode at the end: the switch keeps the
    rows, the final header says WAL (file-format bytes 18/19 == 2) exactly like a database the reference wrote, and a
    second connection keeps the file in WAL mode (the switch then reports "wal" and the driver stays in WAL mode)."""
    path = str(tmp_path / "j.db")
    kp = np.arange(20, dtype=np.float32).reshape(10, 2)
    db = core.Database(path)
    db.write_keypoints(1, kp)
    assert open(path, "rb").read(20)[18:20] == b"\x02\x02"          # Open() set WAL, like cpp/database.cc:88-92
    assert db._set_journal_mode("TRUNCATE") == "truncate"
    db.write_keypoints(2, kp)
    db.write_image_pair_flow(2, 1, np.arange(5, dtype=np.uint32), kp[:5], np.ones(5, np.float32))
    assert open(path, "rb").read(20)[18:20] == b"\x01\x01"
    assert db._set_journal_mode("WAL") == "wal"
    db.close()
    assert open(path, "rb").read(20)[18:20] == b"\x02\x02"
    con = sqlite3.connect(path)
    assert con.execute("select count(*) from keypoints").fetchone()[0] == 2
    assert con.execute("select count(*) from optical_flow").fetchone()[0] == 1
    assert con.execute("PRAGMA journal_mode").fetchone()[0] == "wal"
    db2 = core.Database(path)                                        # with a second connection open the mode cannot change
    con.execute("select * from keypoints").fetchall()
    assert db2._set_journal_mode("TRUNCATE") == "wal"              # refused: stays WAL, no error
    assert db2._set_journal_mode("WAL") == "wal"
    db2.close()
    con.close()


def test_bulk_writer_connection_writes_the_same_file(core, tmp_path):
    """The analysis' writer connection hands its page writes to worker threads (csrc/host/async_write_vfs.h).  The same
    inserts through it and through a plain connection give byte-identical files; rows written earlier in a transaction can be
    read back inside it (a read of the file waits for the pending pages); a rollback leaves nothing; the WAL switch at the end
    and a reader on the default VFS see everything."""
    rng = np.random.default_rng(5)
    frames = []
    for f in range(1, 13):
        n = int(rng.integers(9000, 12000))          # ~90 KB of keypoints, 4 x ~130 KB of flows per frame: well over a page
        kp = rng.random((n, 2), dtype=np.float32) * 1000
        flows = []
        for to in (f - 2, f - 1, f + 1, f + 2):
            m = int(rng.integers(n // 2, n))
            flows.append((to, np.sort(rng.choice(n, m, replace=False)).astype(np.uint32), rng.random((m, 2), dtype=np.float32), rng.random(m, dtype=np.float32)))
        frames.append((f, kp, flows))

    def fill(db, check_reads):
        assert db._set_journal_mode("TRUNCATE") == "truncate"
        for start in range(0, len(frames), 5):                      # transactions of 5, 5 and 2 frames
            db._begin()
            for f, kp, flows in frames[start:start + 5]:
                db.write_keypoints(f, kp)
                for to, idx, tgt, err in flows:
                    db.write_image_pair_flow(f, to, idx, tgt, err)
                if check_reads and f % 4 == 0:                       # inside the transaction, pages of it already spilled
                    assert np.array_equal(db.read_keypoints(f), kp)
                    got = db.read_image_pair_flow(f, flows[2][0])
                    assert np.array_equal(got.src_kps_indices, flows[2][1]) and np.array_equal(got.tgt_kps, flows[2][2])
            db._commit()
        # a transaction that is rolled back: nothing of it may stay
        db._begin()
        db.write_keypoints(99, frames[0][1])
        db.write_image_pair_flow(99, 98, *frames[0][2][0][1:])
        db._rollback()
        assert not db.keypoints_exist(99)
        assert db._set_journal_mode("WAL") == "wal"
        db.close()

    before = core._async_write_counters()
    fill(core.Database._open_bulk_writer(str(tmp_path / "a.db")), True)
    after = core._async_write_counters()
    fill(core.Database(str(tmp_path / "b.db")), False)
    assert core._async_write_counters()["deferred_writes"] == after["deferred_writes"]     # the plain connection defers nothing
    a, b = open(tmp_path / "a.db", "rb").read(), open(tmp_path / "b.db", "rb").read()
    assert len(a) > 12 * 500_000 and a == b
    assert after["deferred_writes"] - before["deferred_writes"] > len(a) // 65536 // 2    # most pages went through the workers
    assert after["deferred_bytes"] - before["deferred_bytes"] >= (after["deferred_writes"] - before["deferred_writes"]) * 4096
    con = sqlite3.connect(str(tmp_path / "a.db"))
    assert con.execute("PRAGMA integrity_check").fetchone()[0] == "ok"
    assert con.execute("select count(*), sum(rows) from keypoints").fetchone() == (12, sum(len(k) for _, k, _ in frames))
    row = con.execute("select src_keypoints_indices, tgt_keypoints, flow_errors from optical_flow where image_id_from=7 and image_id_to=8").fetchone()
    to, idx, tgt, err = frames[6][2][2]
    assert to == 8 and row[0] == idx.tobytes() and row[1] == tgt.tobytes() and row[2] == err.tobytes()
    con.close()

    # a second bulk-writer connection on a file that already has one is an ordinary connection (no second set of workers)
    first = core.Database._open_bulk_writer(str(tmp_path / "c.db"))
    second = core.Database._open_bulk_writer(str(tmp_path / "c.db"))
    first.write_keypoints(1, frames[0][1])
    assert np.array_equal(second.read_keypoints(1), frames[0][1])
    second.close()
    first.close()

    # POLYCHASE_DB_WRITE_THREADS=0: the bulk writer is an ordinary connection too
    os.environ["POLYCHASE_DB_WRITE_THREADS"] = "0"
    try:
        c0 = core._async_write_counters()["deferred_writes"]
        db = core.Database._open_bulk_writer(str(tmp_path / "d.db"))
        db._set_journal_mode("TRUNCATE")
        db.write_keypoints(1, frames[0][1])
        db.close()
        assert core._async_write_counters()["deferred_writes"] == c0
    finally:
        del os.environ["POLYCHASE_DB_WRITE_THREADS"]


def test_bulk_writer_reports_a_failed_deferred_write_at_the_commit(tmp_path):
    """A page write that fails on the worker thread (file size limit here) must fail the transaction: the commit raises, the
    journal rolls the file back, and the database is intact with exactly the rows committed before.  In a subprocess: the
    limit is per process."""
    import subprocess
    import textwrap
    script = textwrap.dedent("""
        import os, resource, signal, sqlite3, sys
        import numpy as np
        sys.path.insert(0, os.path.join(%r, "polychase_amd", "core"))
        import polychase_core as core
        path = sys.argv[1]
        signal.signal(signal.SIGXFSZ, signal.SIG_IGN)          # write(2) then fails with EFBIG instead of killing the process
        kp = (np.arange(2 * 30000, dtype=np.float32).reshape(-1, 2))          # 240 KB per row
        db = core.Database._open_bulk_writer(path)
        assert db._set_journal_mode("TRUNCATE") == "truncate"
        db._begin(); db.write_keypoints(1, kp); db.write_keypoints(2, kp); db._commit()
        before = core._async_write_counters()["deferred_writes"]
        resource.setrlimit(resource.RLIMIT_FSIZE, (2 * 1024 * 1024, resource.getrlimit(resource.RLIMIT_FSIZE)[1]))
        failed = False
        try:
            db._begin()
            for f in range(3, 40):                                           # 9 MB: far beyond the limit
                db.write_keypoints(f, kp)
            db._commit()
        except Exception as e:
            failed = True
            print("raised:", str(e)[:120])
        assert failed, "the commit went through although the pages could not be written"
        assert core._async_write_counters()["deferred_writes"] > before
        try:
            db._rollback()
        except Exception:
            pass
        try:
            db.close()
        except Exception:
            pass
        resource.setrlimit(resource.RLIMIT_FSIZE, (resource.RLIM_INFINITY, resource.getrlimit(resource.RLIMIT_FSIZE)[1]))
        con = sqlite3.connect(path)
        assert con.execute("PRAGMA integrity_check").fetchone()[0] == "ok"
        assert [r[0] for r in con.execute("select image_id from keypoints order by image_id")] == [1, 2]
        assert con.execute("select keypoints from keypoints where image_id = 2").fetchone()[0] == kp.tobytes()
        print("intact")
    """) % ROOT
    r = subprocess.run([sys.executable, "-c", script, str(tmp_path / "full.db")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "intact" in r.stdout, r.stdout + r.stderr
