"""GPU: the multi-GPU analysis of a C++ host -- csrc/host/multi_gpu.cc (GenerateOpticalFlowDatabaseMultiGpu) over the RCCL
entry points of the C ABI (include/polychase_hip.h: pc_comm_*), launched by polychase_amd/lib/polychase_multi_gpu
(tools/multi_gpu/multi_gpu_analyze.cc): no Python and no torch underneath.

The database must not depend on the number of ranks (keypoint order, flow order and float bits identical for
any number of ranks; the sharded loop is the reference's cpp/opticalflow.cc:209-321).  A one-GPU box runs several ranks on
GPU 0 with the payload over the TCP control connection (`--transport tcp`, a testing aid: RCCL refuses two ranks on one
device); RCCL itself runs on a one-rank communicator here and between two ranks where a second GPU exists."""
import ctypes as C
import json
import os
import socket
import sqlite3
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from polychase_amd import build, hip  # noqa: E402


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _dump(path):
    con = sqlite3.connect(path)
    k = list(con.execute("select rowid, image_id, rows, keypoints from keypoints order by rowid"))
    f = list(con.execute("select rowid, image_id_from, image_id_to, rows, src_keypoints_indices, tgt_keypoints, flow_errors "
                         "from optical_flow order by rowid"))
    con.close()
    return k, f


def _run(tmp_path, name, gpus, *extra, frames=22, size=(320, 240)):
    tool = build.multi_gpu_tool_path()
    assert os.path.exists(tool), "python -m polychase_amd.build builds polychase_amd/lib/polychase_multi_gpu"
    db = str(tmp_path / f"{name}.db")
    env = {k: v for k, v in os.environ.items() if k not in ("POLYCHASE_RANK", "POLYCHASE_DEVICE")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([tool, "--gpus", str(gpus), "--database", db, "--width", str(size[0]), "--height", str(size[1]), "--frames", str(frames),
                        "--max-level", "2", "--port", str(_free_port()), "--piece-frames", "3", *extra],
                       text=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-1500:], r.stderr[-3000:])
    return db, json.loads(lines[0])


def test_database_does_not_depend_on_the_number_of_ranks(tmp_path):
    """1 rank (the single-GPU driver) vs 2 and 3 ranks that share GPU 0 (payload over TCP): byte-identical rows in the same
    order; the ranks > 0 really sent pieces (several per rank: 3 frames each)."""
    one, j1 = _run(tmp_path, "one", 1)
    assert j1["world_size"] == 1 and j1["keypoint_rows"] == 22 and j1["flow_rows"] > 100
    ref = _dump(one)
    assert len(ref[0]) == 22
    for n in (2, 3):
        db, j = _run(tmp_path, f"tcp{n}", n, "--transport", "tcp", "--share-gpu")
        assert j["world_size"] == n and j["pieces_received"] >= 2 * (n - 1) and j["bytes_received"] > 0
        assert j["keypoint_rows"] == j1["keypoint_rows"] and j["flow_rows"] == j1["flow_rows"]
        assert _dump(db) == ref, f"{n} ranks wrote another database than one rank"


def test_the_cpp_database_equals_the_python_products(tmp_path):
    """the same clip through polychase_core.generate_optical_flow_database (what the addon calls) -> the same rows as the
    C++ multi-rank run: the procedural clip of the tool is restated here with the same integer arithmetic"""
    sys.path.insert(0, os.path.join(ROOT, "polychase_amd", "core"))
    import polychase_core as core
    w, h, n = 160, 120, 12

    def lattice(x, y, cell):
        def hsh(a, b):
            a = a.astype(np.uint64)
            b = b.astype(np.uint64)
            v = ((a * 0x9E3779B1) & 0xFFFFFFFF) ^ (((b + 0x7F4A7C15) & 0xFFFFFFFF) * 0x85EBCA77 & 0xFFFFFFFF)
            v ^= v >> 15
            v = (v * 0xC2B2AE3D) & 0xFFFFFFFF
            v ^= v >> 13
            return (v & 255).astype(np.int64)
        cx, cy, fx, fy = x // cell, y // cell, x % cell, y % cell
        top = hsh(cx, cy) * (cell - fx) + hsh(cx + 1, cy) * fx
        bot = hsh(cx, cy + 1) * (cell - fx) + hsh(cx + 1, cy + 1) * fx
        return (top * (cell - fy) + bot * fy) // (cell * cell)

    def frame(t):
        ys, xs = np.mgrid[0:h, 0:w].astype(np.int64)
        ox, oy = 64 + t, 64 + t // 2
        v = (2 * lattice(xs + ox, ys + oy, 6) + lattice(xs + ox + 1000, ys + oy + 500, 17) + 1) // 3
        return np.ascontiguousarray(np.repeat(v.astype(np.uint8)[:, :, None], 3, axis=2))

    fo = core.OpticalFlowOptions()
    fo.max_level = 2
    py_db = str(tmp_path / "py.db")
    core.generate_optical_flow_database(core.VideoInfo(w, h, 1, n), lambda f: frame(f - 1), None, py_db, core.GFTTOptions(), fo)
    cpp_db, _ = _run(tmp_path, "cpp", 2, "--transport", "tcp", "--share-gpu", frames=n, size=(w, h))
    assert _dump(cpp_db) == _dump(py_db)
    # ... and through the binding of the same entry point with one rank
    one_db = str(tmp_path / "binding.db")
    r = core.generate_optical_flow_database_multi_gpu(core.VideoInfo(w, h, 1, n), lambda f: frame(f - 1), None, one_db, 1, 0,
                                                       flow_options=fo)
    assert r["shard"] == (1, 1 + n) and not r["cancelled"]
    assert _dump(one_db) == _dump(py_db)


def test_a_failing_rank_ends_every_rank_with_an_error(tmp_path):
    """ranks 1 and 2 cannot produce their frames (the frames file ends inside rank 1's shard; rank 0's shard and its halo
    are complete): rank 0 must report WHICH rank failed and all processes must end -- nobody waits for a credit or a piece
    for ever"""
    w, h, n = 160, 120, 30
    short = tmp_path / "short.rgb"
    short.write_bytes(bytes(w * h * 3 * 20))         # frames 1..20: rank 0 = [1, 11) + halo to 18 is fine, rank 1 = [11, 21) is not
    tool = build.multi_gpu_tool_path()
    r = subprocess.run([tool, "--gpus", "3", "--database", str(tmp_path / "x.db"), "--width", str(w), "--height", str(h), "--frames", str(n),
                        "--max-level", "1", "--port", str(_free_port()), "--transport", "tcp", "--share-gpu", "--frames-file", str(short)],
                       text=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode != 0
    assert "rank 1 failed" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")], r.stderr[-2000:]


def test_rccl_entry_points_on_a_one_rank_communicator():
    """pc_comm_*: librccl loaded at the first call, ncclGetUniqueId, ncclCommInitRank, and the two all-gathers of
    pc_comm_all_gather_log -- every RCCL call of the C++ stitch except send / recv (which need a peer), on the one GPU of a test box"""
    L = hip.load()
    ctx = hip.Context(0)
    ident = (C.c_ubyte * 128)()
    hip._check(L.pc_comm_unique_id(ident))
    assert any(ident)
    comm = C.c_void_p()
    hip._check(L.pc_comm_create(ctx._h, ident, 1, 0, C.byref(comm)))
    assert L.pc_comm_world_size(comm) == 1 and L.pc_comm_rank(comm) == 0
    n, slot = 100_003, 131_072
    src, dst = C.c_void_p(), C.c_void_p()
    hip._check(L.pc_peer_buffer_alloc(0, slot, C.byref(src)))
    hip._check(L.pc_peer_buffer_alloc(0, slot, C.byref(dst)))
    payload = np.random.default_rng(5).integers(0, 256, n, dtype=np.uint8)
    import torch
    t = torch.from_numpy(payload).cuda()
    hip._check(L.pc_peer_copy_async(0, src, C.c_void_p(t.data_ptr()), n, None))
    torch.cuda.synchronize()
    sizes = (C.c_uint64 * 1)()
    hip._check(L.pc_comm_all_gather_log(comm, src, n, dst, slot, sizes))
    assert sizes[0] == n
    back = np.zeros(n, np.uint8)
    hip._check(L.pc_peer_buffer_download(0, back.ctypes.data_as(C.c_void_p), dst, n))
    assert np.array_equal(back, payload)
    assert L.pc_comm_all_gather_log(comm, src, slot + 1, dst, slot, sizes) == -4        # PC_E_CAPACITY: piece larger than the slot
    assert L.pc_comm_send(comm, src, 16, 0) == -1                                      # PC_E_INVALID: a rank cannot send to itself
    L.pc_comm_destroy(comm)
    L.pc_peer_buffer_free(0, src)
    L.pc_peer_buffer_free(0, dst)
    ctx.close()


def test_two_gpus_rccl_send_recv(tmp_path):
    """the product transport: ncclSend / ncclRecv between two GPUs, the same database as one rank"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    one, _ = _run(tmp_path, "one", 1)
    two, j = _run(tmp_path, "rccl2", 2, "--transport", "rccl")
    assert j["pieces_received"] >= 2 and _dump(two) == _dump(one)
