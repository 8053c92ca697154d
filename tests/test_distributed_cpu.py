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
