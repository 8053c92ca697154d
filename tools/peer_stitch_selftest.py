#!/usr/bin/env python3
"""peer_stitch_selftest.py -- distributed.PeerLogStitch between the ranks of a torchrun job whose ranks all sit on GPU 0 (gloo):
export / map / probe / push three pieces / read every rank's log back and compare.  The ranks' logs have different sizes on
purpose (the benchmark's do: each rank sizes its log from its own frames' keypoint counts).  Exit code 0 = every rank saw
every rank's bytes.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 tools/peer_stitch_selftest.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main() -> int:
    import numpy as np
    import torch
    import torch.distributed as dist

    from polychase_amd import distributed as D

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    side = dist.new_group(backend="gloo")
    n = int(os.environ.get("SELFTEST_LOG_BYTES", 1 << 20)) + 4096 * rank
    rng = np.random.default_rng(100 + rank)
    content = rng.integers(0, 256, n, dtype=np.uint8)
    log = torch.from_numpy(content).cuda()
    st = D.make_log_stitch(log, side_group=side, prefer="peer")
    if not isinstance(st, D.PeerLogStitch):
        print(f"rank {rank}: fell back to {type(st).__name__}", flush=True)
        return 2
    used = 300000 + 1000 * rank
    bad = 0
    for region in range(2):          # the benchmark reuses the stitch for every timed region
        st.reset()
        st.gather(0, 4096)
        st.gather(4096, 100000)
        st.gather(100000, used)
        st.finish()
        dist.barrier()
        for r, (buf, size) in enumerate(st.rank_logs()):
            want = np.random.default_rng(100 + r).integers(0, 256, int(os.environ.get("SELFTEST_LOG_BYTES", 1 << 20)) + 4096 * r,
                                                           dtype=np.uint8)[:300000 + 1000 * r]
            if size != len(want) or not np.array_equal(buf, want):
                print(f"rank {rank}, region {region}: slot {r} differs (size {size}, expected {len(want)})", flush=True)
                bad += 1
        dist.barrier()
    st.close()
    dist.barrier()
    dist.destroy_process_group()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
