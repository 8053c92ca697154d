#!/usr/bin/env python3
"""peer_stitch_selftest.py -- distributed.PeerLogStitch between the ranks of a torchrun job whose ranks all sit on GPU 0 (gloo):
export / map / probe / push two pieces / read every rank's log back.  Logs of different sizes per rank on purpose.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 tools/peer_stitch_selftest.py
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
from polychase_amd import distributed as D
def main():
    rank=int(os.environ["RANK"]); world=int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    side = dist.new_group(backend="gloo")
    log=torch.full((int(os.environ.get("DBG_LOG_BYTES", 1<<20)) + 4096 * rank,), rank+10, dtype=torch.uint8, device="cuda")
    st=D.make_log_stitch(log, side_group=side, prefer="peer")
    print(rank, type(st).__name__, getattr(st, "probe_error", None), flush=True)
    if isinstance(st, D.PeerLogStitch):
        st.reset(); st.gather(0, 4096); st.gather(4096, 10000); st.finish(); dist.barrier()
        for r,(buf,used) in enumerate(st.rank_logs()): print(rank, "rank_logs", r, used, buf[:3], buf[-3:], flush=True)
        st.close()
    dist.barrier()
main()
