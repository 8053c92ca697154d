cd /tmp
for rep in 1 2; do for cap in 256 192 128 96 64; do
POLYCHASE_TRACK_LM_BLOCKS=$cap python $GRAFT_REPO_ROOT/tests/c5_endtoend.py --frames 300 --oracle-frames 0 --out $GRAFT_REPO_ROOT/gpurun_out/lmcap_${cap}_$rep.json > /tmp/lmcap.log 2>&1 || tail -3 /tmp/lmcap.log
done; done
