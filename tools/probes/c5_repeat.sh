#!/bin/bash
# Run ON the GPU box: C5 end to end N times -> gpurun_out/<tag>_<i>.json   tools/probes/c5_repeat.sh <tag> [n]
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
TAG=${1:-c5rep}; N=${2:-3}
cd /tmp
for i in $(seq $N); do python $ROOT/tests/c5_endtoend.py --frames 300 --oracle-frames 0 --out $ROOT/gpurun_out/${TAG}_$i.json > /tmp/c5rep.log 2>&1 || tail -3 /tmp/c5rep.log; done
