#!/bin/bash
# L1 (TCP) / texture-addresser (TA) counters of the LK launch: is the staging bound by the vector-memory front end?
# usage (GPU box): tools/lk_tcp_counters.sh [c2|c3] [arith]   -> gpurun_out/lk_tcp_<cfg>_<arith>.txt
cfg=${1:-c2}; arith=${2:-opencv_x86}; win=${3:-10}
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/lk_tcp_${cfg}_${arith}_w${win}.txt; : > $out
for set in "GRBM_GUI_ACTIVE TA_BUSY_avr TA_BUSY_max TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_GATE_EN1_sum" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_TCR_TCP_STALL_CYCLES_sum TD_TD_BUSY_sum" \
           "SQ_WAVES SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_INSTS_VALU"; do
  d=/tmp/pmc_$RANDOM; rm -rf $d
  timeout 120 rocprofv3 --pmc $set --kernel-trace -d $d -o r --output-format csv -- \
      python $GRAFT_REPO_ROOT/tools/lk_bench.py --config $cfg --reps 3 --arith $arith --window $win > /dev/null 2>$d.err || { echo "FAILED: $set" >> $out; tail -3 $d.err >> $out; continue; }
  f=$(find $d -name '*counter_collection.csv' | head -1)
  python3 - "$f" >> $out <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if 'lk3_kernel' not in r['Kernel_Name'] and 'lk4_kernel' not in r['Kernel_Name'] and 'lk_' not in r['Kernel_Name']: continue
    k = (r['Kernel_Name'].split('(')[0][-40:], r['Counter_Name'])
    acc[k][0] += float(r['Counter_Value']); acc[k][1] += 1
# rows are per (dispatch, counter[, dimension instance]); report the per-dispatch mean
disp = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if 'lk3_kernel' in r['Kernel_Name'] or 'lk4_kernel' in r['Kernel_Name']: disp[r['Dispatch_Id']] += 0
nd = max(1, len(disp))
for (kn, cn), (v, n) in sorted(acc.items()):
    print(f"{kn:42s} {cn:44s} per-dispatch {v / nd:16.1f}  (rows {n}, dispatches {nd})")
PY
done
cat $out
