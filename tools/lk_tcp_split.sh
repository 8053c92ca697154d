cd /tmp; export TMPDIR=/tmp
for v in default istage2 jstage2 unpaired; do
  lib=""; [ $v != default ] && lib=$GRAFT_REPO_ROOT/polychase_amd/lib/variants/libpolychase_hip_$v.so
  d=/tmp/p_$v; rm -rf $d
  POLYCHASE_HIP_LIB=$lib timeout 120 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum SQ_INSTS_VMEM_RD TA_BUSY_avr SQ_INSTS_VMEM_WR --kernel-trace -d $d -o r --output-format csv -- python $GRAFT_REPO_ROOT/tools/lk_bench.py --config c2 --reps 3 --arith canonical > /dev/null 2>&1
  f=$(find $d -name '*counter_collection.csv' | head -1)
  python3 - "$f" $v <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float); disp=set()
for r in csv.DictReader(open(sys.argv[1])):
    if 'lk3_kernel' not in r['Kernel_Name']: continue
    acc[r['Counter_Name']] += float(r['Counter_Value']); disp.add(r['Dispatch_Id'])
n=len(disp)
print(sys.argv[2], {k: round(v/n/20320,1) for k,v in acc.items()}, "per wave; dispatches", n)
PY
done
