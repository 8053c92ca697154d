run() { python bench.py --no-cpu-baseline --no-c3 --no-end-to-end --config $1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(round(d[\"value\"],1), round(d[\"ms_per_step\"],4), round(d[\"roofline\"][\"avg_launch_ms\"],4), round(d[\"roofline\"][\"launch_overlap\"],3), {k: round(v,3) for k,v in d.get(\"kernel_ms_per_frame\").items()})"; }
for c in c2 c3; do
echo "== $c L1 two lanes, gate on"; run $c
echo "== $c L1 gate off"; POLYCHASE_LK_GATE=0 run $c

echo "== $c L3 two lanes + detect stream, 8 queues, gate on"; GPU_MAX_HW_QUEUES=8 POLYCHASE_DETECT_STREAMS=1 run $c
echo "== $c L3 on 4 queues"; POLYCHASE_DETECT_STREAMS=1 run $c
done
