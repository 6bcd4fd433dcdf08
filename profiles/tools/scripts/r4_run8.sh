cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_raster_hip.py -q -m gpu -k "backward or gradient or overflow or fp16" 2>&1 | grep -E "passed|failed|Error|^E  |^FAILED" | cut -c1-600 | head -20 | tee gpurun_out/r4_raster_tests.log
for i in 1 2; do
for v in 1 0; do
echo -n "FS_RASTER_BWD_DEPTH=$v: "
FS_RASTER_BWD_DEPTH=$v python bench.py --sections raster --no-cpu-baseline --no-graph --mode train --views 8 --steps 10 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value'],1), {k: round(v,4) for k,v in d['kernel_ms_per_view'].items()})"
done; done 2>&1 | tee gpurun_out/r4_bwd_depth_ab.txt
