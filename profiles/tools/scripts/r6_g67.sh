cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for cfg in "16 2" "8 2" "4 2" "8 3" "4 3" "16 3"; do
set -- $cfg
FREESPLAT_RASTER_BATCH=$1 FREESPLAT_RASTER_STREAMS=$2 python bench.py --sections raster --no-cpu-baseline --no-graph 2>/dev/null | tail -2 | head -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('batch $1 streams $2', round(d['value'],1), {k: round(v,4) for k,v in d['kernel_ms_per_view'].items()})"
done; done
