cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_raster_hip.py tests/test_configs_4_5.py -x -q -m gpu 2>&1 | tail -5 )
run() { python bench.py --sections raster --no-cpu-baseline --no-graph "$@" 2>/dev/null | python -c "
import sys, json
L=[l for l in sys.stdin if l.startswith('{')]; d=json.loads(L[-2])
print(round(d['value'],1), {k: round(v,4) for k,v in d['kernel_ms_per_view'].items()})"; }
for i in 1 2; do
  echo -n "static  fwd  : "; FS_RASTER_TILE_ORDER=0 run
  echo -n "heaviest fwd : "; run
done 2>&1 | tee gpurun_out/r5_tile_order_ab.txt
for i in 1 2; do
  echo -n "static  train: "; FS_RASTER_TILE_ORDER=0 run --mode train --views 8 --steps 10
  echo -n "heaviest train: "; run --mode train --views 8 --steps 10
done 2>&1 | tee -a gpurun_out/r5_tile_order_ab.txt
echo -n "static  closeup: "; FS_RASTER_TILE_ORDER=0 run --workload c3_closeup_968x1296_1M --views 4 --steps 5 --warmup 1 | tee -a gpurun_out/r5_tile_order_ab.txt
echo -n "heaviest closeup: "; run --workload c3_closeup_968x1296_1M --views 4 --steps 5 --warmup 1 | tee -a gpurun_out/r5_tile_order_ab.txt
echo -n "static  c2: "; FS_RASTER_TILE_ORDER=0 run --workload c2_640x480_300k | tee -a gpurun_out/r5_tile_order_ab.txt
echo -n "heaviest c2: "; run --workload c2_640x480_300k | tee -a gpurun_out/r5_tile_order_ab.txt
