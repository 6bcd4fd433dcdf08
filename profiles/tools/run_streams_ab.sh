mkdir -p gpurun_out/r3d
python -m pytest tests/test_raster_hip.py tests/test_raster_oracle.py -x -q 2>&1 | tail -8 > gpurun_out/r3d/raster_test.log; tail -3 gpurun_out/r3d/raster_test.log
for s in 2 3 1 2; do
FREESPLAT_RASTER_STREAMS=$s python bench.py --sections raster --no-cpu-baseline --no-graph 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('streams',$s, round(d['value'],1), {k:round(v,4) for k,v in d['kernel_ms_per_view'].items()}, round(d['roofline']['avg_launch_ms'],4))" | tee -a gpurun_out/r3d/streams.log
done
