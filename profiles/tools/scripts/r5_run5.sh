cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 ) > gpurun_out/r5_gpu_tests.log; tail -4 gpurun_out/r5_gpu_tests.log
AB_ARGS="--workload c3_closeup_968x1296_1M --views 4 --steps 5 --warmup 1" AB_LIBS="networks=freesplat_amd/libfreesplat_hip_lsold.so,partitioned=" timeout 600 python profiles/tools/raster_ab.py > gpurun_out/r5_long_sort_ab.txt 2>&1; cat gpurun_out/r5_long_sort_ab.txt
timeout 600 rocprofv3 --kernel-trace -d gpurun_out/c3_step_trace -o x --output-format csv -- python bench_c3_step.py --trace-steps 2 --warmup 2 > gpurun_out/c3_step_trace.log 2>&1
python profiles/tools/c3_step_glue.py gpurun_out/c3_step_trace 2 > gpurun_out/r5_c3_step_glue.json 2> gpurun_out/c3_step_glue.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r5_c3_step_glue.json"))
print({k:d[k] for k in ("kernels_in_window","per_step_ms","window_ms_per_step","gpu_idle_ms_per_step","glue_frac_of_hotpath_gpu_time","launches_per_step")})
for k in d["top_glue_kernels"][:10]: print(k)
PY
timeout 600 python bench_c3_step.py --steps 3 > gpurun_out/r5_c3_step.json 2> gpurun_out/r5_c3_step.err; head -c 1800 gpurun_out/r5_c3_step.json
