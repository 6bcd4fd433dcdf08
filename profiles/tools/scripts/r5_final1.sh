cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 ) > gpurun_out/r5_gpu_tests.log; tail -4 gpurun_out/r5_gpu_tests.log
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/c3_step_trace -o x --output-format csv -- python bench_c3_step.py --trace-steps 2 --warmup 3 > gpurun_out/c3_step_trace.log 2>&1
python profiles/tools/c3_step_glue.py gpurun_out/c3_step_trace 2 > gpurun_out/r5_c3_step_glue.json 2> gpurun_out/c3_step_glue.err && cp gpurun_out/r5_c3_step_glue.json profiles/r5_c3_step_glue.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5_c3_step_glue.json"))
print({k:d[k] for k in ("kernels_in_window","per_step_ms","window_ms_per_step","gpu_idle_ms_per_step","glue_frac_of_hotpath_gpu_time","launches_per_step")})
for k in d["top_glue_kernels"][:8]: print(k)
PY
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r5_bench_stdout.log 2> gpurun_out/r5_bench_stderr.log
tail -c 600 gpurun_out/r5_bench_stderr.log
tail -n 1 gpurun_out/r5_bench_stdout.log | head -c 4200; echo
profiles/run_rocprof.sh r5 > gpurun_out/r5_rocprof.log 2>&1
profiles/run_rocprof_train.sh r5 > gpurun_out/r5_rocprof_train.log 2>&1
python profiles/summarize.py gpurun_out/prof_r5 r5 "python bench.py --steps 5 --warmup 2 --sections raster --no-graph --no-cpu-baseline --no-profile --min-time 0" > gpurun_out/r5_summarize.log 2>&1; tail -12 gpurun_out/r5_summarize.log
python profiles/summarize.py gpurun_out/prof_r5_train r5_train "python bench.py --steps 3 --warmup 1 --sections raster --no-graph --no-cpu-baseline --no-profile --min-time 0 --mode train" > gpurun_out/r5_summarize_train.log 2>&1; tail -12 gpurun_out/r5_summarize_train.log
mkdir -p gpurun_out/profiles_r5 && cp profiles/r5_*kernel_stats.csv profiles/r5_*hbm_traffic.json profiles/r5_*sq_counters.json profiles/r5_c3_step_glue.json gpurun_out/profiles_r5/ 2>/dev/null; ls gpurun_out/profiles_r5
