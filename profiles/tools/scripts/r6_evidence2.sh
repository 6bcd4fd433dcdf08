# Round-6 evidence refresh after the last kernel changes (GPU box, repo root)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6_profiles
CMD="python bench.py --steps 5 --warmup 2 --sections raster --no-graph --no-cpu-baseline --no-profile --min-time 0"
bash profiles/run_rocprof_train.sh r6 > /dev/null 2>&1
python profiles/summarize.py gpurun_out/prof_r6_train r6_train "$CMD --mode train" > /dev/null 2>&1
for which in c3 fvt10 native; do
  rm -rf /tmp/prof_x
  rocprofv3 --kernel-trace -d /tmp/prof_x -o x --output-format csv -- python profiles/tools/cv_train_prof.py $which 6 > /tmp/cvt.log 2>&1
  python profiles/tools/kstats.py /tmp/prof_x "rocprofv3 --kernel-trace -- python profiles/tools/cv_train_prof.py $which 6   ($(grep 'train step' /tmp/cvt.log))" | head -14 > profiles/r6_cv_train_${which}_kernel_stats.csv
done
bash profiles/tools/fwd_traffic.sh r6 cvt_native_K1 cvt_c3scale_K2 cvt_fvt10_K8 > gpurun_out/r6_profiles/fwd_traffic_cvt.log 2>&1
cp profiles/r6_* gpurun_out/r6_profiles/ 2>/dev/null
rm -rf gpurun_out/prof_r6_train gpurun_out/traffic_r6
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r6_profiles/r6_gpu_tests.log 2>&1
tail -3 gpurun_out/r6_profiles/r6_gpu_tests.log
timeout 1200 python bench.py > gpurun_out/r6_profiles/r6_bench_stdout.log 2>&1
tail -1 gpurun_out/r6_profiles/r6_bench_stdout.log > gpurun_out/r6_profiles/r6_bench_headline.json
cp gpurun_out/bench_full.json gpurun_out/r6_profiles/r6_bench.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
tail -4 gpurun_out/r6_profiles/fwd_traffic_cvt.log
