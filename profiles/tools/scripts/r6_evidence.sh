# Round-6 evidence pass (GPU box, repo root): kernel stats / HBM traffic / SQ counters per kernel family, copied to gpurun_out/r6_profiles/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6_profiles
CMD="python bench.py --steps 5 --warmup 2 --sections raster --no-graph --no-cpu-baseline --no-profile --min-time 0"
bash profiles/run_rocprof.sh r6 > /dev/null 2>&1
python profiles/summarize.py gpurun_out/prof_r6 r6 "$CMD" > /dev/null 2>&1
bash profiles/run_rocprof_train.sh r6 > /dev/null 2>&1
python profiles/summarize.py gpurun_out/prof_r6_train r6_train "$CMD --mode train" > /dev/null 2>&1
for which in c3 fvt10 native; do
  rm -rf /tmp/prof_x
  rocprofv3 --kernel-trace -d /tmp/prof_x -o x --output-format csv -- python profiles/tools/cv_train_prof.py $which 6 > /tmp/cvt.log 2>&1
  python profiles/tools/kstats.py /tmp/prof_x "rocprofv3 --kernel-trace -- python profiles/tools/cv_train_prof.py $which 6   ($(grep 'train step' /tmp/cvt.log))" | head -14 > profiles/r6_cv_train_${which}_kernel_stats.csv
done
: > profiles/r6_ptf_train_kernel_stats.csv
for shape in "2 384 512" "3 968 1296"; do
  rm -rf /tmp/prof_x
  rocprofv3 --kernel-trace -d /tmp/prof_x -o x --output-format csv -- python profiles/tools/ptf_train_prof.py $shape > /tmp/ptft.log 2>&1
  python profiles/tools/kstats.py /tmp/prof_x "rocprofv3 --kernel-trace -- python profiles/tools/ptf_train_prof.py $shape   ($(tail -1 /tmp/ptft.log))" | head -14 >> profiles/r6_ptf_train_kernel_stats.csv
done
bash profiles/tools/fwd_traffic.sh r6 > gpurun_out/r6_profiles/fwd_traffic.log 2>&1
cp profiles/r6_* gpurun_out/r6_profiles/ 2>/dev/null
rm -rf gpurun_out/prof_r6 gpurun_out/prof_r6_train gpurun_out/traffic_r6
ls gpurun_out/r6_profiles
