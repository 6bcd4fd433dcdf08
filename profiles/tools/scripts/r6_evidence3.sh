# Round-6 final evidence refresh (GPU box, repo root): PTF kernel stats, GPU suite, bench, smoke
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6_profiles
: > profiles/r6_ptf_train_kernel_stats.csv
for shape in "2 384 512" "3 968 1296"; do
  rm -rf /tmp/prof_x
  rocprofv3 --kernel-trace -d /tmp/prof_x -o x --output-format csv -- python profiles/tools/ptf_train_prof.py $shape > /tmp/ptft.log 2>&1
  python profiles/tools/kstats.py /tmp/prof_x "rocprofv3 --kernel-trace -- python profiles/tools/ptf_train_prof.py $shape   ($(grep 'ms/step' /tmp/ptft.log | tail -1))" | head -14 >> profiles/r6_ptf_train_kernel_stats.csv
done
for which in c3 fvt10 native; do
  rm -rf /tmp/prof_x
  rocprofv3 --kernel-trace -d /tmp/prof_x -o x --output-format csv -- python profiles/tools/cv_train_prof.py $which 6 > /tmp/cvt.log 2>&1
  python profiles/tools/kstats.py /tmp/prof_x "rocprofv3 --kernel-trace -- python profiles/tools/cv_train_prof.py $which 6   ($(grep 'train step' /tmp/cvt.log))" | head -14 > profiles/r6_cv_train_${which}_kernel_stats.csv
done
cp profiles/r6_ptf_train_kernel_stats.csv profiles/r6_cv_train_*_kernel_stats.csv gpurun_out/r6_profiles/
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r6_profiles/r6_gpu_tests.log 2>&1
tail -3 gpurun_out/r6_profiles/r6_gpu_tests.log
timeout 1200 python bench.py > gpurun_out/r6_profiles/r6_bench_stdout.log 2>&1
tail -1 gpurun_out/r6_profiles/r6_bench_stdout.log > gpurun_out/r6_profiles/r6_bench_headline.json
cp gpurun_out/bench_full.json gpurun_out/r6_profiles/r6_bench.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
