# Round-6 final evidence refresh (GPU box, repo root)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6_profiles
O=gpurun_out/r6_profiles
: > $O/r6_ptf_train_kernel_stats.csv
for shape in "2 384 512" "3 968 1296"; do
  rm -rf /tmp/prof_x
  rocprofv3 --kernel-trace -d /tmp/prof_x -o x --output-format csv -- python profiles/tools/ptf_train_prof.py $shape > /tmp/ptft.log 2>&1
  python profiles/tools/kstats.py /tmp/prof_x "rocprofv3 --kernel-trace -- python profiles/tools/ptf_train_prof.py $shape   ($(grep 'ms/step' /tmp/ptft.log | tail -1))" | head -14 >> $O/r6_ptf_train_kernel_stats.csv
done
: > $O/r6_encoder_tail_kernel_stats.csv
for shape in "2 192 256" "3 484 648"; do
  rm -rf /tmp/prof_x
  rocprofv3 --kernel-trace -d /tmp/prof_x -o x --output-format csv -- python profiles/tools/depth_tail_prof.py $shape > /tmp/dt.log 2>&1
  python profiles/tools/kstats.py /tmp/prof_x "rocprofv3 --kernel-trace -- python profiles/tools/depth_tail_prof.py $shape   ($(grep 'ms/step' /tmp/dt.log | tail -1))" | head -6 >> $O/r6_encoder_tail_kernel_stats.csv
done
for which in c3 fvt10 native; do
  rm -rf /tmp/prof_x
  rocprofv3 --kernel-trace -d /tmp/prof_x -o x --output-format csv -- python profiles/tools/cv_train_prof.py $which 6 > /tmp/cvt.log 2>&1
  python profiles/tools/kstats.py /tmp/prof_x "rocprofv3 --kernel-trace -- python profiles/tools/cv_train_prof.py $which 6   ($(grep 'train step' /tmp/cvt.log))" | head -14 > $O/r6_cv_train_${which}_kernel_stats.csv
done
rm -rf /tmp/prof_x
rocprofv3 --kernel-trace -d /tmp/prof_x -o x --output-format csv -- python bench.py --steps 5 --warmup 2 --sections raster --no-graph --no-cpu-baseline --no-profile --min-time 0 > /dev/null 2>&1
python profiles/tools/kstats.py /tmp/prof_x "rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --sections raster --no-graph --no-cpu-baseline --no-profile --min-time 0" | head -10 > $O/r6_kernel_stats.csv
rm -rf /tmp/prof_x
rocprofv3 --kernel-trace -d /tmp/prof_x -o x --output-format csv -- python bench.py --steps 5 --warmup 2 --sections raster --no-graph --no-cpu-baseline --no-profile --min-time 0 --mode train > /dev/null 2>&1
python profiles/tools/kstats.py /tmp/prof_x "rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --sections raster --no-graph --no-cpu-baseline --no-profile --min-time 0 --mode train" | head -12 > $O/r6_train_kernel_stats.csv
rm -rf gpurun_out/c3_step_trace
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/c3_step_trace -o x --output-format csv -- python bench_c3_step.py --trace-steps 2 --warmup 3 > gpurun_out/c3_step_trace.log 2>&1
python profiles/tools/c3_step_glue.py gpurun_out/c3_step_trace 2 > $O/r6_c3_step_glue.json 2> gpurun_out/c3_step_glue.err
rm -rf gpurun_out/c3_step_trace
timeout 1500 python -m pytest tests -m gpu -q > $O/r6_gpu_tests.log 2>&1
tail -3 $O/r6_gpu_tests.log
timeout 1200 python bench.py > $O/r6_bench_stdout.log 2>&1
tail -1 $O/r6_bench_stdout.log > $O/r6_bench_headline.json
cp gpurun_out/bench_full.json $O/r6_bench.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python -c "
import json
h=json.load(open('$O/r6_bench_headline.json')); print(h['value'], h['sections']['train']['value'])
d=json.load(open('$O/r6_c3_step_glue.json')); print(d['per_step_ms'])
"
