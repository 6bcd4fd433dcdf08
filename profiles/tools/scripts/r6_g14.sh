cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_cost_volume_hip.py -m gpu -x -q > gpurun_out/g14_tests.log 2>&1
tail -15 gpurun_out/g14_tests.log
for which in c3 fvt10 native; do for v in 1 0; do echo "BWD16=$v"; FS_CV_BWD16=$v timeout 600 python profiles/tools/cv_train_prof.py $which 6 2>&1 | tail -2; done; done
cd /tmp
for v in 1 0; do FS_CV_BWD16=$v rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/g14_prof_$v -o c3 -- python $GRAFT_REPO_ROOT/profiles/tools/cv_train_prof.py c3 6 > /dev/null 2>&1; done
cd $GRAFT_REPO_ROOT
for v in 1 0; do echo "== BWD16=$v"; python profiles/tools/kstats.py $(find gpurun_out/g14_prof_$v -name "*kernel_stats.csv" | head -1) 2>/dev/null | head -8; done
