set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_cost_volume_hip.py -q -m gpu -x -k "backward" 2>&1 | tail -30 > gpurun_out/r4_cv_tests.log
cat gpurun_out/r4_cv_tests.log
timeout 600 python profiles/tools/cv_bwd_form_ab.py 2>&1 | tee gpurun_out/r4_cv_bwd_form_ab.txt
cd /tmp && export TMPDIR=/tmp
CV_ONE=fvt10_K8 timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_cv_fvt10 -o cv -- python $GRAFT_REPO_ROOT/profiles/tools/cv_bwd_form_ab.py > $GRAFT_REPO_ROOT/gpurun_out/prof_cv_fvt10.log 2>&1
tail -3 $GRAFT_REPO_ROOT/gpurun_out/prof_cv_fvt10.log
find $GRAFT_REPO_ROOT/gpurun_out/prof_cv_fvt10 -name "*kernel_stats*" | head
f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_cv_fvt10 -name "*kernel_stats.csv" | head -1); head -12 "$f" | cut -c1-220
