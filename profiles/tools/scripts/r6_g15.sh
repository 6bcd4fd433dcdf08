cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_cost_volume_hip.py -m gpu -x -q -k "grad or backward" > gpurun_out/g15_tests.log 2>&1
tail -5 gpurun_out/g15_tests.log
for which in c3 fvt10 native; do for v in 1 0; do echo "BWD16=$v"; FS_CV_BWD16=$v timeout 600 python profiles/tools/cv_train_prof.py $which 6 2>&1 | tail -1; done; done
cd /tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/g15_prof -o c3 -- python $GRAFT_REPO_ROOT/profiles/tools/cv_train_prof.py c3 6 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python profiles/tools/kstats.py gpurun_out/g15_prof | head -6
