cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_cost_volume_hip.py -m gpu -x -q -k "grad or backward" > gpurun_out/g18_tests.log 2>&1
tail -5 gpurun_out/g18_tests.log
for which in c3 fvt10 native; do for v in 2 1 0; do echo "BWD16=$v"; FS_CV_BWD16=$v timeout 600 python profiles/tools/cv_train_prof.py $which 6 2>&1 | tail -1; done; done
for which in c3 fvt10; do
rm -rf /tmp/prof_x
rocprofv3 --kernel-trace -d /tmp/prof_x -o x --output-format csv -- python profiles/tools/cv_train_prof.py $which 4 2>&1 | grep "train step"
python profiles/tools/kstats.py /tmp/prof_x | head -5
done
