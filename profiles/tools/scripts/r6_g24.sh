cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_cost_volume_hip.py -m gpu -x -q 2>&1 | tail -3
for which in c3 fvt10 native; do for sv in 0 1; do
echo "== SAVE=$sv $which"
rm -rf /tmp/prof_x
FREESPLAT_CV_SAVE=$sv rocprofv3 --kernel-trace -d /tmp/prof_x -o x --output-format csv -- python profiles/tools/cv_train_prof.py $which 4 2>&1 | grep "train step"
python profiles/tools/kstats.py /tmp/prof_x | grep "bwd_kernel\|cost_volume16_kernel\|src_grad\|proj_kernel"
done; done
