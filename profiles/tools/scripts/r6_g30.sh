cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_cost_volume_hip.py -m gpu -x -q -k "grad or backward or sweep or two_pass" 2>&1 | tail -3
for form in 1 0; do
for which in c3 fvt10 native; do
echo "== FORM=$form $which"
rm -rf /tmp/prof_x
FS_CV_SG_FORM=$form rocprofv3 --kernel-trace -d /tmp/prof_x -o x --output-format csv -- python profiles/tools/cv_train_prof.py $which 4 2>&1 | grep "train step"
python profiles/tools/kstats.py /tmp/prof_x | grep "src_grad"
done; done
