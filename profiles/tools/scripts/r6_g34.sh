cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for lib in "" _nopf; do
for which in fvt10 c3; do
echo "== lib$lib $which SAVE=1"
rm -rf /tmp/prof_x
FREESPLAT_CV_SAVE=1 FREESPLAT_LIB=$GRAFT_REPO_ROOT/freesplat_amd/libfreesplat_hip$lib.so rocprofv3 --kernel-trace -d /tmp/prof_x -o x --output-format csv -- python profiles/tools/cv_train_prof.py $which 4 2>&1 | grep "train step"
python profiles/tools/kstats.py /tmp/prof_x | grep "16_bwd"
done; done
FREESPLAT_CV_SAVE=1 python -m pytest tests/test_cost_volume_hip.py -m gpu -x -q -k "grad or backward" 2>&1 | tail -2
