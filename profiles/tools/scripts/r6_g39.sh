cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
FREESPLAT_LIB=$GRAFT_REPO_ROOT/freesplat_amd/libfreesplat_hip_quadw.so python -m pytest tests/test_cost_volume_hip.py -m gpu -x -q -k "grad or backward" 2>&1 | tail -2
for rep in 1 2; do
for lib in "" _quadw; do
for which in c3 fvt10; do
echo "== lib$lib $which"
rm -rf /tmp/prof_x
FREESPLAT_LIB=$GRAFT_REPO_ROOT/freesplat_amd/libfreesplat_hip$lib.so rocprofv3 --kernel-trace -d /tmp/prof_x -o x --output-format csv -- python profiles/tools/cv_train_prof.py $which 6 2>&1 | grep "train step"
python profiles/tools/kstats.py /tmp/prof_x | grep "16_bwd"
done; done; done
