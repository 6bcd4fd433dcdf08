cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for lib in "" _swz0 _swz0bl0 _swz0tf4; do
for which in c3 fvt10; do
echo "== lib$lib (FS_CV_BWD16=1) $which"
rm -rf /tmp/prof_x
FS_CV_BWD16=1 FREESPLAT_LIB=$GRAFT_REPO_ROOT/freesplat_amd/libfreesplat_hip$lib.so rocprofv3 --kernel-trace -d /tmp/prof_x -o x --output-format csv -- python profiles/tools/cv_train_prof.py $which 4 2>&1 | grep "train step"
python profiles/tools/kstats.py /tmp/prof_x | grep bwd_kernel
done
done
FS_CV_BWD16=1 FREESPLAT_LIB=$GRAFT_REPO_ROOT/freesplat_amd/libfreesplat_hip_swz0.so python -m pytest tests/test_cost_volume_hip.py -m gpu -x -q -k "grad or backward" 2>&1 | tail -2
