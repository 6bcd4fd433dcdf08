cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for lib in trace trace1; do
FS_CV_BWD16=1 FREESPLAT_LIB=$GRAFT_REPO_ROOT/freesplat_amd/libfreesplat_hip_$lib.so python profiles/tools/cv_bwd16_trace.py 2>&1 | tail -1
done
