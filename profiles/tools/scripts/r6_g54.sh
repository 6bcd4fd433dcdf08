cd $GRAFT_REPO_ROOT
FREESPLAT_LIB=$GRAFT_REPO_ROOT/freesplat_amd/libfreesplat_hip_trace.so python profiles/cv_phase_trace.py 2>/dev/null | head -3
echo "== K=2 forced deep shape"
FREESPLAT_LIB=$GRAFT_REPO_ROOT/freesplat_amd/libfreesplat_hip_trace.so FS_CV_FWD_SHAPE=4 python profiles/cv_phase_trace.py 2>/dev/null | sed -n 2p
