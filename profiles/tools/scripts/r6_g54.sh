cd $GRAFT_REPO_ROOT
FREESPLAT_LIB=$GRAFT_REPO_ROOT/freesplat_amd/libfreesplat_hip_trace.so python profiles/cv_phase_trace.py 3 2 242 324 128
FREESPLAT_LIB=$GRAFT_REPO_ROOT/freesplat_amd/libfreesplat_hip_trace.so python profiles/cv_phase_trace.py 10 8 96 128 128
