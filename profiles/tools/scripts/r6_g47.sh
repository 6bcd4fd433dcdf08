cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
AB_LIBS="base=freesplat_amd/lib_base.so,w3=freesplat_amd/libfreesplat_hip_w3.so" python profiles/tools/raster_ab.py train
