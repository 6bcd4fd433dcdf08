cd $GRAFT_REPO_ROOT
FREESPLAT_LIB=$PWD/freesplat_amd/libfreesplat_hip_w7.so timeout 600 python -m pytest tests/test_raster_hip.py -x -q -m gpu 2>&1 | tail -40
