cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_raster_hip.py -m gpu -q -x 2>&1 | tail -3
AB_LIBS="base=freesplat_amd/lib_base.so,new=" python profiles/tools/raster_ab.py train
