cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python profiles/tools/tile_list_hist.py 2>&1 | tail -8
FREESPLAT_LIB=$PWD/freesplat_amd/libfreesplat_hip_w7.so timeout 600 python -m pytest tests/test_raster_hip.py -x -q -m gpu 2>&1 | tail -2
AB_VARIANTS="base:|w7:FREESPLAT_LIB=freesplat_amd/libfreesplat_hip_w7.so" AB_REPEAT=3 timeout 900 python profiles/tools/raster_env_ab.py > gpurun_out/g4_ab.log 2>&1
cat gpurun_out/g4_ab.log
