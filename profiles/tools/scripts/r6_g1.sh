cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_raster_hip.py tests/test_abi.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/g1_tests.log
AB_VARIANTS="legacy:FREESPLAT_PREPROCESS=legacy|b16:FREESPLAT_RASTER_BATCH=16|b8:FREESPLAT_RASTER_BATCH=8|b4:FREESPLAT_RASTER_BATCH=4|pw5b8:FREESPLAT_LIB=freesplat_amd/libfreesplat_hip_pw5.so" timeout 900 python profiles/tools/raster_env_ab.py > gpurun_out/g1_ab.log 2>&1
cat gpurun_out/g1_tests.log gpurun_out/g1_ab.log
