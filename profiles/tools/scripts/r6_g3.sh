cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_raster_hip.py tests/test_abi.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/g3_tests.log
cat gpurun_out/g3_tests.log
for w in 5 6 7; do FREESPLAT_LIB=$PWD/freesplat_amd/libfreesplat_hip_pw$w.so timeout 600 python -m pytest tests/test_raster_hip.py -x -q -m gpu -k "bit_exact or full_size or closeup or long_list" 2>&1 | tail -2; done
AB_VARIANTS="prev6:FREESPLAT_LIB=freesplat_amd/libfreesplat_hip_prev6.so|new:|pw5:FREESPLAT_LIB=freesplat_amd/libfreesplat_hip_pw5.so|pw6:FREESPLAT_LIB=freesplat_amd/libfreesplat_hip_pw6.so|pw7:FREESPLAT_LIB=freesplat_amd/libfreesplat_hip_pw7.so" timeout 900 python profiles/tools/raster_env_ab.py > gpurun_out/g3_ab.log 2>&1
cat gpurun_out/g3_ab.log
