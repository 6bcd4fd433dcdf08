cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in w7 w7s w8; do FREESPLAT_LIB=$PWD/freesplat_amd/libfreesplat_hip_$v.so timeout 600 python -m pytest tests/test_raster_hip.py -x -q -m gpu 2>&1 | tail -2; done > gpurun_out/g6_tests.log 2>&1
AB_VARIANTS="base:|w7:FREESPLAT_LIB=freesplat_amd/libfreesplat_hip_w7.so|w7s:FREESPLAT_LIB=freesplat_amd/libfreesplat_hip_w7s.so|w8:FREESPLAT_LIB=freesplat_amd/libfreesplat_hip_w8.so" AB_REPEAT=3 timeout 900 python profiles/tools/raster_env_ab.py > gpurun_out/g6_ab.log 2>&1
AB_VARIANTS="base:|w7:FREESPLAT_LIB=freesplat_amd/libfreesplat_hip_w7.so|w8:FREESPLAT_LIB=freesplat_amd/libfreesplat_hip_w8.so" AB_REPEAT=2 timeout 900 python profiles/tools/raster_env_ab.py train > gpurun_out/g6_ab_train.log 2>&1
AB_ARGS="--workload c3_closeup_968x1296_1M --views 4 --steps 5" AB_VARIANTS="base:|w7:FREESPLAT_LIB=freesplat_amd/libfreesplat_hip_w7.so|w8:FREESPLAT_LIB=freesplat_amd/libfreesplat_hip_w8.so" AB_REPEAT=2 timeout 900 python profiles/tools/raster_env_ab.py > gpurun_out/g6_ab_closeup.log 2>&1
cat gpurun_out/g6_tests.log gpurun_out/g6_ab.log; echo train; cat gpurun_out/g6_ab_train.log; echo closeup; cat gpurun_out/g6_ab_closeup.log
