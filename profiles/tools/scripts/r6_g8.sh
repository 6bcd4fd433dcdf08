cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/g8_tests.log
for v in opf opf7 bw7; do FREESPLAT_LIB=$PWD/freesplat_amd/libfreesplat_hip_$v.so timeout 600 python -m pytest tests/test_raster_hip.py -x -q -m gpu -k "backward or gradient or grad" 2>&1 | tail -2; done > gpurun_out/g8_var_tests.log 2>&1
AB_VARIANTS="base:|opf:FREESPLAT_LIB=freesplat_amd/libfreesplat_hip_opf.so|bw7:FREESPLAT_LIB=freesplat_amd/libfreesplat_hip_bw7.so|opf7:FREESPLAT_LIB=freesplat_amd/libfreesplat_hip_opf7.so" AB_REPEAT=3 timeout 900 python profiles/tools/raster_env_ab.py train > gpurun_out/g8_ab_train.log 2>&1
cat gpurun_out/g8_tests.log gpurun_out/g8_var_tests.log gpurun_out/g8_ab_train.log
