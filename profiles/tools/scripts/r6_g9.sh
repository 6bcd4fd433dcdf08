cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
FREESPLAT_LIB=$PWD/freesplat_amd/libfreesplat_hip_gru2.so timeout 900 python -m pytest tests/test_ptf_hip.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/g9_tests.log
for shape in "3 968 1296" "2 384 512"; do
  tag=$(echo $shape | tr ' ' '_')
  for lib in base gru2; do
    L=$PWD/freesplat_amd/libfreesplat_hip.so; [ $lib = gru2 ] && L=$PWD/freesplat_amd/libfreesplat_hip_gru2.so
    FREESPLAT_LIB=$L timeout 600 rocprofv3 --kernel-trace -d gpurun_out/ptf_${tag}_$lib -o x --output-format csv -- python profiles/tools/ptf_train_prof.py $shape > gpurun_out/g9_ptf_${tag}_$lib.log 2>&1
    python profiles/tools/kstats.py gpurun_out/ptf_${tag}_$lib "fold $shape $lib" | head -12 > gpurun_out/g9_ptf_${tag}_${lib}_stats.csv
    rm -rf gpurun_out/ptf_${tag}_$lib
  done
done
cat gpurun_out/g9_tests.log; cat gpurun_out/g9_ptf_*_stats.csv | cut -c1-130; grep "ms/step" gpurun_out/g9_ptf_*.log
timeout 900 python -m pytest tests/test_configs_4_5.py -x -q -m gpu -k oracle_inverts 2>&1 | grep -E "assert|Error|passed|failed" | head -20
AB_VARIANTS="base:|opf:FREESPLAT_LIB=freesplat_amd/libfreesplat_hip_opf.so" AB_REPEAT=3 timeout 900 python profiles/tools/raster_env_ab.py train > gpurun_out/g9_ab_train.log 2>&1
cat gpurun_out/g9_ab_train.log
