cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python profiles/tools/tile_list_hist.py > gpurun_out/g5_hist.log 2>&1
FREESPLAT_LIB=$PWD/freesplat_amd/libfreesplat_hip_w7.so timeout 600 python -m pytest tests/test_raster_hip.py -x -q -m gpu 2>&1 | tail -2 > gpurun_out/g5_w7_tests.log
AB_VARIANTS="base:|w7:FREESPLAT_LIB=freesplat_amd/libfreesplat_hip_w7.so" AB_REPEAT=3 timeout 900 python profiles/tools/raster_env_ab.py > gpurun_out/g5_ab.log 2>&1
for shape in c3 fvt10; do
  for lib in base sgfloor; do
    L=$PWD/freesplat_amd/libfreesplat_hip.so; [ $lib = sgfloor ] && L=$PWD/freesplat_amd/libfreesplat_hip_sgfloor.so
    FREESPLAT_LIB=$L timeout 600 rocprofv3 --kernel-trace -d gpurun_out/cvt_${shape}_$lib -o x --output-format csv -- python profiles/tools/cv_train_prof.py $shape 4 > gpurun_out/g5_cvt_${shape}_$lib.log 2>&1
    python profiles/tools/kstats.py gpurun_out/cvt_${shape}_$lib "$shape $lib" > gpurun_out/g5_cvt_${shape}_${lib}_stats.csv
    rm -rf gpurun_out/cvt_${shape}_$lib
  done
done
tail -3 gpurun_out/g5_hist.log gpurun_out/g5_w7_tests.log; cat gpurun_out/g5_ab.log; cat gpurun_out/g5_cvt_*_stats.csv | cut -c1-150; grep "train step" gpurun_out/g5_cvt_*.log
