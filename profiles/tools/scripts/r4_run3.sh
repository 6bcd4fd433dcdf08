cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_cost_volume_hip.py -q -m gpu -k "backward or golden or native_size" 2>&1 | grep -E "passed|failed|AssertionError|^FAILED" | cut -c1-600 > gpurun_out/r4_cv_tests.log
cat gpurun_out/r4_cv_tests.log
FREESPLAT_LIB=$PWD/freesplat_amd/libfreesplat_hip_sgstats.so timeout 300 python profiles/tools/cv_sg_stats.py small native_K1 c3scale_K2 fvt10_K8 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_cv_sg_stats.txt
timeout 600 python profiles/tools/cv_bwd_form_ab.py 2>&1 | tee gpurun_out/r4_cv_bwd_form_ab.txt
cd /tmp && export TMPDIR=/tmp
for wl in fvt10_K8 c3scale_K2 native_K1; do
CV_ONE=$wl timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_cv_$wl -o cv -- python $GRAFT_REPO_ROOT/profiles/tools/cv_bwd_form_ab.py > $GRAFT_REPO_ROOT/gpurun_out/prof_cv_$wl.log 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_cv_$wl -name "*kernel_stats.csv" | head -1); echo $wl; head -8 "$f" | cut -c1-200
done
