cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in "" sg_NO_LOAD sg_NO_CLAIM sg_NO_RMW sg_NL_NR; do
  for wl in fvt10_K8 c3scale_K2; do
    if [ -z "$v" ]; then lib=$PWD/freesplat_amd/libfreesplat_hip.so; else lib=$PWD/freesplat_amd/libfreesplat_hip_$v.so; fi
    echo -n "variant=${v:-shipped} $wl: "
    FREESPLAT_LIB=$lib CV_ONE=$wl timeout 120 python profiles/tools/cv_bwd_form_ab.py 2>&1 | grep -v amdgpu.ids | tail -1
  done
done 2>&1 | tee gpurun_out/r4_sg_variants.txt
timeout 900 python -m pytest tests/test_cost_volume_hip.py -q -m gpu -k "backward_tight and (k2_60x80 or native)" 2>&1 | grep -E "passed|failed|AssertionError|^FAILED" | cut -c1-600 | tee gpurun_out/r4_cv_tests2.log
