cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_cost_volume_hip.py -q -m gpu -k "reference_gradients or backward_forms or (backward_tight and (k8 or k3_c16 or k2_behind))" 2>&1 | grep -E "passed|failed|Error|^E  |^FAILED" | cut -c1-500 | head -10
for i in 1 2; do for v in prev ""; do
  if [ -z "$v" ]; then lib=$PWD/freesplat_amd/libfreesplat_hip.so; else lib=$PWD/freesplat_amd/libfreesplat_hip_$v.so; fi
  for wl in fvt10_K8 c3scale_K2 native_K1; do
    echo -n "${v:-new} $wl: "; FREESPLAT_LIB=$lib CV_ONE=$wl timeout 120 python profiles/tools/cv_bwd_form_ab.py 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-40
  done
done; done 2>&1 | tee gpurun_out/r4_sg_claimpad_ab.txt
