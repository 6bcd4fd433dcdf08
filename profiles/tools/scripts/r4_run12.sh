cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_cost_volume_hip.py -q -m gpu -k "reference_gradients or backward_forms or (backward_tight and (k8 or k3_c16 or k2_behind))" 2>&1 | grep -E "passed|failed|Error|^E  |^FAILED" | cut -c1-500 | head -10
cd /tmp && export TMPDIR=/tmp
for v in tileouter ""; do
  if [ -z "$v" ]; then lib=$GRAFT_REPO_ROOT/freesplat_amd/libfreesplat_hip.so; else lib=$GRAFT_REPO_ROOT/freesplat_amd/libfreesplat_hip_$v.so; fi
  for wl in fvt10_K8 c3scale_K2; do
    FREESPLAT_LIB=$lib CV_ONE=$wl timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_ord_${v:-new}_$wl -o cv -- python $GRAFT_REPO_ROOT/profiles/tools/cv_bwd_form_ab.py > /dev/null 2>&1
    f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_ord_${v:-new}_$wl -name "*kernel_stats.csv" | head -1)
    echo -n "${v:-new} $wl: "; grep "cv_src_grad" "$f" | awk -F'","' '{print "src_grad avg us", $4/1000}'
  done
done 2>&1 | tee $GRAFT_REPO_ROOT/gpurun_out/r4_sg_order_ab.txt
