cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
FREESPLAT_LIB=$PWD/freesplat_amd/libfreesplat_hip_ps2.so timeout 600 python -m pytest tests/test_cost_volume_hip.py -q -m gpu -k "reference_gradients or backward_forms or (backward_tight and (k8 or k3_c16 or k2_behind) and two_pass)" 2>&1 | grep -E "passed|failed|Error|^E  |^FAILED" | cut -c1-500 | head -10
cd /tmp && export TMPDIR=/tmp
for v in ps2 ps4 ps8 ""; do
  if [ -z "$v" ]; then lib=$GRAFT_REPO_ROOT/freesplat_amd/libfreesplat_hip.so; else lib=$GRAFT_REPO_ROOT/freesplat_amd/libfreesplat_hip_$v.so; fi
  for wl in fvt10_K8 c3scale_K2; do
    FREESPLAT_LIB=$lib CV_ONE=$wl timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_ps_${v:-base}_$wl -o cv -- python $GRAFT_REPO_ROOT/profiles/tools/cv_bwd_form_ab.py > /dev/null 2>&1
    f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_ps_${v:-base}_$wl -name "*kernel_stats.csv" | head -1)
    echo -n "${v:-base} $wl: "; python -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'cv_src_grad' in r['Name']: print('src_grad', round(float(r['AverageNs'])/1e3,1), 'us')"
  done
done 2>&1 | tee $GRAFT_REPO_ROOT/gpurun_out/r4_sg_plane_stride_ab.txt
