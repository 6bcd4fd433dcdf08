cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_cost_volume_hip.py -q -m gpu -k "reference_gradients or backward_forms or random_shapes or plane_chunks or (backward_tight and (k8 or k3_c16 or k2_behind or oblique))" 2>&1 | grep -E "passed|failed|^E  |^FAILED" | cut -c1-500 | head -10
cd /tmp && export TMPDIR=/tmp
for sv in 1 2 4; do
  for wl in fvt10_K8 c3scale_K2; do
    FS_CV_SG_S=$sv CV_ONE=$wl timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_sgs_${sv}_$wl -o cv -- python $GRAFT_REPO_ROOT/profiles/tools/cv_bwd_form_ab.py > /dev/null 2>&1
    f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_sgs_${sv}_$wl -name "*kernel_stats.csv" | head -1)
    echo -n "S=$sv $wl: "; python -c "
import csv
for r in csv.DictReader(open('$f')):
    if 'cv_src_grad' in r['Name']: print('src_grad', round(float(r['AverageNs'])/1e3,1), 'us')"
  done
done 2>&1 | tee $GRAFT_REPO_ROOT/gpurun_out/r4_sg_nsrc_ab.txt
