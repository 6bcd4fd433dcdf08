cd /tmp && export TMPDIR=/tmp
for v in nodw ""; do
  if [ -z "$v" ]; then lib=$GRAFT_REPO_ROOT/freesplat_amd/libfreesplat_hip.so; else lib=$GRAFT_REPO_ROOT/freesplat_amd/libfreesplat_hip_$v.so; fi
  for wl in c3scale_K2 native_K1; do
    FREESPLAT_LIB=$lib CV_ONE=$wl timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_p1_${v:-base}_$wl -o cv -- python $GRAFT_REPO_ROOT/profiles/tools/cv_bwd_form_ab.py > /dev/null 2>&1
    f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_p1_${v:-base}_$wl -name "*kernel_stats.csv" | head -1)
    echo -n "${v:-base} $wl: "; python -c "
import csv
for r in csv.DictReader(open('$f')):
    if 'cost_volume_bwd' in r['Name']: print('pass1', round(float(r['AverageNs'])/1e3,1), 'us')"
  done
done
