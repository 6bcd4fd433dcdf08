cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for wl in fvt10_K8 c3scale_K2 native_K1; do
  ( cd /tmp; CV_ONE=$wl timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r4cv_$wl -o cv -- python $GRAFT_REPO_ROOT/profiles/tools/cv_bwd_form_ab.py > /dev/null 2>&1 )
done
bash profiles/run_rocprof_encoder.sh r4 > /dev/null 2>&1
ls gpurun_out | head
