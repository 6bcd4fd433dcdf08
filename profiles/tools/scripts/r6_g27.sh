cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6_profiles
timeout 900 python -m pytest tests/test_cost_volume_hip.py -m gpu -x -q 2>&1 | tail -3
bash profiles/tools/fwd_traffic.sh r6 cv_native_K1 cv_c3scale_K2 cv_fvt10_K8 cv_fvt10_K8_cl > gpurun_out/r6_profiles/fwd_traffic_cv.log 2>&1
tail -6 gpurun_out/r6_profiles/fwd_traffic_cv.log
cp profiles/r6_traffic.json gpurun_out/r6_profiles/r6_traffic.json
rm -rf gpurun_out/traffic_r6
