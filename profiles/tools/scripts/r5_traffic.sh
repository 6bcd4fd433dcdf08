cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
bash profiles/tools/fwd_traffic.sh r5 2>&1 | tail -30
cp profiles/r5_traffic.json gpurun_out/r5_traffic.json
