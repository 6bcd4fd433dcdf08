cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for lib in _a _c _e _a _c _e; do
echo "== lib$lib"
for w in cv_c3scale_K2 cv_fvt10_K8 cv_fvt10_K8_cl; do
rm -rf /tmp/prof_x
FREESPLAT_LIB=$GRAFT_REPO_ROOT/freesplat_amd/libfreesplat_hip$lib.so rocprofv3 --kernel-trace -d /tmp/prof_x -o x --output-format csv -- python profiles/tools/fwd_traffic.py run $w 6 > /dev/null 2>&1
python profiles/tools/kstats.py /tmp/prof_x | grep "fs::cost_volume16_kernel" | sed "s/^/$w /"
done; done
