cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for lib in _a _b _c _d _e; do
echo "== lib$lib"
FREESPLAT_LIB=$GRAFT_REPO_ROOT/freesplat_amd/libfreesplat_hip$lib.so timeout 600 python -m pytest tests/test_cost_volume_hip.py -m gpu -q -x 2>&1 | tail -1
for which in c3 fvt10 native; do
rm -rf /tmp/prof_x
FREESPLAT_LIB=$GRAFT_REPO_ROOT/freesplat_amd/libfreesplat_hip$lib.so rocprofv3 --kernel-trace -d /tmp/prof_x -o x --output-format csv -- python profiles/tools/cv_train_prof.py $which 4 > /tmp/cvt.log 2>&1
python profiles/tools/kstats.py /tmp/prof_x | grep "fs::cost_volume16_kernel" | sed "s/^/$which /"
done; done
