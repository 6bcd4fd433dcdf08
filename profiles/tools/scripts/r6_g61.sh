cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for rep in 1 2 3; do
for lib in "" _t4; do
rm -rf /tmp/prof_x
FREESPLAT_LIB=$GRAFT_REPO_ROOT/freesplat_amd/libfreesplat_hip$lib.so rocprofv3 --kernel-trace -d /tmp/prof_x -o x --output-format csv -- python profiles/tools/cv_train_prof.py c3 6 > /tmp/cvt.log 2>&1
echo "lib$lib $(grep 'train step' /tmp/cvt.log) $(python profiles/tools/kstats.py /tmp/prof_x | grep 'cost_volume16_bwd' | cut -d, -f1-4)"
done; done
