cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for lib in "" _prio1 _prio2 _tf4; do
for which in c3 fvt10; do
echo "== lib$lib $which"
rm -rf /tmp/prof_x
FREESPLAT_LIB=$GRAFT_REPO_ROOT/freesplat_amd/libfreesplat_hip$lib.so rocprofv3 --kernel-trace -d /tmp/prof_x -o x --output-format csv -- python profiles/tools/cv_train_prof.py $which 4 2>&1 | grep "train step"
python profiles/tools/kstats.py /tmp/prof_x | grep bwd_kernel
done
done
FREESPLAT_LIB=$GRAFT_REPO_ROOT/freesplat_amd/libfreesplat_hip_trace.so python profiles/tools/cv_bwd16_trace.py 2>&1 | tail -1
