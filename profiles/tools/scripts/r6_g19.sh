cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for lib in onewave_notaps onewave_nodw; do
for which in c3 fvt10; do
echo "== $lib (FS_CV_BWD16=1) $which"
rm -rf /tmp/prof_x
FS_CV_BWD16=1 FREESPLAT_LIB=$GRAFT_REPO_ROOT/freesplat_amd/libfreesplat_hip_$lib.so rocprofv3 --kernel-trace -d /tmp/prof_x -o x --output-format csv -- python profiles/tools/cv_train_prof.py $which 4 2>&1 | grep "train step"
python profiles/tools/kstats.py /tmp/prof_x | grep bwd_kernel
done
done
