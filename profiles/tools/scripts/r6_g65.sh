cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for ch in 1 2 1 2 4; do
rm -rf /tmp/prof_x
FS_CV_SG_CHUNKS=$ch rocprofv3 --kernel-trace -d /tmp/prof_x -o x --output-format csv -- python profiles/tools/cv_train_prof.py c3 6 > /tmp/cvt.log 2>&1
echo "chunks=$ch $(grep 'train step' /tmp/cvt.log) $(python profiles/tools/kstats.py /tmp/prof_x | grep 'cv_src_grad' | cut -d, -f1-3)"
done
