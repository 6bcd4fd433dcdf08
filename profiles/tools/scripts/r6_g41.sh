cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for lib in "" _g3; do
for shape in "2 384 512" "3 968 1296"; do
echo "== lib$lib $shape"
rm -rf /tmp/prof_x
FREESPLAT_LIB=$GRAFT_REPO_ROOT/freesplat_amd/libfreesplat_hip$lib.so rocprofv3 --kernel-trace -d /tmp/prof_x -o x --output-format csv -- python profiles/tools/ptf_train_prof.py $shape 2>&1 | grep "ms/step"
python profiles/tools/kstats.py /tmp/prof_x | grep "gru_bwd"
done; done
