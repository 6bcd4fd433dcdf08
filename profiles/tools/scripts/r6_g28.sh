cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ptf_hip.py tests/test_ptf_gru_hip.py -m gpu -x -q 2>&1 | tail -3
for lib in "" _la0 _la6; do
for shape in "2 384 512" "3 968 1296"; do
echo "== lib$lib $shape"
rm -rf /tmp/prof_x
FREESPLAT_LIB=$GRAFT_REPO_ROOT/freesplat_amd/libfreesplat_hip$lib.so rocprofv3 --kernel-trace -d /tmp/prof_x -o x --output-format csv -- python profiles/tools/ptf_train_prof.py $shape 2>&1 | grep "ms/step"
python profiles/tools/kstats.py /tmp/prof_x | grep "gru_bwd\|gru_kernel"
done; done
