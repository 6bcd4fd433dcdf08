cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ptf_hip.py tests/test_abi.py tests/test_composed_dropin.py -m gpu -x -q 2>&1 | tail -5
for v in 1 0; do
for shape in "2 384 512" "3 968 1296"; do
echo "== FS_GRU_BWD16=$v $shape"
rm -rf /tmp/prof_x
FS_GRU_BWD16=$v rocprofv3 --kernel-trace -d /tmp/prof_x -o x --output-format csv -- python profiles/tools/ptf_train_prof.py $shape 2>&1 | grep "ms/step"
python profiles/tools/kstats.py /tmp/prof_x | grep "gru_bwd"
done; done
