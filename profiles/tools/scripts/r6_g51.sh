cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_ptf_hip.py -m gpu -q -x 2>&1 | tail -2
for sv in 1 0; do
echo "== FREESPLAT_PTF_DW_STREAM=$sv"
for shape in "2 384 512" "10 384 512" "3 968 1296"; do
FREESPLAT_PTF_DW_STREAM=$sv python profiles/tools/ptf_train_prof.py $shape 2>&1 | grep "ms/step" | tail -1
done; done
