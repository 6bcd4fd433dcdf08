cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_ptf_hip.py -m gpu -q -x 2>&1 | tail -4
for sv in 1 0; do
echo "== FREESPLAT_GRU_SAVE=$sv"
for shape in "2 384 512" "3 968 1296"; do
rm -rf /tmp/prof_x
FREESPLAT_GRU_SAVE=$sv rocprofv3 --kernel-trace -d /tmp/prof_x -o x --output-format csv -- python profiles/tools/ptf_train_prof.py $shape > /tmp/pt.log 2>&1
grep "ms/step" /tmp/pt.log | tail -1
python profiles/tools/kstats.py /tmp/prof_x | grep "fs::ptf_gru"
done; done
