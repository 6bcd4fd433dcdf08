cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_ptf_hip.py tests/test_composed_dropin.py -m gpu -q 2>&1 | tail -2
for sp in 1 0; do
echo "== FS_PTF_WS_BWD_SPLIT=$sp"
rm -rf /tmp/prof_x
FS_PTF_WS_BWD_SPLIT=$sp rocprofv3 --kernel-trace -d /tmp/prof_x -o x --output-format csv -- python profiles/tools/ptf_train_prof.py 3 968 1296 > /tmp/pt.log 2>&1
grep "ms/step" /tmp/pt.log | tail -1
python profiles/tools/kstats.py /tmp/prof_x | grep "write_state_bwd"
rm -rf /tmp/prof_x
FS_PTF_WS_BWD_SPLIT=$sp rocprofv3 --kernel-trace -d /tmp/prof_x -o x --output-format csv -- python profiles/tools/ptf_train_prof.py 10 384 512 > /tmp/pt.log 2>&1
grep "ms/step" /tmp/pt.log | tail -1
python profiles/tools/kstats.py /tmp/prof_x | grep "write_state_bwd"
done
