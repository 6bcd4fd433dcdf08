cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ptf_hip.py tests/test_abi.py tests/test_composed_dropin.py tests/test_configs_4_5.py -m gpu -x -q 2>&1 | tail -5
for v in 1 0; do
echo "== FS_GRU_FWD16=$v train 3 968 1296"
rm -rf /tmp/prof_x
FS_GRU_FWD16=$v rocprofv3 --kernel-trace -d /tmp/prof_x -o x --output-format csv -- python profiles/tools/ptf_train_prof.py 3 968 1296 2>&1 | grep "ms/step"
python profiles/tools/kstats.py /tmp/prof_x | grep "gru"
echo "== FS_GRU_FWD16=$v inference fold 3 views"
rm -rf /tmp/prof_x
FS_GRU_FWD16=$v rocprofv3 --kernel-trace -d /tmp/prof_x -o x --output-format csv -- python profiles/tools/fwd_traffic.py run ptf_3_views 5 > /dev/null 2>&1
python profiles/tools/kstats.py /tmp/prof_x | grep "gru\|write_state"
done
