cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_cost_volume_hip.py tests/test_configs_4_5.py tests/test_composed_dropin.py -m gpu -q 2>&1 | tail -2
for which in c3 fvt10 native; do
rm -rf /tmp/prof_x
rocprofv3 --kernel-trace -d /tmp/prof_x -o x --output-format csv -- python profiles/tools/cv_train_prof.py $which 6 > /tmp/cvt.log 2>&1
grep "train step" /tmp/cvt.log
python profiles/tools/kstats.py /tmp/prof_x | grep "fs::c" | head -4
done
for w in cv_c3scale_K2 cv_fvt10_K8 cv_fvt10_K8_cl; do
rm -rf /tmp/prof_x
rocprofv3 --kernel-trace -d /tmp/prof_x -o x --output-format csv -- python profiles/tools/fwd_traffic.py run $w 6 > /dev/null 2>&1
python profiles/tools/kstats.py /tmp/prof_x | grep "fs::cost_volume16_kernel" | sed "s/^/$w /"
done
