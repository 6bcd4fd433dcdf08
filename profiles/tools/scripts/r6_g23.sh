cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for lib in "" _f0 _f1 _f2w3 _f4; do
for which in c3 fvt10; do
echo "== lib$lib $which"
rm -rf /tmp/prof_x
FREESPLAT_LIB=$GRAFT_REPO_ROOT/freesplat_amd/libfreesplat_hip$lib.so rocprofv3 --kernel-trace -d /tmp/prof_x -o x --output-format csv -- python profiles/tools/cv_prof.py $which 2>&1 | grep "fwd ms"
python profiles/tools/kstats.py /tmp/prof_x | grep "cost_volume16_kernel"
done
done
python -m pytest tests/test_cost_volume_hip.py -m gpu -x -q 2>&1 | tail -2
