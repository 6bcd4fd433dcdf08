cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_depth_tail.py -m gpu -q 2>&1 | tail -2
for lib in "" _u1 _c1; do
echo "== lib$lib"
for shape in "2 192 256" "3 484 648"; do
rm -rf /tmp/prof_x
FREESPLAT_LIB=$GRAFT_REPO_ROOT/freesplat_amd/libfreesplat_hip$lib.so rocprofv3 --kernel-trace -d /tmp/prof_x -o x --output-format csv -- python profiles/tools/depth_tail_prof.py $shape > /tmp/dt.log 2>&1
python profiles/tools/kstats.py /tmp/prof_x | grep "fs::depth_up"
done; done
