cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for lib in "" _n1 _n2; do
for shape in "2 192 256" "3 484 648"; do
rm -rf /tmp/prof_x
FREESPLAT_LIB=$GRAFT_REPO_ROOT/freesplat_amd/libfreesplat_hip$lib.so rocprofv3 --kernel-trace -d /tmp/prof_x -o x --output-format csv -- python profiles/tools/depth_tail_prof.py $shape > /tmp/dt.log 2>&1
echo "lib$lib $shape $(python profiles/tools/kstats.py /tmp/prof_x | grep 'depth_upsample' | cut -d, -f1-3)"
done; done
