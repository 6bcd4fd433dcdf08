cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for lib in "" _f2 _f4; do
echo "== lib$lib inference fold 3 views / 10 views"
for w in ptf_3_views ptf_10_views; do
rm -rf /tmp/prof_x
FREESPLAT_LIB=$GRAFT_REPO_ROOT/freesplat_amd/libfreesplat_hip$lib.so rocprofv3 --kernel-trace -d /tmp/prof_x -o x --output-format csv -- python profiles/tools/fwd_traffic.py run $w 5 > /dev/null 2>&1
python profiles/tools/kstats.py /tmp/prof_x | grep "gru"
done; done
echo "== old 32-pair"
for w in ptf_3_views ptf_10_views; do
rm -rf /tmp/prof_x
FS_GRU_FWD16=0 rocprofv3 --kernel-trace -d /tmp/prof_x -o x --output-format csv -- python profiles/tools/fwd_traffic.py run $w 5 > /dev/null 2>&1
python profiles/tools/kstats.py /tmp/prof_x | grep "gru"
done
