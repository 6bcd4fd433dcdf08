cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for w in ptf_10_views ptf_3_views; do
rm -rf /tmp/prof_x
rocprofv3 --kernel-trace -d /tmp/prof_x -o x --output-format csv -- python profiles/tools/fwd_traffic.py run $w 5 > /tmp/ft.log 2>&1
echo "== $w"
python profiles/tools/kstats.py /tmp/prof_x | head -10
done
