cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for rep in 1 2; do
for lib in "" _nt; do
rm -rf /tmp/prof_x
FREESPLAT_LIB=$GRAFT_REPO_ROOT/freesplat_amd/libfreesplat_hip$lib.so rocprofv3 --kernel-trace -d /tmp/prof_x -o x --output-format csv -- python profiles/tools/ptf_train_prof.py 3 968 1296 > /tmp/pt.log 2>&1
echo "lib$lib $(grep 'ms/step' /tmp/pt.log | tail -1)"
python profiles/tools/kstats.py /tmp/prof_x | grep "fs::ptf_gru16\|bwd16\|gru_dw_k" | cut -d, -f1-3
done; done
