cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
echo "== noinline (default)"; timeout 300 python profiles/tools/scripts/dbg_long_sort.py 2>&1 | grep -v amdgpu.ids | tail -20
echo "== inline"; FREESPLAT_LIB=$PWD/freesplat_amd/libfreesplat_hip_pinl.so timeout 300 python profiles/tools/scripts/dbg_long_sort.py 2>&1 | grep -v amdgpu.ids | tail -20
