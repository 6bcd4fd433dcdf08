#!/bin/bash
# One GPU call that decides whether a kernel variant is kept: parity tests on the variant build, then an A/B against the
# in-tree library inside the same session (box-to-box variation is +-5 %, larger than most kernel deltas).
#   make -C freesplat_amd/csrc VARIANT=x EXTRA=-DFS_SOMETHING          (here; the .so travels with the snapshot)
#   gpurun --timeout 120 -- 'profiles/tools/variant_check.sh x "backward" cv'
# $1 variant tag (freesplat_amd/libfreesplat_hip_$1.so)   $2 pytest -k expression   $3 cv | raster | train
# Round 3's figures for budgeting (charged seconds incl. ~10 s of box overhead): cost-volume tests -k backward 3 s,
# the whole cost-volume test file 30 s, cv_ab.py 12 s, raster_ab.py ~25 s, raster_ab.py train ~40 s.
export TMPDIR=/tmp
TAG=$1; KEXPR=$2; WHICH=${3:-cv}
LIB=$PWD/freesplat_amd/libfreesplat_hip_$TAG.so
[ -f "$LIB" ] || { echo "no $LIB"; exit 2; }
case $WHICH in
  cv)     FILES="tests/test_cost_volume_hip.py"; AB="python profiles/tools/cv_ab.py";;
  raster) FILES="tests/test_raster_hip.py";      AB="python profiles/tools/raster_ab.py";;
  train)  FILES="tests/test_raster_hip.py";      AB="python profiles/tools/raster_ab.py train";;
  *) echo "cv | raster | train"; exit 2;;
esac
mkdir -p gpurun_out
FREESPLAT_LIB=$LIB timeout 300 python -m pytest $FILES -q -x -k "$KEXPR" 2>&1 | tail -5 | tee gpurun_out/variant_${TAG}_tests.log
grep -q "passed" gpurun_out/variant_${TAG}_tests.log && ! grep -q "failed\|error" gpurun_out/variant_${TAG}_tests.log || { echo "PARITY FAILED: not timing"; exit 1; }
AB_LIBS="base=,$TAG=freesplat_amd/libfreesplat_hip_$TAG.so" timeout 300 $AB 2>&1 | tee gpurun_out/variant_${TAG}_ab.log
