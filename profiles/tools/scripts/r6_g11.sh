cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in ts ts3; do FREESPLAT_LIB=$PWD/freesplat_amd/libfreesplat_hip_$v.so timeout 900 python -m pytest tests/test_cost_volume_hip.py tests/test_configs_4_5.py -x -q -m gpu -k "not fold" 2>&1 | tail -3; done > gpurun_out/g11_tests.log 2>&1
for rep in 1 2; do for v in base ts ts3; do
  L=$PWD/freesplat_amd/libfreesplat_hip.so; [ $v != base ] && L=$PWD/freesplat_amd/libfreesplat_hip_$v.so
  for shape in fvt10 c3 native; do echo -n "$v "; FREESPLAT_LIB=$L python profiles/tools/cv_prof.py $shape 2>&1 | tail -1; done
done; done > gpurun_out/g11_ab.log 2>&1
cat gpurun_out/g11_tests.log gpurun_out/g11_ab.log
