cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
AB_LIBS="base=,th4=freesplat_amd/libfreesplat_hip_sgth4.so" timeout 900 python profiles/tools/cv_lib_ab.py c3scale_K2 fvt10_K8 native_K1 2>&1 | tee gpurun_out/r5_cv_sg_th4_ab.txt
( FREESPLAT_LIB=$PWD/freesplat_amd/libfreesplat_hip_sgth4.so timeout 600 python -m pytest tests/test_cost_volume_hip.py -x -q -m gpu -k "backward" 2>&1 | tail -4 )
timeout 600 python bench_c3_step.py > gpurun_out/r5_c3_step.json 2> gpurun_out/r5_c3_step.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r5_c3_step.json"))
print({k:d[k] for k in ("ms_per_step","ms_each_step","wall_ms_per_step","library_kernel_ms","non_library_ms","glue_ms","glue_frac_of_gpu_time")}); print(d["library_kernel_ms_by_stage"])
PY
