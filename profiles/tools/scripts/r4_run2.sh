cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
profiles/tools/_bin/lds_atomic_rate 2>&1 | tee gpurun_out/r4_lds_atomic_rate.txt
FREESPLAT_LIB=$PWD/freesplat_amd/libfreesplat_hip_sgstats.so timeout 300 python profiles/tools/cv_sg_stats.py 2>&1 | tee gpurun_out/r4_cv_sg_stats.txt
