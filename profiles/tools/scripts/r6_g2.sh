cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
AB_VARIANTS="b16:|pw5:FREESPLAT_LIB=freesplat_amd/libfreesplat_hip_pw5.so|pw6:FREESPLAT_LIB=freesplat_amd/libfreesplat_hip_pw6.so|s3:FREESPLAT_RASTER_STREAMS=3|s4:FREESPLAT_RASTER_STREAMS=4" timeout 900 python profiles/tools/raster_env_ab.py > gpurun_out/g2_ab.log 2>&1
cat gpurun_out/g2_ab.log
bash profiles/tools/pmc_passes.sh gpurun_out/pmc_r6a "" -- python bench.py --steps 3 --warmup 1 --sections raster --no-graph --no-cpu-baseline --no-profile --min-time 0 > gpurun_out/g2_pmc.log 2>&1
cat gpurun_out/g2_pmc.log
