cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
AB_VARIANTS="b16:|b8:FREESPLAT_RASTER_BATCH=8|b4:FREESPLAT_RASTER_BATCH=4|b2:FREESPLAT_RASTER_BATCH=2|legacy:FREESPLAT_PREPROCESS=legacy|s3b4:FREESPLAT_RASTER_BATCH=4;FREESPLAT_RASTER_STREAMS=3" AB_REPEAT=2 timeout 900 python profiles/tools/raster_env_ab.py > gpurun_out/g12_ab.log 2>&1
cat gpurun_out/g12_ab.log
