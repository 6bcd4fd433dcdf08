cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_raster_hip.py tests/test_cost_volume_hip.py tests/test_ptf_hip.py tests/test_multi_rank_one_gpu.py -q -m gpu -k "backward or gradient or reference_gradients or two_ranks" 2>&1 | grep -E "passed|failed|Error|^E  |^FAILED" | cut -c1-700 | head -20 | tee gpurun_out/r4_tests_d.log
AB_LIBS="prev=freesplat_amd/libfreesplat_hip_prevbwd.so,new=" python profiles/tools/raster_ab.py train 2>&1 | tee gpurun_out/r4_bwd_opacity_ab.txt
