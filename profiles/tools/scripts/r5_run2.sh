cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_raster_hip.py -x -q -m gpu 2>&1 | tail -8 ) > gpurun_out/r5_raster_tests.log; tail -3 gpurun_out/r5_raster_tests.log
profiles/tools/_bin/valu_issue_rate > gpurun_out/r5_valu_issue_rate.txt 2>&1; grep -E "16 chains|e64|v_fma_f32 " gpurun_out/r5_valu_issue_rate.txt
AB_LIBS="w4=freesplat_amd/libfreesplat_hip_w4.so,w5=freesplat_amd/libfreesplat_hip_w5.so,new=" timeout 600 python profiles/tools/raster_ab.py > gpurun_out/r5_blend_waves_ab.txt 2>&1; cat gpurun_out/r5_blend_waves_ab.txt
AB_LIBS="b8=freesplat_amd/libfreesplat_hip_b8.so,new=" timeout 600 python profiles/tools/raster_ab.py train > gpurun_out/r5_bwd_waves_ab.txt 2>&1; cat gpurun_out/r5_bwd_waves_ab.txt
AB_ARGS="--workload c3_closeup_968x1296_1M --views 4 --steps 5 --warmup 1" AB_LIBS="networks=freesplat_amd/libfreesplat_hip_lsold.so,partitioned=" timeout 600 python profiles/tools/raster_ab.py > gpurun_out/r5_long_sort_ab.txt 2>&1; cat gpurun_out/r5_long_sort_ab.txt
timeout 600 rocprofv3 --kernel-trace -d gpurun_out/c3_step_trace -o x --output-format csv -- python bench_c3_step.py --trace-steps 2 --warmup 2 > gpurun_out/c3_step_trace.log 2>&1
python profiles/tools/c3_step_glue.py gpurun_out/c3_step_trace 2 > gpurun_out/r5_c3_step_glue.json 2> gpurun_out/c3_step_glue.err; head -c 1500 gpurun_out/r5_c3_step_glue.json; tail -3 gpurun_out/c3_step_glue.err
