cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ptf_hip.py -q -m gpu -k "gru" 2>&1 | grep -E "passed|failed|^E  |^FAILED" | cut -c1-600 | head -10
AB_LIBS="base=,ch64=freesplat_amd/libfreesplat_hip_ch64.so,ch16=freesplat_amd/libfreesplat_hip_ch16.so" python profiles/tools/ptf_ab.py 2>&1 | tee gpurun_out/r4_gru_chunk_ab.txt
