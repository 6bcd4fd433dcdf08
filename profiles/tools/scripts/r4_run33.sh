cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
FREESPLAT_LIB=$GRAFT_REPO_ROOT/freesplat_amd/libfreesplat_hip_biaslds.so timeout 600 python -m pytest tests/test_ptf_hip.py -q -m gpu -k "gru or fold" 2>&1 | grep -E "passed|failed|^E  |^FAILED" | cut -c1-400 | head -6
AB_LIBS="base=,biaslds=freesplat_amd/libfreesplat_hip_biaslds.so" python profiles/tools/ptf_ab.py 2>&1 | tee gpurun_out/r4_gru_bias_lds_ab.txt
