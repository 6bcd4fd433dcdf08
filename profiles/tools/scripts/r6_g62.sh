cd $GRAFT_REPO_ROOT
for env in "FS_GRU_FWD16=0" "FS_GRU_BWD16=0" "FS_PTF_WS_SPLIT=0 FS_PTF_WS_BWD_SPLIT=0" "FREESPLAT_GRU_SAVE=0"; do
echo "== $env (ptf tests)"
env $env timeout 900 python -m pytest tests/test_ptf_hip.py tests/test_composed_dropin.py -m gpu -q 2>&1 | tail -1
done
for env in "FS_CV_FWD_SHAPE=1" "FS_CV_FWD_SHAPE=4" "FS_CV_BWD16=0" "FS_CV_SG_FORM=0" "FREESPLAT_CV_SAVE=1" "FREESPLAT_CV_SAVE=0"; do
echo "== $env (cost volume tests)"
env $env timeout 900 python -m pytest tests/test_cost_volume_hip.py -m gpu -q 2>&1 | tail -1
done
