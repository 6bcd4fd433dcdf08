cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in 1 0; do echo "BWD16=$v"; FS_CV_BWD16=$v timeout 600 python profiles/tools/cv_train_prof.py c3 6 2>&1 | tail -1; done
FS_CV_BWD16=1 bash profiles/tools/pmc_passes.sh gpurun_out/g16_pmc_new bwd_kernel -- python profiles/tools/cv_train_prof.py c3 2 > gpurun_out/g16_pmc_new.txt 2>&1
FS_CV_BWD16=0 bash profiles/tools/pmc_passes.sh gpurun_out/g16_pmc_old bwd_kernel -- python profiles/tools/cv_train_prof.py c3 2 > gpurun_out/g16_pmc_old.txt 2>&1
cat gpurun_out/g16_pmc_new.txt gpurun_out/g16_pmc_old.txt
rm -rf gpurun_out/g16_pmc_new gpurun_out/g16_pmc_old
