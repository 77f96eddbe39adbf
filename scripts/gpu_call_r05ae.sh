R=$GRAFT_REPO_ROOT; TAG=r05ae; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
bash scripts/gpu_ab.sh $TAG 3000 3 daccord_amd/libvar_step8.so default
