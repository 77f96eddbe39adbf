R=$GRAFT_REPO_ROOT; TAG=r05h; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
bash scripts/gpu_ab.sh $TAG 3000 3 daccord_amd/libvar_lines.so daccord_amd/libvar_head.so default
( DACC_LIB=$R/daccord_amd/libdaccord_hip_prof.so timeout 300 python scripts/prof_sites.py 256 14 ) > $O/sites_cfg2_256piles.log 2>&1; head -n 45 $O/sites_cfg2_256piles.log | cut -c1-175
