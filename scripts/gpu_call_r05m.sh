R=$GRAFT_REPO_ROOT; TAG=r05m; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
( DACC_LIB=$R/daccord_amd/libdaccord_hip_prof.so timeout 300 python scripts/prof_sites.py 256 14 ) > $O/sites_cfg2_256piles.log 2>&1; head -n 52 $O/sites_cfg2_256piles.log | cut -c1-175
