R=$GRAFT_REPO_ROOT; TAG=r05y; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
( DACC_LIB=$R/daccord_amd/libdaccord_hip_prof.so timeout 300 python scripts/prof_sites.py 256 14 ) > $O/sites_cfg2_256piles.log 2>&1; head -n 30 $O/sites_cfg2_256piles.log | cut -c1-175
( DACC_LIB=$R/daccord_amd/libdaccord_hip_prof.so timeout 300 python scripts/prof_sites.py 64 14 54 ) > $O/sites_54x_64piles.log 2>&1; head -n 22 $O/sites_54x_64piles.log | cut -c1-175
