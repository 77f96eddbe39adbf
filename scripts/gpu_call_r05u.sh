R=$GRAFT_REPO_ROOT; TAG=r05u; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
( DACC_LIB=$R/daccord_amd/libdaccord_hip_prof.so timeout 300 python scripts/prof_sites.py 64 14 54 ) > $O/sites_54x_64piles.log 2>&1; grep -E "site (28|31|32|33) |phase instances|total" $O/sites_54x_64piles.log | cut -c1-200
