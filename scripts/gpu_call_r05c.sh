R=$GRAFT_REPO_ROOT; TAG=r05c; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
bash scripts/gpu_ab.sh $TAG 3000 3 daccord_amd/libvar_lines.so default
( DACC_LIB=$R/daccord_amd/libdaccord_hip_prof.so timeout 300 python scripts/prof_sites.py 256 14 ) > $O/sites_cfg2_256piles.log 2>&1; grep -E "site (12|13|28|31|32|33) |phase instances|phase F trees|total" $O/sites_cfg2_256piles.log | cut -c1-200
( DACC_LIB=$R/daccord_amd/libdaccord_hip_prof.so timeout 300 python scripts/prof_sites.py 64 14 54 ) > $O/sites_54x_64piles.log 2>&1; grep -E "site (28|31|32|33) |phase instances|total" $O/sites_54x_64piles.log | cut -c1-200
