# GPU call 2 of round 5: the fine-site ledger (profiling build), what PC sampling configurations the box offers
R=$GRAFT_REPO_ROOT; TAG=r05b; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
( timeout 60 rocprofv3-avail list --pc-sampling; timeout 60 rocprofv3-avail info --pc-sampling ) > $O/pc_sampling_avail.log 2>&1; tail -n 12 $O/pc_sampling_avail.log
( DACC_LIB=$R/daccord_amd/libdaccord_hip_prof.so timeout 300 python scripts/prof_sites.py 256 14 ) > $O/sites_cfg2_256piles.log 2>&1; head -n 60 $O/sites_cfg2_256piles.log | cut -c1-200
( DACC_LIB=$R/daccord_amd/libdaccord_hip_prof.so timeout 300 python scripts/prof_sites.py 64 14 54 ) > $O/sites_54x_64piles.log 2>&1; head -n 3 $O/sites_54x_64piles.log | cut -c1-200
