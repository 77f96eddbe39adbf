# A/B of library variants on the GPU box: bash scripts/gpu_ab.sh <tag> <reads> <steps> lib1 lib2 ...   (paths relative to the repo; "default" = the product library)
# every variant runs in its own process on the same synthetic batch; prints step time, tier times and the FASTA digest (which must not move)
R=$GRAFT_REPO_ROOT; TAG=$1; READS=$2; STEPS=$3; shift 3; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for rep in 1 2; do
for lib in "$@"; do
  if [ "$lib" = "default" ]; then unset DACC_LIB; else export DACC_LIB=$R/$lib; fi
  n=$(basename $lib .so)
  ( timeout 300 python scripts/sweep_env.py $READS $STEPS "" ) > $O/ab_${n}_$rep.log 2>&1
  echo "$n rep$rep: $(grep '^{' $O/ab_${n}_$rep.log | tail -n 1 | cut -c1-260)"
done
done
