# phase profiles (shader cycles per window) of heap-code variants, same box, back to back
R=$GRAFT_REPO_ROOT; O=gpurun_out/r2v; mkdir -p $R/$O; cd $R
for V in i0_p0 i1_p1 i0_p1 i2_p0 i2_p1 i0_p0; do
  ( DACC_LIB=$R/daccord_amd/libvar_${V}_prof.so timeout 100 python scripts/prof_phases.py 64 ) > $O/phases_$V.$RANDOM.log 2>&1
done
for f in $O/phases_*.log; do echo $f; grep -v amdgpu $f | grep "total cyc\|F trees\|R blocks\|pair-gen\|pair-replay\|cand-errors" | tail -6; done
