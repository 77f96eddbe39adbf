# Native probe for the last seconds of a round's GPU time: the C++ front end on the files of scripts/make_probe.py, no
# Python.  Each case: FASTA md5 against the oracle's (probe_in/<case>.expected.md5).
O=${PROBE_OUT:-gpurun_out/probe}; mkdir -p $O
export LD_LIBRARY_PATH=$PWD/daccord_amd:$LD_LIBRARY_PATH
while read name args; do
  s=$(date +%s.%N)
  timeout 6 ./daccord_amd/daccord_hip $args probe_in/$name.las probe_in/$name.db > $O/$name.fasta 2> $O/$name.err; rc=$?
  e=$(date +%s.%N)
  got=$(md5sum < $O/$name.fasta | cut -d' ' -f1); want=$(cat probe_in/$name.expected.md5)
  echo "$name rc=$rc got=$got want=$want $( [ "$got" = "$want" ] && echo SAME || echo DIFFERENT ) $(echo "$e - $s" | bc -l 2>/dev/null)" >> $O/result.txt
  rm -f $O/$name.fasta.tmp
done < probe_in/cases.txt
cat $O/result.txt
