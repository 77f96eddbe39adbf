#!/bin/bash
# ORACLE SUPPORT (test infrastructure): compiles the reference's own hot-path headers, where they lie under /root/reference/src
# and unmodified, against the libmaus2 stand-in of this directory (libmaus2/shim.hpp) into oracle/_ref/ (git-ignored; the built
# libraries travel to the GPU box with the snapshot, the reference sources do not and are never copied into the repository).
#   oracle/_ref/libdaccord_ref.so      the reference's DebruijnGraphContainer: k in [3,12] as shipped
#   oracle/_ref/libdaccord_ref_k16.so  the same sources with OUR factory (k16/DebruijnGraphContainer.hpp) so that DebruijnGraph<13..16> exist
# -ffp-contract=off: the path compares FP64 weights exactly.  No-op (exit 0) where /root/reference is absent.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${DACC_REFERENCE:-/root/reference}/src"
OUT="$HERE/../_ref"
if [ ! -f "$REF/HandleContext.hpp" ]; then echo "ref_shim/build.sh: $REF not present, nothing built"; exit 0; fi
mkdir -p "$OUT"
# the estimator functions (handleIndelEstimate<k>, handleIndelEstimateDeep<k>) are lines 271-995 of the driver's translation unit:
# cut out for the duration of the compile only
EXC="$OUT/.estimate_excerpt.$$.hpp"
sed -n '271,995p' "$REF/daccord.cpp" > "$EXC"
grep -q "^double handleIndelEstimate(" "$EXC" || { echo "ref_shim/build.sh: estimator not at the expected lines of daccord.cpp"; rm -f "$EXC"; exit 1; }
# ... and three more pieces of the same file: the two comparators of :997-1011, the heap order of the estimator's pile selection
# (:1406-1412, a local struct of daccord()) and the selection loop itself (:1712-1742, the body that feeds on pdec->getNextOverlap)
EXC2="$OUT/.cmp_excerpt.$$.hpp"; EXC3="$OUT/.pfg_excerpt.$$.hpp"; EXC4="$OUT/.sel_excerpt.$$.hpp"
sed -n '997,1011p' "$REF/daccord.cpp" > "$EXC2"; sed -n '1406,1412p' "$REF/daccord.cpp" > "$EXC3"; sed -n '1712,1742p' "$REF/daccord.cpp" > "$EXC4"
grep -q "^struct OverlapPosComparator" "$EXC2" && grep -q "struct PairFirstGreaterComp" "$EXC3" && head -1 "$EXC4" | grep -q "while ( pdec->getNextOverlap(OVL) )" || { echo "ref_shim/build.sh: selection code not at the expected lines of daccord.cpp"; rm -f "$EXC" "$EXC2" "$EXC3" "$EXC4"; exit 1; }
# ... and the MAIN path's pile selection (ref_select.cpp): :2026-2105 = the heap entry, its block order and the per-thread buffers (with the
# 64 KiB input block size), :2120-2288 = the body of the loop over A reads from lmaxinput to the sort by abpos
EXC5="$OUT/.mainsel_a_excerpt.$$.hpp"; EXC6="$OUT/.mainsel_b_excerpt.$$.hpp"
sed -n '2026,2105p' "$REF/daccord.cpp" > "$EXC5"; sed -n '2120,2288p' "$REF/daccord.cpp" > "$EXC6"
head -1 "$EXC5" | grep -q "struct OverlapEntry" && grep -q "inputbuffersize = 64\*1024" "$EXC5" && head -1 "$EXC6" | grep -q "uint64_t const rl = RL\[z-minaread\]" && tail -3 "$EXC6" | grep -q "OverlapDataInterfacePosComparator()" || { echo "ref_shim/build.sh: main pile selection not at the expected lines of daccord.cpp"; rm -f "$EXC" "$EXC2" "$EXC3" "$EXC4" "$EXC5" "$EXC6"; exit 1; }
# ... and the read interval of a run: :1119-1224 = the -J / -I branches (parsing included), :1227 = toparead
EXC7="$OUT/.ivl_a_excerpt.$$.hpp"; EXC8="$OUT/.ivl_b_excerpt.$$.hpp"
sed -n '1119,1224p' "$REF/daccord.cpp" > "$EXC7"; sed -n '1227,1227p' "$REF/daccord.cpp" > "$EXC8"
head -1 "$EXC7" | grep -q 'if ( arg.uniqueArgPresent("J") )' && grep -q "int64_t const toparead = maxaread >= 0 ? maxaread + 1 : maxaread;" "$EXC8" || { echo "ref_shim/build.sh: read interval code not at the expected lines of daccord.cpp"; rm -f "$EXC" "$EXC2" "$EXC3" "$EXC4" "$EXC5" "$EXC6" "$EXC7" "$EXC8"; exit 1; }
# ... the error rates derived from the estimator's counts (:1867-1878: len, numerr, est_cor, p_i, p_d)
EXC10="$OUT/.rates_excerpt.$$.hpp"
sed -n '1867,1878p' "$REF/daccord.cpp" > "$EXC10"
head -1 "$EXC10" | grep -q "uint64_t const len = GAS.matches + GAS.mismatches + GAS.deletions;" && grep -q "double const p_d = " "$EXC10" || { echo "ref_shim/build.sh: rate formulas not at the expected lines of daccord.cpp"; rm -f "$EXC10"; exit 1; }
# ... and the defaults of the options (:106-169, the getDefault* functions behind the help text and the argument parser)
EXC9="$OUT/.defaults_excerpt.$$.hpp"
sed -n '106,169p' "$REF/daccord.cpp" > "$EXC9"
head -1 "$EXC9" | grep -q "static uint64_t getDefaultVerbose()" && grep -q "static uint64_t getDefaultMaxFilterFreq()" "$EXC9" || { echo "ref_shim/build.sh: option defaults not at the expected lines of daccord.cpp"; rm -f "$EXC" "$EXC2" "$EXC3" "$EXC4" "$EXC5" "$EXC6" "$EXC7" "$EXC8" "$EXC9"; exit 1; }
trap 'rm -f "$EXC" "$EXC2" "$EXC3" "$EXC4" "$EXC5" "$EXC6" "$EXC7" "$EXC8" "$EXC9" "$EXC10"' EXIT
FLAGS="-O2 -std=c++17 -fPIC -fopenmp -ffp-contract=off -shared -DNDEBUG -w -DDACC_REF_ESTIMATE_EXCERPT=\"$EXC\" -DDACC_REF_CMP_EXCERPT=\"$EXC2\" -DDACC_REF_PFG_EXCERPT=\"$EXC3\" -DDACC_REF_SEL_EXCERPT=\"$EXC4\" -DDACC_REF_MAINSEL_A_EXCERPT=\"$EXC5\" -DDACC_REF_MAINSEL_B_EXCERPT=\"$EXC6\" -DDACC_REF_IVL_A_EXCERPT=\"$EXC7\" -DDACC_REF_IVL_B_EXCERPT=\"$EXC8\" -DDACC_REF_DEFAULTS_EXCERPT=\"$EXC9\" -DDACC_REF_RATES_EXCERPT=\"$EXC10\""
g++ $FLAGS -I"$HERE" -I"$REF" -o "$OUT/libdaccord_ref.so" "$HERE/ref_capi.cpp" "$HERE/ref_select.cpp" &
g++ $FLAGS -DDACC_REF_K16 -I"$HERE/k16" -I"$HERE" -I"$REF" -o "$OUT/libdaccord_ref_k16.so" "$HERE/ref_capi.cpp" "$HERE/ref_select.cpp" &
wait
ls -la "$OUT"
