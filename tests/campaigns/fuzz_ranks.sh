#!/bin/bash
# N-rank campaign on ONE GPU through examples/trc_ranks (no Python in the ranks): random rank counts, ragged and tiny frames (fewer pixels
# than ranks: empty slices), both splits, with and without the grouped SPPM pass; the program checks its composed frame against the 1-rank
# frame (tiles) or the rank-ordered mean of the shards (samples) itself and exits non-zero otherwise.   usage: bash tests/campaigns/fuzz_ranks.sh <a> <b>
A=${1:-0}; B=${2:-40}; bad=0
[ -x examples/trc_ranks ] || make example > /dev/null
for ((s=A; s<B; s++)); do
  RANDOM=$((s * 7919 + 13))
  n=$((2 + RANDOM % 8)); 
  case $((RANDOM % 5)) in 0) W=$((1 + RANDOM % 7)); H=$((1 + RANDOM % 5));; 1) W=$((9 + RANDOM % 40)); H=$((5 + RANDOM % 30));; *) W=$((64 + RANDOM % 300)); H=$((40 + RANDOM % 200));; esac
  if [ $((RANDOM % 2)) = 0 ]; then mode="--samples"; spp=$((n * (1 + RANDOM % 4))); extra=""; else mode="--host-collectives"; spp=$((1 + RANDOM % 20)); extra=$([ $((RANDOM % 3)) = 0 ] && echo "--sppm $((1 + RANDOM % 3))"); fi
  if ! timeout 300 examples/trc_ranks --ranks $n $mode --size $W $H --spp $spp $extra --out /dev/null > /tmp/fr_$$.log 2>&1; then
    bad=$((bad + 1)); echo "FAILED seed $s: --ranks $n $mode --size $W $H --spp $spp $extra"; tail -3 /tmp/fr_$$.log
  fi
done
echo "seeds $A..$((B - 1)): $((B - A - bad)) passed, $bad FAILED"
