#!/bin/bash
# usage: cold_loop.sh N [ENV=VAL ...] : N cold runs, prints the count of losses that differ from the reproducible value
N=$1; shift
bad=0
for i in $(seq 1 $N); do
  v=$(env "$@" python tools/probe/coldrun3.py 2>&1 | grep LOSS | cut -d" " -f2)
  if [ "$v" != "5.892194747924805" ]; then bad=$((bad+1)); echo "  run $i: $v"; fi
done
echo "$* : $bad / $N differ"
