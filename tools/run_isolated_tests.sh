#!/bin/bash
# Run the given pytest node ids one per process (GPU box), stopping at the first crash (rc other than 0/1).
# usage: tools/run_isolated_tests.sh OUTDIR nodeid...
out=$1; shift
mkdir -p "$out"
i=0
for t in "$@"; do
  i=$((i+1))
  log="$out/iso_$i.log"
  echo "== $t" > "$log"
  LIBC_FATAL_STDERR_=1 timeout -k 10 400 python -m pytest "$t" -x -q -m gpu -rA >> "$log" 2>&1
  rc=$?
  echo "$t rc=$rc"
  tail -5 "$log" | cut -c1-200
  if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "stopping: crash in $t"; exit $rc; fi
done
exit 0
