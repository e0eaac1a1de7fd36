#!/bin/bash
# A/B on the GPU box: does the priority class of the (unused) upload streams change the resident C2 rate?
for p in normal high normal high normal high; do
  if [ $p = normal ]; then unset ICELK_COPY_PRIORITY; else export ICELK_COPY_PRIORITY=$p; fi
  echo "== copy streams: $p"; bash tools/bench_repeat.sh 2 200 20 | cut -c1-100
done
