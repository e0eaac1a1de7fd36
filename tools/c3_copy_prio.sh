#!/bin/bash
# A/B on the GPU box: the streams of icelk_upload_gray_async -- high priority, created at the first upload (default) -- against
# the normal-priority pair of before (ICELK_COPY_PRIORITY=normal); C3 four times each, twice
for p in high normal high normal; do
  if [ $p = high ]; then unset ICELK_COPY_PRIORITY; else export ICELK_COPY_PRIORITY=$p; fi
  echo "== upload streams: $p"; bash tools/c3_repeat.sh 4 | cut -c1-60
done
