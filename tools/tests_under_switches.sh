#!/bin/bash
# Runs on the GPU box: the GPU suite with one A/B switch forced for every handle of the run; one summary block per switch
for sw in ICELK_HOST_TAIL=1 ICELK_NO_TEMPLATE_REUSE=1 ICELK_STRIP_WAVES=11 ICELK_NO_ORDER=1 ICELK_COPY_PRIORITY=normal ICELK_PYR_AHEAD_WIDE=1 ICELK_NO_STREAM_PROBE=1; do
  echo "== $sw"
  env $sw timeout -k 10 900 python -m pytest tests -m gpu -q 2>&1 | grep -E "^FAILED|passed|failed" | cut -c1-160
done
