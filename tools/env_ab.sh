#!/bin/bash
# A/B of an environment switch on one GPU box: tools/env_ab.sh VAR "v1 v2" [band_proxy args...]  (interleaved, twice each)
var=$1; vals=$2; shift 2
for rep in 1 2; do
  for v in $vals; do
    export $var=$v
    echo "# $var=$v: $(python3 tools/band_proxy.py "$@" 2>&1 | grep '^{' | sed 's/.*"event_us_per_call": \([0-9.]*\).*/\1 us/')"
  done
done
