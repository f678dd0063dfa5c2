#!/bin/bash
# round 6 A/B: builds variants/libzmi355_NAME.so for each "NAME:-DFLAG ..." argument and probes each (GPU box)
mkdir -p gpurun_out variants
export ZMI_TUNING=1 PROBE_S=${PROBE_S:-16384} PROBE_LEVELS=${PROBE_LEVELS:-6} PROBE_SCRATCH_GIB=${PROBE_SCRATCH_GIB:-70}
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  [ -f variants/libzmi355_$name.so ] || tools/build_variant.sh $name $flags > /dev/null 2>&1
  echo "== $name ($flags)"
  PROBE_LIB=variants/libzmi355_$name.so timeout 600 python tools/gpu_probe.py 2>&1 | grep "deflate\|inflate\|class"
done
