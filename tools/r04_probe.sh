#!/bin/bash
# kernel A/B on the GPU box: tools/r04_probe.sh TAG "lib[:ENV=V,ENV=V][:probe flags]" ...   (lib = product | variants/NAME)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd "$R" || exit 1
export ZMI_TUNING=1
i=0
for spec in "$@"; do
  IFS=':' read -r lib envs flags <<< "$spec"
  i=$((i+1))
  L=""; [ "$lib" != "product" ] && L="ZMI_LIB=variants/libzmi355_$lib.so"
  E=$(echo "$envs" | tr ',' ' ')
  F=${flags:---shards 16384 --levels 6 --reps 2 --host-verify 2}
  echo "== [$i] $lib $E $F"
  env $L $E timeout 150 python tools/gpu_fast_probe.py $F --tag "$lib $E" > "$O/probe_$i.log" 2>&1 || echo "   (probe $i: rc $? -- timed out or failed)"
  grep -v "^JSON\|^# lib" "$O/probe_$i.log" | tail -14
done
