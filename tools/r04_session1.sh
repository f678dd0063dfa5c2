#!/bin/bash
# Round 4, first GPU call: validation of everything built on the CPU so far + the baseline numbers of the kernel work.
#   gpurun --timeout 2400 -- 'bash tools/r04_session1.sh'
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04a
mkdir -p "$O"
cd "$R" || exit 1
export ZMI_TUNING=1
P="python tools/gpu_fast_probe.py --shards 16384 --levels 6 --reps 2 --host-verify 2"
# 1. kernel A/B first (the numbers the next hours of work hang on)
ZMI_LIB=variants/libzmi355_r03.so $P --class-times --classes --tag r03 > "$O/probe_r03.log" 2>&1; tail -12 "$O/probe_r03.log" | grep -v JSON
$P --class-times --classes --tag new > "$O/probe_new.log" 2>&1; grep -v JSON "$O/probe_new.log" | tail -12
ZMI_LIB=variants/libzmi355_noopt.so $P --class-times --classes --tag noopt > "$O/probe_noopt.log" 2>&1; grep -v JSON "$O/probe_noopt.log" | tail -12
for kv in "ZMI_BARREN_CHAIN=0" "ZMI_MIN_LIVE=16" "ZMI_MIN_LIVE=24" "ZMI_CHAIN=3" "ZMI_CHAIN=5" "ZMI_CHAIN=6 ZMI_MIN_LIVE=24" "ZMI_MIN_SUB_SPAN=0" "ZMI_BLOCK_TOKENS=8192" "ZMI_PRODUCERS=3" "ZMI_BARREN_CHAIN=0 ZMI_MIN_LIVE=16 ZMI_PRODUCERS=3"; do
  tag=$(echo "$kv" | tr ' =' '__')
  env $kv ZMI_LIB=variants/libzmi355_noopt.so $P --tag "$kv" > "$O/probe_$tag.log" 2>&1
  echo "== $kv"; grep "^L6" "$O/probe_$tag.log"
done
ZMI_LIB=variants/libzmi355_noopt.so python tools/gpu_fast_probe.py --shards 16384 --levels 1,9 --reps 1 --real --tag levels > "$O/probe_levels.log" 2>&1; grep -v JSON "$O/probe_levels.log" | tail -12
# 2. correctness on hardware
python -c "import __graft_entry__ as g; g.smoke()" > "$O/smoke.log" 2>&1; echo "smoke rc=$?"
timeout 1500 python -m pytest tests -m gpu -q -x > "$O/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -5 "$O/pytest_gpu.log"
# 3. the bench through its own launcher (small)
timeout 600 python bench.py --gpus 1 --shards 4096 --steps 1 --warmup 1 --no-extras --no-cpu --scratch-gib 18 > "$O/bench_small.json" 2> "$O/bench_small.err"; echo "bench rc=$?"; cut -c1-600 "$O/bench_small.json"
