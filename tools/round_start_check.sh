#!/bin/bash
# First GPU call of a round (one box acquisition for everything that was only emulator-validated before):
#   gpurun --timeout 2400 -- 'bash tools/round_start_check.sh r02a'
# smoke, the -m gpu suite, the differential ABI probes on the device path, the default bench line, then the
# rocprofv3 kernel-trace + PMC collection of tools/prof_final.sh.  Everything lands under gpurun_out/TAG.
TAG=${1:-r02a}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd "$R" || exit 1
python -c "import __graft_entry__ as g; g.smoke()" > "$O/smoke.log" 2>&1; echo "smoke rc=$?"
timeout 1500 python -m pytest tests -m gpu -q > "$O/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -3 "$O/pytest_gpu.log"
python tools/abi_probe_inflate.py --gpu > "$O/abi_probe_inflate.log" 2>&1; grep -c "^OK" "$O/abi_probe_inflate.log"
python tools/abi_probe_deflate.py --gpu > "$O/abi_probe_deflate.log" 2>&1; grep -v "^OK" "$O/abi_probe_deflate.log"
python tools/gpu_abi_latency.py > "$O/abi_latency.csv" 2>&1; cat "$O/abi_latency.csv"
timeout 600 python bench.py > "$O/bench.json" 2> "$O/bench.err"; echo "bench rc=$?"; cat "$O/bench.json"
bash tools/prof_final.sh "$TAG" > "$O/rocprofv3_summary.csv" 2> "$O/prof.err"; echo "prof rc=$?"; head -12 "$O/rocprofv3_summary.csv"
