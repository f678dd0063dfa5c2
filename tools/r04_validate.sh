#!/bin/bash
# full validation on the GPU box: smoke, the -m gpu suite, the default bench line
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r04v}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd "$R" || exit 1
python -c "import __graft_entry__ as g; g.smoke()" > "$O/smoke.log" 2>&1; echo "smoke rc=$?"; tail -1 "$O/smoke.log"
timeout 1500 python -m pytest tests -m gpu -q > "$O/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -4 "$O/pytest_gpu.log"
timeout 900 python bench.py > "$O/bench.json" 2> "$O/bench.err"; echo "bench rc=$?"; cat "$O/bench.json"; tail -5 "$O/bench.err"
