run() { echo "$@"; env "$@" PROBE_LEVELS=${LV:-6} PROBE_S=2048 timeout 300 python tools/gpu_probe.py 2>&1 | grep -E "deflate"; }
run ZMI_CHAIN=8
run ZMI_CHAIN=7
run ZMI_CHAIN=6
LV=1,3,9 run ZMI_X=1
