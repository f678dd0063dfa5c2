set -e
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "inflate or roundtrip or golden" 2>&1 | tail -2
PROBE_CLASSES=1 PROBE_LEVELS=6 PROBE_S=2048,8192 timeout 600 python tools/gpu_probe.py 2>&1 | grep -E "inflate" | sed "s/^/r8192 /" | tee gpurun_out/probe_inf2.log
cp zlib_rs_amd/libzmi355.so /tmp/keep.so
for r in 4096 16384; do cp variants/libzmi355_r$r.so zlib_rs_amd/libzmi355.so; PROBE_LEVELS=6 PROBE_S=2048,8192 timeout 600 python tools/gpu_probe.py 2>&1 | grep -E "inflate" | sed "s/^/r$r /" | tee -a gpurun_out/probe_inf2.log; done
