#!/bin/bash
# round 6: default-mode inflate() at small pieces, decode-kernel selections side by side (ZMI_INF_MW_MAX under ZMI_TUNING)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python - <<PY
import sys
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tests")
import bench
o = bench._oracle()
data = b"".join(o.gen_shard(i, 1 << 20) for i in range(4))
rc, gz = o.deflate(data, 6, 2)
open("/tmp/in.gz", "wb").write(gz)
open("/tmp/in.len", "w").write(str(len(data)))
PY
gcc -O2 -o /tmp/chunk_sweep $R/tools/chunk_sweep.c -ldl
export LD_LIBRARY_PATH=$R/zlib_rs_amd:$LD_LIBRARY_PATH
for mw in default 0 jump; do
  echo "== ZMI_INF_MW_MAX=$mw"
  if [ $mw = jump ]; then ZMI_TUNING=1 ZMI_INF_JUMP=1 /tmp/chunk_sweep $R/zlib_rs_amd/libz_mi355.so /tmp/in.gz $(cat /tmp/in.len) 31 1024 4096 16384 65536 262144
  elif [ $mw = default ]; then /tmp/chunk_sweep $R/zlib_rs_amd/libz_mi355.so /tmp/in.gz $(cat /tmp/in.len) 31 1024 4096 16384 65536 262144
  else ZMI_TUNING=1 ZMI_INF_MW_MAX=$mw /tmp/chunk_sweep $R/zlib_rs_amd/libz_mi355.so /tmp/in.gz $(cat /tmp/in.len) 31 1024 4096 16384 65536 262144; fi
done
