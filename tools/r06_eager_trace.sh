#!/bin/bash
# round 6: what one default-mode inflate() call of a small piece costs -- kernel trace + HIP API trace of the reference's chunk sweep
# (tools/chunk_sweep.c) at ONE chunk size on the drop-in.  usage (GPU box): tools/r06_eager_trace.sh [chunk] -> gpurun_out/r06_eager_*
set -e
CH=${1:-4096}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python - <<PY
import sys, os
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
/tmp/chunk_sweep $R/zlib_rs_amd/libz_mi355.so /tmp/in.gz $(cat /tmp/in.len) 31 $CH > $R/gpurun_out/r06_eager_plain.txt 2>&1
rocprofv3 --kernel-trace --hip-runtime-trace --stats -d /tmp/prof_eager -o eager -- /tmp/chunk_sweep $R/zlib_rs_amd/libz_mi355.so /tmp/in.gz $(cat /tmp/in.len) 31 $CH > $R/gpurun_out/r06_eager_rocprof.txt 2>&1 || true
find /tmp/prof_eager -type f > $R/gpurun_out/r06_eager_files.txt 2>&1
python3 - > $R/gpurun_out/r06_eager_stats.txt 2>&1 <<PY
import sqlite3, glob
con = sqlite3.connect(glob.glob("/tmp/prof_eager/*.db")[0]); cur = con.cursor()
names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
print("kernel,calls,total_us,avg_us")
for r in cur.execute("select name, count(*), sum(duration), avg(duration) from kernels group by name order by sum(duration) desc"):
    print("%s,%d,%.1f,%.2f" % (r[0].split('(')[0][:70], r[1], r[2] / 1e3, r[3] / 1e3))
for v in ("regions", "regions_and_samples"):
    if v in names:
        cols = [c[1] for c in cur.execute("pragma table_info(%s)" % v)]
        print(v, cols)
        try:
            print("api,calls,total_us,avg_us")
            for r in cur.execute("select name, count(*), sum(end - start), avg(end - start) from %s group by name order by sum(end - start) desc limit 25" % v):
                print("%s,%d,%.1f,%.2f" % (r[0][:60], r[1], r[2] / 1e3, r[3] / 1e3))
        except Exception as ex:
            print("query failed", ex)
        break
PY
