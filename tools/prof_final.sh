#!/bin/bash
# Round-end profile collection (run on the GPU box via gpurun): kernel-trace stats + HBM traffic PMC
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r01}
cd $R
mkdir -p $O/$TAG/trace $O/$TAG/fetch $O/$TAG/write $O/$TAG/sq
rocprofv3 --kernel-trace --stats -d $O/$TAG/trace -o t -- python bench.py --shards 16384 --steps 2 --warmup 1 --no-cpu --verify 0 --no-extras > $O/$TAG/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/$TAG/fetch -o p -- python bench.py --shards 16384 --steps 1 --warmup 0 --no-cpu --verify 0 --no-extras > $O/$TAG/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/$TAG/write -o p -- python bench.py --shards 16384 --steps 1 --warmup 0 --no-cpu --verify 0 --no-extras > $O/$TAG/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU -d $O/$TAG/sq -o p -- python bench.py --shards 16384 --steps 1 --warmup 0 --no-cpu --verify 0 --no-extras > $O/$TAG/sq.log 2>&1
python3 - <<PY
import sqlite3, glob, json
out = {}
con = sqlite3.connect(glob.glob("$O/$TAG/trace/*.db")[0]); cur = con.cursor()
print("# rocprofv3 --kernel-trace --stats -- python bench.py --shards 16384 --steps 2 --warmup 1 --no-cpu --verify 0 --no-extras (durations in ns; one deflate launch group = 16384 shards = 16 GiB raw, the launch size of the full bench)")
print("kernel,calls,total_us,avg_us,pct")
for r in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    if "zmi_" in r[0]:
        print('%s,%d,%.1f,%.1f,%.2f' % (r[0].split('(')[0].replace('void ', ''), r[1], r[2], r[3], r[4]))
print("# PMC passes: python bench.py --shards 16384 --steps 1 --warmup 0 (one launch of each deflate kernel = 16384 shards = 16 GiB raw; inflate kernels: a 64-stream warm-up launch + one 16384-stream launch)")
print("kernel,counter,sum_over_dispatches,dispatches")
traffic = {}
for d in ("fetch", "write", "sq"):
    con = sqlite3.connect(glob.glob("$O/$TAG/%s/*.db" % d)[0]); cur = con.cursor()
    for r in cur.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection where kernel_name like '%zmi_%' group by kernel_name, counter_name"):
        k = r[0].split('(')[0].replace('void ', '').split('<')[0]
        k = 'zmi_lz77_kernel' if k.startswith('zmi_lz77_kernel') else k
        print('%s,%s,%.0f,%d' % (k, r[1], r[2], r[3]))
        if r[1] in ("FETCH_SIZE", "WRITE_SIZE"):
            # inflate kernels: a 64-stream warm-up launch + the 16384-stream launch -> the sum is the large launch's traffic (+0.4 %)
            traffic.setdefault(k, {})[r[1]] = r[2] if "inflate" in k else r[2] / r[3]
# gfx950: FETCH_SIZE counts 64 B per 128 B request for wide coalesced reads -> double it (MI355X_MICROARCH.md, HBM)
import datetime
res = {k: {"fetch_bytes_per_launch": v.get("FETCH_SIZE", 0) * 1024 * 2, "write_bytes_per_launch": v.get("WRITE_SIZE", 0) * 1024,
           "shards_per_launch": 16384} for k, v in traffic.items()}
res["_collected"] = "$TAG, " + datetime.date.today().isoformat()
json.dump(res, open("$O/$TAG/traffic.json", "w"), indent=1)
PY
