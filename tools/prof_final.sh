#!/bin/bash
# Round-end evidence collection (run on the GPU box via gpurun, on the SHIPPED commit: no kernel change after it):
#   gpurun --timeout 2400 -- 'bash tools/prof_final.sh r03'
# 1. rocprofv3 --kernel-trace --stats of the DEFAULT bench line, extras included (configs[2] inflate, level 1 / 9 sweep, real
#    data, stream ABI): every kernel gets rows per launch size (grid), so the 64-stream warm-ups do not pollute the averages
# 2. HBM traffic: separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of one 16 384-shard deflate launch + its round trip
# 3. SQ counters of the same launch
# Everything lands under gpurun_out/TAG; the summaries are printed to stdout (-> profiles/TAG_rocprofv3_summary.csv).
cd /tmp && export TMPDIR=/tmp
# bench.py as the rank itself (WORLD_SIZE set: no launcher in between -- rocprofv3 follows the process it starts)
export WORLD_SIZE=1 RANK=0 LOCAL_RANK=0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r05}
cd $R
mkdir -p $O/$TAG/trace $O/$TAG/fetch $O/$TAG/write $O/$TAG/sq
rocprofv3 --kernel-trace --stats -d $O/$TAG/trace -o t -- python bench.py --steps 2 --warmup 1 --no-cpu --verify 0 > $O/$TAG/trace_bench.json 2> $O/$TAG/trace.log
SMALL="python bench.py --shards 16384 --steps 1 --warmup 0 --no-cpu --verify 0 --no-extras --no-stitch"
rocprofv3 --pmc FETCH_SIZE -d $O/$TAG/fetch -o p -- $SMALL > $O/$TAG/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/$TAG/write -o p -- $SMALL > $O/$TAG/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU -d $O/$TAG/sq -o p -- $SMALL > $O/$TAG/sq.log 2>&1
python3 - <<PY
import sqlite3, glob, json, datetime
def short(n):
    n = n.split('(')[0].replace('void ', '')
    return n
con = sqlite3.connect(glob.glob("$O/$TAG/trace/*.db")[0]); cur = con.cursor()
print("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu --verify 0   (the default bench line incl. its extras;")
print("# rows per kernel AND launch size: grid = workgroups; deflate launch group = 16384 shards = 16 GiB raw; inflate launches of 16384 / 4096 / 3 / 64 streams)")
print("kernel,grid_workgroups,calls,total_us,avg_us")
rows = cur.execute("select name, grid_x / workgroup_x, count(*), sum(duration), avg(duration) from kernels where name like '%zmi_%' group by name, grid_x / workgroup_x order by sum(duration) desc").fetchall()
for r in rows:
    print('%s,%d,%d,%.1f,%.1f' % (short(r[0]), r[1], r[2], r[3] / 1e3, r[4] / 1e3))
print("# PMC passes: python bench.py --shards 16384 --steps 1 --warmup 0 --no-extras --no-stitch (one launch of each deflate kernel = 16384 shards = 16 GiB raw;")
print("# inflate kernels: the round trip, one 16384-stream launch + a 64-stream warm-up)")
print("kernel,counter,sum_over_dispatches,dispatches")
traffic = {}
for d in ("fetch", "write", "sq"):
    con = sqlite3.connect(glob.glob("$O/$TAG/%s/*.db" % d)[0]); cur = con.cursor()
    for r in cur.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection where kernel_name like '%zmi_%' group by kernel_name, counter_name"):
        k = short(r[0]).split('<')[0]
        print('%s,%s,%.0f,%d' % (k, r[1], r[2], r[3]))
        if r[1] in ("FETCH_SIZE", "WRITE_SIZE"):
            # inflate kernels: a 64-stream warm-up launch + the 16384-stream launch -> the sum is the large launch's traffic (+0.4 %)
            traffic.setdefault(k, {})[r[1]] = r[2] if ("inflate" in k or "jump" in k) else r[2] / r[3]
# gfx950: FETCH_SIZE counts 64 B per 128 B request for wide coalesced reads -> double it (MI355X_MICROARCH.md, HBM)
res = {k: {"fetch_bytes_per_launch": v.get("FETCH_SIZE", 0) * 1024 * 2, "write_bytes_per_launch": v.get("WRITE_SIZE", 0) * 1024,
           "shards_per_launch": 16384} for k, v in traffic.items()}
res["_collected"] = "$TAG, " + datetime.date.today().isoformat()
import sys
sys.path.insert(0, "$R")
import bench
res["_csrc_sha16"] = bench.csrc_sha16()   # bench.py replays this file only against the kernel sources it was collected from
json.dump(res, open("$O/$TAG/traffic.json", "w"), indent=1)
PY
# (the caller redirects stdout into gpurun_out/TAG/rocprofv3_summary.csv: the issue file is made from it afterwards)
echo "python tools/make_issue_json.py gpurun_out/$TAG/rocprofv3_summary.csv profiles/${TAG}_issue.json $TAG" > $O/$TAG/make_issue.cmd
