cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_enc; mkdir -p $O/a $O/b
cd $R
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU -d $O/a -o p -- python bench.py --shards 16384 --steps 1 --warmup 0 --no-cpu --verify 0 --no-extras > $O/a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE -d $O/b -o p -- python bench.py --shards 16384 --steps 1 --warmup 0 --no-cpu --verify 0 --no-extras > $O/b.log 2>&1
python3 - <<PY
import sqlite3, glob
for d in ("a","b"):
    fs = glob.glob("$O/%s/*.db" % d)
    if not fs: print("no db", d); continue
    con = sqlite3.connect(fs[0]); cur = con.cursor()
    for r in cur.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection where kernel_name like '%zmi_%' group by kernel_name, counter_name"):
        print(r[0].split('(')[0][:40], r[1], "%.4g" % r[2], r[3])
PY
tail -3 $O/b.log
