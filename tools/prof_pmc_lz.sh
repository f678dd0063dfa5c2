#!/bin/bash
# PMC passes focused on the deflate kernels (run on the GPU box via gpurun)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-v3}
cd $R
run() { n=$1; shift; mkdir -p $O/pmc_$TAG/$n; rocprofv3 --pmc "$@" -d $O/pmc_$TAG/$n -o p -- python bench.py --shards 1024 --steps 1 --warmup 0 --no-cpu --verify 0 --no-extras > $O/pmc_$TAG/$n.log 2>&1; }
run a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
run b SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
run c SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_ATOMIC SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH
python3 - <<PY
import sqlite3, glob
for db in sorted(glob.glob("$O/pmc_$TAG/*/*.db")):
    con=sqlite3.connect(db); cur=con.cursor()
    for r in cur.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection where kernel_name like 'zmi_lz77%' or kernel_name like 'zmi_encode%' or kernel_name like 'zmi_compact%' group by kernel_name, counter_name"):
        print('%s,%s,%.0f,%d'%(r[0].split('(')[0],r[1],r[2],r[3]))
PY
