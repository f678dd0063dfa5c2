#!/bin/bash
# rocprofv3 passes for the deflate pipeline (run on the GPU box via gpurun)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O/prof_trace $O/prof_pmc1 $O/prof_pmc2 $O/prof_pmc3
cd $R
rocprofv3 --kernel-trace --stats -d $O/prof_trace -o trace -- python bench.py --shards 2048 --steps 2 --warmup 1 --no-cpu --verify 0 > $O/prof_trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d $O/prof_pmc1 -o pmc1 -- python bench.py --shards 1024 --steps 1 --warmup 0 --no-cpu --verify 0 > $O/prof_pmc1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU -d $O/prof_pmc2 -o pmc2 -- python bench.py --shards 1024 --steps 1 --warmup 0 --no-cpu --verify 0 > $O/prof_pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/prof_pmc3 -o pmc3 -- python bench.py --shards 1024 --steps 1 --warmup 0 --no-cpu --verify 0 > $O/prof_pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/prof_pmc3 -o pmc4 -- python bench.py --shards 1024 --steps 1 --warmup 0 --no-cpu --verify 0 > $O/prof_pmc4.log 2>&1
rocprofv3 -L > $O/counters_list.txt 2>&1
find $O -name "*.csv" | head -30
