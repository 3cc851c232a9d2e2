cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_b4; mkdir -p $O
rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(SQC?_[A-Z0-9_]+)" | sort -u > $O/counters.txt
wc -l $O/counters.txt
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/p1 -- python3 tools/batch_query.py 4 --reps=2 > $O/p1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p2 -- python3 tools/batch_query.py 4 --reps=2 > $O/p2.log 2>&1
timeout 300 rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_INPUT_VALID_READYB SQ_INST_CYCLES_SMEM SQC_TC_REQ SQC_TC_DATA_READ_REQ SQ_INSTS_SMEM_NORM --kernel-trace --output-format csv -d $O/p3 -- python3 tools/batch_query.py 4 --reps=2 > $O/p3.log 2>&1
python tools/pmc_summary.py $O/p1/*/*_counter_collection.csv $O/p2/*/*_counter_collection.csv $O/p3/*/*_counter_collection.csv > $O/sq_b4.json 2>$O/summary.err
rm -rf $O/p1 $O/p2 $O/p3
tail -5 $O/p3.log; head -c 600 $O/summary.err
