cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/mfma_pmc
rm -rf $O; mkdir -p $O
cd $R
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/p1 -- python3 tools/sweep_mfma_check.py --reps=2 > $O/p1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p2 -- python3 tools/sweep_mfma_check.py --reps=2 > $O/p2.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS --kernel-trace --output-format csv -d $O/p3 -- python3 tools/sweep_mfma_check.py --reps=2 > $O/p3.log 2>&1
python tools/pmc_summary.py $O/p1/*/*_counter_collection.csv $O/p2/*/*_counter_collection.csv $O/p3/*/*_counter_collection.csv > $O/summary.json 2>$O/summary.err
tail -3 $O/p3.log
rm -rf $O/p1 $O/p2 $O/p3
