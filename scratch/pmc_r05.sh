#!/bin/bash
# rocprofv3 PMC passes (program directly after `--`; counters in passes of their own, with --kernel-trace only) over four proves of
# benchmark/1600k: per kernel family the launches of the LAST prove (witness resident, kernels of the prove serialised by the
# profiler) — duration, waves, VALU instructions, VALU-active / busy cycles, HBM-side fetch and write bytes.
# output: gpurun_out/r05_pmc_kernels.txt (copied to profiles/ by hand)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/scratch/pmc_child.py > /dev/null 2>&1   # inputs cached in /tmp before the profiled runs
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmcA -- python3 $R/scratch/pmc_child.py > $R/gpurun_out/pmcA.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmcB -- python3 $R/scratch/pmc_child.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmcC -- python3 $R/scratch/pmc_child.py > /dev/null 2>&1
cd $R
python3 scratch/pmc_r05_summary.py > gpurun_out/r05_pmc_kernels.txt 2> gpurun_out/r05_pmc_kernels.err
rm -rf gpurun_out/pmcA gpurun_out/pmcB gpurun_out/pmcC
cat gpurun_out/r05_pmc_kernels.txt
