#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box into gpurun_out/prof/ (copy what is to be judged into profiles/):
#   solo / pipelined kernel stats of bench.py, MFMA-busy + GRBM passes of the encoder (tools/pmc_mfma.py reduces them).
# Each step appends a line to gpurun_out/prof/progress.log so a long run never looks hung.
set -e
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/prof; rm -rf $O; mkdir -p $O
say() { echo "[$(date +%T)] $*" | tee -a $O/progress.log; }
say solo stats
rocprofv3 --kernel-trace --stats --output-format csv -d $O/solo -- python3 bench.py --no-pipeline --no-cpu-baseline --no-x4 --no-extras > $O/bench_solo.json 2> $O/bench_solo.err
say pipelined stats
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pipe -- python3 bench.py --no-cpu-baseline --no-x4 --no-extras > $O/bench_pipe.json 2> $O/bench_pipe.err
say mfma busy
rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES -d $O/busy -- python3 tools/enc_bench.py 2 > /dev/null 2>&1
say grbm
rocprofv3 --kernel-trace --output-format csv --pmc GRBM_GUI_ACTIVE -d $O/act -- python3 tools/enc_bench.py 2 > /dev/null 2>&1
python3 tools/pmc_mfma.py $O/busy $O/act $O/pmc_mfma.json | tee -a $O/progress.log
cp $(ls $O/solo/*/*kernel_stats.csv | head -1) $O/kernel_stats_solo.csv
cp $(ls $O/pipe/*/*kernel_stats.csv | head -1) $O/kernel_stats_pipelined.csv
say done
