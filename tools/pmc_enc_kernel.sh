set -e
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/pmc_enc; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $O/p1 -- python3 tools/enc_bench.py 2 > $O/p1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAIT_INST_ANY -d $O/p2 -- python3 tools/enc_bench.py 2 > $O/p2.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES -d $O/p3 -- python3 tools/enc_bench.py 2 > $O/p3.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_WAVES -d $O/p4 -- python3 tools/enc_bench.py 2 > $O/p4.log 2>&1
for p in p1 p2 p3 p4; do python3 tools/pmc_view.py $O/$p ${1:-flash_attn}; done
