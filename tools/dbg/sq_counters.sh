# gpurun -- bash tools/dbg/sq_counters.sh [SIZE]: issue / wait / LDS counters of the pipeline's kernels
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
S=${1:-768}
O=gpurun_out/sq_$S; rm -rf $O; mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1 || true
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/p1 -- python bench.py --size $S --steps 1 --warmup 0 --cpu-sample 0 --no-encode > $O/p1.log 2>&1 || tail -3 $O/p1.log
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES -d $O/p2 -- python bench.py --size $S --steps 1 --warmup 0 --cpu-sample 0 --no-encode > $O/p2.log 2>&1 || tail -3 $O/p2.log
python tools/pmc_summary.py $O $O/summary.json > $O/summary.txt 2>&1
cat $O/summary.txt | cut -c1-900
