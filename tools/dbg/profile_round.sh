# gpurun --timeout 1200 -- bash tools/dbg/profile_round.sh : the passes behind profiles/rNN/
# (reduce afterwards with: python tools/profile_summary.py gpurun_out/prof_round profiles/rNN 1024)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_round; rm -rf $O; mkdir -p $O
B="python bench.py --steps 1 --warmup 0 --cpu-sample 0 --bm4dnet 0 --end-to-end 0"
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/trace -- python bench.py --steps 3 --warmup 1 --cpu-sample 0 --bm4dnet 0 --end-to-end 0 > $O/trace.log 2>&1 && echo trace ok
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -- $B > $O/fetch.log 2>&1 && echo fetch ok
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE -d $O/write -- $B > $O/write.log 2>&1 && echo write ok
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/sq1 -- $B > $O/sq1.log 2>&1 && echo sq1 ok
timeout -k 10 400 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES -d $O/sq2 -- $B > $O/sq2.log 2>&1 && echo sq2 ok
# round 3 extras: encode legs alone (both formats, per kernel), the R-D sweep with real bytes, the fuzz run
if [ "$1" != "short" ]; then
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/exac_trace -- python tools/bench_exac.py 1024 5 > $O/bench_exac.json 2> $O/bench_exac.err && echo exac ok
timeout -k 10 500 python tools/rd_sweep.py 1024 > $O/rd_sweep_1024.json 2> $O/rd_sweep.log && echo rd ok
fi
timeout -k 10 260 python tools/fuzz_parity.py 200 4 > $O/fuzz_parity.log 2>&1; tail -2 $O/fuzz_parity.log
du -sh $O; echo profiling done
