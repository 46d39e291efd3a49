set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_round; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; tail -1 $O/gpu_tests.log
timeout -k 10 300 python tools/stress_stage.py 384 > $O/stress.log 2>&1; tail -1 $O/stress.log
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json
rocprofv3 --kernel-trace --stats -d $O/trace -- python bench.py --steps 3 --warmup 1 --cpu-sample 0 > $O/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/fetch -- python bench.py --steps 1 --warmup 0 --cpu-sample 0 > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/write -- python bench.py --steps 1 --warmup 0 --cpu-sample 0 > $O/write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES -d $O/sq1 -- python bench.py --steps 1 --warmup 0 --cpu-sample 0 > $O/sq1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d $O/sq2 -- python bench.py --steps 1 --warmup 0 --cpu-sample 0 > $O/sq2.log 2>&1
echo profiling done
