# L1 / L2 counters of the bench kernels (separate --pmc passes; run through gpurun)
# (a third pass with TA_*_STALLED_BY_TC / TCP_TCR_TCP_STALL counters aborted inside rocprofv3 and hung the run: not collected)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_cache; rm -rf $O; mkdir -p $O
B="python bench.py --size 768 --steps 1 --warmup 0 --cpu-sample 0"
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum -d $O/p1 -- $B > $O/p1.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_REQ_sum -d $O/p2 -- $B > $O/p2.log 2>&1
echo done
