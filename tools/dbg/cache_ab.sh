# L1 / L2 counters of the stage kernels with and without the interleaved Wiener volume (768^3), + the stamp profile
# (every rocprofv3 under `timeout -k 10`: an over-full --pmc list once made rocprofv3 abort and sit until the
#  silence guard killed the call -- README; a failed pass is reported, the script goes on to the next)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_cache; rm -rf $O; mkdir -p $O
for pv in 1 0; do
  export EXABM4D_STAGE_PAIRVOL=$pv
  B="python bench.py --size 768 --steps 1 --warmup 0 --cpu-sample 0 --bm4dnet 0 --no-encode"
  timeout -k 10 400 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum -d $O/pv${pv}_p1 -- $B > $O/pv${pv}_p1.log 2>&1 && echo "pv$pv p1 ok" || echo "pv$pv p1 FAILED (see $O/pv${pv}_p1.log)"
  timeout -k 10 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_REQ_sum -d $O/pv${pv}_p2 -- $B > $O/pv${pv}_p2.log 2>&1 && echo "pv$pv p2 ok" || echo "pv$pv p2 FAILED (see $O/pv${pv}_p2.log)"
done
unset EXABM4D_STAGE_PAIRVOL
SIZE=512 timeout -k 10 300 python tools/dbg/stamps.py > $O/stamps_512.log 2>&1 || echo "stamps FAILED"; tail -4 $O/stamps_512.log
echo done
