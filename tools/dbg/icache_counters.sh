# instruction-cache counters of the bench kernels (one pass; run through gpurun)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_icache; rm -rf $O; mkdir -p $O
timeout -k 10 100 rocprofv3 -L > $O/avail.txt 2>&1
grep -o -i "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_IFETCH[A-Z_]*\|SQC_INST[A-Z_]*" $O/avail.txt | sort -u | tr '\n' ' '; echo
B="python bench.py --steps 1 --warmup 0 --cpu-sample 0 --bm4dnet 0"
timeout -k 10 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES -d $O/p1 -- $B > $O/p1.log 2>&1 && echo p1 ok
python tools/pmc_summary.py $O/p1 $O/icache.json && python - <<'P'
import json
d = json.load(open("gpurun_out/prof_icache/icache.json"))
for k, v in d.items():
    c = {n: x["per_launch_mean"] for n, x in v.items()}
    if c.get("SQC_ICACHE_REQ", 0) > 1e7:
        print(k[:70], {n: f"{x:.3g}" for n, x in c.items()}, "miss ratio %.4f" % (c.get("SQC_ICACHE_MISSES", 0) / c["SQC_ICACHE_REQ"]))
P
