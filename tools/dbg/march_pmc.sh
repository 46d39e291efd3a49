# HBM / L2 counters of block matching alone, one tile per workgroup (0) against the march (8 blocks).
# Counter groups as in cache_counters.sh / profile_round.sh (larger groups do not fit one pass).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_march; rm -rf $O; mkdir -p $O
for m in 0 8; do
  timeout -k 10 150 rocprofv3 --pmc FETCH_SIZE -d $O/m$m/fetch -- python tools/dbg/bm_march_probe.py 1024 $m > $O/m$m.fetch.log 2>&1 || exit 1
  echo "fetch pass $m done" >> $O/progress.log
  timeout -k 10 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_REQ_sum -d $O/m$m/tcc -- python tools/dbg/bm_march_probe.py 1024 $m > $O/m$m.tcc.log 2>&1 || exit 1
  echo "tcc pass $m done" >> $O/progress.log
  python tools/pmc_summary.py $O/m$m $O/m$m.json
done
python - <<'P'
import json
for m in (0, 8):
    d = json.load(open(f"gpurun_out/prof_march/m{m}.json"))
    for k, v in d.items():
        if "bm_tile" in k:
            print(m, k[:60], {c: round(x["per_launch_mean"] / 1e6, 2) for c, x in v.items()})
P
