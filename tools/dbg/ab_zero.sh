mkdir -p gpurun_out/r4
for i in 1 2; do
for z in 0 1; do
  EXABM4D_ZERO_OVERLAP=$z timeout -k 10 200 python bench.py --size 1024 --steps 3 --warmup 1 --cpu-sample 0 --bm4dnet 0 --end-to-end 0 --no-encode > gpurun_out/r4/ab_zero_${z}_$i.json 2> gpurun_out/r4/ab_zero_${z}_$i.err || { echo "zero $z FAILED"; tail -3 gpurun_out/r4/ab_zero_${z}_$i.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r4/ab_zero_${z}_$i.json"))
p=d["phase_ms"]
print("zero_overlap=$z", round(d["ms_per_step"],1), {k:round(v,1) for k,v in p.items()}, d["psnr"]["gpu_equals_cpu"] if "psnr" in d else None)
PY
done
done
