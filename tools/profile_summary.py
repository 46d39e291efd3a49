"""Reduce one profiling round (tools/dbg/profile_round.sh) to the files bench.py and the docs cite:

    python tools/profile_summary.py gpurun_out/prof_round profiles/r02 1024

writes profiles/r02/kernel_stats_bench_default_<N>.csv (rocprofv3 --kernel-trace --stats),
profiles/r02/pmc_counters_<N>.json (per-kernel means of the separate --pmc passes) and
profiles/latest_counters.json: per bench phase the kernel-trace average, HBM bytes per launch =
(2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md section HBM)
and the VALU counters (SQ_INSTS_VALU; busy cycles = SQ_ACTIVE_INST_VALU quad-cycles x 4)."""
import csv
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PHASES = {           # bench phase -> (kernel-name fragment, launches of it per bench step)
    "counts_from_u16": "OpCountsFromU16",
    # uint16 pipelines: both matching passes run in the integer tile kernel (stage 2 on the basic estimate
    # rounded to counts); its per-launch means are means over the two launches of a step
    "blockmatch_ht": "bm_tile16_kernel",
    "blockmatch_wie": "bm_tile16_kernel",     # (the rounded estimate is written by normalize_basic)
    "stage_ht": "stage_half_kernel<false>",
    "stage_wie": "stage_half_kernel<true>",
    # (round 3: the pipelines' normalisations carry the z pass of the denominator convolution)
    "normalize_basic": "normalize_zconv_kernel<4, false>",
    "normalize_out": "normalize_zconv_kernel<4, true>",
    # EXAC v2 legs are three kernels each (model, coder, pack; the pack kernel serves both legs, so its
    # per-launch mean is the mean over the two): time and counters are summed over the fragments
    "encode_u16": ["rans2_model_strips_kernel", "rans2_code_kernel<2>", "rans2_pack_kernel"],
    "encode_idx": ["rans2_model_rows32_kernel", "rans2_code_kernel<4>", "rans2_pack_kernel"],
    "dct_quantise": "dctq_forward",
}


def newest_database_only(src):
    """gpurun MERGES what a call wrote into the local gpurun_out/: the databases of earlier profiling rounds stay
    next to the new ones and pmc_summary.py would average over all of them (kernels that no longer run
    included).  Keep the newest database of every pass."""
    import glob
    for d in sorted(os.listdir(src)):
        dbs = sorted(glob.glob(os.path.join(src, d, "**", "*_results.db"), recursive=True), key=os.path.getmtime)
        for old in dbs[:-1]:
            print("removing the database of an earlier round:", old)
            os.remove(old)


def main():
    src, dst, size = sys.argv[1], sys.argv[2], int(sys.argv[3])
    os.makedirs(dst, exist_ok=True)
    newest_database_only(src)
    stats_csv = os.path.join(dst, f"kernel_stats_bench_default_{size}.csv")
    pmc_json = os.path.join(dst, f"pmc_counters_{size}.json")
    subprocess.run([sys.executable, os.path.join(HERE, "pmc_summary.py"), os.path.join(src, "trace"),
                    os.path.join(dst, "_trace_unused.json"), stats_csv], check=True,
                   stdout=subprocess.DEVNULL)
    os.remove(os.path.join(dst, "_trace_unused.json"))
    pmc_dirs = [d for d in ("fetch", "write", "sq1", "sq2") if os.path.isdir(os.path.join(src, d))]
    tmp = os.path.join(src, "_pmc_only")
    os.makedirs(tmp, exist_ok=True)
    for d in pmc_dirs:
        link = os.path.join(tmp, d)
        if not os.path.exists(link):
            os.symlink(os.path.abspath(os.path.join(src, d)), link)
    subprocess.run([sys.executable, os.path.join(HERE, "pmc_summary.py"), tmp, pmc_json], check=True,
                   stdout=subprocess.DEVNULL)
    with open(pmc_json) as f:
        pmc = json.load(f)
    avg = {}
    with open(stats_csv, newline="") as f:
        for row in csv.DictReader(f):
            avg[row["Name"]] = float(row["AverageNs"]) / 1e6
    commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True,
                            cwd=os.path.dirname(HERE)).stdout.strip()
    kernels = {}
    for phase, frags in PHASES.items():
        frags = [frags] if isinstance(frags, str) else list(frags)
        names = [next((k for k in pmc if f in k), None) for f in frags]
        tnames = [next((k for k in avg if f in k), None) for f in frags]
        if all(n is None for n in names) and all(t is None for t in tnames):
            continue

        def val(counter):
            got = [pmc[n][counter]["per_launch_mean"] for n in names if n and counter in pmc.get(n, {})]
            return sum(got) if len(got) == len([n for n in names if n]) and got else None

        rec = {"kernel": " + ".join(n or t or "?" for n, t in zip(names, tnames)),
               "avg_ms": sum(avg[t] for t in tnames if t) if any(tnames) else None}
        if len(frags) > 1:
            rec["parts_ms"] = {t: avg[t] for t in tnames if t}
        if val("FETCH_SIZE") is not None and val("WRITE_SIZE") is not None:
            rec["hbm_bytes"] = (2.0 * val("FETCH_SIZE") + val("WRITE_SIZE")) * 1024.0
        if val("SQ_INSTS_VALU") is not None:
            rec["valu_insts"] = val("SQ_INSTS_VALU")
        if val("SQ_ACTIVE_INST_VALU") is not None:
            rec["valu_busy_cycles"] = 4.0 * val("SQ_ACTIVE_INST_VALU")
        for extra in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_INSTS_LDS",
                      "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_SALU"):
            if val(extra) is not None:
                rec[extra] = val(extra)
        kernels[phase] = rec
    out = {"_comment": "per bench phase: rocprofv3 kernel-trace average and per-launch PMC means of "
                       "`python bench.py` (tools/dbg/profile_round.sh); written by tools/profile_summary.py",
           "volume": [size] * 3, "source": os.path.relpath(dst, os.path.dirname(HERE)), "commit": commit,
           "kernels": kernels}
    with open(os.path.join(os.path.dirname(HERE), "profiles", "latest_counters.json"), "w") as f:
        json.dump(out, f, indent=1)
    for k, v in kernels.items():
        print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a != "kernel"})


if __name__ == "__main__":
    main()
