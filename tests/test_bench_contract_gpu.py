"""The benchmark line the driver parses: `python bench.py` at a small size in a child process must
print exactly one JSON line with the contract's keys (task statement, section "Measurement"), the
roofline and cpu_baseline objects, and a step that includes the encode legs."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--size", "64", "--steps", "2",
                          "--warmup", "1", "--bm4dnet", "0", *extra], capture_output=True, text=True,
                         timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_default_line_has_the_contract_keys():
    d = _run()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["unit"] == "voxels/s" and d["data"] == "synthetic" and "workload" in d["config"]
    assert abs(d["value"] - 64 ** 3 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] == "port" and c["value"] > 0
    assert 0 < c["share_of_box"] <= 1 and c["cores"] <= c["cpus_visible"]
    # the PCIe-inclusive figures ride with the line, outside `value`
    e = d["end_to_end"]
    assert e["h2d_ms"] > 0 and e["denoise_ms"] > 0 and e["d2h_ms"] > 0
    assert abs(e["sequential_ms"] - (e["h2d_ms"] + e["denoise_ms"] + e["d2h_ms"])) < 1e-6 * e["sequential_ms"] + 1e-3
    assert e["streamed_chunk_local"]["ms"] > 0
    # "denoised+encoded": the encode legs are inside the timed step
    for phase in ("blockmatch_ht", "stage_ht", "blockmatch_wie", "stage_wie", "encode_u16", "dct_quantise",
                  "encode_idx"):
        assert d["phase_ms"][phase] > 0, phase
    e = d["encoded"]
    assert e["cratio_denoised"] > e["cratio_raw"] > 1.0
    if e.get("cratio_zstd5_shuffle") is not None:             # libzstd present: the reference's codec family
        assert e["cratio_denoised_same_chunks"] >= e["cratio_zstd5_shuffle"]
    # PSNR vs the CPU port on the same 256^3 volume (BASELINE.json's metric names it): < 0.01 dB apart
    p = d["psnr"]
    # round 4: integer aggregation sums -- the GPU's uint16 volume IS the port's
    assert p["delta_db"] == 0.0 and p["frac_differing"] == 0.0
    assert p["max_abs_u16"] == 0 and p["frac_beyond_one_count"] == 0.0
    assert p["gpu_equals_cpu"] is True and p["second_launch_identical"] is True and p["gpu_vs_cpu"] is None
    assert p["gpu_vs_clean"] > p["noisy_vs_clean"] + 5.0
    assert c["encode"]["exac_v2_port"]["cratio"] > 3.0


@pytest.mark.gpu
def test_no_encode_flag_says_so():
    d = _run("--no-encode", "--cpu-sample", "0")
    assert d["config"]["encode"] == "none" and "encode_u16" not in d["phase_ms"] and "cpu_baseline" not in d


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["volumes", "slabs", "chunks"])
def test_gpus_flag_launches_its_own_ranks(mode):
    """`python bench.py --gpus 2` with no launcher around it (how the driver calls it) must start two
    ranks itself: n_gpus == 2 in the line and twice the voxels of one rank.  BENCH_REHEARSAL=1 puts
    both ranks on this box's one GPU with a gloo rendezvous (never the measured configuration)."""
    env = dict(os.environ, BENCH_REHEARSAL="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    extra = ["--mode", mode] + (["--chunk", "32"] if mode == "chunks" else [])
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size", "64",
                          "--steps", "1", "--warmup", "1", "--bm4dnet", "0", "--cpu-sample", "0", *extra],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout[-2000:]        # ONE line on stdout: gloo's own chatter goes to stderr
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert abs(d["value"] - 2 * 64 ** 3 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]


@pytest.mark.gpu
def test_chunks_mode_takes_a_per_rank_shape():
    """--shape planes,rows,columns: the layout BASELINE config 4 runs at (256,4096,4096 per rank); here
    one rank, one layer of 2 x 3 chunks of 32^3."""
    d = _run("--mode", "chunks", "--chunk", "32", "--shape", "32,64,96")
    assert d["config"]["volume"] == [32, 64, 96]
    assert abs(d["value"] - 32 * 64 * 96 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    assert 15.0 < d["residual_std"] < 30.0 and d["rank0_lossless_cratio"] > 1.0


@pytest.mark.gpu
def test_volumes_mode_survives_a_dead_rccl_rendezvous():
    """The default mode has no data-path collective: when RCCL does not come up (BENCH_REHEARSAL=2 forces
    that, both ranks on this box's one GPU) the ranks meet over gloo and the line says so; the slab mode,
    which exchanges device planes, fails instead."""
    env = dict(os.environ, BENCH_REHEARSAL="2")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size", "64", "--steps", "1",
           "--warmup", "1", "--bm4dnet", "0", "--cpu-sample", "0"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and "gloo" in d["config"]["rendezvous"]
    out = subprocess.run(cmd + ["--mode", "slabs"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode != 0


@pytest.mark.gpu
def test_a_failing_rank_fails_the_launcher():
    env = dict(os.environ, BENCH_REHEARSAL="1", BENCH_FAIL_RANK="1")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size", "64",
                          "--steps", "1", "--warmup", "0", "--bm4dnet", "0", "--cpu-sample", "0"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode != 0


def test_launcher_parent_never_touches_the_gpu_or_torch():
    """CPU: the parent of `--gpus N` only spawns; with a bogus interpreter argument list the children
    die at once and the parent reports it -- without having imported torch or libexabm4d.so."""
    code = ("import sys, bench\n"
            "sys.argv = ['bench.py', '--gpus', '2', '--definitely-not-a-flag']\n"
            "rc = bench.launch_ranks(2)\n"
            "assert rc != 0, rc\n"
            "assert 'torch' not in sys.modules and 'aind_exaspim_image_compression._native' not in sys.modules\n"
            "print('parent clean, rc', rc)\n")
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=ROOT,
                         env=env)
    assert out.returncode == 0 and "parent clean" in out.stdout, out.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["volumes", "slabs", "chunks"])
def test_two_real_gpus_over_rccl(mode):
    """Where the box has at least two GPUs: the same launcher WITHOUT the rehearsal switch -- one rank
    per GPU, `nccl` (= RCCL) rendezvous, the halo exchanges of the slab / chunk modes over xGMI
    (HaloExchange: int16 planes as bytes, batch_isend_irecv).  Skipped on the one-GPU boxes the builder
    and the driver's test tier get; the driver's scaling run is the first place it executes."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "BENCH_REHEARSAL"):
        env.pop(k, None)
    extra = ["--mode", mode] + (["--chunk", "64"] if mode == "chunks" else [])
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size", "128",
                          "--steps", "1", "--warmup", "1", "--bm4dnet", "0", "--cpu-sample", "0", *extra],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")][0])
    assert d["n_gpus"] == 2 and 15.0 < d["residual_std"] < 30.0        # a denoised volume came back on rank 0
