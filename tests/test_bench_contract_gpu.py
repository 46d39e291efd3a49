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
    # "denoised+encoded": the encode legs are inside the timed step
    for phase in ("blockmatch_ht", "stage_ht", "blockmatch_wie", "stage_wie", "encode_u16", "dct_quantise",
                  "encode_idx"):
        assert d["phase_ms"][phase] > 0, phase
    assert d["encoded"]["cratio_denoised"] > d["encoded"]["cratio_raw"] > 1.0


@pytest.mark.gpu
def test_no_encode_flag_says_so():
    d = _run("--no-encode", "--cpu-sample", "0")
    assert d["config"]["encode"] == "none" and "encode_u16" not in d["phase_ms"] and "cpu_baseline" not in d
