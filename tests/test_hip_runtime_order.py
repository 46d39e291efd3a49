"""One HIP runtime per process, whatever the import order.  The reference's
callers import `bm4d` before torch (machine_learning/data_handling.py:12, then inference.py's
torch); PyTorch-ROCm wheels bundle their own libamdhip64.so and the copy loaded second finds no
devices.  `_native.lib()` settles it; these tests run each order in a fresh interpreter."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "aind-exaspim-image-compression_amd")
PRE = f"import sys; sys.path[:0] = [{ROOT!r}, {PKG!r}]\n"


def _py(code, timeout=600):
    r = subprocess.run([sys.executable, "-c", PRE + code], capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    return r.stdout


@pytest.mark.parametrize("first", ["native", "torch"])
def test_one_runtime_is_mapped_in_either_order(first):
    a = "from aind_exaspim_image_compression import _native; _native.lib()"
    b = "import torch"
    code = "\n".join([a, b] if first == "native" else [b, a]) + \
        "\nfrom aind_exaspim_image_compression import _native\nprint(len(_native._mapped_hip_runtimes()))\n"
    assert _py(code).strip().endswith("1")


@pytest.mark.gpu
def test_bm4d_first_then_torch_cuda():
    out = _py(
        "import numpy as np\n"
        "from bm4d import bm4d\n"
        "z = np.random.default_rng(0).normal(100, 24, (32, 32, 32)).astype(np.float32)\n"
        "y = bm4d(z, 24.0)\n"
        "assert y.shape == z.shape and float(np.var(y)) < 0.5 * float(np.var(z))\n"
        "import torch\n"
        "t = torch.zeros(4).cuda() + 1\n"
        "assert torch.cuda.device_count() >= 1 and float(t.sum().item()) == 4.0\n"
        "y2 = bm4d(z, 24.0)\n"
        "assert np.array_equal(np.asarray(y), np.asarray(y2)) or np.allclose(y, y2, atol=1e-3)\n"
        "print('ok')\n")
    assert out.strip().endswith("ok")


@pytest.mark.gpu
def test_torch_cuda_first_then_bm4d():
    out = _py(
        "import numpy as np, torch\n"
        "t = torch.ones(4).cuda()\n"
        "from bm4d import bm4d\n"
        "z = np.random.default_rng(0).normal(100, 24, (32, 32, 32)).astype(np.float32)\n"
        "y = bm4d(z, 24.0)\n"
        "assert float(np.var(y)) < 0.5 * float(np.var(z)) and float((t * 2).sum().item()) == 8.0\n"
        "print('ok')\n")
    assert out.strip().endswith("ok")
