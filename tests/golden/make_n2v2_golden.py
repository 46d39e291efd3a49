"""Writes tests/golden/n2v2.npz + n2v2_state.json by IMPORTING THE REFERENCE's N2V2UNet
(machine_learning/unet3d.py:392-475; MaxBlurPool3D :493-535, UpNoSkip3D :538-571).  Run in the
build container only (the reference does not exist on the GPU box):

    PYTHONPATH=/root/reference/src python tests/golden/make_n2v2_golden.py

Data only: the inputs are regenerated from seeds by the tests, the expected outputs are stored.
"""
import json
import os

import numpy as np
import torch

from aind_exaspim_image_compression.machine_learning import unet3d as ref_unet

HERE = os.path.dirname(os.path.abspath(__file__))

if __name__ == "__main__":
    torch.manual_seed(0)
    model = ref_unet.N2V2UNet()
    model.eval()
    torch.set_num_threads(8)
    out = {}
    for name, shape in (("cube32", (1, 1, 32, 32, 32)), ("odd", (1, 1, 33, 32, 35))):
        x = torch.randn(*shape, generator=torch.Generator().manual_seed(1))
        with torch.no_grad():
            out[name] = model(x).numpy()
    np.savez_compressed(os.path.join(HERE, "n2v2.npz"), **out)
    sd = model.state_dict()
    with open(os.path.join(HERE, "n2v2_state.json"), "w") as f:
        json.dump({"abs_sums": {k: float(v.double().abs().sum()) for k, v in sd.items()},
                   "shapes": {k: list(v.shape) for k, v in sd.items()},
                   "n_params": int(sum(p.numel() for p in model.parameters()))}, f, indent=1)
    print("n2v2 fixtures written", {k: v.shape for k, v in out.items()})
