"""Import shim: ``from bm4d import bm4d`` resolves to the MI355X implementation, so the
reference's call sites (machine_learning/data_handling.py:12, evaluate.py:11) need no edit when
this directory is on ``sys.path`` in place of the third-party wheel."""
from aind_exaspim_image_compression.bm4d import BM4DProfile, bm4d  # noqa: F401

__all__ = ["bm4d", "BM4DProfile"]
