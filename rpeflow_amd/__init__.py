"""rpeflow_amd -- RPEFlow's data-parallel hot path on MI355X (gfx950).

``rpeflow_amd.csrc`` exports the four operators of the reference's
``models/csrc/__init__.py:1`` with the same signatures; they call hand-written
HIP kernels in ``librpeflow_hip.so`` through a C ABI (``include/rpeflow_hip.h``).
There is no CPU or PyTorch fallback: a missing library or a CPU tensor raises.
"""
__version__ = "0.1.0"
