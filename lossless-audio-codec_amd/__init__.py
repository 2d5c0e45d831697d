"""MI355X-native LAC block-encode path: HIP kernels + C ABI (csrc/, liblacx.so), ctypes mirror of the
reference's encoder classes (lacx.py) and the integer-only synthetic PCM generator (synth.py).

The directory name is not a Python identifier; load it with `__graft_entry__.load_pkg()`, which
registers it as the package `lac_amd`.
"""
from . import lacx, synth  # noqa: F401
