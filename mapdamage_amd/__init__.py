"""mapdamage_amd — MI355X-native per-read damage tabulation (the mapDamage hot path).

Only what the path needs lives here: the SoA batch model (``batch``), the synthetic
workloads (``synth``), the ctypes binding of the C-ABI library built from ``csrc/``
(``engine``) and the host-side mirror of the reference's accumulator interface
(``statistics``).  Importing the package does not import torch or load the HIP library.
"""

__version__ = "0.1.0"
