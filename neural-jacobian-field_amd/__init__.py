"""MI355X-native volumetric-rendering hot path for Neural Jacobian Fields.

Only what the path needs lives here: ``csrc/`` (HIP kernels + the C ABI declared in
``include/njf_hip.h``), ``hip.py`` (ctypes binding, fails loudly without the library),
``packing.py`` (weight/feature-map layouts for the fused kernels) and the host-side mirror of
the reference interface (``model.py``, ``decoder.py``, ``ray_samplers.py``, ``geometry.py``).
"""

__version__ = "0.1.0"
