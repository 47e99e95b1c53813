"""MI355X-native volumetric-rendering hot path for Neural Jacobian Fields.

Only what the path needs lives here: ``csrc/`` (HIP kernels + the C ABI declared in ``include/njf_hip.h``),
``hip.py`` (ctypes binding, fails loudly without the library; weight and feature-map layouts are produced by the
library's own ``njf_pack_*`` / ``njf_project_*`` entry points) and the host-side mirror of the reference interface:
``model.py``, ``decoder.py``, ``encoder.py``, ``ray_samplers.py``, ``geometry.py``, ``config.py``, ``model_wrapper.py``,
``inference/`` (the reference's names and state-dict keys), plus what the widening steps of SURVEY.md 8f added --
``training.py`` (backward passes), ``inverse_dynamics.py``, ``visualization.py``, ``parallel.py`` (ray sharding over RCCL),
``renderer.py`` (a facade over ``Model`` for callers that hold a feature map) and ``synthetic.py`` (seeded benchmark inputs).
"""

__version__ = "0.1.0"
