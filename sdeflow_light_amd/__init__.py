"""sdeflow_light_amd — MI355X-native hot path of sdeflow-light / MSGM.

Score-matching training step and reverse-SDE sampling on hand-written gfx950
HIP kernels behind a C ABI (``libmsgm_hip.so``, ``include/msgm_hip.h``).
The Python modules mirror the reference's file and class names
(``NN``, ``NNUnet1D``, ``NNUnet``, ``SDEs``, ``sde_scheme``) so a driver can
switch by changing its imports.  There is no CPU fallback: the kernels need
the built extension and a GPU and fail loudly otherwise.
"""
__version__ = "0.1.0"
