"""rapidcfd-dev_amd -- MI355X-native lduMatrix / fvMatrix compute engine.

The directory name carries a hyphen (it is the reference's name), so import it
through ``__graft_entry__.load_package()`` which registers it as
``rapidcfd_dev_amd``.  Only the hot path of SimFlowCFD/RapidCFD-dev lives here
(SURVEY.md section 8): ``csrc/`` (HIP kernels + the C ABI of include/mi_ldu.h),
``engine.py`` (ctypes binding), ``synthetic.py`` (hex-box inputs),
``parallel.py`` (one-rank-per-GPU domain decomposition driver).
"""
from . import synthetic  # noqa: F401
from . import engine  # noqa: F401
