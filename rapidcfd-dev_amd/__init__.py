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


def __getattr__(name):  # ``pkg.parallel`` pulls in torch.distributed: imported on first use
    if name == "parallel":
        from importlib import import_module
        return import_module(__name__ + ".parallel")
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
