"""polychase_amd: the MI355X-native video-analysis path behind polychase_core (see README.md / DESIGN.md)."""
import os as _os

# The engine's streams need hardware queues of their own (csrc/hip/api.hip: pc_runtime_defaults says why).  The library sets this
# default when it is loaded; Python hosts usually initialise HIP through torch BEFORE the library is loaded, so the package sets
# it on import as well -- import polychase_amd (or set the variable) before the first torch.cuda call.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
