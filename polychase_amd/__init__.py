"""polychase_amd: the MI355X-native video-analysis path behind polychase_core (see README.md / DESIGN.md)."""
import os as _os

# The engine's streams need hardware queues of their own (include/polychase_hip.h: pc_runtime_init says why).  The library sets
# this default in pc_runtime_init / pc_context_create; Python hosts usually initialise HIP through torch BEFORE that, so the
# package sets it on import as well -- import polychase_amd (or set the variable) before the first torch.cuda call.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
