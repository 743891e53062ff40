"""MI355X-native kernels of the SuNeRF ray-march renderer (binding layer; see include/sunerf_hip.h)."""
from .lib import LIB_PATH, EXPORTED_SYMBOLS, SunerfHipError, load  # noqa: F401
