"""sam6d_hip -- host side of the MI355X-native SAM-6D geometric-matching path (libsam6d_hip.so over ctypes)."""
from . import _lib  # noqa: F401


def lib_path():
    return _lib.LIB_PATH


def require_lib():
    """Load libsam6d_hip.so or raise (no fallback path exists)."""
    return _lib.load()
