"""Loader of ``csrc/libhironaka_hip.so`` (the C ABI of ``include/hironaka_hip.h``).

There is NO fallback: if the library is missing or a symbol is absent the import of any
operator raises.  A CPU "just works" path would silently void every parity claim.
"""
import ctypes as C
import os
import subprocess

# torch FIRST: it brings its own libamdhip64; loading ours afterwards resolves the HIP runtime to
# that already-loaded copy, so kernels, streams and device pointers all live in ONE runtime.
# (Loading libhironaka_hip.so first would pull /opt/rocm's runtime in beside torch's and every
# launch on a torch stream would fail.)
import torch  # noqa: F401

from . import _abi as A

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libhironaka_hip.so")
_lib = None


class HironakaHipError(RuntimeError):
    def __init__(self, status: int, where: str = ""):
        self.status = status
        text = A.STATUS_TEXT.get(status, "unknown status")
        super().__init__(f"{where + ': ' if where else ''}libhironaka_hip status {status} ({text})")


def build(force: bool = False) -> str:
    """hipcc --offload-arch=gfx950 the kernels (cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-s", f"-j{min(8, os.cpu_count() or 1)}"]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd)
    return LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hironaka_amd has no CPU fallback by design)")
        handle = C.CDLL(LIB_PATH)
        A.bind(handle)  # AttributeError if a declared symbol is not exported
        got = handle.hk_abi_version()
        if got != A.HK_ABI_VERSION:
            raise ImportError(f"libhironaka_hip ABI {got} != python binding {A.HK_ABI_VERSION}")
        _lib = handle
    return _lib


def check(status: int, where: str = "") -> None:
    if status != A.HK_OK:
        raise HironakaHipError(status, where)
