"""Locating / building the in-tree shared libraries.  No fallbacks: a missing library is an error."""
import ctypes
import os
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)


class NativeLibraryError(RuntimeError):
    pass


def _load(name, make_target):
    path = os.path.join(PKG_DIR, name)
    if not os.path.exists(path):
        # build in tree (hipcc cross-compiles gfx950 without a GPU); never silently substitute
        try:
            subprocess.check_call(["make", "-C", ROOT, make_target], stdout=subprocess.DEVNULL)
        except Exception as e:  # pragma: no cover
            raise NativeLibraryError("%s is not built and `make %s` failed: %s" % (name, make_target, e))
    try:
        return ctypes.CDLL(path)
    except OSError as e:
        raise NativeLibraryError("cannot load %s: %s" % (path, e))


def load_rt(variant=None):
    # torch ships its own copy of the HIP runtime (torch/lib/libamdhip64.so, soname libamdhip64.so.7).
    # Two HIP/HSA runtimes cannot share one process, so when torch is installed it is imported FIRST:
    # librt_mi355x.so's DT_NEEDED libamdhip64.so.7 then binds to the copy torch already loaded and the
    # tensors handed to rt_trace_shard live in the same runtime as the kernels that fill them.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    # RT_LIB_VARIANT=<suffix> loads librt_mi355x_<suffix>.so (kernel experiments built next to the product)
    suffix = variant or os.environ.get("RT_LIB_VARIANT")
    if suffix == "alt":
        return _load("librt_mi355x_alt.so", "alt")
    if suffix:
        return ctypes.CDLL(os.path.join(PKG_DIR, "librt_mi355x_%s.so" % suffix))
    return _load("librt_mi355x.so", "vulkan_raytracing_amd/librt_mi355x.so")


def load_host():
    return _load("librt_host.so", "vulkan_raytracing_amd/librt_host.so")
