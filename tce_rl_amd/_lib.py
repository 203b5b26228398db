"""ctypes binding of libtce_hip.so (the C-ABI drop-in boundary, include/tce_hip.h).

There is no CPU fallback: if the library is missing, loading raises.
"""
import ctypes
import os
import re

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
# (TCE_HIP_LIB: another build of the same library -- scripts/mlpw_variant.py -- for A / B runs)
LIB_PATH = os.environ.get("TCE_HIP_LIB") or os.path.join(_PKG, "libtce_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_PKG), "include", "tce_hip.h")

_lib = None

_CTYPES = {
    "int": ctypes.c_int, "int64_t": ctypes.c_int64, "uint64_t": ctypes.c_uint64,
    "float": ctypes.c_float,
    "double": ctypes.c_double, "void": None,
}


def _parse_header(path):
    """Prototype table {name: (restype, [argtypes])} from the C header, so the
    Python side can never drift from include/tce_hip.h."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    protos = {}
    for m in re.finditer(r"([\w\s\*]+?)\b(tce_\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if "*" in ret:
            restype = ctypes.c_char_p if "char" in ret else ctypes.c_void_p
        else:
            restype = _CTYPES[ret.replace("const", "").strip()]
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    ty = a.replace("const", "").split()[0]
                    argtypes.append(_CTYPES[ty])
        protos[name] = (restype, argtypes)
    return protos


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "tce_rl_amd: %s is missing. Build it with "
            "`python -m tce_rl_amd.build` (needs hipcc); there is no CPU "
            "fallback for the hot path." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in _parse_header(HEADER_PATH).items():
        fn = getattr(lib, name)          # AttributeError if not exported
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise RuntimeError("%s failed (%d): %s" % (
            name, rc, lib.tce_last_error().decode()))


def ptr(t):
    return None if t is None else t.data_ptr()


# torch.cuda.current_stream() builds a Stream object through three layers of
# device-index helpers: 8 us per call, paid by every C call (400 + 200 per step on
# the sharded black-box path, which is host-bound).  The raw handle of the same
# stream comes from two C calls.
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def stream():
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def sfx(dtype):
    if dtype == torch.float32:
        return "f32"
    if dtype == torch.float64:
        return "f64"
    raise NotImplementedError("dtype %s (only float32/float64, like the "
                              "reference: util_data_structure.py:70-75)" % dtype)


def check_dev(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "tce_rl_amd ops run on a HIP device only (got a %s tensor); "
                "there is no CPU fallback" % t.device)
