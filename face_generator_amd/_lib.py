"""ctypes binding of libfacegen_hip.so, generated from include/facegen_hip.h (the single source of truth).

There is NO fallback: if the shared library is missing the import of any compute entry fails loudly.
"""
import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(HERE), "include", "facegen_hip.h")
_LIB = None

FG_BWD_PARAM_GRADS = 1
FG_BWD_INPUT_GRAD = 2

LAYER_TYPES = dict(LINEAR=1, VIEW=2, PRELU=3, UPSAMPLE2X=4, CONV=5, BATCHNORM=6, SPATIAL_DROPOUT=7, AVGPOOL2=8,
                   DROPOUT=9, SIGMOID=10, LEAKYRELU=11, MAXPOOL2=12)


class FgError(RuntimeError):
    pass


class LayerSpec(ctypes.Structure):
    _fields_ = [("type", ctypes.c_int), ("a", ctypes.c_int), ("b", ctypes.c_int), ("c", ctypes.c_int),
                ("d", ctypes.c_int), ("p", ctypes.c_float), ("q", ctypes.c_float)]


_CTYPES = [
    (r"^const fg_layer_spec\*$", ctypes.POINTER(LayerSpec)),
    (r"^const float\* const\*$", ctypes.POINTER(ctypes.c_void_p)),
    (r"^(fg_ctx|fg_net|fg_comm|fg_gan|void|float)\*\*$", ctypes.POINTER(ctypes.c_void_p)),
    (r"^const char\*$", ctypes.c_char_p),
    (r"^char\*$", ctypes.c_char_p),
    (r"^long long\*$", ctypes.POINTER(ctypes.c_longlong)),
    (r"^(const )?double\*$", ctypes.c_void_p),
    (r"^(const )?(fg_ctx|fg_net|fg_comm|fg_gan|void|float|int)\*$", ctypes.c_void_p),
    (r"^int$", ctypes.c_int),
    (r"^long long$", ctypes.c_longlong),
    (r"^size_t$", ctypes.c_size_t),
    (r"^float$", ctypes.c_float),
    (r"^double$", ctypes.c_double),
    (r"^uint64_t$", ctypes.c_uint64),
]


def _ctype(t):
    t = re.sub(r"\s+", " ", t.strip()).replace(" *", "*")
    for pat, ct in _CTYPES:
        if re.match(pat, t):
            return ct
    raise FgError("facegen_hip.h: cannot map C type %r" % t)


def parse_header(path=HEADER):
    """-> {name: (restype, [argtypes], [argnames])} for every function declared in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    decls = {}
    for m in re.finditer(r"(?:^|\n)\s*((?:const\s+)?[A-Za-z_][\w ]*?[\s\*]+)(fg_\w+)\s*\(([^;{]*?)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        args = args.strip()
        argtypes, argnames = [], []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                mm = re.match(r"^(.*?)(\w+)$", a)
                argtypes.append(_ctype(mm.group(1)))
                argnames.append(mm.group(2))
        decls[name] = (_ctype(ret), argtypes, argnames)
    return decls


def lib_path():
    return os.environ.get("FACEGEN_HIP_LIB", os.path.join(HERE, "libfacegen_hip.so"))


def load_library():
    """Load libfacegen_hip.so and attach signatures.  Raises FgError if it is not built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise FgError("libfacegen_hip.so not found at %s -- run `python -m face_generator_amd.build` "
                      "(there is no CPU fallback)" % path)
    # Share ONE HIP runtime with torch: torch ships its own libamdhip64.so.7 / libhsa-runtime64 and must be loaded
    # first so that libfacegen_hip.so binds to the same instance (device pointers, streams and the KFD handle are
    # per-runtime; a second runtime in the process sees no device).  A Lua host has no torch and simply uses /opt/rocm.
    import torch  # noqa: F401
    lib = ctypes.CDLL(path)
    for name, (ret, argtypes, _) in parse_header().items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise FgError("libfacegen_hip.so does not export %s declared in facegen_hip.h" % name)
        fn.restype = ret
        fn.argtypes = argtypes
    _LIB = lib
    return lib
