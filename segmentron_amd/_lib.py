"""ctypes binding of libsegmentron_hip.so (the C-ABI declared in include/segmentron_hip.h).

The header is the single source of truth: prototypes are parsed from it and turned into ctypes
``argtypes``.  There is NO fallback: if the library is missing or an entry point fails, a
``RuntimeError`` is raised — the product never silently routes through PyTorch/CPU ops.
"""
import ctypes
import os
import re

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
# SEGMENTRON_HIP_LIB: load another build of the same C-ABI (kernel A/B experiments)
LIB_PATH = os.environ.get("SEGMENTRON_HIP_LIB") or os.path.join(_PKG, "libsegmentron_hip.so")
HEADER_PATH = os.path.join(_ROOT, "include", "segmentron_hip.h")

# SEG_TRACE_CALLS=1: print every C-ABI launch and synchronise after it (locating a GPU fault)
_TRACE = os.environ.get("SEG_TRACE_CALLS") == "1"

_CTYPES = {"int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float,
           "double": ctypes.c_double}


def parse_header(path=HEADER_PATH):
    """-> {name: (restype, [(ctype, argname), ...])} for every prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"(const\s+char\s*\*|int)\s+(seg_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        restype = ctypes.c_char_p if "char" in ret else ctypes.c_int
        argl = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                if "*" in a:
                    argl.append((ctypes.c_void_p, a.split("*")[-1].strip()))
                else:
                    ty, nm = a.rsplit(" ", 1)
                    argl.append((_CTYPES[ty.replace("const ", "").strip()], nm))
        protos[name] = (restype, argl)
    return protos


class _Lib:
    def __init__(self):
        self._dll = None
        self.protos = parse_header()

    def load(self):
        if self._dll is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    "segmentron_amd: %s not found — build it with `python -c 'import "
                    "__graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). There is "
                    "no CPU/PyTorch fallback." % LIB_PATH)
            dll = ctypes.CDLL(LIB_PATH)
            for name, (restype, args) in self.protos.items():
                fn = getattr(dll, name)  # AttributeError if the header and the .so disagree
                fn.restype = restype
                fn.argtypes = [a[0] for a in args]
            self._dll = dll
        return self._dll

    def call(self, name, *args):
        dll = self.load()
        if _TRACE:  # debugging aid: name + scalar arguments of every launch, synchronised
            import sys
            import torch
            sys.stderr.write("[seg] %s %s\n" % (name, " ".join(
                str(a) for a in args if isinstance(a, (int, float)) and abs(a) < (1 << 31))))
            sys.stderr.flush()
        rc = getattr(dll, name)(*args)
        if rc != 0:
            raise RuntimeError("%s failed (%d): %s" % (name, rc, dll.seg_last_error().decode()))
        if _TRACE:
            torch.cuda.synchronize()

    def query(self, name, *args):
        return getattr(self.load(), name)(*args)


LIB = _Lib()
