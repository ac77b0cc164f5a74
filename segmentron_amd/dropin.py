"""Drop-in overlay: make `import segmentron` serve the MI355X hot path AND everything else of the
reference checkout (solver, data pipeline, utils, CLI options) side by side.

The reference's entry points (tools/train.py:17-29, tools/eval.py:17-24, tools/demo.py:9-15)
import `segmentron.{config, models.model_zoo}` — the hot path, implemented here — and
`segmentron.{solver.*, utils.*, data.dataloader}` — Python plumbing this project deliberately
does not re-implement (SURVEY.md §2 OUT).  `install()` therefore
  1. aliases every module of `segmentron_amd` that mirrors a reference module under the
     `segmentron.` name (`segmentron.config is segmentron_amd.config`, ...), and
  2. puts a meta-path finder in front that resolves any OTHER `segmentron.*` module from the
     reference checkout's own files (found through $SEGMENTRON_REFERENCE_ROOT, sys.path, or the
     directory of the running tools/ script), executed under its normal package name so its
     relative imports (`from ..config import cfg`) bind to the aliases of (1).
The reference tree is optional: without it only the hot-path modules exist (and importing e.g.
`segmentron.solver` raises ModuleNotFoundError that says why).  Nothing of the reference is copied.
"""
import importlib
import importlib.abc
import importlib.machinery
import importlib.util
import os
import pkgutil
import sys

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG_DIR)
SHIM_DIR = os.path.join(_PKG_DIR, "shims")

# modules of segmentron_amd with no reference counterpart (never aliased)
_PRIVATE = {"_lib", "hip_ops", "functional", "parallel", "dropin", "shims", "torch_ops", "graph",
            "csrc"}
# reference packages that replace our minimal stand-ins when the reference tree is available:
# segmentron_amd.data.dataloader only carries the NUM_CLASS table the model constructors read
_PREFER_REFERENCE = ("data",)


def _is_reference_root(d):
    return (os.path.isfile(os.path.join(d, "segmentron", "__init__.py"))
            and os.path.isdir(os.path.join(d, "segmentron", "solver"))
            and os.path.realpath(d) != os.path.realpath(_ROOT))


def find_reference_root():
    """Directory that CONTAINS the reference's `segmentron/` package, or None."""
    cands = []
    env = os.environ.get("SEGMENTRON_REFERENCE_ROOT")
    if env:
        if not _is_reference_root(env):
            raise ImportError("SEGMENTRON_REFERENCE_ROOT=%r does not contain the reference's "
                              "segmentron/ package" % env)
        return os.path.abspath(env)
    main = sys.argv[0] if sys.argv and sys.argv[0] else ""
    if main and os.path.isfile(main):  # <root>/tools/train.py -> <root>
        cands.append(os.path.dirname(os.path.dirname(os.path.abspath(main))))
    cands.extend(p or os.getcwd() for p in sys.path)
    for d in cands:
        if os.path.isdir(d) and _is_reference_root(d):
            return os.path.abspath(d)
    return None


class _ReferenceFinder(importlib.abc.MetaPathFinder):
    """Resolves `segmentron.<x>` that segmentron_amd does not implement from the reference tree."""

    def __init__(self):
        self._root, self._seen = None, None

    @property
    def root(self):
        if self._root is None and self._seen != tuple(sys.path):  # re-scan when sys.path grew
            self._seen = tuple(sys.path)
            self._root = find_reference_root()
        return self._root

    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith("segmentron."):
            return None
        root = self.root
        if root is None:
            raise ModuleNotFoundError(
                "%s is not part of the MI355X hot path (segmentron_amd) and no reference "
                "SegmenTron checkout was found to serve it from: put the checkout on sys.path "
                "behind this repository or set SEGMENTRON_REFERENCE_ROOT" % fullname,
                name=fullname)
        parts = fullname.split(".")
        search = os.path.join(root, *parts[:-1])
        return importlib.machinery.PathFinder.find_spec(fullname, [search])


_FINDER = _ReferenceFinder()


def _alias_names():
    """Relative names of every segmentron_amd module that mirrors a reference module."""
    import segmentron_amd
    out = []
    pre = "segmentron_amd."
    for m in pkgutil.walk_packages(segmentron_amd.__path__, prefix=pre,
                                   onerror=lambda name: None):
        rel = m.name[len(pre):]
        top = rel.split(".")[0]
        if top in _PRIVATE or top.startswith("lib"):  # libsegmentron_hip.so is not a module
            continue
        out.append(rel)
    return out


def install_shims():
    """torchvision / thop are imported by the reference's data pipeline and tools/ scripts
     but are not in the ROCm image: minimal stand-ins for exactly the names the
    reference uses live in segmentron_amd/shims and are put at the END of sys.path, so a real
    installation always wins."""
    missing = [n for n in ("torchvision", "thop") if importlib.util.find_spec(n) is None]
    if missing and SHIM_DIR not in sys.path:
        sys.path.append(SHIM_DIR)
    return missing


def install(package):
    """Called by segmentron/__init__.py with the `segmentron` module object."""
    ref = _FINDER.root
    aliased = []
    for rel in _alias_names():
        if ref is not None and rel.split(".")[0] in _PREFER_REFERENCE:
            continue
        mod = importlib.import_module("segmentron_amd." + rel)
        sys.modules["segmentron." + rel] = mod
        aliased.append(rel)
        if "." not in rel:
            setattr(package, rel, mod)
    if _FINDER not in sys.meta_path:
        sys.meta_path.insert(0, _FINDER)
    install_shims()
    if ref is not None:
        # package-level names of the reference's segmentron/utils/__init__.py:4-5, served lazily
        utils = sys.modules["segmentron.utils"]

        def _utils_getattr(name, _lazy={"download": "download", "check_sha1": "download",
                                        "makedirs": "filesystem"}):
            if name in _lazy:
                return getattr(importlib.import_module("segmentron.utils." + _lazy[name]), name)
            raise AttributeError("module 'segmentron.utils' has no attribute %r" % name)
        utils.__getattr__ = _utils_getattr
    package.__reference_root__ = ref
    package.__aliased__ = aliased
    return ref
