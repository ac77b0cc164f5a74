#!/usr/bin/env python3
"""Dump the REFERENCE's default configuration tree (segmentron/config/settings.py) to
tests/golden/cfg_defaults.json (dev container only): the key set and default values are the
drop-in contract with the reference's yaml files and command lines."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import ref_import  # noqa: E402


def plain(node):
    if isinstance(node, dict):
        return {k: plain(v) for k, v in node.items() if k != "_immutable" and not k.startswith("__")}
    if isinstance(node, tuple):
        return {"__tuple__": [plain(v) for v in node]}
    if isinstance(node, list):
        return [plain(v) for v in node]
    return node


def main():
    ref_import._install_stubs()
    sys.path.insert(0, ref_import.REFERENCE_ROOT)
    for name in [n for n in sys.modules if n == "segmentron" or n.startswith("segmentron.")]:
        del sys.modules[name]
    from segmentron.config import cfg
    out = plain(dict(cfg))
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "cfg_defaults.json"), "w"), indent=1,
              sort_keys=True)
    print("top-level keys:", sorted(out))


if __name__ == "__main__":
    main()
