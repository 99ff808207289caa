"""loads the product binding (ligero-prover_amd/__init__.py) by path -- the directory name has a hyphen"""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load():
    if "ligero_prover_amd" in sys.modules:
        return sys.modules["ligero_prover_amd"]
    path = os.path.join(ROOT, "ligero-prover_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location("ligero_prover_amd", path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules["ligero_prover_amd"] = mod
    spec.loader.exec_module(mod)
    return mod
