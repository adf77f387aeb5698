"""Import helper: the package directory is `vision-processor_amd/` (hyphen, per the repo layout contract), which
Python cannot import by name.  `load()` registers it as module `vision_processor_amd`."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "vision-processor_amd")
NAME = "vision_processor_amd"


def load():
    if NAME in sys.modules:
        return sys.modules[NAME]
    spec = importlib.util.spec_from_file_location(NAME, os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[NAME] = mod
    spec.loader.exec_module(mod)
    return mod
