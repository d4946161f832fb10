"""Load the binding over ANOTHER build of the library (development only): vs = devlib.load()  (tools/dev/libvstab_dev.so, built by `make -C video-annotator_amd dev`).
The package itself always loads lib/libvstab.so; this is the only way around that, and it takes explicit code."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(path=None):
    path = os.path.abspath(path or os.path.join(ROOT, "tools", "dev", "libvstab_dev.so"))
    spec = importlib.util.spec_from_file_location("vstab_devlib", os.path.join(ROOT, "video-annotator_amd", "__init__.py"),
                                                  submodule_search_locations=[os.path.join(ROOT, "video-annotator_amd")])
    mod = importlib.util.module_from_spec(spec)
    mod.__dict__["_VSTAB_LIB_OVERRIDE"] = path
    sys.modules["vstab_devlib"] = mod
    spec.loader.exec_module(mod)
    return mod
