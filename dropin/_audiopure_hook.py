"""Import hook for the two TOP-LEVEL modules of the reference that ``dropin/`` replaces.

``python adaptive_attack_eval.py`` puts the script's directory -- the reference checkout -- first on ``sys.path``, ahead
of PYTHONPATH, so the checkout's own ``acoustic_system.py`` (and ``M5Net.py``, once ``create_model`` has inserted
``./audio_models/M5``) would shadow the same-named files here.  The packages need no help: the reference's
``diffusion_models`` / ``audio_models`` / ``robustness_eval`` directories have no ``__init__.py`` (namespace portions), and
a regular package found later on the path takes precedence over them.  For the two plain modules a meta-path finder,
installed by ``dropin/sitecustomize.py`` (automatic with ``PYTHONPATH=.../dropin``) or by ``dropin/run.py``, answers first.
"""
import importlib.abc
import importlib.util
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
TOP_LEVEL = {"acoustic_system": "acoustic_system.py", "M5Net": "M5Net.py"}


class _Finder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        rel = TOP_LEVEL.get(fullname)
        if rel is None:
            return None
        return importlib.util.spec_from_file_location(fullname, os.path.join(HERE, rel))


def install():
    root = os.path.dirname(HERE)
    for p in (root, HERE):                      # audiopure_amd itself, and the shim packages
        if p not in sys.path:
            sys.path.append(p)
    if not any(isinstance(f, _Finder) for f in sys.meta_path):
        sys.meta_path.insert(0, _Finder())
