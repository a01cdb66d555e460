"""Imported by the interpreter at start-up when ``dropin/`` is on PYTHONPATH: installs the top-level-module hook
(``_audiopure_hook``), so ``PYTHONPATH=<repo>/dropin:<repo> python adaptive_attack_eval.py ...`` needs no other change.
A site-wide ``sitecustomize`` that this one shadows is chained to."""
import importlib.util
import os
import sys

import _audiopure_hook

_audiopure_hook.install()

_here = os.path.dirname(os.path.abspath(__file__))
for _p in sys.path:                             # chain to the next sitecustomize on the path, if any
    _f = os.path.join(_p or ".", "sitecustomize.py")
    if os.path.isfile(_f) and os.path.abspath(os.path.dirname(_f)) != _here:
        _spec = importlib.util.spec_from_file_location("_chained_sitecustomize", _f)
        try:
            _spec.loader.exec_module(importlib.util.module_from_spec(_spec))
        except Exception:                       # a broken site hook must not take the interpreter down
            pass
        break
