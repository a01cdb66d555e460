"""Launcher for interpreters that skip ``sitecustomize`` (``-S`` / ``-I``) or environments that cannot set PYTHONPATH:

    python <repo>/dropin/run.py adaptive_attack_eval.py --defense Diffusion --t 5 ...

installs the hook, puts the shim packages on the path and runs the reference script unchanged as ``__main__``."""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _audiopure_hook  # noqa: E402

if __name__ == "__main__":
    if len(sys.argv) < 2:
        raise SystemExit("usage: run.py <reference script> [args...]")
    _audiopure_hook.install()
    script = os.path.abspath(sys.argv[1])
    sys.argv = sys.argv[1:]
    sys.path.insert(0, os.path.dirname(script))            # what `python script.py` would have done
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(1, here)                               # shim packages right behind the script's directory
    runpy.run_path(script, run_name="__main__")
