"""Drop-in package: reference import paths -> audiopure_amd.

Only the hot-path modules are replaced; every other module of the reference's same-named package must stay importable,
so the package path is extended over all same-named directories on ``sys.path`` (this directory first)."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
