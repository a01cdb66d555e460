from .model import KWSModel  # noqa: F401
