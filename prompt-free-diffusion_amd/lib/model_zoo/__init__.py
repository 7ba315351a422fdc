from .common.get_model import get_model, register  # noqa: F401
