from .prodmp import ProDMP, get_mp  # noqa: F401
