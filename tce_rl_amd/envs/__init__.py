from .synthetic import SyntheticTCEEnv, SyntheticBBEnv, make_env  # noqa: F401
