from .build import SGD, build_optimizer  # noqa: F401
