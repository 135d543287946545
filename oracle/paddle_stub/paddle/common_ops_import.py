from .static import Variable  # noqa: F401
