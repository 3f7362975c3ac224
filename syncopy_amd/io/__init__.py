"""`.spy` container ingress / egress: the step before and after the hot path."""
from .spy_container import FILE_EXT, load, save  # noqa: F401
