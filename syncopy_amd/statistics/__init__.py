from .summary_stats import mean  # noqa: F401
