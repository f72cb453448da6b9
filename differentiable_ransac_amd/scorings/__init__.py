from .msac_score import MSACScore  # noqa: F401
