from ciaosr_amd.restorer import BasicRestorer  # noqa: F401
