from ciaosr_amd.restorer import CiaoSR  # noqa: F401
