from ciaosr_amd.nonlocal_attn import CrossScaleAttention  # noqa: F401
