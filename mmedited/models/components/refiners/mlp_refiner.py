from ciaosr_amd.mlp import MLPRefiner  # noqa: F401
