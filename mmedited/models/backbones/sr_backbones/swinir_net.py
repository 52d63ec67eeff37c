from ciaosr_amd.encoders.swinir import SwinIR  # noqa: F401
