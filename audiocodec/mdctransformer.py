from audiocodec_amd.mdctransformer import MDCTransformer  # noqa: F401
