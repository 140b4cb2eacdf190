from audiocodec_amd.psychoacoustic import PsychoacousticModel  # noqa: F401
