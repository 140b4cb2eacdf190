"""audiocodec_amd -- MI355X-native MDCT + psychoacoustic-masking hot path of korneelvdbroek/audiocodec.

Python host classes with the reference's names and signatures over hand-written HIP kernels
(``libaudiocodec_amd.so``, C ABI in ``include/audiocodec_amd.h``).  No CPU fallback.
"""

from .mdctransformer import MDCTransformer
from .psychoacoustic import PsychoacousticModel
from .codec import AudioCodec, StreamingMDCT
from .workspace import Workspace

__all__ = ["MDCTransformer", "PsychoacousticModel", "AudioCodec", "StreamingMDCT", "Workspace"]
__version__ = "0.1.0"
