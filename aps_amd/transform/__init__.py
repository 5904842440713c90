from aps_amd.transform.asr import FeatureTransform as AsrTransform
from aps_amd.transform.enh import FeatureTransform as EnhTransform
from aps_amd.transform.spatial import DfTransform, FixedBeamformer
from aps_amd.transform.utils import STFT, iSTFT, forward_stft, inverse_stft

__all__ = ["AsrTransform", "EnhTransform", "FixedBeamformer", "DfTransform", "STFT", "iSTFT", "forward_stft", "inverse_stft"]
