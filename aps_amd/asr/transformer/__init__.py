from aps_amd.asr.transformer.encoder import TransformerEncoder

__all__ = ["TransformerEncoder"]
