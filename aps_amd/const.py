"""
The numeric constants of the reference surface that the hot path uses (values as in
aps/const.py): the float32 machine epsilon that floors logs / magnitudes / normalisers, the int16
full scale of the wav reader, and the mask fill values of the attention layers.
"""
import torch as th

_F32 = th.finfo(th.float32)

EPSILON = float(_F32.eps)        # 2**-23 = 1.1920929e-07
MIN_F32 = float(_F32.min)        # key-padding fill of the reference's own attention path
NEG_INF = -float("inf")          # additive attention masks
MAX_INT16 = 2**15 - 1            # 16-bit PCM full scale
IGNORE_ID = -1                   # padding id of target sequences
