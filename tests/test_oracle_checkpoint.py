"""
CPU: the oracle composed end to end -- joint_oracle (STFT -> features -> LSTM masks -> MVDR ->
log-mel -> Transformer encoder) + encoder_oracle.transformer_decoder -- against the activations the
reference recorded for a checkpoint it wrote itself (tests/golden/checkpoint_enh_xfmr.npz,
make_golden.py gen_checkpoints: `asr@enh_xfmr` through the reference's load_checkpoint).
"""
import torch

from oracle import encoder_oracle as eo
from oracle import joint_oracle as jo
from tests.conftest import assert_close, golden


def test_oracle_reproduces_the_reference_checkpoint_forward():
    g = golden("checkpoint_enh_xfmr")
    conf = g.cfg
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    enh_t, asr_t, nnet = conf["enh_transform"], conf["asr_transform"], conf["nnet_conf"]
    enc_arch = nnet["enc_kwargs"]["arch_kwargs"]
    with torch.no_grad():
        out = jo.joint_forward(sd, g["wav"], g["lens"], frame_len=enh_t["frame_len"],
                               frame_hop=enh_t["frame_hop"], window=enh_t["window"],
                               ipd_index=enh_t["ipd_index"], num_mels=asr_t["num_mels"],
                               rnn_layers=nnet["enh_kwargs"]["num_layers"],
                               enc_layers=nnet["enc_kwargs"]["num_layers"],
                               nhead=enc_arch["nhead"], arch="xfmr", pose="abs",
                               pre_norm=enc_arch.get("pre_norm", False))
        assert out["enc_len"].tolist() == g["enc_len"].tolist()
        assert_close(out["enc_ctc"], g["enc_ctc"], 2e-5, "encoder / CTC branch")
        dec_arch = nnet["dec_kwargs"]["arch_kwargs"]
        dec = eo.transformer_decoder(sd, out["enc_out"], out["enc_len"], g["tgt"], g["tgt_len"],
                                     nnet["dec_kwargs"]["num_layers"], dec_arch["nhead"],
                                     pre_norm=dec_arch.get("pre_norm", False),
                                     prefix="asr.decoder.")
    assert dec.shape == g["dec_out"].shape
    assert_close(dec, g["dec_out"], 2e-5, "decoder output")
