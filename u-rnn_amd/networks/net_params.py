"""Architecture table -> layer specs + cells (mirror of reference net_params.py:5-141, same return shape)."""
from collections import OrderedDict

from .ConvRNN import CGRU_cell


def get_network_params(use_checkpoint, input_height=500, input_width=500, input_channels=63, net_cfg=None):
    if net_cfg is not None:
        enc, dec = net_cfg["encoder"], net_cfg["decoder"]
        enc_conv, enc_gru, down, enc_k = enc["conv_out_channels"], enc["gru_channels"], enc["downsample_factors"], enc["filter_size"]
        dec_gru, dec_conv, up, dec_k = dec["gru_channels"], dec["conv_out_channels"], dec["upsample_factors"], dec["filter_size"]
    else:
        enc_conv, enc_gru, down, enc_k = [16, 64, 96], [64, 96, 96], [1, 2, 2], 1
        dec_gru, dec_conv, up, dec_k = [96, 96, 64], [96, 96, 16], [2, 2, 1], 1
    n = len(enc_gru)
    scales, s = [], 1
    for f in down:
        s *= f
        scales.append(s)
    enc_hw = [(input_height // scales[k], input_width // scales[k]) for k in range(n)]
    dec_hw = [enc_hw[n - 1 - k] for k in range(n)]

    enc_in = [input_channels] + list(enc_gru[:-1])
    encoder_convs = []
    for k in range(n):
        spec = OrderedDict()
        spec[f"conv{k+1}_leaky_1"] = [enc_in[k], enc_conv[k], enc_k, 1, 0]
        if down[k] > 1:
            spec["avgpool"] = [down[k], down[k], 0]
        encoder_convs.append(spec)
    encoder_grus = [CGRU_cell(use_checkpoint, enc_hw[k], enc_conv[k], enc_k, enc_gru[k], "encoder") for k in range(n)]

    dec_in = [enc_gru[n - 1 - k] for k in range(n)]
    decoder_convs = []
    for k in range(n):
        spec = OrderedDict()
        if up[k] > 1:
            spec[f"deconv{k+1}_leaky_1"] = [dec_in[k], dec_conv[k], dec_k + 1, up[k], 0]
        else:
            spec[f"conv{k+1}_leaky_1"] = [dec_in[k], dec_conv[k], dec_k, 1, 0]
        decoder_convs.append(spec)
    gru_in = [dec_conv[0]] + [dec_conv[k - 1] for k in range(1, n)]
    decoder_grus = [CGRU_cell(use_checkpoint, dec_hw[k], gru_in[k], dec_k, dec_gru[k], "decoder") for k in range(n)]
    return [encoder_convs, encoder_grus], [decoder_convs, decoder_grus]
