"""state_dict shapes of the two U-Nets (test infrastructure, like the rest of ``oracle/``): the keys and shapes
the reference modules create (NNUnet1D.py:53-107; model/unet.py:338-446, NNUnet.py:88-94), used with
``oracle.det_params.det_state_dict`` to build identical deterministic parameters on the oracle and the HIP side."""
from __future__ import annotations

from . import nets_ref as N


def unet1d_shapes(L, pre, base=32, mults=(1, 2, 4), emb=128):
    """state_dict shapes of UNet1D (NNUnet1D.py:53-107)."""
    sh = {"time_mlp.0.weight": (emb, 1), "time_mlp.0.bias": (emb,), "time_mlp.2.weight": (emb, emb), "time_mlp.2.bias": (emb,)}
    if pre:
        sh.update({"scale_embed.0.weight": (emb, 1), "scale_embed.0.bias": (emb,), "scale_embed.2.weight": (emb, emb), "scale_embed.2.bias": (emb,)})
    chs = [base * m for m in mults]
    cin = 1
    for i, c in enumerate(chs):
        sh.update({f"enc_blocks.{i}.net.0.weight": (c, cin + emb, 3), f"enc_blocks.{i}.net.0.bias": (c,),
                   f"enc_blocks.{i}.net.2.weight": (c, c, 3), f"enc_blocks.{i}.net.2.bias": (c,),
                   f"downs.{i}.weight": (c, c, 4), f"downs.{i}.bias": (c,)})
        cin = c
    sh.update({"middle.net.0.weight": (cin, cin + emb, 3), "middle.net.0.bias": (cin,),
               "middle.net.2.weight": (cin, cin, 3), "middle.net.2.bias": (cin,)})
    for i, c in enumerate(reversed(chs)):
        sh.update({f"up_convs.{i}.weight": (cin, c, 4), f"up_convs.{i}.bias": (c,),
                   f"dec_blocks.{i}.net.0.weight": (c, 2 * c + emb, 3), f"dec_blocks.{i}.net.0.bias": (c,),
                   f"dec_blocks.{i}.net.2.weight": (c, c, 3), f"dec_blocks.{i}.net.2.bias": (c,)})
        cin = c
    sh.update({"final.weight": (1, cin, 1), "final.bias": (1,)})
    return sh


def unet2d_shapes(cfg: N.UNet2DConfig, prefix=""):
    """state_dict shapes of UNetModel(+LogNorm) (model/unet.py:338-446)."""
    mc, ted = cfg.model_channels, cfg.model_channels * 4
    sh = {}

    def lin(k, o, i):
        sh[k + ".weight"], sh[k + ".bias"] = (o, i), (o,)

    def conv(k, o, i, ks):
        sh[k + ".weight"], sh[k + ".bias"] = (o, i, ks, ks), (o,)

    def res(k, cin, cout):
        sh[k + ".in_layers.0.weight"], sh[k + ".in_layers.0.bias"] = (cin,), (cin,)
        conv(k + ".in_layers.2", cout, cin, 3)
        lin(k + ".emb_layers.1", cout, ted)
        sh[k + ".out_layers.0.weight"], sh[k + ".out_layers.0.bias"] = (cout,), (cout,)
        conv(k + ".out_layers.3", cout, cout, 3)
        if cin != cout:
            conv(k + ".skip_connection", cout, cin, 1)

    def attn(k, c):
        sh[k + ".norm.weight"], sh[k + ".norm.bias"] = (c,), (c,)
        sh[k + ".qkv.weight"], sh[k + ".qkv.bias"] = (3 * c, c, 1), (3 * c,)
        sh[k + ".proj_out.weight"], sh[k + ".proj_out.bias"] = (c, c, 1), (c,)

    lin("time_embed.0", ted, mc); lin("time_embed.2", ted, ted)
    if cfg.use_log_norm:
        lin("scale_embed.0", ted, mc); lin("scale_embed.2", ted, ted)
    inp, mid, out = N.unet2d_plan(cfg)
    ch = mc * cfg.channel_mult[0]
    conv("input_blocks.0.0", ch, cfg.in_channels, 3)
    chans = [ch]
    level = 0
    bi = 1
    for level, mult in enumerate(cfg.channel_mult):
        for _ in range(cfg.num_res_blocks):
            kinds = inp[bi]
            res(f"input_blocks.{bi}.0", ch, mult * mc); ch = mult * mc
            if "attn" in kinds:
                attn(f"input_blocks.{bi}.1", ch)
            chans.append(ch); bi += 1
        if level != len(cfg.channel_mult) - 1:
            conv(f"input_blocks.{bi}.0.op", ch, ch, 3); chans.append(ch); bi += 1
    res("middle_block.0", ch, ch); attn("middle_block.1", ch); res("middle_block.2", ch, ch)
    bi = 0
    for level, mult in list(enumerate(cfg.channel_mult))[::-1]:
        for i in range(cfg.num_res_blocks + 1):
            kinds = out[bi]
            res(f"output_blocks.{bi}.0", ch + chans.pop(), mc * mult); ch = mc * mult
            for j, kd in enumerate(kinds):
                if kd == "attn":
                    attn(f"output_blocks.{bi}.{j}", ch)
                if kd == "up":
                    conv(f"output_blocks.{bi}.{j}.conv", ch, ch, 3)
            bi += 1
    sh["out.0.weight"], sh["out.0.bias"] = (ch,), (ch,)
    conv("out.2", cfg.out_channels, mc * cfg.channel_mult[0], 3)
    return {prefix + k: v for k, v in sh.items()}
