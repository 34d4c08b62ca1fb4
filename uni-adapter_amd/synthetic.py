"""Synthetic inputs of the reference's shapes for bench.py / smoke tests (there are no datasets or
checkpoints offline): Uni3D-layout random weights, unit-sphere point clouds, text banks.
Shapes: models/point_encoder.py:129-186, timm eva02_large_patch14_448 (SURVEY section 8 cfg table)."""
import math

import torch


def uni3d_state_dict(seed=0, trans_dim=1024, embed_dim=1024, enc_dim=512, depth=24, hidden=None, in_ch=6,
                     device='cpu', std=0.02):
    """Random-init weights in the reference state-dict key layout (utils/load_models.py:56-59)."""
    g = torch.Generator(device=device).manual_seed(seed)
    C = trans_dim
    H = hidden if hidden is not None else int(trans_dim * 8 / 3)
    P = 'point_encoder.'
    sd = {}
    rn = lambda *s, scale=1.0: torch.randn(*s, generator=g, device=device) * scale

    def lin(name, co, ci):
        sd[P + name + '.weight'] = rn(co, ci, scale=1 / math.sqrt(ci)); sd[P + name + '.bias'] = rn(co, scale=std)

    def ln(name, c):
        sd[P + name + '.weight'] = 1 + rn(c, scale=0.05); sd[P + name + '.bias'] = rn(c, scale=0.05)

    def conv(name, co, ci):
        sd[P + name + '.weight'] = rn(co, ci, 1, scale=1 / math.sqrt(ci)); sd[P + name + '.bias'] = rn(co, scale=std)

    def bn(name, c):
        sd[P + name + '.weight'] = 1 + rn(c, scale=0.1); sd[P + name + '.bias'] = rn(c, scale=0.1)
        sd[P + name + '.running_mean'] = rn(c, scale=0.1)
        sd[P + name + '.running_var'] = 1 + 0.2 * torch.rand(c, generator=g, device=device)

    conv('encoder.first_conv.0', 128, in_ch); bn('encoder.first_conv.1', 128); conv('encoder.first_conv.3', 256, 128)
    conv('encoder.second_conv.0', 512, 512); bn('encoder.second_conv.1', 512); conv('encoder.second_conv.3', enc_dim, 512)
    lin('encoder2trans', C, enc_dim); lin('trans2embed', embed_dim, C)
    sd[P + 'cls_token'] = rn(1, 1, C, scale=std); sd[P + 'cls_pos'] = rn(1, 1, C, scale=0.1)
    lin('pos_embed.0', 128, 3); lin('pos_embed.2', C, 128)
    for i in range(depth):
        b = f'visual.blocks.{i}.'
        ln(b + 'norm1', C); ln(b + 'norm2', C)
        lin(b + 'attn.q_proj', C, C); lin(b + 'attn.v_proj', C, C); lin(b + 'attn.proj', C, C)
        sd[P + b + 'attn.k_proj.weight'] = rn(C, C, scale=1 / math.sqrt(C))
        lin(b + 'mlp.fc1_g', H, C); lin(b + 'mlp.fc1_x', H, C); ln(b + 'mlp.norm', H); lin(b + 'mlp.fc2', C, H)
    ln('visual.fc_norm', C)
    sd['logit_scale'] = torch.tensor(math.log(1 / 0.07), device=device)
    return sd


def ulip_state_dict(seed=0, trans_dim=384, embed_dim=512, enc_dim=256, depth=12, device='cpu', std=0.02):
    """Random-init weights in the ULIP-2 PointBERT key layout (models/ulip/pointbert/point_encoder.py:121-170,
    models/ulip/ulip_model.py:11-13)."""
    g = torch.Generator(device=device).manual_seed(seed)
    C = trans_dim
    P = 'point_encoder.'
    sd = {}
    rn = lambda *s, scale=1.0: torch.randn(*s, generator=g, device=device) * scale

    def lin(name, co, ci, bias=True):
        sd[P + name + '.weight'] = rn(co, ci, scale=1 / math.sqrt(ci))
        if bias:
            sd[P + name + '.bias'] = rn(co, scale=std)

    def ln(name, c):
        sd[P + name + '.weight'] = 1 + rn(c, scale=0.05); sd[P + name + '.bias'] = rn(c, scale=0.05)

    def conv(name, co, ci):
        sd[P + name + '.weight'] = rn(co, ci, 1, scale=1 / math.sqrt(ci)); sd[P + name + '.bias'] = rn(co, scale=std)

    def bn(name, c):
        sd[P + name + '.weight'] = 1 + rn(c, scale=0.1); sd[P + name + '.bias'] = rn(c, scale=0.1)
        sd[P + name + '.running_mean'] = rn(c, scale=0.1)
        sd[P + name + '.running_var'] = 1 + 0.2 * torch.rand(c, generator=g, device=device)

    conv('encoder.first_conv.0', 128, 3); bn('encoder.first_conv.1', 128); conv('encoder.first_conv.3', 256, 128)
    conv('encoder.second_conv.0', 512, 512); bn('encoder.second_conv.1', 512); conv('encoder.second_conv.3', enc_dim, 512)
    lin('reduce_dim', C, enc_dim)
    sd[P + 'cls_token'] = rn(1, 1, C, scale=std); sd[P + 'cls_pos'] = rn(1, 1, C, scale=0.1)
    lin('pos_embed.0', 128, 3); lin('pos_embed.2', C, 128)
    for i in range(depth):
        b = f'blocks.blocks.{i}.'
        ln(b + 'norm1', C); ln(b + 'norm2', C)
        lin(b + 'attn.qkv', 3 * C, C, bias=False); lin(b + 'attn.proj', C, C)
        lin(b + 'mlp.fc1', 4 * C, C); lin(b + 'mlp.fc2', C, 4 * C)
    ln('norm', C)
    sd['pc_projection'] = rn(2 * C, embed_dim, scale=1 / math.sqrt(2 * C))
    sd['logit_scale'] = torch.tensor(math.log(1 / 0.07), device=device)
    return sd


def openshape_state_dict(seed=0, dim=512, depth=12, heads=8, mlp_dim=1536, sa_dim=256, in_dim=6, out_dim=1280, device='cpu'):
    """Random-init weights in the key layout of OpenShape's `Projected(PointPatchTransformer, nn.Linear)`
    (models/openshape/ppta.py:83-147; 'scaling 4' = PointBERT-ViT-G: 384 patches, radius 0.2, 64 samples)."""
    g = torch.Generator(device=device).manual_seed(seed)
    rn = lambda *s, scale=0.02: torch.randn(*s, generator=g, device=device) * scale
    sd = {}
    last = in_dim + 3
    for i, c in enumerate([64, 64, sa_dim]):
        sd[f'ppat.sa.mlp_convs.{i}.weight'] = rn(c, last, 1, 1, scale=1 / math.sqrt(last)); sd[f'ppat.sa.mlp_convs.{i}.bias'] = rn(c)
        sd[f'ppat.sa.mlp_bns.{i}.weight'] = 1 + rn(c, scale=0.1); sd[f'ppat.sa.mlp_bns.{i}.bias'] = rn(c, scale=0.1)
        sd[f'ppat.sa.mlp_bns.{i}.running_mean'] = rn(c, scale=0.1)
        sd[f'ppat.sa.mlp_bns.{i}.running_var'] = 1 + 0.2 * torch.rand(c, generator=g, device=device)
        last = c
    sd['ppat.lift.0.weight'] = rn(dim, sa_dim + 3, 1, scale=1 / math.sqrt(sa_dim + 3)); sd['ppat.lift.0.bias'] = rn(dim)
    sd['ppat.lift.2.weight'] = 1 + rn(dim, scale=0.05); sd['ppat.lift.2.bias'] = rn(dim, scale=0.05)
    sd['ppat.cls_token'] = rn(dim, scale=1.0)
    inner = heads * 64
    for l in range(depth):
        P = f'ppat.transformer.layers.{l}.'
        for j in (0, 1):
            sd[P + f'{j}.norm.weight'] = 1 + rn(dim, scale=0.05); sd[P + f'{j}.norm.bias'] = rn(dim, scale=0.05)
        sd[P + '0.fn.to_qkv.weight'] = rn(3 * inner, dim, scale=1 / math.sqrt(dim))
        sd[P + '0.fn.to_out.0.weight'] = rn(dim, inner, scale=1 / math.sqrt(inner)); sd[P + '0.fn.to_out.0.bias'] = rn(dim)
        sd[P + '1.fn.net.0.weight'] = rn(mlp_dim, dim, scale=1 / math.sqrt(dim)); sd[P + '1.fn.net.0.bias'] = rn(mlp_dim)
        sd[P + '1.fn.net.3.weight'] = rn(dim, mlp_dim, scale=1 / math.sqrt(mlp_dim)); sd[P + '1.fn.net.3.bias'] = rn(dim)
    sd['proj.weight'] = rn(out_dim, dim, scale=1 / math.sqrt(dim)); sd['proj.bias'] = rn(out_dim)
    return sd


def point_clouds(n, n_points=1024, seed=42, device='cpu'):
    """randn clouds scaled into the unit sphere, rgb = ones (data/tta_datasets.py:127)."""
    g = torch.Generator(device=device).manual_seed(seed)
    xyz = torch.randn(n, n_points, 3, generator=g, device=device)
    xyz = xyz / xyz.norm(dim=-1).amax(dim=1)[:, None, None]
    return xyz, torch.ones(n, n_points, 3, device=device)


def text_bank(K, D, seed=0, device='cpu'):
    g = torch.Generator(device=device).manual_seed(seed)
    return torch.nn.functional.normalize(torch.randn(K, D, generator=g, device=device), dim=1)
