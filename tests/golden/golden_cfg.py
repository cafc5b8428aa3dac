"""Model configurations shared by the golden generator and the tests (plain dicts, our own).

UNET_FULL / VQ_FULL restate configs/frido/layout2i/frido_f8f4_coco_seg.yaml:21-77 of the
reference; the *_SMALL variants are reduced-width versions of the same architecture so that CPU
parity checks finish in seconds.  *_SMALL3 is the 3-scale variant of BASELINE config 5.
"""
import copy

from frido_amd.configs import (UNET_F8F4 as UNET_FULL, VQ_F8F4 as VQ_FULL, BERT_FULL, frido_cfg,  # noqa: E402,F401
                               UNET_F16F8, VQ_F16F8, UNET_512, VQ_512)

UNET_SMALL = dict(
    use_split_head=True, split_embed_dim_list=[3, 3], use_SPADE_norm=True, image_size=16,
    in_channels=6, out_channels=6, model_channels=32, attention_resolutions=[4, 2],
    num_res_blocks=1, channel_mult=[1, 2, 3], num_head_channels=32,
    use_spatial_transformer=True, transformer_depth=1, context_dim=64, num_stage=2)

# transformer_depth = 2 (attention.py:274-277: two BasicTransformerBlocks per SpatialTransformer) -- no shipped config uses it; r05 fixture
UNET_SMALL_D2 = dict(UNET_SMALL, transformer_depth=2)

UNET_SMALL3 = dict(
    use_split_head=True, split_embed_dim_list=[3, 3, 3], use_SPADE_norm=True, image_size=16,
    in_channels=9, out_channels=9, model_channels=32, attention_resolutions=[2],
    num_res_blocks=1, channel_mult=[1, 2], num_head_channels=32,
    use_spatial_transformer=True, transformer_depth=1, context_dim=64, num_stage=3)

VQ_SMALL = dict(
    embed_dim=[3, 3], n_embed=[64, 96],
    edconfig=dict(multiscale=2, double_z=False, z_channels=[3, 3], resolution=64, in_channels=3,
                  out_ch=3, ch=32, ch_mult=[1, 1, 2, 2], num_res_blocks=1, attn_resolutions=[16],
                  dropout=0.0),
    ddconfig=dict(double_z=False, z_channels=6, resolution=64, in_channels=3, out_ch=3, ch=32,
                  ch_mult=[1, 2, 2], num_res_blocks=1, attn_resolutions=[16], dropout=0.0))

VQ_SMALL3 = dict(
    embed_dim=[3, 3, 3], n_embed=[64, 64, 64],
    edconfig=dict(multiscale=3, double_z=False, z_channels=[3, 3, 3], resolution=64, in_channels=3,
                  out_ch=3, ch=32, ch_mult=[1, 1, 2, 2, 2], num_res_blocks=1, attn_resolutions=[16],
                  dropout=0.0),
    ddconfig=dict(double_z=False, z_channels=9, resolution=64, in_channels=3, out_ch=3, ch=32,
                  ch_mult=[1, 2, 2], num_res_blocks=1, attn_resolutions=[16], dropout=0.0))

BERT_SMALL = dict(n_embed=64, n_layer=2, vocab_size=128, max_seq_len=16, use_tokenizer=False)
