"""Model configurations as plain dicts.

UNET_F8F4 / VQ_F8F4 restate configs/frido/layout2i/frido_f8f4_coco_seg.yaml:21-77 of the reference (BASELINE.json
configs 1, 2, 4); UNET_F16F8 / VQ_F16F8 restate configs/frido/t2i/frido_f16f8_coco_clip.yaml (config 3).
"""
import copy

UNET_F8F4 = dict(
    use_split_head=True, split_embed_dim_list=[3, 3], use_SPADE_norm=True, image_size=64,
    in_channels=6, out_channels=6, model_channels=192, attention_resolutions=[8, 4, 2],
    num_res_blocks=2, channel_mult=[1, 2, 3, 5], num_head_channels=32,
    use_spatial_transformer=True, transformer_depth=1, context_dim=640, num_stage=2)

VQ_F8F4 = dict(
    embed_dim=[3, 3], n_embed=[4096, 4096],
    edconfig=dict(multiscale=2, double_z=False, z_channels=[3, 3], resolution=256, in_channels=3,
                  out_ch=3, ch=128, ch_mult=[1, 1, 2, 4], num_res_blocks=2, attn_resolutions=[64],
                  dropout=0.0),
    ddconfig=dict(double_z=False, z_channels=6, resolution=256, in_channels=3, out_ch=3, ch=128,
                  ch_mult=[1, 2, 4], num_res_blocks=2, attn_resolutions=[64], dropout=0.0))

UNET_F16F8 = dict(
    use_split_head=True, split_embed_dim_list=[4, 4], use_SPADE_norm=True, image_size=32,
    in_channels=8, out_channels=8, model_channels=192, attention_resolutions=[8, 4, 2],
    num_res_blocks=2, channel_mult=[1, 2, 3, 5], num_head_channels=32,
    use_spatial_transformer=True, transformer_depth=1, context_dim=768, num_stage=2)

# first stage of the t2i config (frido_f16f8_coco_clip.yaml:47-77): f16f8, 8192 x 2 codes of dim 4, decoder attention at 32 x 32
VQ_F16F8 = dict(
    embed_dim=[4, 4], n_embed=[8192, 8192],
    edconfig=dict(multiscale=2, double_z=False, z_channels=[4, 4], resolution=256, in_channels=3,
                  out_ch=3, ch=128, ch_mult=[1, 1, 2, 2, 4], num_res_blocks=2, attn_resolutions=[32],
                  dropout=0.0),
    ddconfig=dict(double_z=False, z_channels=8, resolution=256, in_channels=3, out_ch=3, ch=128,
                  ch_mult=[1, 1, 2, 4], num_res_blocks=2, attn_resolutions=[32], dropout=0.0))

# BASELINE config 5 (512 x 512, 3-scale pyramid).  The reference ships no such YAML (SURVEY.md §8d item 5 defines it):
# f8f4's denoiser on a 128 x 128 latent with 9 channels in three stages, first stage with three 4096-entry codebooks and
# decoder attention on the 128 x 128 latent plane (16384 keys).
UNET_512 = dict(UNET_F8F4, split_embed_dim_list=[3, 3, 3], image_size=128, in_channels=9, out_channels=9, num_stage=3)

VQ_512 = dict(
    embed_dim=[3, 3, 3], n_embed=[4096, 4096, 4096],
    edconfig=dict(multiscale=3, double_z=False, z_channels=[3, 3, 3], resolution=512, in_channels=3,
                  out_ch=3, ch=128, ch_mult=[1, 1, 2, 2, 4], num_res_blocks=2, attn_resolutions=[128],
                  dropout=0.0),
    ddconfig=dict(double_z=False, z_channels=9, resolution=512, in_channels=3, out_ch=3, ch=128,
                  ch_mult=[1, 2, 4], num_res_blocks=2, attn_resolutions=[128], dropout=0.0))

BERT_FULL = dict(n_embed=640, n_layer=32, max_seq_len=96, use_tokenizer=False)


def frido_cfg(ucfg, vcfg, bcfg,
              unet_target="frido.modules.diffusionmodules.pyunet.PyUNetModel",
              vq_target="taming.models.msvqgan.VQModelInterface",
              bert_target="frido.modules.encoders.modules.BERTEmbedder"):
    """kwargs for FridoDiffusion(...) mirroring the reference YAML's `model.params`."""
    return dict(
        adopted_scale_factor=True, noise_mix_ratio=0.1, first_stage_key="image",
        cond_stage_key="objects_bbox", linear_start=0.0015, linear_end=0.0155, num_timesteps_cond=1,
        log_every_t=200, timesteps=1000, loss_type="l1", image_size=ucfg["image_size"],
        channels=ucfg["in_channels"], cond_stage_trainable=True, conditioning_key="crossattn",
        scale_by_std=True, monitor="val/loss",
        unet_config=dict(target=unet_target, params=copy.deepcopy(ucfg)),
        first_stage_config=dict(target=vq_target, params=dict(
            copy.deepcopy(vcfg), lossconfig=dict(target="taming.modules.losses.DummyLoss"))),
        cond_stage_config=dict(target=bert_target, params=copy.deepcopy(bcfg)))
