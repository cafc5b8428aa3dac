"""Architecture walk of the denoiser and the MS-VQGAN: which blocks exist, in which order, with
which state_dict prefixes and channel counts.  Pure host logic (no tensors); it is what both the
HIP plan builders and the oracle iterate over, and it is pinned by the golden fixtures' key lists.

Restates the constructor loops of the reference:
  frido/modules/diffusionmodules/pyunet.py:575-803   (PyUNetModel.__init__)
  taming/modules/diffusionmodules/model.py:435-510   (MSEncoder.__init__)
  taming/modules/diffusionmodules/model.py:548-616   (Decoder.__init__)
"""
from dataclasses import dataclass, field
from typing import List, Optional

__all__ = ["UNetArch", "unet_arch", "DecoderArch", "decoder_arch", "EncoderArch", "encoder_arch"]


@dataclass
class Blk:
    kind: str            # 'res' | 'st' | 'down' | 'up'
    prefix: str          # state_dict prefix relative to the U-Net root, e.g. 'input_blocks.3.0'
    cin: int
    cout: int


@dataclass
class UNetArch:
    model_channels: int
    time_embed_dim: int
    context_dim: Optional[int]
    num_stage: int
    splits: List[int]
    use_spade: bool
    use_split_head: bool
    in_channels: int
    input_blocks: List[List[Blk]] = field(default_factory=list)
    middle: List[Blk] = field(default_factory=list)
    output_blocks: List[List[Blk]] = field(default_factory=list)
    skip_channels: List[int] = field(default_factory=list)   # channels pushed on the skip stack
    image_size: int = 64
    transformer_depth: int = 1     # BasicTransformerBlocks per SpatialTransformer (attention.py:274-277)


def unet_arch(cfg) -> UNetArch:
    mc = cfg["model_channels"]
    mult = list(cfg.get("channel_mult", (1, 2, 4, 8)))
    nres = cfg["num_res_blocks"]
    attn_res = set(cfg["attention_resolutions"])
    use_st = cfg.get("use_spatial_transformer", False)
    if not use_st:
        raise NotImplementedError("only the SpatialTransformer denoiser (every shipped Frido config) is built")
    if cfg.get("resblock_updown", False) or cfg.get("use_scale_shift_norm", False):
        raise NotImplementedError("resblock_updown / use_scale_shift_norm are not used by any Frido config")
    depth = int(cfg.get("transformer_depth", 1))
    if depth < 1:
        raise ValueError("transformer_depth must be >= 1")
    a = UNetArch(model_channels=mc, time_embed_dim=4 * mc, context_dim=cfg.get("context_dim"),
                 num_stage=cfg.get("num_stage", 1), splits=list(cfg.get("split_embed_dim_list", [])),
                 use_spade=cfg.get("use_SPADE_norm", False), use_split_head=cfg.get("use_split_head", False),
                 in_channels=cfg["in_channels"], image_size=cfg.get("image_size", 64), transformer_depth=depth)
    # pyunet.py:600-609: with the split head input_blocks starts empty, otherwise block 0 is the conv
    idx = 0 if a.use_split_head else 1
    chans = [mc]
    ch = mc
    ds = 1
    for level, m in enumerate(mult):
        for _ in range(nres):
            blk = [Blk("res", f"input_blocks.{idx}.0", ch, m * mc)]
            ch = m * mc
            if ds in attn_res:
                blk.append(Blk("st", f"input_blocks.{idx}.1", ch, ch))
            a.input_blocks.append(blk)
            chans.append(ch)
            idx += 1
        if level != len(mult) - 1:
            a.input_blocks.append([Blk("down", f"input_blocks.{idx}.0", ch, ch)])
            chans.append(ch)
            idx += 1
            ds *= 2
    a.skip_channels = list(chans)
    a.middle = [Blk("res", "middle_block.0", ch, ch), Blk("st", "middle_block.1", ch, ch),
                Blk("res", "middle_block.2", ch, ch)]
    oidx = 0
    for level, m in list(enumerate(mult))[::-1]:
        for i in range(nres + 1):
            ich = chans.pop()
            blk = [Blk("res", f"output_blocks.{oidx}.0", ch + ich, mc * m)]
            ch = mc * m
            j = 1
            if ds in attn_res:
                blk.append(Blk("st", f"output_blocks.{oidx}.{j}", ch, ch))
                j += 1
            if level and i == nres:
                blk.append(Blk("up", f"output_blocks.{oidx}.{j}", ch, ch))
                ds //= 2
            a.output_blocks.append(blk)
            oidx += 1
    return a


# ------------------------------------------------------------------------------------------------
@dataclass
class VBlk:
    kind: str            # 'res' | 'attn' | 'up' | 'down'
    prefix: str
    cin: int
    cout: int


@dataclass
class DecoderArch:
    z_channels: int
    block_in: int
    out_ch: int
    z_res: int
    body: List[VBlk] = field(default_factory=list)   # everything between conv_in and norm_out
    last_ch: int = 0


def decoder_arch(dd, prefix="decoder") -> DecoderArch:
    ch, mult = dd["ch"], list(dd["ch_mult"])
    nres = dd["num_res_blocks"]
    attn_res = set(dd["attn_resolutions"])
    nlev = len(mult)
    block_in = ch * mult[-1]
    res = dd["resolution"] // 2 ** (nlev - 1)
    d = DecoderArch(z_channels=dd["z_channels"], block_in=block_in, out_ch=dd["out_ch"], z_res=res)
    d.body += [VBlk("res", f"{prefix}.mid.block_1", block_in, block_in),
               VBlk("attn", f"{prefix}.mid.attn_1", block_in, block_in),
               VBlk("res", f"{prefix}.mid.block_2", block_in, block_in)]
    for lvl in reversed(range(nlev)):
        bout = ch * mult[lvl]
        for i in range(nres + 1):
            d.body.append(VBlk("res", f"{prefix}.up.{lvl}.block.{i}", block_in, bout))
            block_in = bout
            if res in attn_res:
                d.body.append(VBlk("attn", f"{prefix}.up.{lvl}.attn.{i}", block_in, block_in))
        if lvl != 0:
            d.body.append(VBlk("up", f"{prefix}.up.{lvl}.upsample", block_in, block_in))
            res *= 2
    d.last_ch = block_in
    return d


@dataclass
class EncoderArch:
    ch: int
    in_channels: int
    multiscale: int
    down: List[List[VBlk]] = field(default_factory=list)      # per level: blocks incl. trailing 'down'
    level_out_ch: List[int] = field(default_factory=list)
    heads: List[List[VBlk]] = field(default_factory=list)     # mid_ms[i] blocks
    head_ch: List[int] = field(default_factory=list)
    z_channels: List[int] = field(default_factory=list)


def encoder_arch(ed, prefix="encoder") -> EncoderArch:
    ch, mult = ed["ch"], list(ed["ch_mult"])
    nres = ed["num_res_blocks"]
    attn_res = set(ed["attn_resolutions"])
    ms = ed["multiscale"]
    e = EncoderArch(ch=ch, in_channels=ed["in_channels"], multiscale=ms, z_channels=list(ed["z_channels"]))
    in_mult = [1] + mult
    res = ed["resolution"]
    for lvl in range(len(mult)):
        bin_, bout = ch * in_mult[lvl], ch * mult[lvl]
        blocks = []
        for i in range(nres):
            blocks.append(VBlk("res", f"{prefix}.down.{lvl}.block.{i}", bin_, bout))
            bin_ = bout
            if res in attn_res:
                blocks.append(VBlk("attn", f"{prefix}.down.{lvl}.attn.{i}", bin_, bin_))
        if lvl != len(mult) - 1:
            blocks.append(VBlk("down", f"{prefix}.down.{lvl}.downsample", bin_, bin_))
            res //= 2
        e.down.append(blocks)
        e.level_out_ch.append(bout)
    for i, m in enumerate(in_mult[-ms:]):
        c = ch * m
        e.heads.append([VBlk("res", f"{prefix}.mid_ms.{i}.block_1", c, c),
                        VBlk("attn", f"{prefix}.mid_ms.{i}.attn_1", c, c),
                        VBlk("res", f"{prefix}.mid_ms.{i}.block_2", c, c)])
        e.head_ch.append(c)
    return e
