"""Op-level program builder: turns "conv3x3 of this NHWC operand with that weight" into FridoOp
descriptors, owning the packed-weight cache and the activation pool.  The U-Net / VQGAN plans
(unet_plan.py, vqgan_plan.py) are written against this API; the kernel unit tests drive it directly.
"""
import contextlib
import ctypes as C_
import os

import torch

from . import _lib
from .engine import (Operand, POperand, F32, Alias, Pool, Prog, pack_matrix, pack_conv_weight, rup)

ACT_NONE, ACT_RELU, ACT_SILU = 0, 1, 2
UP2_PHASES = os.environ.get("FRIDO_UP2_PHASES", "1") != "0"       # Upsample convs as four 2x2 phase convolutions
GN_FUSED = os.environ.get("FRIDO_GN_FUSED", "1") != "0"          # one-launch GroupNorm (norm.hip gn_fused_kernel)
GN_CONV_TINY = os.environ.get("FRIDO_GN_CONV_TINY", "1") != "0"     # r06: the eps head (GroupNorm + SiLU + conv3x3 to 3 / 4 channels) as ONE f32 VALU launch (tile 40)
GN_FUSED_MAX_HW = int(os.environ.get("FRIDO_GN_FUSED_MAX_HW", "256"))   # larger planes: gn_stats + gn_apply are faster
# r04: a split-K GEMM whose output goes straight into a one-launch GroupNorm leaves its reduction to that launch (no splitk_reduce)
SK_DEFER = os.environ.get("FRIDO_SK_DEFER", "1") != "0"
GN_EPI_STATS = os.environ.get("FRIDO_GN_EPI_STATS", "1") != "0"   # GroupNorm partial sums from the producing GEMM's epilogue (bf16x3 f32 stream)
LN_IN_ATTN = os.environ.get("FRIDO_LN_IN_ATTN", "1") != "0"       # norm2 / norm3 of a transformer block from the attention kernel's epilogue (A/B switch)
ATTN_FLASH = os.environ.get("FRIDO_ATTN_FLASH", "1") != "0"       # flash-style kernel for long key sequences (flash.hip)
# below this many keys the score matrix is small and the batched GEMM -> softmax -> GEMM chain fills the chip better than
# one workgroup per 64 queries (measured at B = 16: 256 keys x d = 576: 37 us vs 49 us; 1024 keys x d = 384: 102 vs 82 us)
ATTN_FLASH_MIN_KEYS = int(os.environ.get("FRIDO_ATTN_FLASH_MIN_KEYS", "512"))
# SHORT key sequences (cross-attention: 26 / 92 / 1 tokens) on planes with at least this many queries per sample also take the
# flash kernel (one 32-key tile, no cross-wave score exchange) instead of the 16-query short-key kernel; 0 = never
ATTN_FLASH_SHORT_NQ = int(os.environ.get("FRIDO_ATTN_FLASH_SHORT_NQ", "0"))
# GroupNorm-apply fused into the 3x3 conv that consumes it (csrc/convgn.inc, two-plane mode, 64^2 / 32^2 planes): "1" = where the
# launch fills the chip (>= 224 workgroups), "0" = never (gn_apply + ring conv: the r03 path), "force" = wherever the kernel applies
GN_CONV = os.environ.get("FRIDO_GN_CONV", "1")
# fused kernel with K split over slices on the 16 x 16 planes (48 tiles): built, tested, measured -- a tie per launch (100 us vs 100.5 us for
# gn_fused + the split-K ring conv) and -2.3 % end to end (profiles/r04_gnconv_splitk_ab.txt: few chunks per slice leave the prologue's
# un-overlapped staging and the 196-KB partial-sum epilogue uncovered), so it stays OFF; the C ABI keeps the option (FridoGemm.splitk on tiles 20 / 21)
GN_CONV_SPLITK = os.environ.get("FRIDO_GN_CONV_SPLITK", "0") != "0"
ATTN_SKIP_DEAD_STREAM = os.environ.get("FRIDO_ATTN_SKIP_DEAD_STREAM", "1") != "0"      # r05: cross-attention does not store f32 rows nobody reads (A/B switch)
GN_CONV_PREFER = int(os.environ.get("FRIDO_GN_CONV_PREFER", "256"))      # A/B: which tile height is tried first where both fill the chip


class Builder:
    def __init__(self, device, nsplit, weights=None, ws_tag="", planes="f16"):
        self.device = torch.device(device)
        self.nsplit = nsplit
        self.planes = planes            # two-plane element format = which build of the library this builder's programs run on (_lib.use_planes)
        self.ws_tag = ws_tag
        self.w = weights or {}          # name -> f32 tensor on device (reference state_dict naming)
        self.pool = Pool(self.device)
        self.prog = Prog(self.device, nsplit, ws_tag)
        self._wcache = {}
        self._persist = []              # tensors that must outlive the builder's programs
        # bf16 mode keeps the residual stream itself in bf16 (it then doubles as the MFMA operand: no pack passes);
        # bf16x3 (fp32-class parity mode) keeps an f32 stream
        self.stream_bf16 = nsplit == 1

    @contextlib.contextmanager
    def persist_scope(self):
        """Persistent tensors created inside the scope (hoisted SPADE maps, V^T operands, bias / table tensors) are collected in
        the yielded list INSTEAD of the builder's own: the caller (a SamplerEngine) owns them, so dropping the engine -- e.g. its
        eviction from the per-denoiser LRU cache -- frees their HBM.  The packed-weight cache stays shared."""
        outer, mine = self._persist, []
        self._persist = mine
        try:
            yield mine
        finally:
            self._persist = outer

    # ---- programs ------------------------------------------------------------------------------
    def new_prog(self):
        self.prog = Prog(self.device, self.nsplit, self.ws_tag)
        self.prog.keep.append(self)
        return self.prog

    # ---- allocation -----------------------------------------------------------------------------
    def f32(self, rows, C):
        """Residual-stream activation (bf16 in bf16 mode)."""
        return F32(self.pool, rows, C, bf16=self.stream_bf16)

    def f32_strict(self, rows, C):
        """Always-f32 scratch (attention scores, tiny heads)."""
        return F32(self.pool, rows, C, bf16=False)

    def op(self, rows, K, batch=1):
        return POperand(self.pool, rows, K, self.nsplit, batch=batch)

    def persistent_op(self, rows, K, batch=1, zero=True):
        o = Operand(rows, K, self.nsplit, self.device, zero=zero, batch=batch)
        self._persist.append(o)
        return o

    def persistent_f32(self, *shape, zero=False):
        t = (torch.zeros if zero else torch.empty)(*shape, dtype=torch.float32, device=self.device)
        self._persist.append(t)
        return t

    def h64(self, name):
        """Weight as a float64 HOST matrix [N][K]: the plan-time algebra (folded projections) is done on the CPU so that the
        device trace holds nothing but this library's kernels (no rocBLAS / Tensile fp64 GEMMs)."""
        w = self.w[name].detach().to("cpu", torch.float64)
        return w.reshape(w.shape[0], -1) if w.dim() > 1 else w

    def to_dev(self, t):
        return t.to(torch.float32).contiguous().to(self.device)

    def dev_f32(self, name):
        t = self.w[name]
        if t.dtype != torch.float32 or not t.is_contiguous():
            t = t.float().contiguous()
            self.w[name] = t
        return t

    # ---- weights ----------------------------------------------------------------------------------
    def conv_weight(self, name):
        key = ("conv", name)
        if key not in self._wcache:
            self._wcache[key] = pack_conv_weight(self.w[name], self.nsplit)
        return self._wcache[key]

    def lin_weight(self, name, rows=None):
        """Linear / 1x1-conv weight [N][K] (optionally a row slice) as an operand."""
        key = ("lin", name, rows)
        if key not in self._wcache:
            w = self.w[name]
            w = w.reshape(w.shape[0], -1)
            if rows is not None:
                w = w[rows[0]:rows[1]]
            self._wcache[key] = pack_matrix(w, self.nsplit)
        return self._wcache[key]

    def folded_vo_weight(self, v_name, o_name):
        """Single-head attention: W_o (P (X W_v^T + b_v)) + b_o = P (X (W_o W_v)^T) + (W_o b_v + b_o), because the rows of P
        sum to one.  Returns (operand of W_o W_v, device pointer of the folded bias); the product is formed in float64."""
        key = ("vo", v_name, o_name)
        if key not in self._wcache:
            wv, wo = self.h64(v_name + ".weight"), self.h64(o_name + ".weight")
            bias = self.h64(o_name + ".bias").clone() if (o_name + ".bias") in self.w else torch.zeros(wo.shape[0], dtype=torch.float64)
            if (v_name + ".bias") in self.w:
                bias += wo @ self.h64(v_name + ".bias")
            self._wcache[key] = (pack_matrix(self.to_dev(wo @ wv), self.nsplit), self.to_dev(bias))
        wop, bias = self._wcache[key]
        return wop, bias.data_ptr()

    def chained_linear(self, a, a2, first, second, residual):
        """y = second(first(a) + a2) + residual with both Linears in ONE GEMM: [a | a2] . [W_s W_f | W_s]^T + (W_s b_f + b_s)
        (a2 is read as a second A operand along K)."""
        key = ("chain", first, second)
        if key not in self._wcache:
            wf, ws = self.h64(first + ".weight"), self.h64(second + ".weight")
            bias = ws @ self.h64(first + ".bias") + self.h64(second + ".bias")
            k1, k2 = rup(wf.shape[1], 64), rup(ws.shape[1], 64)
            w = torch.zeros((ws.shape[0], k1 + k2), dtype=torch.float64)
            w[:, :wf.shape[1]] = ws @ wf
            w[:, k1:k1 + ws.shape[1]] = ws
            self._wcache[key] = (pack_matrix(self.to_dev(w), self.nsplit), self.to_dev(bias), k1, k2)
        wop, bias, k1, k2 = self._wcache[key]
        # a2: a bf16 stream activation (bf16 mode), or an operand copy of the f32 stream (bf16x3 mode: the cross-attention kernel
        # writes it next to the stream, see attention(also_op=True))
        a2k = a2.C if getattr(a2, "bf16", False) else a2.K
        assert a.K == k1 and a2k == k2, (a.K, k1, a2k, k2)
        M = a.rows * getattr(a, "batch", 1)
        res = self.f32(M, wop.rows)
        self.prog.gemm(M, wop.rows, k1, a, wop, ldb=k1 + k2, bias=bias.data_ptr(), residual=residual.ptr, ldr=residual.C,
                       res_bf16=getattr(residual, "bf16", False), out_f32=res.ptr, ldo=wop.rows, out_bf16=res.bf16,
                       A2=a2, lda2=a2k, K2=k2, gn_part=self._parts_for(res, M, wop.rows, residual=residual))
        self._parts_done(res)
        return res

    def folded_qk_weight(self, q_name, k_name, side):
        """Single-head scores S = (x W_q^T + b_q)(y W_k^T + b_k)^T.  Terms constant along the key axis cancel in the softmax,
        so S ~ x (W_q^T W_k) y^T + (W_k^T b_q) . y.  side="q": returns W' = W_k^T W_q and b' = W_k^T b_q with
        S = (x W'^T + b') y^T (the keys are the raw input rows);  side="k": returns W'' = W_q^T W_k with S = x (y W''^T)^T
        (only valid without a query bias: the projected keys can be cached and the queries are the raw input rows)."""
        key = ("qk", q_name, k_name, side)
        if key not in self._wcache:
            wq, wk = self.h64(q_name + ".weight"), self.h64(k_name + ".weight")
            bq = self.h64(q_name + ".bias") if (q_name + ".bias") in self.w else None
            if side == "q":
                bias = self.to_dev(wk.t() @ bq) if bq is not None else None
                self._wcache[key] = (pack_matrix(self.to_dev(wk.t() @ wq), self.nsplit), bias)
            else:
                assert bq is None, "a query bias cannot be folded into the key side"
                self._wcache[key] = (pack_matrix(self.to_dev(wq.t() @ wk), self.nsplit), None)
        wop, bias = self._wcache[key]
        return wop, (bias.data_ptr() if bias is not None else None)

    def cat_lin_weight(self, key, names):
        if key not in self._wcache:
            w = torch.cat([self.w[n].reshape(self.w[n].shape[0], -1) for n in names], dim=0)
            self._wcache[key] = pack_matrix(w, self.nsplit)
        return self._wcache[key]

    def geglu_weight(self, wname):
        """GEGLU projection [2H][K] with rows regrouped in 16-row blocks [a_0..15 | gate_0..15 | a_16..31 | ...] so that
        a value and its gate land in the same lane of adjacent MFMA column tiles; returns (operand, bias ptr, H)."""
        key = ("geglu", wname)
        if key not in self._wcache:
            w = self.w[wname + ".weight"].float()
            bvec = self.w[wname + ".bias"].float()
            H = w.shape[0] // 2
            assert H % 16 == 0
            idx = torch.arange(H, device=w.device).view(-1, 16)
            perm = torch.stack([idx, idx + H], dim=1).reshape(-1)
            bp = bvec[perm].contiguous()
            self._wcache[key] = (pack_matrix(w[perm], self.nsplit), bp, H)
        return self._wcache[key]

    def linear_geglu(self, a, wname):
        """operand [M][H] = a_proj * gelu(gate_proj) of the GEGLU projection `wname` (attention.py:37-44) in ONE GEMM."""
        wop, bp, H = self.geglu_weight(wname)
        M = a.rows * getattr(a, "batch", 1)
        o = self.op(M, H)
        self.prog.gemm(M, 2 * H, wop.K, a, wop, bias=bp.data_ptr(), out_op=o.ptr, ldoo=H, oo_lo=o.lo, geglu=1)
        return o

    def bias(self, name):
        return self.dev_f32(name).data_ptr() if name in self.w else None

    # ---- ops -----------------------------------------------------------------------------------
    def conv_plus_skip_weight(self, wname, skip_wname):
        """[Cout][9*Cin_pad | Cskip_pad64] weight of a 3x3 conv with a 1x1 skip conv appended along K, and the summed bias."""
        key = ("conv+skip", wname, skip_wname)
        if key not in self._wcache:
            w = self.w[wname + ".weight"].float()
            co, ci, kh, kw = w.shape
            cp = rup(ci, 32)
            wk = torch.zeros((co, kh, kw, cp), dtype=torch.float32, device=w.device)
            wk[..., :ci] = w.permute(0, 2, 3, 1)
            ws = self.w[skip_wname + ".weight"].float().reshape(co, -1)
            k2 = rup(ws.shape[1], 64)
            wsk = torch.zeros((co, k2), dtype=torch.float32, device=w.device)
            wsk[:, :ws.shape[1]] = ws
            bsum = (self.w[wname + ".bias"].float() + self.w[skip_wname + ".bias"].float()).contiguous()
            self._wcache[key] = (pack_matrix(torch.cat([wk.reshape(co, -1), wsk], dim=1), self.nsplit), cp, k2, bsum)
        return self._wcache[key]

    def conv_plus_skip(self, a, raw, B, H, W, wname, skip_wname, out="f32"):
        """out = conv3x3(a) + conv1x1(raw) in ONE implicit GEMM (the 1x1 rides along as an extra K range)."""
        wop, cp, k2, bsum = self.conv_plus_skip_weight(wname, skip_wname)
        w = self.w[wname + ".weight"]
        co, _, kh, kw = w.shape
        assert a.K == cp and raw.K % 64 == 0 and raw.K == k2 and (kh * kw * cp) % 64 == 0
        M = B * H * W
        geom = dict(Hs=H, Ws=W, Cin=cp, Hl=H, Wl=W, Ho=H, Wo=W, kh=kh, kw=kw, stride=1, pad=1, up_shift=0, dn_shift=0)
        res = self.f32(M, co)
        self.prog.gemm(M, co, kh * kw * cp, a, wop, ldb=kh * kw * cp + k2, conv=geom, bias=bsum.data_ptr(), out_f32=res.ptr, ldo=co,
                       out_bf16=res.bf16, A2=raw, lda2=raw.K, K2=k2, gn_part=self._parts_for(res, M, co))
        self._parts_done(res)
        return res

    def up2_phase_weights(self, wname):
        """nearest x2 upsample -> conv3x3 (pad 1) as four 2x2 PHASE convolutions on the source plane: output pixel
        (2y+a, 2x+b) reads source rows {y-1, y} (a = 0) or {y, y+1} (a = 1) -- and likewise columns -- with the 3x3 taps that
        land on the same source pixel summed (float64).  Returns the operand [4][Cout][4 Cin_pad] of the phases
        (a, b) = (0,0), (0,1), (1,0), (1,1) (top / left padding of a phase: 1 - a, 1 - b)."""
        key = ("up2", wname)
        if key not in self._wcache:
            w = self.w[wname + ".weight"].detach().to("cpu", torch.float64)            # [Cout][Cin][3][3]
            co, ci = w.shape[:2]
            cp = rup(ci, 32)
            rows = {0: [w[:, :, 0], w[:, :, 1] + w[:, :, 2]], 1: [w[:, :, 0] + w[:, :, 1], w[:, :, 2]]}     # [ty] -> [Cout][Cin][3(kx)]
            mats = []
            for a_ in (0, 1):
                for b_ in (0, 1):
                    wk = torch.zeros((co, 2, 2, cp), dtype=torch.float64)
                    for ty in (0, 1):
                        r = rows[a_][ty]                      # [Cout][Cin][3]
                        cols = [r[:, :, 0], r[:, :, 1] + r[:, :, 2]] if b_ == 0 else [r[:, :, 0] + r[:, :, 1], r[:, :, 2]]
                        for tx in (0, 1):
                            wk[:, ty, tx, :ci] = cols[tx]
                    mats.append(wk.reshape(co, -1))
            self._wcache[key] = (pack_matrix(self.to_dev(torch.cat(mats, dim=0)), self.nsplit), cp)      # [4 Cout][4 Cin_pad]
        return self._wcache[key]

    def upsample_conv(self, a, B, Hs, Ws, wname):
        """Upsample block: nearest x2 + conv3x3.  Four 2x2 phase convolutions when the plane allows it, else the 9-tap
        conv with the up-sampling folded into its addressing."""
        if UP2_PHASES and Hs & (Hs - 1) == 0 and Ws & (Ws - 1) == 0:
            return self.conv_up2(a, B, Hs, Ws, wname)
        return self.conv(a, B, Hs, Ws, wname, up=1)

    def conv_up2(self, a, B, Hs, Ws, wname):
        """Upsample (nearest x2) + conv3x3 of the NHWC operand `a` -> stream activation [B*2Hs*2Ws][Cout] (4/9 of the MACs of
        the conv on the upsampled plane; pyunet.py:110-121, taming model.py:49-53)."""
        wop, cp = self.up2_phase_weights(wname)
        co = self.w[wname + ".weight"].shape[0]
        assert a.K == cp and Hs & (Hs - 1) == 0 and Ws & (Ws - 1) == 0
        res = self.f32(B * 4 * Hs * Ws, co)
        geom = dict(Hs=Hs, Ws=Ws, Cin=cp, Hl=Hs, Wl=Ws, Ho=Hs, Wo=Ws, kh=2, kw=2, stride=1, pad=1, padx=1, up_shift=0, dn_shift=0,
                    up2_phase=5)
        # one launch: batch index = phase (weights co * 4 cp apart), every phase reads the same source and interleaves its rows
        self.prog.gemm(B * Hs * Ws, co, 4 * cp, a, wop, batch=4, b_bs=co * 4 * cp, conv=geom, bias=self.bias(wname + ".bias"),
                       out_f32=res.ptr, ldo=co, out_bf16=res.bf16)
        return res

    def conv(self, a, B, Hs, Ws, wname, *, stride=1, pad=1, up=0, dn=0, Ho=None, Wo=None, bias=True,
             rowvec=None, act=ACT_NONE, residual=None, out="f32", alpha=1.0):
        """3x3 / 1x1 convolution of the NHWC operand `a` ([B*Hs*Ws][Cin_pad]).
        up: nearest x2^up up-sampling folded in (pyunet.py:119); dn: source sampled at stride 2^dn
        (spade_norm.py:52 nearest down-resize).  Returns F32 or POperand [B*Ho*Wo][Cout]."""
        wop, cp = self.conv_weight(wname + ".weight")
        w = self.w[wname + ".weight"]
        co, _, kh, kw = w.shape
        assert a.K == cp, (wname, a.K, cp)
        Hl, Wl = (Hs << up) >> dn, (Ws << up) >> dn
        if Ho is None:
            Ho = (Hl + 2 * pad - kh) // stride + 1
            Wo = (Wl + 2 * pad - kw) // stride + 1
        M = B * Ho * Wo
        geom = dict(Hs=Hs, Ws=Ws, Cin=cp, Hl=Hl, Wl=Wl, Ho=Ho, Wo=Wo, kh=kh, kw=kw, stride=stride, pad=pad,
                    up_shift=up, dn_shift=dn)
        return self._gemm_out(M, co, kh * kw * cp, a, wop, conv=geom, bias=self.bias(wname + ".bias") if bias else None,
                              rowvec=rowvec, act=act, residual=residual, out=out, alpha=alpha)

    def linear(self, a, wname, *, M=None, bias=True, act=ACT_NONE, residual=None, out="f32", rowvec=None,
               wop=None, bias_ptr=None, alpha=1.0, N=None):
        """y[M][N] = a[M][K] @ W[N][K]^T (+bias).  `a` is an operand with K == W's padded K."""
        wop = wop or self.lin_weight(wname + ".weight")
        M = M if M is not None else a.rows * getattr(a, "batch", 1)
        N = N or wop.rows
        assert a.K == wop.K, (wname, a.K, wop.K)
        if bias_ptr is None and bias:
            bias_ptr = self.bias(wname + ".bias")
        return self._gemm_out(M, N, wop.K, a, wop, bias=bias_ptr, rowvec=rowvec, act=act, residual=residual, out=out,
                              alpha=alpha)

    def _gemm_out(self, M, N, K, a, wop, *, conv=None, bias=None, rowvec=None, act=0, residual=None, out="f32",
                  alpha=1.0):
        kw = {}
        res = None
        if out == "f32":
            res = self.f32(M, N)
            kw.update(out_f32=res.ptr, ldo=N, out_bf16=res.bf16)
        elif out == "f32_strict":
            res = self.f32_strict(M, N)
            kw.update(out_f32=res.ptr, ldo=N)
        elif out == "op":
            res = self.op(M, rup(N, 32))
            assert rup(N, 32) == N, "operand outputs must have N % 32 == 0 (pad columns would be garbage)"
            kw.update(out_op=res.ptr, ldoo=res.K, oo_lo=res.lo)
        elif isinstance(out, tuple) and out[0] == "u8":      # ("u8", uint8 tensor [M][N], mode): image output fused into the epilogue
            _, res, mode = out
            kw.update(out_u8=res.data_ptr(), ldu8=N, u8_mode=mode)
        elif isinstance(out, tuple):        # ("f32"|"op", existing buffer)
            kind, res = out
            if kind == "f32":
                kw.update(out_f32=res.ptr, ldo=res.C, out_bf16=getattr(res, "bf16", False))
            else:
                kw.update(out_op=res.ptr, ldoo=res.K, oo_lo=res.lo)
        if residual is not None:
            kw.update(residual=residual.ptr, ldr=residual.C, res_bf16=getattr(residual, "bf16", False))
        if rowvec is not None:
            kw.update(rowvec=rowvec["ptr"], rows_per_vec=rowvec["rows_per_vec"], ldv=rowvec["ld"],
                      rowvec_step=rowvec.get("step"))
        stream_out = out == "f32" or (isinstance(out, tuple) and out[0] == "f32" and isinstance(res, F32))
        if stream_out and res.gn_part is not None:      # an existing activation is being overwritten: its old partial sums are stale
            self.pool.release(res.gn_part)
            res.gn_part = None
        if stream_out and res.C == N:
            kw["gn_part"] = self._parts_for(res, M, N, act=act, rowvec=rowvec, residual=residual,
                                            up2=bool(conv and conv.get("up2_phase")))
        self.prog.gemm(M, N, K, a, wop, bias=bias, act=act, alpha=alpha, conv=conv, **kw)
        if stream_out:
            self._parts_done(res)
        return res

    def _parts_for(self, res, M, N, *, act=0, rowvec=None, residual=None, up2=False):
        """Partial-sum buffer for the GEMM about to write the f32 stream activation `res` (FridoGemm.gn_part), or None when the
        launch does not take the store-from-registers f32 epilogue (mirror of the library's check in frido_gemm)."""
        if not (GN_EPI_STATS and self.nsplit == 2 and not getattr(res, "bf16", True) and act == 0 and not up2 and N % 8 == 0 and M % 32 == 0):
            return None
        if rowvec is not None and (rowvec.get("rows_per_vec") or 0) < (1 << 29):
            return None
        if residual is not None and (getattr(residual, "bf16", False) or residual.C % 8):
            return None
        res.gn_part = self.pool.alloc((M // 32) * N * 8)
        return res.gn_part.data_ptr()

    def _parts_done(self, res):
        """After Prog.gemm: the tuner may have chosen split-K, which drops the partial sums."""
        if res is not None and getattr(res, "gn_part", None) is not None and not self.prog.ops[-1][1].gn_part:
            self.pool.release(res.gn_part)
            res.gn_part = None

    def gn_conv_tile(self, x1, x2, B, H, W, co, raw=None):
        """(tile, splitk) of the fused GroupNorm + 3x3 conv kernel for this plane -- FridoGemm tile 20 (256-row tiles) or 21 (128-row),
        K split over `splitk` slices where the tiles alone would leave the chip idle (16 x 16 planes) -- or (0, 1) when the launch
        does not qualify / cannot fill the chip (the caller then emits groupnorm() + conv())."""
        if GN_CONV == "0" or self.nsplit != 2 or self.device.type != "cuda":
            return 0, 1
        tensors = [x1] + ([x2] if x2 is not None else []) + [r for r in (raw or ()) if r is not None]
        if any(getattr(t, "bf16", False) or t.C % 32 for t in tensors):
            return 0, 1
        C = x1.C + (x2.C if x2 is not None else 0)
        if C > 960 or C % 32 or co % 192 or W not in (16, 32, 64) or (H * W) % 128:
            return 0, 1
        M = B * H * W
        order = ((20, 256, 396), (21, 128, 204)) if GN_CONV_PREFER == 256 else ((21, 128, 204), (20, 256, 396))
        fits = [(tile, bm) for tile, bm, slots in order if not ((H * W) % bm or bm % W or (bm // W + 2) * (W + 2) > slots)]
        for tile, bm in fits:
            if GN_CONV == "force" or (M // bm) * (co // 192) >= 224:
                return tile, 1
        if GN_CONV_SPLITK:
            for tile, bm in fits:       # split-K: ~256 workgroups, every slice at least two 32-channel chunks (18 k-steps)
                tiles = (M // bm) * (co // 192)
                sk = min(-(-256 // tiles), (C // 32) // 2, 8)
                if sk >= 2 and tiles * sk >= 192:
                    return tile, sk
        return 0, 1

    def gn_conv(self, tile, x1, x2, B, H, W, norm_w, eps, conv_w, *, gamma=None, beta=None, act=ACT_SILU, rowvec=None, residual=None,
                skip=None, splitk=1):
        """out = conv3x3(act(GroupNorm32(cat(x1, x2)) [* (1 + gamma) + beta])) [+ conv1x1_skip(cat(raw1, raw2))] + bias [+ rowvec]
        [+ residual] in ONE launch after the GroupNorm statistics (pyunet.py:262-300; taming model.py:117-137): the normalised
        operand is produced inside the conv kernel, gn_apply and its 8 B / element round trip do not exist.
        norm_w: parameter prefix of the norm (weight / bias), conv_w: of the conv; skip = (raw1, raw2 or None, skip conv prefix)."""
        HW, M = H * W, B * H * W
        C = x1.C + (x2.C if x2 is not None else 0)
        part, S = self.gn_stats(x1, x2, B, HW)
        if skip is not None:
            raw1, raw2, skip_w = skip
            wop, cp, k2, bsum = self.conv_plus_skip_weight(conv_w, skip_w)
            rawC = raw1.C + (raw2.C if raw2 is not None else 0)
            assert cp == C and k2 == rawC, (cp, C, k2, rawC)
            bias_ptr, ldb = bsum.data_ptr(), 9 * cp + k2
        else:
            wop, cp = self.conv_weight(conv_w + ".weight")
            assert cp == C, (cp, C)
            k2, bias_ptr, ldb = 0, self.bias(conv_w + ".bias"), 9 * cp
        co = self.w[conv_w + ".weight"].shape[0]
        gn = dict(gn_x1=x1.ptr, gn_C1=x1.C, gn_x2=x2.ptr if x2 is not None else None, gn_C2=x2.C if x2 is not None else 0,
                  gn_partials=part.data_ptr(), gn_nsplit_px=S, gn_groups=32, gn_eps=eps, gn_weight=self.bias(norm_w + ".weight"),
                  gn_bias=self.bias(norm_w + ".bias"), gn_gamma=gamma.ptr if gamma is not None else None,
                  gn_beta=beta.ptr if beta is not None else None, gn_act=act)
        if skip is not None:
            gn.update(raw_x1=raw1.ptr, raw_C1=raw1.C, raw_x2=raw2.ptr if raw2 is not None else None, raw_C2=raw2.C if raw2 is not None else 0)
        geom = dict(Hs=H, Ws=W, Cin=cp, Hl=H, Wl=W, Ho=H, Wo=W, kh=3, kw=3, stride=1, pad=1, up_shift=0, dn_shift=0)
        res = self.f32(M, co)
        kw = {}
        if residual is not None:
            kw.update(residual=residual.ptr, ldr=residual.C)
        if rowvec is not None:
            kw.update(rowvec=rowvec["ptr"], rows_per_vec=rowvec["rows_per_vec"], ldv=rowvec["ld"], rowvec_step=rowvec.get("step"))
        self.prog.gemm(M, co, 9 * cp, None, wop, ldb=ldb, conv=geom, bias=bias_ptr, out_f32=res.ptr, ldo=co, K2=k2, tile=tile, gn=gn,
                       gn_part=self._parts_for(res, M, co, rowvec=rowvec, residual=residual) if splitk <= 1 else None, **kw)
        if splitk > 1:        # partial sums per slice -> workspace; the library launches splitk_reduce (bias, vectors, residual) behind the kernel
            from . import tune
            st = self.prog.ops[-1][1]
            st.splitk, st.sk_mode = splitk, 0
            st.ws = tune.workspace_for(st, self.device, self.prog.ws_tag + (":s1" if self.prog._sid else ""))
        self.pool.release(part)
        return res

    def gn_conv_tiny_ok(self, x, B, H, W, co):
        """The fused GroupNorm + SiLU + 3x3 conv with a TINY output width (FridoGemm tile 40: the denoiser's eps head) applies."""
        return (GN_CONV_TINY and self.nsplit == 2 and self.device.type == "cuda" and not getattr(x, "bf16", False) and co in (3, 4)
                and x.C % 32 == 0 and x.C <= 960 and W in (16, 32, 64) and (H * W) % 256 == 0 and (256 // W + 2) * (W + 2) <= 396)

    def gn_conv_tiny(self, x, B, H, W, norm_w, eps, conv_w, out, act=ACT_SILU):
        """out[B*H*W][co] = conv3x3(act(GroupNorm32(x))) + bias with co in {3, 4}, ONE launch on the f32 VALU after the GroupNorm statistics
        (csrc/convgn.hip conv3x3_gn_tiny_kernel; pyunet.py:775-803 `out`).  Weights go in as f32 [C / 32][9 taps][co][32 channels]."""
        key = ("tinyconv", conv_w)
        if key not in self._wcache:
            w = self.w[conv_w + ".weight"].float()                      # [co][C][3][3]
            co, C = w.shape[0], w.shape[1]
            wt = w.permute(1, 2, 3, 0).reshape(C // 32, 32, 9, co).permute(0, 2, 3, 1).contiguous()      # [C/32][9][co][32]
            self._wcache[key] = self.to_dev(wt)
        wt = self._wcache[key]
        co, C = self.w[conv_w + ".weight"].shape[0], x.C
        part, S = self.gn_stats(x, None, B, H * W)
        gn = dict(gn_x1=x.ptr, gn_C1=C, gn_partials=part.data_ptr(), gn_nsplit_px=S, gn_groups=32, gn_eps=eps,
                  gn_weight=self.bias(norm_w + ".weight"), gn_bias=self.bias(norm_w + ".bias"), gn_act=act, w_f32=wt.data_ptr())
        geom = dict(Hs=H, Ws=W, Cin=C, Hl=H, Wl=W, Ho=H, Wo=W, kh=3, kw=3, stride=1, pad=1, up_shift=0, dn_shift=0)
        self.prog.gemm(B * H * W, co, 9 * C, None, (None, 0), ldb=9 * C, conv=geom, bias=self.bias(conv_w + ".bias"), out_f32=out.ptr, ldo=out.C,
                       tile=40, gn=gn)
        self.pool.release(part)
        return out

    def gn_stats(self, x1, x2, B, HW):
        p1, p2 = getattr(x1, "gn_part", None), getattr(x2, "gn_part", None) if x2 is not None else None
        if p1 is not None and (x2 is None or p2 is not None) and HW % 32 == 0:
            # statistics from the producers' per-channel partial sums (3 % of the tensor's bytes): one tiny launch
            part = self.pool.alloc(B * 32 * 2 * 8)
            self.prog.emit("FRIDO_OP_GN_STATS", x1=x1.ptr, C1=x1.C, x2=x2.ptr if x2 is not None else None,
                           C2=x2.C if x2 is not None else 0, B=B, HW=HW, groups=32, nsplit_px=1, partials=part.data_ptr(), x_bf16=0,
                           p1=p1.data_ptr(), p2=p2.data_ptr() if p2 is not None else None)
            return part, 1
        S = max(1, min(64, HW // 4, max(1, 1024 // B)))     # ~1024 workgroups, at least 4 pixels each
        part = self.pool.alloc(B * S * 32 * 2 * 8)
        xb = getattr(x1, "bf16", False)
        assert x2 is None or getattr(x2, "bf16", False) == xb
        self.prog.emit("FRIDO_OP_GN_STATS", x1=x1.ptr, C1=x1.C, x2=x2.ptr if x2 is not None else None,
                       C2=x2.C if x2 is not None else 0, B=B, HW=HW, groups=32, nsplit_px=S, partials=part.data_ptr(),
                       x_bf16=int(xb))
        return part, S

    def _deferred_splitk(self, x1, M, x1_dead):
        """r04: if the op emitted LAST is the split-K GEMM that writes x1, its reduction moves into the GroupNorm launch being emitted
        (FridoGemm.sk_mode 2 + FridoGnApply.sk_*): the slices are added, the GEMM's epilogue applied and the statistics taken from
        the same registers -- one launch (splitk_reduce) and one round trip of the tensor less.  Only the IMMEDIATE successor
        qualifies: nothing can have read x1, or reused the workspace, in between.  Returns the sk_* fields (or {})."""
        if not (SK_DEFER and self.nsplit == 2 and self.prog.ops and not getattr(x1, "bf16", False)):
            return {}
        kind, st = self.prog.ops[-1]
        if (kind != _lib.OP_KINDS["FRIDO_OP_GEMM"] or st.splitk <= 1 or st.sk_mode != 0 or st.tile in (9, 10, 20, 21)
                or getattr(st, "_sid", 0) != self.prog._sid):
            return {}
        if (st.out_f32 != x1.ptr or st.M != M or st.N != x1.C or st.ldo != st.N or st.N % 8 or st.batch != 1 or st.act or st.geglu
                or st.row_bias or st.out_op or st.out_u8 or st.out_bf16 or st.up2_phase or (st.residual and st.res_bf16)
                or (st.rowvec and (st.rows_per_vec <= 0 or st.ldv % 8)) or (st.residual and st.ldr % 8)):
            # (r05, advisor: ldv / ldr % 8 -- sk_finish8 mirrors splitk_reduce8, the vec8 form launch_splitk_reduce takes only for
            #  8-element aligned strides; anything else keeps its reduce launch)
            return {}
        st.sk_mode = 2
        self.prog._packed = None
        return dict(sk_ws=st.ws + _lib.SPLITK_HEADER_BYTES, sk_n=st.splitk, sk_alpha=st.alpha, sk_bias=st.bias, sk_rowvec=st.rowvec,
                    sk_rowvec_step=st.rowvec_step, sk_rows_per_vec=st.rows_per_vec, sk_ldv=st.ldv, sk_residual=st.residual,
                    sk_ldr=st.ldr, sk_out=None if x1_dead else st.out_f32)

    def groupnorm(self, x1, x2, B, HW, wname, eps, *, gamma=None, beta=None, act=ACT_NONE, want_raw=False,
                  out_f32=False, x1_dead=False):
        """GroupNorm(32) [+SPADE] [+SiLU] of the (virtually concatenated) NHWC f32 input -> operand.  x1_dead: nothing after this
        op reads x1 (lets a deferred split-K reduction skip materialising it)."""
        C = x1.C + (x2.C if x2 is not None else 0)
        # (r05) a deferred split-K reduction moves the READ of the producing GEMM's residual into THIS launch -- but the residual's
        # owner may have released it at plan time right after emitting the GEMM (unet_plan: the transformer block's input is freed as
        # soon as the block's last GEMM is emitted), and the operand buffers allocated below come out of the same size class: the
        # GroupNorm would then write its output planes over rows other workgroups have not read yet (found by tools/verify_deferred.py
        # on BASELINE config 3 at batch 32: latent error 0.1).  Hold such a buffer back until this op is emitted.
        held = None
        if SK_DEFER and self.nsplit == 2 and self.prog.ops and self.prog.ops[-1][0] == _lib.OP_KINDS["FRIDO_OP_GEMM"]:
            prev = self.prog.ops[-1][1]
            if prev.splitk > 1 and prev.residual and prev.out_f32 == x1.ptr:
                held = self.pool.hold(prev.residual)
        a = self.op(B * HW, C)
        xb = getattr(x1, "bf16", False)
        alias_raw = want_raw and xb and x2 is None          # a bf16 activation already IS its own operand
        raw = None if (not want_raw or alias_raw) else self.op(B * HW, C)
        of = self.f32_strict(B * HW, C) if out_f32 else None
        kw = dict(x1=x1.ptr, C1=x1.C, x2=x2.ptr if x2 is not None else None, C2=x2.C if x2 is not None else 0, B=B, HW=HW,
                  groups=32, eps=eps, weight=self.bias(wname + ".weight"), bias=self.bias(wname + ".bias"),
                  gamma=gamma.ptr if gamma is not None else None, beta=beta.ptr if beta is not None else None,
                  act=act, nsplit=self.nsplit, out_op=a.ptr, out_lo=a.lo,
                  raw_op=raw.ptr if raw is not None else None, raw_lo=raw.lo if raw is not None else 0,
                  out_f32=of.ptr if of is not None else None, x_bf16=int(xb),
                  gb_bf16=int(getattr(gamma, "bf16", False)) if gamma is not None else 0)
        # one launch (statistics + apply from registers) wherever a (sample, group-chunk) slice fits a workgroup
        _, probe = _lib.make_op("FRIDO_OP_GN_FUSED", **kw)
        if GN_FUSED and HW <= GN_FUSED_MAX_HW and _lib.lib().frido_gn_fused_chunk(C_.byref(probe), None) > 0:
            kw.update(self._deferred_splitk(x1, B * HW, x1_dead))
            self.prog.emit("FRIDO_OP_GN_FUSED", **kw)
            part = None
        else:
            part, S = self.gn_stats(x1, x2, B, HW)
            self.prog.emit("FRIDO_OP_GN_APPLY", nsplit_px=S, partials=part.data_ptr(), **kw)
        if part is not None:
            self.pool.release(part)
        if held is not None:
            self.pool.release(held)
        if alias_raw:
            raw = Alias(x1)
        if out_f32:
            return a, raw, of
        return a, raw

    def layernorm(self, x, wname, eps=1e-5):
        a = self.op(x.rows, x.C)
        self.prog.emit("FRIDO_OP_LAYERNORM", x=x.ptr, rows=x.rows, C=x.C, eps=eps, weight=self.bias(wname + ".weight"),
                       bias=self.bias(wname + ".bias"), nsplit=self.nsplit, out_op=a.ptr, out_lo=a.lo,
                       x_bf16=int(getattr(x, "bf16", False)))
        return a

    def softmax(self, s, rows, N, ld, Npad):
        p = self.op(rows, Npad)
        self.prog.emit("FRIDO_OP_SOFTMAX", x=s.ptr, rows=rows, N=N, ld=ld, Npad=Npad, nsplit=self.nsplit, out_op=p.ptr,
                       out_lo=p.lo)
        return p

    def geglu(self, g, H):
        o = self.op(g.rows, H)
        self.prog.emit("FRIDO_OP_GEGLU", x=g.ptr, rows=g.rows, H=H, nsplit=self.nsplit, out_op=o.ptr, out_lo=o.lo)
        return o

    def pack(self, src_ptr, B, HW, Csrc, c0, Cuse, *, nchw=False, scale=1.0, out=None):
        cp = rup(Cuse, 32)
        o = out or self.op(B * HW, cp)
        self.prog.emit("FRIDO_OP_PACK", src=src_ptr, B=B, HW=HW, Csrc=Csrc, c0=c0, Cuse=Cuse, Cpad=cp, nchw=int(nchw),
                       scale=scale, nsplit=self.nsplit, out_op=o.ptr, out_lo=o.lo)
        return o

    def to_operand(self, x):
        """activation [rows][C] -> operand (C % 32 == 0); a bf16 activation is its own operand."""
        if getattr(x, "bf16", False):
            return Alias(x)
        return self.pack(x.ptr, 1, x.rows, x.C, 0, x.C)

    def relayout(self, src_ptr, dst_ptr, B, HW, Csrc, c0, Cuse, Cdst, d0, to_nchw):
        self.prog.emit("FRIDO_OP_RELAYOUT", src=src_ptr, dst=dst_ptr, B=B, HW=HW, Csrc=Csrc, c0=c0, Cuse=Cuse, Cdst=Cdst,
                       d0=d0, to_nchw=int(to_nchw))

    # ---- attention (single head, unfused: QK^T -> softmax -> PV on the MFMA GEMM) -----------------
    def attention(self, q, ldq, k, ldk, vT, B, Nq, Nk, d, *, q_off=0, k_off=0, bias_ptr=None, residual=None, stream=False,
                  also_op=False, ln=None, stream_dead=False):
        """q: operand rows [B*Nq] (row stride ldq, column offset q_off), k: operand rows [B*Nk], vT: operand
        [B][d][Nk_pad] (zero beyond Nk).  Returns operand O [B*Nq][d].  scale = d ** -0.5 (attention.py:158).
        stream=True (the out projection is folded into vT): returns the residual-stream activation O + bias + residual.
        ln=(weight name, eps) with stream=True in bf16x3 mode: where the kernel's workgroups own whole rows (flash kernel with
        d = 256 / 384, short-key kernel with >= 256 workgroups) the LayerNorm of the result comes back as the operand `res.ln_copy`
        from the same launch; otherwise the attribute is absent and the caller runs layernorm().
        stream_dead=True (r05): the caller reads ONLY res.op_copy / res.ln_copy; when the launch produces both, the f32 stream rows
        are not stored (res.stream_skipped = True) -- 20 % of the launch's bytes on the 32 x 32 plane."""
        Np = rup(Nk, 32)
        aligned = ldq % 8 == 0 and ldk % 8 == 0 and q_off % 8 == 0 and k_off % 8 == 0
        small = Nk <= 128 and Nq % 16 == 0 and d % 32 == 0 and aligned
        flash_ok = ATTN_FLASH and aligned and _lib.lib().frido_attn_flash_supported(d)
        short_flash = small and flash_ok and ATTN_FLASH_SHORT_NQ > 0 and Nq >= ATTN_FLASH_SHORT_NQ
        small = small and not short_flash
        flash = not small and flash_ok and (Nk >= ATTN_FLASH_MIN_KEYS or Nk > 4096 or short_flash)
        if small or flash:
            # one launch, scores stay on chip: the short-key kernel (cross-attention, 8x8 planes) or the flash-style kernel
            kind = "FRIDO_OP_ATTN_SMALL" if small else "FRIDO_OP_ATTN_FLASH"
            kw = dict(Q=q.ptr + 2 * q_off, q_lo=q.lo, ldq=ldq, K=k.ptr + 2 * k_off, k_lo=k.lo, k_bs=Nk * ldk, ldk=ldk, VT=vT.ptr,
                      vt_lo=vT.lo, vt_bs=d * Np, ldvt=Np, B=B, Nq=Nq, Nk=Nk, d=d, dv=d, nsplit=self.nsplit, alpha=float(d) ** -0.5)
            if stream:
                res = self.f32(B * Nq, d)
                assert residual is None or getattr(residual, "bf16", False) == res.bf16
                if ln is not None and LN_IN_ATTN and not res.bf16 and self.nsplit == 2 and (
                        (flash and self._flash_ln_ok(d)) or (small and B * (Nq // 16) >= 256)):
                    n = self.op(B * Nq, d)
                    kw.update(ln_op=n.ptr, ln_lo=n.lo, ld_ln=d, ln_w=self.bias(ln[0] + ".weight"), ln_b=self.bias(ln[0] + ".bias"),
                              ln_eps=ln[1])
                    res.ln_copy = n
                if also_op and small and not res.bf16:
                    # the short-key kernel also leaves the stream values as an operand (hi / lo planes): returns (stream, operand)
                    o = self.op(B * Nq, d)
                    kw.update(out_op=o.ptr, out_lo=o.lo, ldo=d)
                    res.op_copy = o
                if (stream_dead and ATTN_SKIP_DEAD_STREAM and small and getattr(res, "op_copy", None) is not None
                        and getattr(res, "ln_copy", None) is not None):
                    kw["skip_act_store"] = 1
                    res.stream_skipped = True
                self.prog.emit(kind, out_act=res.ptr, ld_act=d, residual=residual.ptr if residual is not None else None,
                               ldr=residual.C if residual is not None else 0, bias=bias_ptr, act_bf16=int(res.bf16), **kw)
                return res
            o = self.op(B * Nq, d)
            self.prog.emit(kind, out_op=o.ptr, out_lo=o.lo, ldo=d, **kw)
            return o
        if Nk > 4096:
            raise _lib.FridoHipError(f"attention over {Nk} keys with head dim {d}: the flash kernel is instantiated for "
                                     "d in {128, 256, 384, 512, 576} only and the score-matrix path stops at 4096 keys")
        s = self.f32_strict(B * Nq, Nk)
        self.prog.gemm(Nq, Nk, d, (q.ptr + 2 * q_off, q.lo), (k.ptr + 2 * k_off, k.lo), batch=B, lda=ldq, ldb=ldk,
                       a_bs=Nq * ldq, b_bs=Nk * ldk, alpha=float(d) ** -0.5, out_f32=s.ptr, of_bs=Nq * Nk, ldo=Nk)
        p = self.softmax(s, B * Nq, Nk, Nk, Np)
        s.free()
        if stream:
            res = self.f32(B * Nq, d)
            kw = {}
            if residual is not None:
                kw.update(residual=residual.ptr, ldr=residual.C, res_bs=Nq * residual.C, res_bf16=getattr(residual, "bf16", False))
            self.prog.gemm(Nq, d, Np, p, vT, batch=B, lda=Np, ldb=Np, a_bs=Nq * Np, b_bs=d * Np, bias=bias_ptr, out_f32=res.ptr,
                           of_bs=Nq * d, ldo=d, out_bf16=res.bf16, **kw)
            p.free()
            return res
        o = self.op(B * Nq, d)
        self.prog.gemm(Nq, d, Np, p, vT, batch=B, lda=Np, ldb=Np, a_bs=Nq * Np, b_bs=d * Np, out_op=o.ptr,
                       oo_bs=Nq * d, ldoo=d, oo_lo=o.lo)
        p.free()
        return o

    @staticmethod
    def _flash_ln_ok(d):
        L = _lib.lib()
        return bool(L.frido_attn_flash_ln_supported(d)) if hasattr(L, "frido_attn_flash_ln_supported") else d in (256, 384)

    def v_transposed(self, x, ldx, wop, B, Nk, d, *, bias_ptr=None, out=None, x_off=0):
        """vT[z][d][Nk_pad] = (W_v @ x[z]^T): the value projection written transposed so that PV is an
        NT GEMM.  `out` must be a zero-initialised persistent operand (pad columns stay zero)."""
        Np = rup(Nk, 32)
        vT = out or self.persistent_op(d, Np, batch=B, zero=True)
        # bias of a transposed projection is per ROW (per output channel)
        self.prog.gemm(d, Nk, wop.K, wop, (x.ptr + 2 * x_off, x.lo), batch=B, lda=wop.K, ldb=ldx, a_bs=0, b_bs=Nk * ldx,
                       out_op=vT.ptr, oo_bs=d * Np, ldoo=Np, oo_lo=vT.lo, row_bias=bias_ptr)
        return vT
