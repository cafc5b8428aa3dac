"""Host side of the HIP engine: device buffers (torch tensors are used purely as HBM allocations),
weight packing into MFMA operand layouts, and a program builder that emits FridoOp descriptors for
the native executor (frido_run / frido_graph_capture in libfrido_hip.so).

Nothing here computes on the CPU and nothing falls back to torch ops: every tensor that reaches the
model output is produced by a kernel of libfrido_hip.so.
"""
import ctypes as C

import torch

from . import _lib

import contextlib
import os
SIDE_STREAM = os.environ.get("FRIDO_SIDE_STREAM", "0") != "0"   # independent projections of an attention block on a side stream: measured -2.9 % (the fork / join nodes cost more than the overlap buys), off by default
GEMM_FLAGS = int(os.environ.get("FRIDO_GEMM_FLAGS", "0"))     # FridoGemm.flags A/B switches (include/frido_hip.h)
# Staggered start (r05 experiment, SHIPPED in r06: the library is built with -DFRIDO_STAGGER_RT=1; results are
# unchanged bit for bit): start delay in microseconds of the second resident slot (dispatch ids 256 .. 511) of a multi-round two-per-CU
# GEMM launch (>= 768 workgroups unless FRIDO_STAGGER_MIN_WG says otherwise), so that one slot's prologue / epilogue runs under the
# other's k-loop from then on.  12 us: + 2.1 % end to end (interleaved, profiles/r06_stagger_*.txt; 8 us on the r05 boxes: + 2.0 ... 2.5 %).
STAGGER_US = float(os.environ.get("FRIDO_STAGGER_US", "12"))         # quarter-microsecond resolution, at most 63.75; 0 = off
STAGGER_MIN_WG = int(os.environ.get("FRIDO_STAGGER_MIN_WG", "0"))
STAGGER_MODE = int(os.environ.get("FRIDO_STAGGER_MODE", "0"))        # 0: dispatch ids 256..511 wait; 1: every other workgroup of an XCD (control)
STAGGER_8W = int(os.environ.get("FRIDO_STAGGER_8W", "0"))            # 1: the one-workgroup-per-CU kernels too (odd XCDs start late; igemm_shared.h)
GEMM_FLAGS |= (((int(round(STAGGER_US * 4)) & 255) << 8) | (((STAGGER_MIN_WG // 64) & 255) << 16) | ((STAGGER_MODE & 3) << 24)
               | ((STAGGER_8W & 1) << 26))
# (r06) column-panel tile order of the ring GEMM kernel (igemm.hip; FridoGemm.flags bit 27): keeps a launch's weight panel L2-resident per XCD.
# Built on a PMC finding (the GEGLU projection fetches 162 MB per launch for 30 MB of operands, L2 hit rate 81 %: profiles/r06_pmc_l2_by_instance_*.json);
# measured: -5 % per GEGLU launch in the back-to-back microbenchmark, +0.06 % end to end (three interleaved pairs, profiles/r06_panels_*.txt) -- the L2
# misses are not what paces these k-loops.  OFF by default: the r05 tile order stays the validated one.
GEMM_PANELS = os.environ.get("FRIDO_GEMM_PANELS", "0") != "0"
GEMM_FLAGS |= (1 << 27) if GEMM_PANELS else 0
BF16X3 = 2   # nsplit: hi + residual plane, 3 MFMAs per product (≈ fp32 accuracy)
BF16 = 1     # nsplit: plain bf16 operands


def rup(x, m):
    return (x + m - 1) // m * m


def plane_dtype(nsplit):
    """torch dtype of an operand plane: bf16 for one-plane operands; for the two-plane (bf16x3 precision) operands whatever the
    library was built for (frido_x3_plane_format: fp16 hi + fp16 lo since r03, bf16 pairs before)."""
    if nsplit == 2 and _lib.lib().frido_x3_plane_format() == 1:
        return torch.float16
    return torch.bfloat16


class Operand:
    """operand matrix [rows][K] (K contiguous) of 16-bit elements with `nsplit` planes (plane_dtype); plane p starts p*lo elements in."""

    def __init__(self, rows, K, nsplit, device, zero=False, batch=1):
        self.rows, self.K, self.nsplit, self.batch = rows, K, nsplit, batch
        n = batch * rows * K
        self.lo = rup(n, 8)
        alloc = torch.zeros if zero else torch.empty
        self.t = alloc((nsplit, self.lo), dtype=plane_dtype(nsplit), device=device)

    @property
    def ptr(self):
        return self.t.data_ptr()

    def nbytes(self):
        return self.t.numel() * 2

    def to_f32(self):
        """Debug helper: hi (+ lo) planes as an f32 tensor [batch*rows, K]."""
        n = self.batch * self.rows * self.K
        v = self.t[0, :n].float()
        if self.nsplit == 2:
            v = v + self.t[1, :n].float()
        return v.view(self.batch * self.rows, self.K)


def pack_matrix(w2d, nsplit, kpad=32):
    """f32 [N][K] on device -> Operand with K zero-padded to a multiple of `kpad` (torch's RNE casts)."""
    N, K = w2d.shape
    Kp = rup(K, kpad)
    op = Operand(N, Kp, nsplit, w2d.device, zero=True)
    w = w2d.float()
    dt = plane_dtype(nsplit)
    if dt == torch.float16:
        w = w.clamp(-65504.0, 65504.0)      # saturate like the kernels' split_op (csrc/common.h): an inf plane would turn the product into NaN
    hi = w.to(dt)
    view = op.t[0, :N * Kp].view(N, Kp)
    view[:, :K] = hi
    if nsplit == 2:
        op.t[1, :N * Kp].view(N, Kp)[:, :K] = (w - hi.float()).to(dt)
    return op


def pack_conv_weight(w4d, nsplit):
    """(Cout, Cin, kh, kw) -> [Cout][kh*kw*Cin_pad] with k = (ky*kw + kx)*Cin_pad + c."""
    co, ci, kh, kw = w4d.shape
    cp = rup(ci, 32)
    w = torch.zeros((co, kh, kw, cp), dtype=torch.float32, device=w4d.device)
    w[..., :ci] = w4d.permute(0, 2, 3, 1)
    return pack_matrix(w.reshape(co, kh * kw * cp), nsplit), cp


class Graph:
    def __init__(self, handle, keep):
        self.handle, self.keep = handle, keep
        self.lib = _lib.lib()           # the build that captured the graph (r05: one per plane format) launches and destroys it

    def launch(self, stream):
        _lib.check(self.lib.frido_graph_launch(self.handle, stream), "frido_graph_launch")

    def __del__(self):
        try:
            if self.handle:
                self.lib.frido_graph_destroy(self.handle)
        except Exception:
            pass


class Pool:
    """Plan-time buffer pool: ops of one program run in order on one stream, so a buffer can be handed
    out again as soon as the builder has emitted its last reader."""

    def __init__(self, device):
        self.device = device
        self.free_bufs = {}
        self.total = 0

    def alloc(self, nbytes):
        nbytes = rup(max(nbytes, 16), 256)
        lst = self.free_bufs.get(nbytes)
        if lst:
            return lst.pop()
        self.total += nbytes
        return torch.empty(nbytes, dtype=torch.uint8, device=self.device)

    def release(self, buf):
        self.free_bufs.setdefault(buf.numel(), []).append(buf)

    def hold(self, ptr):
        """Take the FREE buffer that starts at device address `ptr` out of circulation (returns it, or None if no free buffer starts
        there): an op about to be emitted still reads it although its owner released it at plan time (builder.groupnorm with a
        deferred split-K reduction: the GEMM's residual is read by the GroupNorm launch).  release() it after the emission."""
        if not ptr:
            return None
        for lst in self.free_bufs.values():
            for i, t in enumerate(lst):
                if t.data_ptr() == ptr:
                    return lst.pop(i)
        return None


class F32:
    """Activation [rows][C] of the residual stream in a pooled buffer: f32, or (bf16 = True) a single bf16 plane that
    doubles as an MFMA operand (K = C)."""

    def __init__(self, pool, rows, C, bf16=False):
        self.rows, self.C, self.pool, self.bf16 = rows, C, pool, bf16
        self.K, self.lo, self.nsplit, self.batch = C, 0, 1, 1       # operand view (valid when bf16)
        self.buf = pool.alloc(rows * C * (2 if bf16 else 4))
        self.gn_part = None       # per-channel partial sums written by the producing GEMM's epilogue (FridoGemm.gn_part), if any

    @property
    def ptr(self):
        return self.buf.data_ptr()

    def view(self):
        dt, nb = (torch.bfloat16, 2) if self.bf16 else (torch.float32, 4)
        return self.buf[: self.rows * self.C * nb].view(dt).view(self.rows, self.C)

    def to_f32(self):
        return self.view().float()

    def free(self):
        if self.buf is not None:
            self.pool.release(self.buf)
            self.buf = None
        if self.gn_part is not None:
            self.pool.release(self.gn_part)
            self.gn_part = None


class Alias:
    """Non-owning operand view of a bf16 activation (free() is a no-op)."""

    def __init__(self, act):
        self.act = act
        self.rows, self.K, self.lo, self.nsplit, self.batch = act.rows, act.C, 0, 1, 1

    @property
    def ptr(self):
        return self.act.ptr

    def to_f32(self):
        return self.act.to_f32()

    def free(self):
        pass


class POperand:
    """Pooled operand (activation) [rows][K] with nsplit planes."""

    def __init__(self, pool, rows, K, nsplit, batch=1):
        self.rows, self.K, self.nsplit, self.batch, self.pool = rows, K, nsplit, batch, pool
        self.lo = rup(batch * rows * K, 8)
        self.buf = pool.alloc(nsplit * self.lo * 2)

    @property
    def ptr(self):
        return self.buf.data_ptr()

    def to_f32(self):
        n = self.batch * self.rows * self.K
        t = self.buf[: self.nsplit * self.lo * 2].view(plane_dtype(self.nsplit)).view(self.nsplit, self.lo)
        v = t[0, :n].float()
        if self.nsplit == 2:
            v = v + t[1, :n].float()
        return v.view(self.batch * self.rows, self.K)

    def free(self):
        if self.buf is not None:
            self.pool.release(self.buf)
            self.buf = None


class Prog:
    """A list of ops + the tensors they reference (kept alive)."""

    def __init__(self, device, nsplit, ws_tag=""):
        self.device, self.nsplit = device, nsplit
        self.ops = []
        self.keep = []
        self._packed = None
        self.flops = 0
        self._sid = 0
        self.ws_tag = ws_tag          # split-K workspace identity (programs that may run concurrently must not share one)

    # ---- emission helpers -------------------------------------------------------------------
    def emit(self, kind, **kw):
        op = _lib.make_op(kind, **kw)
        op[1]._sid = self._sid          # stream id of the native executor (0 = caller's stream, 1 = its side stream)
        self.ops.append(op)
        self._packed = None

    def sync(self, frm, to):
        """Everything emitted so far for stream `frm` happens before what is emitted later for stream `to`."""
        if SIDE_STREAM:
            self.emit("FRIDO_OP_SYNC", **{"from": frm, "to": to})

    @contextlib.contextmanager
    def side(self):
        """Ops emitted inside run on the executor's side stream, concurrently with what the caller emits for the main stream
        between `with` exit and the next sync(1, 0).  The caller brackets the region: sync(0, 1) before, sync(1, 0) after."""
        prev = self._sid
        self._sid = 1 if SIDE_STREAM else 0
        try:
            yield
        finally:
            self._sid = prev

    def gemm(self, M, N, K, A, B, *, batch=1, lda=None, ldb=None, a_bs=0, b_bs=0, a_lo=None, b_lo=None,
             bias=None, rowvec=None, rows_per_vec=0, ldv=0, rowvec_step=None, alpha=1.0, act=0,
             residual=None, res_bs=0, ldr=0, out_f32=None, of_bs=0, ldo=0, out_op=None, oo_bs=0, ldoo=0,
             oo_lo=0, tile=0, conv=None, row_bias=None, geglu=0, res_bf16=0, out_bf16=0, batch_inner=0, a_bs2=0, b_bs2=0, of_bs2=0, oo_bs2=0, A2=None, lda2=0, K2=0,
             gn_part=None, out_u8=None, ldu8=0, u8_mode=0, gn=None):
        """A/B: objects with .ptr/.lo (Operand / POperand) or (ptr, lo) tuples.  gn: the gn_* / raw_* fields of a fused GroupNorm +
        conv launch (FridoGemm tiles 20 / 21: A is then None, the operand is produced in the kernel)."""
        ap, alo = (A.ptr, A.lo) if hasattr(A, "ptr") else (A if A is not None else (None, 0))
        bp, blo = (B.ptr, B.lo) if hasattr(B, "ptr") else B
        kw = dict(M=M, N=N, K=K, batch=batch, nsplit=self.nsplit, A=ap, a_lo=alo if a_lo is None else a_lo,
                  a_bs=a_bs, lda=lda if lda is not None else K, B=bp, b_lo=blo if b_lo is None else b_lo,
                  b_bs=b_bs, ldb=ldb if ldb is not None else K, alpha=alpha, act=act, tile=tile, geglu=geglu, batch_inner=batch_inner, a_bs2=a_bs2, b_bs2=b_bs2, flags=GEMM_FLAGS,
                  of_bs2=of_bs2, oo_bs2=oo_bs2)
        if conv:
            kw.update(conv=1, **conv)
            kw.setdefault("padx", conv["pad"])
        if A2 is not None:
            kw.update(A2=A2.ptr, a2_lo=A2.lo, lda2=lda2, K2=K2)
        if bias is not None:
            kw["bias"] = bias
        if row_bias is not None:
            kw["row_bias"] = row_bias
        if rowvec is not None:
            kw.update(rowvec=rowvec, rows_per_vec=rows_per_vec, ldv=ldv)
            if rowvec_step is not None:
                kw["rowvec_step"] = rowvec_step
        if residual is not None:
            kw.update(residual=residual, res_bs=res_bs, ldr=ldr, res_bf16=int(res_bf16))
        if out_f32 is not None:
            kw.update(out_f32=out_f32, of_bs=of_bs, ldo=ldo, out_bf16=int(out_bf16))
        if out_op is not None:
            kw.update(out_op=out_op, oo_bs=oo_bs, ldoo=ldoo, oo_lo=oo_lo)
        if gn_part is not None:
            kw["gn_part"] = gn_part
        if out_u8 is not None:
            kw.update(out_u8=out_u8, ldu8=ldu8, u8_mode=u8_mode)
        if gn is not None:
            kw.update(gn)
            if K2:
                kw["K2"] = K2
        self.emit("FRIDO_OP_GEMM", **kw)
        if not tile and self.device.type == "cuda":
            from . import tune
            st = self.ops[-1][1]
            choice = tune.best_tile(st, self.device, torch.cuda.current_stream(self.device).cuda_stream)
            st.tile, st.splitk = choice[0], choice[1]
            if len(choice) > 2 and choice[2] and "FRIDO_STAGGER_US" not in os.environ:
                # (r05, for round 6) a pinned per-signature start delay in quarter microseconds (tools/tune_in_context.py --stagger):
                # FridoGemm.flags bits 8..15 -- it replaces the library-wide default delay for this signature; an explicit FRIDO_STAGGER_US
                # in the environment overrides every pinned delay
                st.flags = (st.flags & ~0xFF00) | ((int(choice[2]) & 255) << 8)
            if st.splitk > 1:
                st.sk_mode = tune.SK_MODE
                if not st.sk_mode:
                    st.gn_part = None   # the split-K reduction kernel does not produce them (the consumer falls back to gn_stats)
                # ops of the executor's side stream run CONCURRENTLY with main-stream ops: they get their own workspace
                st.ws = tune.workspace_for(st, self.device, self.ws_tag + (":s1" if self._sid else ""))
        self.flops += 2 * M * N * (K + K2) * batch * (3 if self.nsplit == 2 else 1)

    # ---- execution ---------------------------------------------------------------------------
    def packed(self):
        if self._packed is None:
            self._packed = _lib.pack_ops(self.ops)
        return self._packed

    def run(self, stream):
        arr = self.packed()
        _lib.check(_lib.lib().frido_run(C.addressof(arr), len(self.ops), stream), "frido_run")

    def run_timed(self, stream):
        """Run with a HIP event around every op on `stream`; returns per-op device milliseconds."""
        arr = self.packed()
        ms = (C.c_float * len(self.ops))()
        _lib.check(_lib.lib().frido_run_timed(C.addressof(arr), len(self.ops), stream, ms), "frido_run_timed")
        return list(ms)

    def capture(self, stream):
        arr = self.packed()
        h = C.c_void_p()
        _lib.check(_lib.lib().frido_graph_capture(C.addressof(arr), len(self.ops), stream, C.byref(h)),
                   "frido_graph_capture")
        return Graph(h, (arr, self))


def current_stream_ptr(device=None):
    return torch.cuda.current_stream(device).cuda_stream


def require_gpu(device):
    device = torch.device(device)
    if device.type != "cuda":
        raise _lib.FridoHipError(
            f"the Frido hot path runs only on an MI355X HIP device (got device '{device}'); there is no CPU path")
    _lib.check(_lib.lib().frido_init(), "frido_init")
    return device
