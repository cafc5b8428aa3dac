"""Runtimes that own compiled HIP programs for a model instance.

  DenoiserRuntime  — PyUNetModel.forward(x, t, context, stage) on the HIP engine (API path);
  SamplerEngine    — the multi-stage DDIM / PLMS loop with per-sample invariants hoisted, the per-step
                     body captured in a hipGraph, a device step counter and coefficient tables
                     (reference: frido/models/diffusion/ddim.py:116-273, plms.py:116-303);
  DecoderRuntime   — VQModelInterface.decode / decode_first_stage on the HIP engine.
"""
import os

import numpy as np
import torch

from . import _lib, config
from .builder import Builder
from .engine import current_stream_ptr, require_gpu
from .schedules import sampler_coef_table
from .unet_plan import UNetStagePlan
from .vqgan_plan import VQDecodePlan, VQEncodePlan

# step bodies per captured DDIM graph (1 = one graph launch per step, the r01-r05 form; r06 default 20: 10 measured +0.17 %, 40 +0.28 % end to end, interleaved,
# profiles/r06_graph_steps_ab.txt); see SamplerEngine._ddim_stage
GRAPH_STEPS = max(1, int(os.environ.get("FRIDO_GRAPH_STEPS", "20")))


def _weights_of(module, device):
    return {k: v.detach().to(device=device, dtype=torch.float32).contiguous() for k, v in module.state_dict().items()}


def _run1(builder, kind, stream, **kw):
    """Launch a single op immediately."""
    from .engine import Prog
    p = Prog(builder.device, builder.nsplit)
    p.emit(kind, **kw)
    p.run(stream)


class DenoiserRuntime:
    def __init__(self, module, cfg, device, precision=None):
        self.planes = config.planes(precision)      # (r05) which build of the library this model runs on: see _lib.use_planes
        with _lib.use_planes(self.planes):
            self.device = require_gpu(device)
        self.cfg = cfg
        self.nsplit = config.nsplit(precision)
        self.b = Builder(self.device, self.nsplit, _weights_of(module, self.device), planes=self.planes)
        self.plans = {}
        self._replicas = {0: self.b}

    def builder_for(self, replica):
        """Builder of sampler replica `replica`.  Replica 0 is the runtime's own; further replicas get their OWN activation
        pool, persistent buffers and split-K workspace (they share the f32 weight tensors, not the scratch), so that their
        captured step bodies can be replayed CONCURRENTLY on different streams (only tools/dual_stream_exp.py does that: the
        measured gain of two concurrent sub-batches was +1 %, so pipeline.sample_images runs ONE replica)."""
        if replica not in self._replicas:
            self._replicas[replica] = Builder(self.device, self.nsplit, self.b.w, ws_tag=f":r{replica}", planes=self.planes)
        return self._replicas[replica]

    @_lib.with_planes
    def forward(self, x, t, context, stage):
        """x (B, Cin, H, W) f32 cuda NCHW, t (B,) int64, context (B, nctx, cd) -> eps (B, nch, H, W)."""
        B, Cin, H, W = x.shape
        nctx = context.shape[1]
        stage = 0 if stage is None else int(stage)
        key = (B, H, W, nctx, stage, Cin)
        st = current_stream_ptr(self.device)
        if key not in self.plans:
            x_state = torch.zeros(B, H * W, Cin, dtype=torch.float32, device=self.device)
            self.plans[key] = UNetStagePlan(self.b, self.cfg, B=B, H=H, W=W, nctx=nctx, stage=stage, x_state=x_state,
                                            temb_rows=B, per_sample_t=True)
        plan = self.plans[key]
        xc = x.contiguous().float()
        _run1(self.b, "FRIDO_OP_RELAYOUT", st, src=xc.data_ptr(), dst=plan.x_state.data_ptr(), B=B, HW=H * W, Csrc=Cin,
              c0=0, Cuse=Cin, Cdst=Cin, d0=0, to_nchw=0)
        plan.set_context(context.to(torch.float32))
        plan.set_timesteps(t.to(torch.int64))
        plan.pre.run(st)
        plan.step.run(st)
        out = torch.empty(B, plan.nch, H, W, dtype=torch.float32, device=self.device)
        _run1(self.b, "FRIDO_OP_RELAYOUT", st, src=plan.eps.data_ptr(), dst=out.data_ptr(), B=B, HW=H * W, Csrc=plan.nch,
              c0=0, Cuse=plan.nch, Cdst=plan.nch, d0=0, to_nchw=1)
        return out


class SamplerEngine:
    """One instance per (denoiser weights, B, latent shape, context length, S, eta, cfg on/off, kind)."""

    def __init__(self, builder: Builder, cfg, **kw):
        self.planes = builder.planes
        with _lib.use_planes(self.planes):
            self._init(builder, cfg, **kw)

    def _init(self, builder: Builder, cfg, *, B, C, H, W, nctx, S, eta, kind, alphas_cumprod, embed_dim, cfg_scale=1.0,
              use_graph=True, num_stage=None, temperature=1.0):
        self.b, self.cfg = builder, cfg
        self.dev = builder.device
        self.B, self.C, self.H, self.W, self.nctx = B, C, H, W, nctx
        self.kind = kind
        self.cfg_scale = float(cfg_scale)
        self.xrep = 2 if self.cfg_scale != 1.0 else 1
        self.embed = list(embed_dim)
        self.num_stage = num_stage if num_stage is not None else cfg.get("num_stage", 1)
        self.use_graph = use_graph
        self.temperature = float(temperature)
        tab, self.t_loop = sampler_coef_table(np.asarray(alphas_cumprod, dtype=np.float32), S, eta, plms=(kind == "plms"))
        self.n_steps = tab.shape[0]
        self.coef = torch.from_numpy(tab).to(self.dev)
        self.step = torch.zeros(1, dtype=torch.int32, device=self.dev)
        self.rng = torch.zeros(2, dtype=torch.int64, device=self.dev)   # {seed, sample0} read by the captured kernels
        self.cfg_dev = torch.ones(1, dtype=torch.float32, device=self.dev)   # guidance scale read by the captured kernels
        if not cfg.get("use_split_head", False):
            raise NotImplementedError("SamplerEngine: the multi-stage loop masks channels per stage, which needs "
                                      "use_split_head=True (every shipped Frido config)")
        self.x = torch.zeros(B, H * W, C, dtype=torch.float32, device=self.dev)
        self.pred_x0 = torch.zeros_like(self.x)
        self.stages = []
        with self.b.persist_scope() as owned:       # this engine owns its plans' persistent buffers: evicting it frees them
            for s in range(self.num_stage):
                plan = UNetStagePlan(self.b, cfg, B=B, H=H, W=W, nctx=nctx, stage=s, x_state=self.x, temb_rows=self.n_steps,
                                     per_sample_t=False, step_ptr=self.step.data_ptr(), xrep=self.xrep)
                self.stages.append(plan)
        self._persist = owned
        self.graphs = {}
        self.noise_buf = None
        self._stream = None
        # PLMS state
        if kind == "plms":
            nmax = max(self.embed[:self.num_stage])
            self.hist_stride = B * H * W * nmax
            self.hist = torch.zeros(4, self.hist_stride, dtype=torch.float32, device=self.dev)    # eps ring, slot = step & 3
            self.x_save = torch.zeros_like(self.x)

    # ---- helpers ---------------------------------------------------------------------------------
    def _stream_ptr(self):
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=self.dev)
        return self._stream

    def _sampler_op(self, s, *, noise_ptr, noise_C, seed, sample0, write_x=1, x_out=None, eps_out=None, hist=(),
                    row_offset=0, hist_mode=0, no_cfg=False):
        plan = self.stages[s]
        start = sum(self.embed[:s])
        nch = self.embed[s]
        BHW = self.B * self.H * self.W
        kw = dict(x=self.x.data_ptr(), B=self.B, HW=self.H * self.W, Cx=self.C, start=start, nch=nch,
                  eps_cond=plan.eps.data_ptr(), cfg_scale=self.cfg_scale, coef=self.coef.data_ptr(),
                  step=self.step.data_ptr(), coef_row_offset=row_offset, temperature=self.temperature,
                  x_out=(x_out if x_out is not None else self.x.data_ptr()), pred_x0=self.pred_x0.data_ptr(),
                  write_x=write_x, seed=seed, sample0=sample0, rng_stream=s + 1, rng_dev=self.rng.data_ptr(),
                  cfg_dev=self.cfg_dev.data_ptr())
        if hist_mode:
            kw.update(hist_ring=self.hist.data_ptr(), hist_stride=self.hist_stride, hist_mode=hist_mode)
        if self.xrep == 2 and not no_cfg:
            kw["eps_uncond"] = plan.eps.data_ptr() + 4 * BHW * nch
        if noise_ptr:
            kw.update(noise=noise_ptr, noise_stride=BHW * noise_C, noise_C=noise_C, noise_c0=start)
        if eps_out is not None:
            kw["eps_out"] = eps_out
        for i, h in enumerate(hist):
            kw[f"hist{i + 1}"] = h
        return kw

    def _upload_noise(self, s, tape):
        """tape: list of per-step NCHW tensors (B, 3(s+1), H, W) in draw order -> the stage's persistent NHWC device buffer
        (fixed address: the captured step body of the tape mode reads it by step index)."""
        Cs = sum(self.embed[:s + 1])
        t = torch.stack([torch.as_tensor(n, dtype=torch.float32) for n in tape])     # [n][B][Cs][H][W]
        assert t.shape[1:] == (self.B, Cs, self.H, self.W), (t.shape, Cs)
        bufs = self.__dict__.setdefault("_tape_bufs", {})
        if s not in bufs:
            bufs[s] = torch.empty(t.shape[0], self.B, self.H, self.W, Cs, dtype=torch.float32, device=self.dev)
        bufs[s].copy_(t.permute(0, 1, 3, 4, 2), non_blocking=False)                 # plumbing: layout + H2D
        return bufs[s].data_ptr(), Cs

    # ---- main entry --------------------------------------------------------------------------------
    @torch.no_grad()
    @_lib.with_planes
    def run(self, cond, uncond=None, *, x_T=None, noise="philox", seed=0, sample0=0, log_every_t=100, callback=None,
            img_callback=None, noise_dropout=0.0, score_corrector=None, corrector_kwargs=None, model=None):
        """Runs all stages.  noise: "philox" (device counter RNG), "torch" (draw from torch's global CPU generator in
        exactly the reference's order -- the stream of a reference run on CPU with the same torch.manual_seed), or a
        callable shape -> tensor replaying a recorded tape.  A supplied x_T is, like in the reference (ddim.py:150-152,
        plms.py:150-152), taken as the FINISHED stage-0 result: stage 0 and its pooling hand-off are skipped (with one
        stage x_T comes back unchanged).
        noise_dropout (ddim.py:260-262; plms.py get_x_prev_and_pred_x0): F.dropout of the update's noise -- its keep mask comes from
        torch's generator right after the randn, so it exists in the host-noise modes only ("torch" / a tape).
        score_corrector (ddim.py:228-230): an arbitrary Python hook between the denoiser and the update -- the DDIM stage then runs
        step by step on the stream (forward program, hook on torch tensors, update kernel) instead of replaying a captured graph.
        Returns (samples NCHW, intermediates dict)."""
        B, C, H, W = self.B, self.C, self.H, self.W
        self._opts = dict(noise_dropout=float(noise_dropout), score_corrector=score_corrector, corrector_kwargs=dict(corrector_kwargs or {}),
                          model=model, cond=cond, uncond=uncond)
        if noise_dropout > 0. and noise == "philox":
            raise NotImplementedError("noise_dropout draws its keep mask from torch's generator: use noise='torch' (or a recorded tape)")
        stream = self._stream_ptr()
        stream.wait_stream(torch.cuda.current_stream(self.dev))
        sp = stream.cuda_stream
        draw = None
        if noise == "torch":
            draw = lambda shape: torch.randn(shape)
        elif callable(noise):
            draw = noise
        with torch.cuda.stream(stream):
            ctx = cond.to(self.dev, torch.float32)
            if self.xrep == 2:
                ctx = torch.cat([ctx, uncond.to(self.dev, torch.float32)], dim=0)
            self.cfg_dev.fill_(self.cfg_scale)
            # ---- x_T (ddim.py:127-130) ----
            if x_T is not None:
                xt = torch.as_tensor(x_T, dtype=torch.float32)
            elif draw is not None:
                xt = draw((B, C, H, W))
            else:
                xt = None
            if xt is not None:
                xd = xt.to(self.dev).contiguous()
                _run1(self.b, "FRIDO_OP_RELAYOUT", sp, src=xd.data_ptr(), dst=self.x.data_ptr(), B=B, HW=H * W, Csrc=C,
                      c0=0, Cuse=C, Cdst=C, d0=0, to_nchw=0)
            else:
                _run1(self.b, "FRIDO_OP_RANDN", sp, dst=self.x.data_ptr(), n=B * H * W * C, per_sample=H * W * C, seed=seed,
                      sample0=sample0, rng_stream=0)
            x0_nchw = self._to_nchw(self.x, sp)
            inter = {"x_inter": [x0_nchw], "pred_x0": [x0_nchw]}
            t_loop = torch.from_numpy(self.t_loop.astype(np.int64)).to(self.dev)
            n = self.n_steps
            for s in range(self.num_stage):
                if x_T is not None and s == 0:
                    continue                 # ddim.py:150-152: "Auto adopt x_T into stage 0" (no denoising, no hand-off)
                plan = self.stages[s]
                Cs = sum(self.embed[:s + 1])
                plan.set_context(ctx)
                plan.set_timesteps(t_loop)
                self.step.zero_()
                plan.pre.run(sp)
                if self.kind == "ddim":
                    self._ddim_stage(s, sp, draw, seed, sample0, inter, log_every_t, callback, img_callback, Cs)
                else:
                    self._plms_stage(s, sp, draw, seed, sample0, inter, log_every_t, callback, img_callback, Cs)
                if self.num_stage != 1:
                    levels = self.num_stage - s - 1
                    if levels > 0:
                        c0, c1 = sum(self.embed[:s]), sum(self.embed[:s + 1])
                        _run1(self.b, "FRIDO_OP_HANDOFF", sp, x=self.x.data_ptr(), B=B, H=H, W=W, Cx=C, c0=c0, c1=c1,
                              levels=levels)
                        # the reference mutates the logged tensor in place (ddim.py:185): mirror that
                        if inter["x_inter"] and getattr(self, "_last_logged_stage", None) == s:
                            inter["x_inter"][-1] = self._to_nchw(self.x, sp)[:, :Cs]
            out = self._to_nchw(self.x, sp)
        torch.cuda.current_stream(self.dev).wait_stream(stream)
        return out, inter

    def _to_nchw(self, nhwc, sp):
        B, H, W = self.B, self.H, self.W
        C = nhwc.shape[-1]
        out = torch.empty(B, C, H, W, dtype=torch.float32, device=self.dev)
        _run1(self.b, "FRIDO_OP_RELAYOUT", sp, src=nhwc.data_ptr(), dst=out.data_ptr(), B=B, HW=H * W, Csrc=C, c0=0,
              Cuse=C, Cdst=C, d0=0, to_nchw=1)
        return out

    def _log(self, s, i, inter, log_every_t, sp, Cs, callback, img_callback):
        n = self.n_steps
        index = n - i - 1
        if callback:
            callback(i)
        if img_callback:
            img_callback(self._to_nchw(self.pred_x0, sp)[:, :Cs], i)
        if index % log_every_t == 0 or index == n - 1:
            inter["x_inter"].append(self._to_nchw(self.x, sp)[:, :Cs])
            inter["pred_x0"].append(self._to_nchw(self.pred_x0, sp)[:, :Cs])
            self._last_logged_stage = s if i == n - 1 else None

    def _ddim_stage(self, s, sp, draw, seed, sample0, inter, log_every_t, callback, img_callback, Cs):
        """ddim.py:155-175.  ONE captured hipGraph per stage (denoiser forward + state update + step-counter bump) is
        replayed every step: the Philox form draws its noise in the update kernel, the tape form (recorded / torch-CPU noise)
        reads the stage's persistent noise buffer by step index."""
        from .engine import Prog
        plan = self.stages[s]
        n = self.n_steps
        opts = getattr(self, "_opts", {})
        p_drop = opts.get("noise_dropout", 0.0)
        if draw is not None and p_drop > 0.:
            base = draw
            # dropout(sigma * noise * temperature) = sigma * temperature * dropout(noise): applied to the host tape, mask drawn right
            # after the step's randn like the reference does
            draw = lambda shape: torch.nn.functional.dropout(base(shape), p=p_drop)
        if opts.get("score_corrector") is not None:
            return self._ddim_stage_with_corrector(s, sp, draw, seed, sample0, inter, log_every_t, callback, img_callback, Cs, opts)
        if draw is not None:
            noise_ptr, noise_C = self._upload_noise(s, [draw((self.B, Cs, self.H, self.W)) for _ in range(n)])
            key = ("ddim_tape", s)
            if key not in self.graphs:
                body = Prog(self.dev, self.b.nsplit)
                body.ops = list(plan.step.ops)
                body.emit("FRIDO_OP_SAMPLER_STEP", **self._sampler_op(s, noise_ptr=noise_ptr, noise_C=noise_C, seed=0, sample0=0))
                body.emit("FRIDO_OP_STEP_ADD", step=self.step.data_ptr(), delta=1)
                body.keep = [plan]
                self.graphs[key] = body.capture(sp) if self.use_graph else body
            g = self.graphs[key]
            launch = (lambda: g.launch(sp)) if self.use_graph else (lambda: g.run(sp))
        else:
            self.rng.copy_(torch.tensor([seed, sample0], dtype=torch.int64))
            key = ("ddim", s)
            if key not in self.graphs:
                full = Prog(self.dev, self.b.nsplit)
                full.ops = list(plan.step.ops)
                full.emit("FRIDO_OP_SAMPLER_STEP", **self._sampler_op(s, noise_ptr=None, noise_C=0, seed=0, sample0=0))
                full.emit("FRIDO_OP_STEP_ADD", step=self.step.data_ptr(), delta=1)
                full.keep = [plan]
                self.graphs[key] = full.capture(sp) if self.use_graph else full
            g = self.graphs[key]
            launch = (lambda: g.launch(sp)) if self.use_graph else (lambda: g.run(sp))
        # (r06) GRAPH_STEPS > 1: the same step body K times in ONE captured graph (the device step counter makes every repetition pick its own
        # timestep / noise slice), replayed wherever the K - 1 steps in between need no host access (log / callbacks): fewer graph launches --
        # the trace shows ~30 us between the last kernel of one replay and the first of the next (profiles/r05_x3_gap_analysis.json)
        K = GRAPH_STEPS if self.use_graph else 1
        gk = None
        if K > 1 and n >= K and callback is None and img_callback is None:
            kkey = key + ("x%d" % K,)
            if kkey not in self.graphs:
                from .engine import Prog
                multi = Prog(self.dev, self.b.nsplit)
                multi.ops = list(g.keep[1].ops) * K          # (Graph.keep = (packed descriptor array, the Prog it was captured from))
                multi.keep = [plan]
                self.graphs[kkey] = multi.capture(sp)
            gk = self.graphs[kkey]
        needs_host = lambda i: (n - i - 1) % log_every_t == 0 or i == 0              # when _log touches the state at step i (index = n - i - 1) without callbacks
        i = 0
        while i < n:
            if gk is not None and i + K <= n and not any(needs_host(j) for j in range(i, i + K - 1)):
                gk.launch(sp)
                self.multi_step_launches = getattr(self, "multi_step_launches", 0) + 1      # (tests: the K-step graph really ran)
                i += K
            else:
                launch()
                i += 1
            self._log(s, i - 1, inter, log_every_t, sp, Cs, callback, img_callback)

    def _corrected_eps(self, s, sp, Cs, t_value, opts):
        """`score_corrector.modify_score(model, e_t, x, t, c, **kwargs)` (ddim.py:228-230, plms.py:236-238) on the eps the forward
        program just wrote: CFG mix (ddim.py:226), frozen channels zero-padded like the reference's e_t, hook on torch tensors
        (NCHW), active channels written back as the (already mixed) conditional eps the update kernel reads."""
        plan = self.stages[s]
        B, H, W = self.B, self.H, self.W
        start, nch = sum(self.embed[:s]), self.embed[s]
        e = plan.eps.view(self.xrep, B, H, W, nch).permute(0, 1, 4, 2, 3)            # [cond | uncond] x (B, nch, H, W)
        e_t = e[0]
        if self.xrep == 2:
            e_t = e[1] + self.cfg_scale * (e_t - e[1])
        e_t = torch.cat((torch.zeros(B, start, H, W, device=self.dev), e_t), dim=1) if start else e_t.contiguous()
        x_now = self._to_nchw(self.x, sp)[:, :Cs]
        t = torch.full((B,), int(t_value), device=self.dev, dtype=torch.long)
        e_new = opts["score_corrector"].modify_score(opts["model"], e_t, x_now, t, opts["cond"], **opts["corrector_kwargs"])
        e_new = e_new.to(torch.float32).contiguous()
        assert e_new.shape == (B, Cs, H, W), "modify_score must return a tensor of e_t's shape"
        _run1(self.b, "FRIDO_OP_RELAYOUT", sp, src=e_new.data_ptr(), dst=plan.eps.data_ptr(), B=B, HW=H * W, Csrc=Cs, c0=start,
              Cuse=nch, Cdst=nch, d0=0, to_nchw=0)

    def _ddim_stage_with_corrector(self, s, sp, draw, seed, sample0, inter, log_every_t, callback, img_callback, Cs, opts):
        """ddim.py:188-273 with the score corrector between the (CFG-mixed) eps and the update: eager, one step at a time --
        forward program, hook, update kernel."""
        from .engine import Prog
        plan = self.stages[s]
        n, B, H, W = self.n_steps, self.B, self.H, self.W
        if draw is not None:
            noise_ptr, noise_C = self._upload_noise(s, [draw((B, Cs, H, W)) for _ in range(n)])
        else:
            noise_ptr, noise_C = None, 0
            self.rng.copy_(torch.tensor([seed, sample0], dtype=torch.int64))
        upd = Prog(self.dev, self.b.nsplit)
        upd.emit("FRIDO_OP_SAMPLER_STEP", **self._sampler_op(s, noise_ptr=noise_ptr, noise_C=noise_C, seed=0, sample0=0, no_cfg=True))
        upd.emit("FRIDO_OP_STEP_ADD", step=self.step.data_ptr(), delta=1)
        t_steps = self.t_loop.astype(np.int64)
        for i in range(n):
            plan.step.run(sp)
            self._corrected_eps(s, sp, Cs, t_steps[i], opts)
            upd.run(sp)
            self._log(s, i, inter, log_every_t, sp, Cs, callback, img_callback)

    def _plms_stage_with_corrector(self, s, sp, draw, seed, sample0, inter, log_every_t, callback, img_callback, Cs, opts):
        """plms.py:156-194,198-303 with the score corrector inside every model evaluation (plms.py:236-238): the op sequences of
        the two captured PLMS programs, run eagerly with the hook after each forward.  The corrected eps is what enters the
        Adams-Bashforth history ring (plms.py:175-177 appends get_model_output's result)."""
        from .engine import Prog
        plan = self.stages[s]
        n = self.n_steps
        self.rng.copy_(torch.tensor([seed, sample0], dtype=torch.int64))
        nbytes = self.x.numel() * 4

        def prog(*ops):
            p = Prog(self.dev, self.b.nsplit)
            for kind, kw in ops:
                p.emit(kind, **kw)
            return p
        step_add = lambda d: ("FRIDO_OP_STEP_ADD", dict(step=self.step.data_ptr(), delta=d))
        upd = lambda mode: ("FRIDO_OP_SAMPLER_STEP", self._sampler_op(s, noise_ptr=None, noise_C=0, seed=0, sample0=0, hist_mode=mode, no_cfg=True))
        save = ("FRIDO_OP_COPY", dict(src=self.x.data_ptr(), dst=self.x_save.data_ptr(), n=nbytes))
        restore = ("FRIDO_OP_COPY", dict(src=self.x_save.data_ptr(), dst=self.x.data_ptr(), n=nbytes))
        first_a = prog(save, upd(1), *([step_add(1)] if n > 1 else []))            # ... then the forward at (x_prev, t_next)
        first_b = prog(*([step_add(-1)] if n > 1 else []), restore, upd(3), step_add(1))
        body = prog(upd(1), step_add(1))
        t_steps = self.t_loop.astype(np.int64)
        p_drop = opts.get("noise_dropout", 0.0)
        for i in range(n):
            if draw is not None:          # eta == 0: the reference still draws (and discards) noise for every update
                for _ in range(2 if i == 0 else 1):
                    nz = draw((self.B, Cs, self.H, self.W))
                    if p_drop > 0.:
                        torch.nn.functional.dropout(nz, p=p_drop)
            plan.step.run(sp)
            self._corrected_eps(s, sp, Cs, t_steps[i], opts)
            if i == 0:
                first_a.run(sp)
                plan.step.run(sp)
                self._corrected_eps(s, sp, Cs, t_steps[min(i + 1, n - 1)], opts)      # t_next (plms.py:164-166)
                first_b.run(sp)
            else:
                body.run(sp)
            self._log(s, i, inter, log_every_t, sp, Cs, callback, img_callback)

    def _plms_stage(self, s, sp, draw, seed, sample0, inter, log_every_t, callback, img_callback, Cs):
        """plms.py:156-194,285-303: Heun-style first step (two denoiser calls), then Adams-Bashforth 2/3/4.  Two captured
        hipGraphs per stage serve every step: `first` = [forward, save x, update with e_t (eps -> ring slot 0), step+1,
        forward at (x_prev, t_next), step-1, restore x, update with (e_t + e_next)/2, step+1]; `body` = [forward, update with
        the ring history selected by the device step counter, step+1].  eta == 0 (plms.py:25-26), so no noise enters the
        update and the same graphs serve the philox / torch / tape modes."""
        from .engine import Prog
        opts = getattr(self, "_opts", {})
        if opts.get("score_corrector") is not None:
            return self._plms_stage_with_corrector(s, sp, draw, seed, sample0, inter, log_every_t, callback, img_callback, Cs, opts)
        plan = self.stages[s]
        n = self.n_steps
        self.rng.copy_(torch.tensor([seed, sample0], dtype=torch.int64))
        nbytes = self.x.numel() * 4

        def build(first):
            p = Prog(self.dev, self.b.nsplit)
            p.ops = list(plan.step.ops)
            if first:
                p.emit("FRIDO_OP_COPY", src=self.x.data_ptr(), dst=self.x_save.data_ptr(), n=nbytes)
                p.emit("FRIDO_OP_SAMPLER_STEP", **self._sampler_op(s, noise_ptr=None, noise_C=0, seed=0, sample0=0, hist_mode=1))
                if n > 1:
                    p.emit("FRIDO_OP_STEP_ADD", step=self.step.data_ptr(), delta=1)
                p.ops += list(plan.step.ops)
                if n > 1:
                    p.emit("FRIDO_OP_STEP_ADD", step=self.step.data_ptr(), delta=-1)
                p.emit("FRIDO_OP_COPY", src=self.x_save.data_ptr(), dst=self.x.data_ptr(), n=nbytes)
                p.emit("FRIDO_OP_SAMPLER_STEP", **self._sampler_op(s, noise_ptr=None, noise_C=0, seed=0, sample0=0, hist_mode=3))
            else:
                p.emit("FRIDO_OP_SAMPLER_STEP", **self._sampler_op(s, noise_ptr=None, noise_C=0, seed=0, sample0=0, hist_mode=1))
            p.emit("FRIDO_OP_STEP_ADD", step=self.step.data_ptr(), delta=1)
            p.keep = [plan]
            return p.capture(sp) if self.use_graph else p

        for name, first in (("plms_first", True), ("plms_body", False)):
            if (name, s) not in self.graphs:
                self.graphs[(name, s)] = build(first)
        g_first, g_body = self.graphs[("plms_first", s)], self.graphs[("plms_body", s)]
        go = (lambda g: g.launch(sp)) if self.use_graph else (lambda g: g.run(sp))
        for i in range(n):
            # eta == 0: the reference still draws (and discards) noise for every update; keep a replayed stream in step
            if draw is not None:
                p_drop = getattr(self, "_opts", {}).get("noise_dropout", 0.0)
                for _ in range(2 if i == 0 else 1):
                    nz = draw((self.B, Cs, self.H, self.W))
                    if p_drop > 0.:
                        torch.nn.functional.dropout(nz, p=p_drop)      # (plms.py get_x_prev_and_pred_x0: the mask draw consumes generator state)
            go(g_first if i == 0 else g_body)
            self._log(s, i, inter, log_every_t, sp, Cs, callback, img_callback)


class DecoderRuntime:
    def __init__(self, module, vq_cfg, device, precision=None):
        self.planes = config.planes(precision)
        with _lib.use_planes(self.planes):
            self.device = require_gpu(device)
        self.cfg = vq_cfg
        self.nsplit = config.nsplit(precision)
        self.b = Builder(self.device, self.nsplit, _weights_of(module, self.device), planes=self.planes)
        self.plans = {}

    U8_MODES = {False: 0, None: 0, True: 1, "np": 1, "pil": 2}

    @_lib.with_planes
    def decode(self, z, inv_scale=None, return_code=False, to_uint8=False, force_codes=None):
        """z (B, Ctot, h, w) NCHW latent -> image (B, 3, H, W); inv_scale: per-scale multiplier (1/scale_factor).
        to_uint8: True / "np" -> the (B, H, W, 3) uint8 array of scripts/sample_diffusion.py:115-121 (custom_to_np), "pil" -> the
        pixel bytes of :103-113 (custom_to_pil's truncating conversion), written by the LAST conv's epilogue (r04: no f32
        image, 4x less to gather / write)."""
        B, Ct, h, w = z.shape
        embed = self.cfg["embed_dim"]
        inv = tuple(float(v) for v in (inv_scale if inv_scale is not None else [1.0] * len(embed)))
        u8 = self.U8_MODES[to_uint8]
        key = (B, h, w, inv, force_codes is not None, u8)
        st = current_stream_ptr(self.device)
        if key not in self.plans:
            z_state = torch.zeros(B, h * w, Ct, dtype=torch.float32, device=self.device)
            self.plans[key] = (z_state, VQDecodePlan(self.b, self.cfg["ddconfig"], embed, self.cfg["n_embed"], B=B, h=h, w=w,
                                                      z_state=z_state, inv_scale=inv, forced=force_codes is not None, u8_mode=u8))
        z_state, plan = self.plans[key]
        if force_codes is not None:      # test hook: decode the given per-scale code maps instead of the argmin's
            for dst, src in zip(plan.force_idx, force_codes):
                dst.copy_(torch.as_tensor(src, dtype=torch.int64).reshape(-1))
        zc = z.contiguous().float()
        _run1(self.b, "FRIDO_OP_RELAYOUT", st, src=zc.data_ptr(), dst=z_state.data_ptr(), B=B, HW=h * w, Csrc=Ct, c0=0,
              Cuse=Ct, Cdst=Ct, d0=0, to_nchw=0)
        plan.prog.run(st)
        if u8:
            img = plan.out_u8.view(B, plan.H, plan.W, plan.a.out_ch).clone()
            return (img, [i.view(B, -1) for i in plan.idx]) if return_code else img
        out = torch.empty(B, plan.a.out_ch, plan.H, plan.W, dtype=torch.float32, device=self.device)
        _run1(self.b, "FRIDO_OP_RELAYOUT", st, src=plan.out_nhwc.data_ptr(), dst=out.data_ptr(), B=B, HW=plan.H * plan.W,
              Csrc=plan.a.out_ch, c0=0, Cuse=plan.a.out_ch, Cdst=plan.a.out_ch, d0=0, to_nchw=1)
        if return_code:
            return out, [i.view(B, -1) for i in plan.idx]
        return out

    @_lib.with_planes
    def encode(self, x, scale=None):
        """x (B, 3, H, W) NCHW image -> pre-quantisation latent (B, sum(embed), H/f, W/f); `scale` (per scale) folds
        get_first_stage_encoding's multiply in."""
        B, Cin, H, W = x.shape
        embed = self.cfg["embed_dim"]
        sc = tuple(float(v) for v in (scale if scale is not None else [1.0] * len(embed)))
        key = ("enc", B, H, W, sc)
        st = current_stream_ptr(self.device)
        if key not in self.plans:
            x_in = torch.zeros(B, Cin, H, W, dtype=torch.float32, device=self.device)
            self.plans[key] = (x_in, VQEncodePlan(self.b, self.cfg, B=B, H=H, W=W, x_in=x_in, scale=sc))
        x_in, plan = self.plans[key]
        x_in.copy_(x)            # plumbing: D2D copy into the plan's fixed input buffer
        plan.prog.run(st)
        return plan.out.clone()
