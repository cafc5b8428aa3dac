"""TIMING EXPERIMENTS ONLY -- not part of the product (frido_amd/ has no such switch).

FRIDO_DEBUG_SKIP=KIND[,KIND...] (e.g. GN_APPLY,LAYERNORM, or GEMM:conv / GEMM:dense) drops those ops from every program
before it is captured into a hipGraph.  The results are garbage; what is measured is what a bucket of launches costs INSIDE the
replayed graph, which per-op event timing overstates (tools/skip_costs.sh).  `bench.py --allow-debug` calls install(); the
line it prints then carries "debug_work_skipped": true and is not a benchmark result."""
import os


def install():
    skip = {k for k in os.environ.get("FRIDO_DEBUG_SKIP", "").split(",") if k}
    if not skip:
        return False
    from frido_amd import _lib, engine
    names = {v: k[len("FRIDO_OP_"):] for k, v in _lib.OP_KINDS.items()}
    real_capture = engine.Prog.capture

    def capture(self, stream):
        kept = []
        for kind, st in self.ops:
            tags = {names[kind]}
            if names[kind] == "GEMM":
                tags.add("GEMM:conv" if st.conv else "GEMM:dense")
            if not tags & skip:
                kept.append((kind, st))
        self.ops, self._packed = kept, None
        return real_capture(self, stream)

    engine.Prog.capture = capture
    return True
