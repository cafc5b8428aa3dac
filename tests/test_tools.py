"""CPU tests of the measurement tools' own logic (no GPU: timers are faked)."""
import importlib.util
import os
import types

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REPO, "tools", name + ".py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _st(M, N, K, tile, **kw):
    d = dict(M=M, N=N, K=K, K2=0, batch=1, nsplit=2, conv=0, tile=tile, splitk=1, up2_phase=0, flags=0)
    d.update(kw)
    return types.SimpleNamespace(**d)


def test_in_context_tuner_keeps_only_clear_wins_and_restores_the_rest():
    tic = _load("tune_in_context")
    GEMM, OTHER = 1, 2
    prog = types.SimpleNamespace(_packed=None, ops=[
        (GEMM, _st(16384, 3072, 384, 2)), (OTHER, None),            # signature A: tile 7 is 20 % faster in context
        (GEMM, _st(4096, 576, 576, 1)), (OTHER, None),              # signature B: tile 5 wins by 0.5 % only -> unchanged
        (GEMM, _st(16384, 3072, 384, 2)), (OTHER, None),            # signature A again
        (GEMM, _st(1024, 960, 960, 3, splitk=4)),                   # split-K launch: never touched
        (GEMM, _st(65536, 192, 1728, 20)),                          # fused GroupNorm + conv tile: never touched
        (GEMM, _st(32, 960, 960, 3)),                               # M < 64: tiles 1 / 2 / 4 are not candidates, tile 6 is "rejected"
    ])
    calls = {"n": 0}

    def time_forward():
        calls["n"] += 1
        row = []
        for kind, st in prog.ops:
            if kind != GEMM:
                row.append(0.010)
            elif st.M == 16384:
                row.append(0.100 if st.tile != 7 else 0.080)
            elif st.M == 4096:
                row.append(0.0500 if st.tile != 5 else 0.04975)
            elif st.M == 32:
                if st.tile == 6:
                    raise RuntimeError("rejected")
                row.append(0.020)
            else:
                row.append(0.030)
        return [[row, row, row]]

    sig = lambda st: (st.M, st.N, st.K)
    kept, total = tic.tune([prog], time_forward, sig, GEMM, min_gain=0.015, min_share=0.002, log=lambda *_: None)
    assert [(k[0], k[1], k[2]) for k in kept] == [((16384, 3072, 384), 2, 7)]
    assert abs(total - (0.2 + 0.05 + 0.03 + 0.03 + 0.02 + 0.03)) < 1e-9
    tiles = [st.tile for kind, st in prog.ops if kind == GEMM]
    assert tiles == [7, 1, 7, 3, 20, 3]                              # winners applied, everything else restored
    assert tic.candidates(prog.ops[6][1]) == [] and tic.candidates(prog.ops[7][1]) == []
    assert 1 not in tic.candidates(prog.ops[8][1]) and 7 not in tic.candidates(prog.ops[8][1])
    assert set(tic.candidates(prog.ops[0][1])) == {1, 2, 3, 4, 5, 6, 18, 19}            # (its own tile, now 7, is not a candidate)


def test_in_context_tuner_sweeps_the_start_delay_of_eligible_launches_only():
    tic = _load("tune_in_context")
    GEMM = 1
    prog = types.SimpleNamespace(_packed=None, ops=[
        (GEMM, _st(16384, 3072, 384, 2)),           # 2048 workgroups of a 4-wave tile: eligible; 6 us (24 quarter-us) is best in context
        (GEMM, _st(4096, 576, 576, 1)),             # 160 workgroups: below the threshold, never delayed
        (GEMM, _st(16384, 384, 1536, 18)),          # 8-wave tile: no second resident slot
    ])

    def time_forward():
        row = []
        for _, st in prog.ops:
            q = (st.flags >> 8) & 255
            if st.M == 16384 and st.N == 3072:
                row.append(0.100 - 0.0005 * q if q <= 24 else 0.100)
            else:
                assert q == 0
                row.append(0.050)
        return [[row, row]]

    kept, _ = tic.tune([prog], time_forward, lambda st: (st.M, st.N, st.K), GEMM, min_gain=0.015, stagger=(8, 16, 24, 32, 48), min_wg=512,
                       log=lambda *_: None)
    assert [(k[0], k[1], k[2], k[5]) for k in kept] == [((16384, 3072, 384), 2, 2, 24)]
    assert [(st.flags >> 8) & 255 for _, st in prog.ops] == [24, 0, 0]
    assert tic.workgroups(prog.ops[0][1]) == 2048 and tic.workgroups(prog.ops[2][1]) is None


def test_pinned_cache_entry_may_carry_a_start_delay(monkeypatch):
    """engine.Prog.gemm applies the optional third element of a tuner choice to FridoGemm.flags bits 8..15 (round-6 plumbing);
    a two-element choice (every entry of today's pinned cache) leaves the flags alone."""
    import torch
    from frido_amd import engine, tune
    monkeypatch.setattr(torch.cuda, "current_stream", lambda dev=None: types.SimpleNamespace(cuda_stream=0))
    for choice, want in (((2, 1, 24), 24 << 8), ((2, 1), 0), ((1, 4, 0), 0)):
        monkeypatch.setattr(tune, "best_tile", lambda st, dev, stream, c=choice: c)
        monkeypatch.setattr(tune, "workspace_for", lambda st, dev, tag="": 0)
        p = engine.Prog(torch.device("cuda"), 2)
        p.gemm(16384, 3072, 384, (0x1000, 64), (0x2000, 64), out_f32=0x3000, ldo=3072)
        st = p.ops[-1][1]
        # a pinned delay REPLACES the library-wide default (engine.STAGGER_US, bits 8..15 of GEMM_FLAGS); without one the default stands
        assert (st.tile, st.splitk) == choice[:2] and st.flags == (((engine.GEMM_FLAGS & ~0xFF00) | want) if want else engine.GEMM_FLAGS)


def test_trace_diff_groups_by_kernel_and_grid(tmp_path, capsys):
    import csv
    td = _load("trace_diff")
    hdr = ["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Workgroup_Size_X", "Workgroup_Size_Y", "Workgroup_Size_Z", "Grid_Size_X",
           "Grid_Size_Y", "Grid_Size_Z"]
    for name, us in (("a", 150), ("b", 135)):
        d = tmp_path / name / "x"
        d.mkdir(parents=True)
        with open(d / "1_kernel_trace.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(hdr)
            for i in range(4):
                w.writerow(["void igemm_kernel<128, 192>(FridoGemm)", 1000 * i, 1000 * i + us * 1000, 256, 1, 1, 256 * 2048, 1, 1])
            w.writerow(["void igemm_kernel<128, 192>(FridoGemm)", 0, 50000, 256, 1, 1, 256 * 768, 1, 1])
    agg = td.load(str(tmp_path / "a"))
    assert agg[("void igemm_kernel<128, 192>(FridoGemm)", 2048, 256)] == [4, 600000]
    assert agg[("void igemm_kernel<128, 192>(FridoGemm)", 768, 256)] == [1, 50000]
    import sys
    sys.argv = ["trace_diff.py", str(tmp_path / "a"), str(tmp_path / "b")]
    td.main()
    out = capsys.readouterr().out
    assert "-0.06" in out and "2048" in out


def test_gap_analysis_sampling_loop_window(tmp_path):
    """(r06) tools/gap_analysis.py `sampling_loop`: the kernels between the first and the last update kernel of one pass, the forwards in
    that window and the GEMM family's summed duration per forward -- what bench.py's `roofline.trace` divides its FLOPs by."""
    import csv
    import json
    import subprocess
    import sys
    rows, t = [], [0]

    def k(name, dur):
        rows.append(dict(Start_Timestamp=t[0], End_Timestamp=t[0] + dur, Kernel_Name=name))
        t[0] += dur + 100
    k("randn_kernel", 1000)
    k("void (anonymous namespace)::igemm_kernel<128, 128, 2, true, 32, false, 1>(FridoGemm)", 70000)          # hoisted pre-pass: outside the window
    for _ in range(5):
        k("void (anonymous namespace)::igemm_kernel<128, 128, 2, false, 32, false, 1>(FridoGemm)", 50000)
        k("void (anonymous namespace)::conv3x3_gn_kernel<256, false, false>(FridoGemm)", 100000)
        k("void (anonymous namespace)::gn_fused_f32_kernel<1024, 1>(FridoGnApply)", 5000)
        k("(anonymous namespace)::sampler_step_kernel(FridoSamplerStep)", 3000)
    k("(anonymous namespace)::vq_kernel(FridoVq)", 1000)
    (tmp_path / "x").mkdir()
    with open(tmp_path / "x" / "1_kernel_trace.csv", "w") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)
    out = subprocess.run([sys.executable, os.path.join(REPO, "tools", "gap_analysis.py"), str(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    loop = json.loads(out.stdout)["sampling_loop"]
    assert loop["forwards"] == 4 and loop["gemm_family_launches"] == 8 and loop["kernels_per_forward"] == 4.0
    assert abs(loop["gemm_family_ms_per_forward"] - 0.15) < 1e-9 and abs(loop["forward_ms_wall"] - 0.158375) < 1e-9


def test_inflight_audit_flags_a_touched_destination_and_guards_the_build(tmp_path):
    """(r06) tools/asm_inflight_check.py: between a hand-issued `global_load_dwordx4 vD` and the `s_waitcnt vmcnt` that covers it nothing may
    read or write vD (conv3x3_gn_kernel rests on the compiler not splitting such a live range).  The audit exits non-zero on a hazard AND when
    it sees no hand-issued load at all; `make -C frido_amd/csrc` runs it on both builds (a violation fails the build), checked here too."""
    import subprocess
    import sys
    tool = os.path.join(REPO, "tools", "asm_inflight_check.py")
    clean = ["_ZN12_GLOBAL__N_117conv3x3_gn_kernelILi256ELb0ELb0EEEv9FridoGemm:", "\tglobal_load_dwordx4 v[10:13], v[2:3], off", "\tv_add_u32_e32 v20, v21, v22",
             "\ts_waitcnt vmcnt(0)", "\tv_add_f32_e32 v30, v10, v11", "\ts_endpgm", ".Lfunc_end0:"]
    hazard = clean[:2] + ["\tv_mov_b32_e32 v40, v11"] + clean[2:]                 # a copy of an in-flight register: stale data
    empty = [clean[0], "\tv_add_u32_e32 v20, v21, v22", "\ts_endpgm", ".Lfunc_end0:"]
    for name, lines, rc in (("clean", clean, 0), ("hazard", hazard, 1), ("empty", empty, 1)):
        f = tmp_path / f"{name}.s"
        f.write_text("\n".join(lines) + "\n")
        out = subprocess.run([sys.executable, tool, str(f)], capture_output=True, text=True)
        assert out.returncode == rc, (name, out.stdout, out.stderr)
    assert "v_mov_b32_e32 v40, v11" in subprocess.run([sys.executable, tool, str(tmp_path / "hazard.s")], capture_output=True, text=True).stdout
    csrc = os.path.join(REPO, "frido_amd", "csrc")
    if os.path.exists(os.path.join(csrc, "..", "libfrido_hip.so")) and os.path.exists("/opt/rocm/bin/hipcc"):
        mk = subprocess.run(["make", "-C", csrc, "audit"], capture_output=True, text=True, env=dict(os.environ, PATH=os.environ.get("PATH", "") + ":/opt/rocm/bin"))
        assert mk.returncode == 0, mk.stdout[-2000:] + mk.stderr[-2000:]
        log = open(os.path.join(csrc, ".audit_f16.log")).read()
        assert log.count("0 instructions touching an in-flight destination") == 8          # all eight instantiations of the fused kernel
