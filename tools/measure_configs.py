#!/usr/bin/env python3
"""The other measurement rows of SURVEY.md 8(d), each next to a CPU time of the oracle (test infrastructure) on a
bounded sample: C2 CCSD iteration at (20,80), C4 UEG N=14 / 57 plane waves (TC integrals on the device + one DCSD
iteration), C5 one EOM-CCSD sigma build at (30,120).  One JSON object per line on stdout.

    python3 tools/measure_configs.py [--skip-cpu]     (needs the GPU; bench.py stays the headline measurement)
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def timed(fn, sync, reps, warm=3):
    import gc
    gc.collect()
    gc.disable()        # (a generation-2 collection is a 38-ms outlier; the solvers switch the collector off for their loops too)
    try:
        return _timed(fn, sync, reps, warm)
    finally:
        gc.enable()


def _timed(fn, sync, reps, warm=3):
    for _ in range(warm):       # lazy set-up (cached permutations, packed integrals) and launch-graph recording
        fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    sync()
    return (time.perf_counter() - t0) / reps


def c2(args):
    from pymes_amd.integral.device import DeviceIntegrals
    from pymes_amd.model import synthetic
    from pymes_amd.solver.ccsd import CCSD
    no, nv = 20, 80
    B, eps = synthetic.factors(no, nv, seed=0, scale=0.15)
    ints = DeviceIntegrals.from_factors(no, B)
    solver = CCSD(no)
    st = quiet(solver.setup, np.diag(eps), ints)
    dt = timed(lambda: quiet(solver.iterate, st), ints.ctx.sync, 20)
    o, v = float(no), float(nv)
    from bench import reference_flops
    flops = reference_flops(no, nv)
    out = {"config": "C2 CCSD iteration, synthetic (nocc=20, nvirt=80, scale 0.15)", "gpu_s": dt,
           "reference_algorithmic_flops": flops, "algorithmic_tflops": flops / dt / 1e12}
    # the whole golden solve of tests/golden/solves.json["syn_20_80"] (the reference itself: 9 iterations, 706 s in the
    # build container): integrals from the factors, set-up, iterations to delta_e = 1e-10, amplitudes back on the host
    ints.ctx.close()
    gold = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                                       "solves.json")))["syn_20_80"]["ccsd"]
    t0 = time.perf_counter()
    ints = DeviceIntegrals.from_factors(no, B)
    s2 = CCSD(no, delta_e=gold["delta_e"])
    res = quiet(s2.solve, np.diag(eps), ints)
    out["golden_solve"] = {"gpu_wall_s": time.perf_counter() - t0, "iterations": s2.iterations, "energy": res["ccsd e"],
                           "energy_minus_reference": res["ccsd e"] - gold["e"],
                           "reference_wall_s": gold.get("reference_seconds"), "reference_iterations": gold["iterations"]}
    if not args.skip_cpu:
        from oracle import cc_oracle as oc
        from oracle.cases import synthetic_case
        so, sv = 10, 40                                     # bounded sample: flops scale as the closed form above
        f, V, _, _ = synthetic_case(so, sv, seed=0, scale=0.15)
        rs_ = np.random.default_rng(1)
        t1, t2 = 0.05 * rs_.standard_normal((sv, so)), 0.05 * rs_.standard_normal((sv, sv, so, so))
        Vb = oc.split_blocks(so, V)
        t0 = time.perf_counter()
        fd = oc.dressed_fock(so, f, t1, Vb)
        Vd = oc.dressed_V(t1, Vb)
        oc.singles_residual(so, fd, t1, t2, Vb)
        oc.ccsd_doubles_residual(so, fd, t2, Vd)
        cpu = time.perf_counter() - t0
        sfl = reference_flops(so, sv)
        out["cpu_baseline"] = {"value": cpu * flops / sfl, "unit": "s", "cores": 1, "kind": "port",
                               "sample": f"oracle residuals + dressing at ({so},{sv}) in {cpu:.2f} s, scaled by the "
                                         "algorithmic-flop ratio"}
    ints.ctx.close()
    return out


def c4(args):
    from pymes_amd.model.ueg import UEG
    from pymes_amd.solver.ccsd import CCSD
    from pymes_amd.integral.device import DeviceIntegrals
    from tests.test_ueg import tc_problem
    nel, rs, cutoff = 14, 1.0, 5
    m = UEG(nel, nel // 2, nel // 2, rs)
    m.init_single_basis(cutoff)
    n_pw = len(m.basis_fns) // 2
    kc = m.L / (2 * np.pi) * 2.3225029893472993 / rs       # test_ccd_dcd.py:99
    t0 = time.perf_counter()
    no, V, f, e_hf, d2, e3, eps_i, eps_a = tc_problem(UEG, nel, rs, cutoff, kc)
    t_int = time.perf_counter() - t0
    ints = DeviceIntegrals.from_V_pqrs(no, V)
    solver = CCSD(no, is_dcsd=True)
    st = quiet(solver.setup, f, ints)
    dt = timed(lambda: quiet(solver.iterate, st), ints.ctx.sync, 20)
    out = {"config": f"C4 UEG N=14 rs=1.0 cutoff=5 ({n_pw} plane waves): TC integrals + DCSD iteration",
           "gpu_integrals_s": t_int, "gpu_iteration_s": dt,
           "note": "integrals = eval_2b_integrals(only_2b) + eval_2b_integrals(effect_2b) + HF + 3-body contractions, "
                   "device kernels + host glue, download included"}
    if not args.skip_cpu:
        from oracle.ueg_oracle import Ueg
        from oracle import cc_oracle as oc
        u = Ueg(nel, rs)
        u.init_basis(cutoff)
        u.k_cutoff = kc
        t0 = time.perf_counter()
        u.two_body("only_2b")
        t_2b = time.perf_counter() - t0
        Vb = oc.split_blocks(no, V)
        t2 = oc.mp2(f.diagonal()[:no], f.diagonal()[no:], Vb["ijab"], Vb["abij"])[1]
        t0 = time.perf_counter()
        oc.ccsd_doubles_residual(no, f, t2, Vb, is_dcsd=True)
        cpu_it = time.perf_counter() - t0
        out["cpu_baseline"] = {"integrals_only_2b_s": t_2b, "dcsd_doubles_residual_s": cpu_it, "cores": 1,
                               "kind": "port", "sample": "oracle two_body('only_2b') (vectorised numpy; the reference's "
                               "Python loops took 53.6 s for this call in the survey container) and one oracle DCSD "
                               "doubles residual at full size"}
    ints.ctx.close()
    return out


def c5(args):
    from pymes_amd.integral.device import DeviceIntegrals
    from pymes_amd.model import synthetic
    from pymes_amd.solver.eom_ccsd import _Sigma
    no, nv = 30, 120
    B, eps = synthetic.factors(no, nv, seed=0)
    ints = DeviceIntegrals.from_factors(no, B)
    ctx = ints.ctx
    ctx.set_orbital_energies(eps[:no], eps[no:])
    t2 = ctx.empty((nv, nv, no, no))
    ctx.mp2(t2, 0.0)
    f = np.diag(eps)
    t0 = time.perf_counter()
    sig = _Sigma(ctx, f, t2)
    ctx.sync()
    t_hoist = time.perf_counter() - t0
    rng = np.random.default_rng(0)
    u1 = ctx.array(rng.standard_normal((nv, no)))
    u2h = rng.standard_normal((nv, nv, no, no))
    u2 = ctx.array(u2h + u2h.transpose(1, 0, 3, 2))          # exchange-symmetric, like every Davidson vector
    flag = sig.exchange_symmetric(u2)                        # decided once per vector by the Davidson driver
    dt = timed(lambda: sig.apply(u1, u2, u2_sym=flag), ctx.sync, 5)
    flops = 2.0 * (6.4e9 + 1.502e12)
    # roofline of the sigma build: executed GEMM flops / summed HIP-event time of the GEMM calls (a pass of its own)
    ctx.stats(reset=True)
    ctx.prof_enable(True)
    ctx.prof_reset()
    reps = 3
    for _ in range(reps):
        sig.apply(u1, u2, u2_sym=flag)
    ctx.sync()
    prof, dma = ctx.prof_query(), ctx.prof_query(kernel_class=1)
    ctx.prof_enable(False)
    peak = 78.6
    tf = lambda q: q["flops"] / (q["ms"] * 1e-3) / 1e12 if q["ms"] > 0 else 0.0
    out = {"config": "C5 one EOM-CCSD sigma build (singles + doubles), synthetic (nocc=30, nvirt=120)", "gpu_s": dt,
           "hoisted_intermediates_once_per_solve_s": t_hoist, "reference_algorithmic_flops": flops,
           "algorithmic_tflops": flops / dt / 1e12,
           "roofline": {"bound": "mfma", "peak": peak, "unit": "TFLOP/s", "kernel": "dgemm_glds_kernel",
                        "achieved": tf(dma), "frac": tf(dma) / peak, "ms_per_sigma": dma["ms"] / reps,
                        "launches_per_sigma": dma["kernel_launches"] / reps,
                        "all_gemm": {"achieved": tf(prof), "frac": tf(prof) / peak, "ms_per_sigma": prof["ms"] / reps,
                                     "calls_per_sigma": prof["launches"] / reps,
                                     "executed_flops_per_sigma": prof["flops"] / reps}}}
    # multi-vector build (what the Davidson driver calls): k symmetric vectors stacked, per-vector time
    kvec = 4
    u1s = [ctx.array(rng.standard_normal((nv, no))) for _ in range(kvec)]
    u2s = []
    for _ in range(kvec):
        h = rng.standard_normal((nv, nv, no, no))
        u2s.append(ctx.array(h + h.transpose(1, 0, 3, 2)))
    del h
    syms = [True] * kvec
    dtk = timed(lambda: sig.apply_many(u1s, u2s, syms), ctx.sync, 5)
    ctx.stats(reset=True)
    ctx.prof_enable(True)
    ctx.prof_reset()
    for _ in range(reps):
        sig.apply_many(u1s, u2s, syms)
    ctx.sync()
    profk, dmak = ctx.prof_query(), ctx.prof_query(kernel_class=1)
    ctx.prof_enable(False)
    out["multi_vector"] = {"k": kvec, "gpu_s_per_vector": dtk / kvec, "many_ok": bool(sig.many_ok),
                           "roofline": {"achieved": tf(dmak), "frac": tf(dmak) / peak, "ms_per_vector": dmak["ms"] / reps / kvec,
                                        "all_gemm": {"achieved": tf(profk), "frac": tf(profk) / peak,
                                                     "ms_per_vector": profk["ms"] / reps / kvec,
                                                     "calls_per_build": profk["launches"] / reps,
                                                     "executed_flops_per_vector": profk["flops"] / reps / kvec}}}
    if not args.skip_cpu:
        from oracle import eom_oracle as eo, cc_oracle as oc
        from oracle.cases import synthetic_case
        so, sv = 12, 48
        fs, V, _, _ = synthetic_case(so, sv, seed=0)
        t2s = 0.05 * rng.standard_normal((sv, sv, so, so))
        Vb = oc.split_blocks(so, V)
        u1s, u2s = rng.standard_normal((sv, so)), rng.standard_normal((sv, sv, so, so))
        t0 = time.perf_counter()
        eo.sigma_singles(so, fs, Vb, u1s, u2s, t2s)
        eo.sigma_doubles(so, fs, Vb, u1s, u2s, t2s)
        cpu = time.perf_counter() - t0
        # leading terms o^2 v^4 (ladder) + o^3 v^3 (40 three-operand terms contracted pairwise)
        scale = (30.0**2 * 120.0**4 + 20 * 30.0**3 * 120.0**3) / (so**2 * sv**4 + 20 * so**3 * sv**3)
        out["cpu_baseline"] = {"value": cpu * scale, "unit": "s", "cores": 1, "kind": "port",
                               "sample": f"oracle sigma (np.einsum optimize=True per term) at ({so},{sv}) in {cpu:.2f} s, "
                                         "scaled by o^2v^4 + 20 o^3v^3"}
    ctx.close()
    return out


def c5dav(args):
    """Config 5 end to end: CCSD ground state -> T1 dressing -> EOM-CCSD Davidson (eom_ccsd.py:46-167) at (30,120), everything
    resident in HBM (DeviceIntegrals -> device amplitudes -> DressedDeviceIntegrals).  Orbital energies with isolated frontier
    levels (as oracle/cases.py::eom_davidson_case: on the dense spectrum of SURVEY 8(d) the reference's driver stalls)."""
    from pymes_amd.integral.device import DeviceIntegrals
    from pymes_amd.model import synthetic
    from pymes_amd.solver.ccsd import CCSD
    from pymes_amd.solver.eom_ccsd import EOM_CCSD
    (no, nv), n_excit = args.davidson_size, 3
    B, _ = synthetic.factors(no, nv, seed=0, scale=args.davidson_scale)
    rng = np.random.default_rng(5)
    eps = np.concatenate([np.sort(np.concatenate([[-1.5], -2.7 - 0.8 * rng.random(no - 1)])),
                          np.sort(np.concatenate([[1.5, 1.9, 2.35], 3.2 + 1.0 * rng.random(nv - 3)]))])
    f = np.diag(eps)
    ints = DeviceIntegrals.from_factors(no, B)
    ctx = ints.ctx
    out = {"config": f"C5 EOM-CCSD Davidson, synthetic (nocc={no}, nvirt={nv}), n_excit = {n_excit}, device-resident chain"}
    t0 = time.perf_counter()
    cc = CCSD(no, delta_e=1e-10)
    res = quiet(cc.solve, f, ints, device_amplitudes=True)
    ctx.sync()
    out["ccsd"] = {"wall_s": time.perf_counter() - t0, "iterations": cc.iterations, "energy": res["ccsd e"]}
    t0 = time.perf_counter()
    fd = cc.get_T1_dressed_fock(f, res["t1"], ints)
    Vd = cc.get_T1_dressed_V(res["t1"], ints)
    ctx.sync()
    out["dressing_11_blocks_s"] = time.perf_counter() - t0

    def run(reuse, phases, max_iter):
        eom = EOM_CCSD(no, n_excit=n_excit)
        eom.max_iter, eom.reuse_sigma, eom.profile_phases = max_iter, reuse, phases
        ctx.sync()
        t0 = time.perf_counter()
        ee = quiet(eom.solve, fd, Vd, res["t2"])
        ctx.sync()
        return eom, ee, time.perf_counter() - t0
    run(True, False, 3)                                               # warm-up: lazy statics, cached transposed blocks
    eom, ee, wall = run(True, False, args.davidson_passes)
    out["davidson"] = {"passes": eom.iterations, "wall_s": wall, "s_per_pass": wall / eom.iterations,
                       "sigma_vectors": eom.timings["sigma_vectors"], "ritz_values_last_pass": [float(x) for x in eom.history[-1]],
                       "converged": bool(eom.iterations < args.davidson_passes)}
    eom, ee, wall_p = run(True, True, args.davidson_passes)            # the same with a synchronisation per phase
    tm = eom.timings
    out["davidson"]["phases"] = {k: tm[k] for k in ("hoist_s", "sigma_s", "orth_s", "subspace_s")}
    out["davidson"]["per_pass"] = eom.pass_log
    full = [p for p in eom.pass_log if p["dim"] == eom.max_dim and p["new_vectors"] == n_excit]
    if full:
        sig_ms = 1e3 * np.mean([p.get("sigma_s", 0.0) for p in full])
        rest_ms = 1e3 * np.mean([p.get("orth_s", 0.0) + p.get("subspace_s", 0.0) for p in full])
        out["davidson"]["pass_at_full_dimension"] = {"dim": eom.max_dim, "sigma_ms": sig_ms, "orth_plus_subspace_ms": rest_ms,
                                                     "pass_over_sigma": (sig_ms + rest_ms) / sig_ms}
    eom_ref, ee_ref, wall_ref = run(False, False, args.davidson_passes)       # the reference's schedule: all vectors anew
    out["davidson_reference_schedule"] = {"passes": eom_ref.iterations, "wall_s": wall_ref,
                                          "s_per_pass": wall_ref / eom_ref.iterations,
                                          "sigma_vectors": eom_ref.timings["sigma_vectors"],
                                          "max_ritz_difference": float(np.abs(np.array(eom_ref.history) - np.array(eom.history)).max())}
    ctx.close()
    return out


def c5feast(args):
    """One FEAST-EOM-CCSD pass (feast_eom_ccsd.py:103-150) at (30,120) on the device-resident chain: 8 quadrature points x
    the current trial vectors, each a GCROT(m,k) solve of (z - H) Q = u through the general (not exchange-symmetric)
    multi-vector sigma build, then the projected eigenproblem.  Reported next to the Davidson numbers of c5dav."""
    from pymes_amd.integral.device import DeviceIntegrals
    from pymes_amd.model import synthetic
    from pymes_amd.solver.ccsd import CCSD
    from pymes_amd.solver.feast_eom_ccsd import FEAST_EOM_CCSD
    no, nv = args.davidson_size
    B, _ = synthetic.factors(no, nv, seed=0, scale=args.davidson_scale)
    rng = np.random.default_rng(5)
    eps = np.concatenate([np.sort(np.concatenate([[-1.5], -2.7 - 0.8 * rng.random(no - 1)])),
                          np.sort(np.concatenate([[1.5, 1.9, 2.35], 3.2 + 1.0 * rng.random(nv - 3)]))])
    f = np.diag(eps)
    ints = DeviceIntegrals.from_factors(no, B)
    ctx = ints.ctx
    cc = CCSD(no, delta_e=1e-10)
    res = quiet(cc.solve, f, ints, device_amplitudes=True)
    fd = cc.get_T1_dressed_fock(f, res["t1"], ints)
    Vd = cc.get_T1_dressed_V(res["t1"], ints)
    out = {"config": f"C5 FEAST-EOM-CCSD passes, synthetic (nocc={no}, nvirt={nv}), window 3.0 +- 0.25, device-resident chain"}
    s = FEAST_EOM_CCSD(no, e_c=3.0, e_r=0.25, n_trial=4, max_iter=args.feast_passes)
    np.random.seed(1)
    ctx.sync()
    t0 = time.perf_counter()
    ev = quiet(s.solve, fd, Vd, res["t2"])
    ctx.sync()
    wall = time.perf_counter() - t0
    mv = [m for _, m in s.linear_solver_info]
    out["feast"] = {"passes": s.iterations, "wall_s": wall, "s_per_pass": wall / s.iterations, "linear_solves": len(mv),
                    "matvecs": int(np.sum(mv)), "matvecs_per_solve_mean": float(np.mean(mv)),
                    "solves_not_converged": int(sum(1 for info, _ in s.linear_solver_info if info != 0)),
                    "ritz_values_last_pass": [[float(np.real(x)), float(np.imag(x))] for x in s.history[-1]],
                    "s_per_matvec": wall / max(1, int(np.sum(mv)))}
    ctx.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--feast-passes", type=int, default=2)
    ap.add_argument("--davidson-passes", type=int, default=40)
    ap.add_argument("--davidson-size", type=lambda t: tuple(int(x) for x in t.split(",")), default=(30, 120))
    ap.add_argument("--davidson-scale", type=float, default=0.12)
    ap.add_argument("--skip-cpu", action="store_true")
    ap.add_argument("--only", default="c2,c4,c5")
    args = ap.parse_args()
    for name in args.only.split(","):
        print(json.dumps({"c2": c2, "c4": c4, "c5": c5, "c5dav": c5dav, "c5feast": c5feast}[name](args)), flush=True)


if __name__ == "__main__":
    main()
