#!/usr/bin/env python3
"""Headline benchmark: wall time of one CCSD iteration at (nocc=50, nvirt=200) on N MI355X,
plus the fp64-MFMA roofline fraction of the dominant kernel and a CPU baseline.

    python bench.py --gpus N --steps K --warmup W          (N > 1: spawns its own N rank processes, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W          (ranks started by the launcher)

A "step" is one full pass of the loop body of pymes/solver/ccsd.py:159-209 (dressed Fock,
T1-dressed V blocks, singles + doubles residuals, amplitude update, DIIS, energy, norms) on
synthetic density-fitted integrals that are formed on the GPU and stay resident in HBM.
For N > 1 the particle-particle ladder is sharded over the ranks on the virtual index
(pymes_amd/dist.py); the problem size is fixed, so scaling is "strong".
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# the pool's host driver supports dmabuf IPC only: without this RCCL's buffer exchange between the rank processes fails with
# `hipIpcGetMemHandle: invalid argument` (set before anything initialises the GPU; a launcher's own setting wins)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6      # 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz (MI355X fp64 matrix = vector peak)


def reference_flops(no, nv, is_dcsd=False):
    """Algorithmic flops of one reference iteration (2 x FMA of its contraction sequence)."""
    o, v = float(no), float(nv)
    if is_dcsd:     # closed forms of the doubles residual, SURVEY §8(d) (ccd.py:164-254)
        doubles = v**4 * o**2 + 5 * o**3 * v**3 + o**4 * v**2 + 2 * o**2 * v**3 + 2 * o**3 * v**2
    else:
        doubles = v**4 * o**2 + 10 * o**3 * v**3 + 2 * o**4 * v**2 + 3 * o**2 * v**3 + 3 * o**3 * v**2
    dressing = 5 * o * v**4 + 28 * o**2 * v**3          # ccsd.py:290-421 as the reference evaluates it
    singles = 2 * o**2 * v**3 + 2 * o**3 * v**2
    return 2.0 * (doubles + dressing + singles)


TRAFFIC_CSV = os.path.join("profiles", "r06", "bench_c3_pmc_hbm_traffic.csv")
# workloads with committed counter passes: (nocc, nvirt) -> summary written by tools/profile_bench.sh
TRAFFIC_CSVS = {(50, 200): TRAFFIC_CSV, (20, 80): os.path.join("profiles", "r06", "bench_c2_pmc_hbm_traffic.csv")}


def kernels_hash():
    import hashlib
    return hashlib.sha256(open(os.path.join(ROOT, "pymes_amd", "csrc", "kernels.hip"), "rb").read()).hexdigest()


def pmc_traffic_per_launch(no, nv, world, kernel_prefix):
    """(HBM bytes per launch of the kernels whose name starts with `kernel_prefix`, provenance string).

    Hardware counters cannot be read from inside the run, so the figure comes from the committed rocprofv3 PMC passes of
    THIS command at the workloads of TRAFFIC_CSVS (tools/profile_bench.sh: FETCH_SIZE doubled per the gfx950
    correction + WRITE_SIZE, separate passes).  The CSV records the sha256 of kernels.hip it was collected with; when
    the kernels have changed since, or for any other workload, the value is null rather than stale."""
    if world != 1 or (no, nv) not in TRAFFIC_CSVS:
        return None, "not collected for this workload"
    TRAFFIC_CSV = TRAFFIC_CSVS[(no, nv)]
    path = os.path.join(ROOT, TRAFFIC_CSV)
    if not os.path.exists(path):
        return None, f"{TRAFFIC_CSV} missing"
    launches, gbytes, recorded = 0, 0.0, None
    for line in open(path):
        if line.startswith("# kernels.hip sha256="):
            recorded = line.split("=", 1)[1].strip()
        # <false, false> (both operands M/N-contiguous) occurs only in the integral build from the factors, not in an iteration
        if line.startswith('"' + kernel_prefix) and "<false, false>" not in line:
            name, n, fetch, write = line.rsplit(",", 3)
            launches += int(n)
            gbytes += int(n) * (float(fetch) + float(write))
    if recorded != kernels_hash():
        return None, f"{TRAFFIC_CSV} is stale (collected with kernels.hip {str(recorded)[:12]}, now {kernels_hash()[:12]})"
    if not launches:
        return None, f"{TRAFFIC_CSV} has no launch of {kernel_prefix}"
    return gbytes / launches * 1e9, f"{TRAFFIC_CSV} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, kernels.hip {recorded[:12]})"


def launch_ranks(n, argv, script=None, timeout_s=1500.0):
    """`python bench.py --gpus N` without a launcher: start N rank processes of this script (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in their environment, rendezvous on 127.0.0.1) and relay rank 0's JSON line.  The parent never
    touches the GPU (no HIP call, no exec of an initialised process); a failing rank takes the others down, and so does
    the overall time limit (a hung collective must end the run with a non-zero status, not stall the driver)."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0 prints the one JSON line (read by a thread so that a dead sibling cannot leave us blocked on the pipe);
    # stderr of every rank is inherited
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    codes = [None] * n
    t_start = time.time()
    while any(c is None for c in codes):
        for r, p in enumerate(procs):
            if codes[r] is None:
                codes[r] = p.poll()
        timed_out = time.time() - t_start > timeout_s
        if timed_out:
            sys.stderr.write(f"bench.py: ranks still running after {timeout_s:.0f} s — stopping them\n")
        if timed_out or any(c not in (None, 0) for c in codes):   # a rank failed: the others would wait for it forever
            time.sleep(0.0 if timed_out else 5.0)
            for r, p in enumerate(procs):
                if codes[r] is None and p.poll() is None:
                    p.kill()                                  # exactly the PIDs started above
                if codes[r] is None:
                    codes[r] = p.wait()
            break
        time.sleep(0.05)
    reader.join(timeout=10.0)
    for ln in b"".join(chunks).decode().splitlines():      # only the JSON line goes to stdout (gloo logs there too)
        (sys.stdout if ln.lstrip().startswith("{") else sys.stderr).write(ln + "\n")
    sys.stdout.flush()
    if any(codes):
        raise SystemExit(f"rank exit codes {codes}")



# ---- the other BASELINE configs, driver-timed -------------------------------------------------------------
# Golden values are LITERALS recorded from the reference's own runs (tests/golden/*.json / *.npz, written by oracle/make_golden*.py
# in the build container) — data, not the oracle; nothing under oracle/ is imported here.
C2_GOLDEN = {"e": -0.40137558021086484, "iterations": 9, "delta_e": 1e-10}            # tests/golden/solves.json["syn_20_80"]["ccsd"]
C4_GOLDEN = {"dcsd": -0.4181690961120107, "passes": 22, "k_cutoff": 1.436091003782944}   # tests/golden/ueg.json["tc_N14_rs1.0_c5"]
# (30,120) Davidson: no reference run exists at this size (the reference would take days); the literals are THIS engine's values
# of round 4 (profiles/r04/configs_c2_c4_c5.jsonl) — a regression check; the driver is reference-pinned at (12,48) and (20,80)
C5_RITZ_TOL = 2e-4        # relative residual of a converged root at the driver's stopping test (measured: profiles/r06)
C5_TIGHT_DE, C5_TIGHT_RITZ_TOL = 1e-12, 1e-6      # the certificate VERDICT r5 asked for: residual < 1e-6 once the driver is carried on
C5_DAVIDSON = {"ccsd_e": -0.32983974045195175, "ee": [3.0060256332720243, 3.409850418963405, 3.8568464914371234], "passes": 26}


def _quiet(fn, *a, **k):
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def _timed(fn, sync, reps, warm=3):
    for _ in range(warm):       # lazy statics and launch-graph recording
        fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    sync()
    return (time.perf_counter() - t0) / reps


def other_configs(device=0):
    """BASELINE configs 2, 4 and 5 behind the headline: per config an iteration / sigma time under this process's clock and a
    result checked against the reference's recorded value.  Returns the `other_configs` object of the JSON line."""
    import gc
    from pymes_amd.integral.device import DeviceIntegrals
    from pymes_amd.model import synthetic
    from pymes_amd.solver.ccsd import CCSD
    out, t_all = {}, time.perf_counter()
    gc.collect()
    gc.disable()
    try:
        # ---- C2: (20,80) CCSD, the reference's own 9-iteration solve, then the iteration time -------------------------------
        no, nv = 20, 80
        B, eps = synthetic.factors(no, nv, seed=0, scale=0.15)
        t0 = time.perf_counter()
        ints = DeviceIntegrals.from_factors(no, B, device=device)
        s2 = CCSD(no, delta_e=C2_GOLDEN["delta_e"], device=device)
        res = _quiet(s2.solve, np.diag(eps), ints)
        out["c2_solve_s"] = time.perf_counter() - t0
        out["c2_energy_minus_reference"] = res["ccsd e"] - C2_GOLDEN["e"]
        out["c2_ok"] = bool(abs(res["ccsd e"] - C2_GOLDEN["e"]) < 1e-9 and s2.iterations == C2_GOLDEN["iterations"])
        solver = CCSD(no, device=device)
        st = _quiet(solver.setup, np.diag(eps), ints)
        for _ in range(4):
            _quiet(solver.iterate, st)
        ints.ctx.sync()
        dt = _timed(lambda: _quiet(solver.iterate, st), ints.ctx.sync, 50, warm=0)
        # executed GEMM flops of one iteration: counted by the host planner, i.e. in an EAGER pass (a replayed launch graph
        # does not go through it); per-GEMM profiling switches the replay and the pipelining off; the second pass counts
        ints.ctx.prof_enable(True)
        for _ in range(2):
            ints.ctx.stats(reset=True)
            _quiet(solver.iterate, st)
        ints.ctx.sync()
        fl = ints.ctx.stats()["gemm_flops"]
        ints.ctx.prof_enable(False)
        out["c2_ms"] = 1e3 * dt
        out["c2_frac"] = fl / dt / 1e12 / FP64_MFMA_PEAK_TFLOPS       # executed GEMM flops over the whole iteration
        out["c2_executed_gemm_flops"] = fl
        ints.ctx.close()
        # ---- C4: UEG N = 14, rs = 1, 57 plane waves, transcorrelated DCSD (test_symmetrised_2body_integral.py:39-222) -------
        from pymes_amd.mean_field import hf
        from pymes_amd.model.ueg import UEG
        nel = 14
        t0 = time.perf_counter()
        m = UEG(nel, nel // 2, nel // 2, 1.0)
        m.init_single_basis(5)
        m.k_cutoff = C4_GOLDEN["k_cutoff"]
        no, n_p = nel // 2, len(m.basis_fns) // 2
        kin = np.array([m.basis_fns[2 * i].kinetic for i in range(n_p)])
        V = _quiet(m.eval_2b_integrals, correlator=m.trunc, is_only_2b=True, sp=0)
        f = hf.construct_hf_matrix(no, np.diag(kin), V)
        Va = _quiet(m.eval_2b_integrals, correlator=m.trunc, is_effect_2b=True, sp=0)
        V = V + 0.5 * (Va + Va.transpose(1, 0, 3, 2))
        f = f + np.diag(_quiet(m.double_contractions_in_3_body))
        out["c4_integrals_s"] = time.perf_counter() - t0
        ints = DeviceIntegrals.from_V_pqrs(no, V, device=device)
        s4 = CCSD(no, delta_e=1e-10, is_dcsd=True, device=device)
        r4 = _quiet(s4.solve, f, ints)
        out["c4_energy_minus_reference"] = r4["ccsd e"] - C4_GOLDEN["dcsd"]
        out["c4_ok"] = bool(abs(r4["ccsd e"] - C4_GOLDEN["dcsd"]) < 1e-9 and s4.iterations == C4_GOLDEN["passes"])
        solver = CCSD(no, is_dcsd=True, device=device)
        st = _quiet(solver.setup, f, ints)
        # (0.2 ms per pass: 20 passes are 4 ms of device time right behind 70 ms of host-side integral glue — the clocks are
        # still ramping; 60 passes after 10 of warm-up)
        out["c4_ms"] = 1e3 * _timed(lambda: _quiet(solver.iterate, st), ints.ctx.sync, 60, warm=10)
        ints.ctx.close()
        # ---- C5: (30,120) EOM-CCSD sigma (the inputs of tests/golden/eom_sigma_30_120.npz, i.e. the reference's own
        # update_singles / update_doubles output), single and k = 4 stacked; then one Davidson solve ----------------------------
        from pymes_amd.solver.eom_ccsd import EOM_CCSD, _Sigma
        no, nv = 30, 120
        g = np.load(os.path.join(ROOT, "tests", "golden", "eom_sigma_30_120.npz"))
        B, eps = synthetic.factors(no, nv, seed=0, scale=float(g["scale"]))
        rng = np.random.default_rng(int(g["seed"]))
        n = no + nv
        fd = np.diag(eps) + 0.02 * rng.standard_normal((n, n))
        t2h = rng.standard_normal((nv, nv, no, no)) * 0.02
        t2h = 0.5 * (t2h + t2h.transpose(1, 0, 3, 2))
        u1h = rng.standard_normal((nv, no)) * 0.3
        u2h = rng.standard_normal((nv, nv, no, no)) * 0.05
        u2h = 0.5 * (u2h + u2h.transpose(1, 0, 3, 2))
        ints = DeviceIntegrals.from_factors(no, B, device=device)
        ctx = ints.ctx
        sig = _Sigma(ctx, fd, ctx.array(t2h))
        u1, u2 = ctx.array(u1h), ctx.array(u2h)
        s1, s2d = sig.apply(u1, u2, u2_sym=True)
        s1h, s2h = s1.get(), s2d.get()
        sc1, sc2 = np.abs(g["sigma1"]).max(), np.abs(g["sigma2_val"]).max()
        err = max(np.abs(s1h - g["sigma1"]).max() / sc1, np.abs(s2h[7:8] - g["sigma2_slab"]).max() / sc2,
                  np.abs(s2h.reshape(-1)[g["sigma2_idx"]] - g["sigma2_val"]).max() / sc2)
        sums = np.array([s2h.sum(), np.abs(s2h).sum(), np.linalg.norm(s2h)])
        out["c5_sigma_rel_err_vs_reference"] = float(err)
        out["c5_ok"] = bool(err < 1e-10 and np.abs(sums - g["sigma2_sums"]).max() < 1e-9 * g["sigma2_sums"][1])
        del s1h, s2h, s1, s2d
        ctx.stats(reset=True)
        # (warm-up: the first build allocates its temporaries — and the host-side comparison above left the GPU idle for a
        # few hundred ms: the first builds run at idle clocks and ONE of the next few stalls ~70 ms while the power state
        # changes — measured, tools/probe_other_configs.py; a solve never has that gap)
        dt = _timed(lambda: sig.apply(u1, u2, u2_sym=True), ctx.sync, 10, warm=30)
        fl = ctx.stats()["gemm_flops"] / 40
        out["c5_sigma_ms"] = 1e3 * dt
        out["c5_frac"] = fl / dt / 1e12 / FP64_MFMA_PEAK_TFLOPS
        kvec = 4
        u1s = [u1] + [ctx.array(rng.standard_normal((nv, no))) for _ in range(kvec - 1)]
        u2s = [u2]
        for _ in range(kvec - 1):
            h = rng.standard_normal((nv, nv, no, no))
            u2s.append(ctx.array(h + h.transpose(1, 0, 3, 2)))
        ctx.stats(reset=True)
        dtk = _timed(lambda: sig.apply_many(u1s, u2s, [True] * kvec), ctx.sync, 5, warm=2)
        flk = ctx.stats()["gemm_flops"] / 7
        out["c5_sigma_k4_ms_per_vector"] = 1e3 * dtk / kvec
        out["c5_k4_frac"] = flk / dtk / 1e12 / FP64_MFMA_PEAK_TFLOPS
        del sig, u1s, u2s, u1, u2
        ctx.close()
        # Davidson: CCSD -> T1 dressing -> EOM, device-resident; orbital energies with isolated frontier levels (on the dense
        # spectrum of the SURVEY 8(d) recipe the reference's driver stalls)
        B, _ = synthetic.factors(no, nv, seed=0, scale=0.12)
        rng = np.random.default_rng(5)
        eps = np.concatenate([np.sort(np.concatenate([[-1.5], -2.7 - 0.8 * rng.random(no - 1)])),
                              np.sort(np.concatenate([[1.5, 1.9, 2.35], 3.2 + 1.0 * rng.random(nv - 3)]))])
        f = np.diag(eps)
        ints = DeviceIntegrals.from_factors(no, B, device=device)
        ctx = ints.ctx
        cc = CCSD(no, delta_e=1e-10, device=device)
        res = _quiet(cc.solve, f, ints, device_amplitudes=True)
        fdd = cc.get_T1_dressed_fock(f, res["t1"], ints)
        Vd = cc.get_T1_dressed_V(res["t1"], ints)
        eom = EOM_CCSD(no, n_excit=3, device=device)
        eom.max_iter = 3
        _quiet(eom.solve, fdd, Vd, res["t2"])                      # warm-up: lazy statics, cached transposed blocks
        eom = EOM_CCSD(no, n_excit=3, device=device)
        eom.max_iter = 60
        ctx.sync()
        t0 = time.perf_counter()
        ee = _quiet(eom.solve, fdd, Vd, res["t2"])
        ctx.sync()
        out["c5_davidson_s"] = time.perf_counter() - t0
        out["c5_davidson_passes"] = eom.iterations
        out["c5_davidson_ok"] = bool(abs(res["ccsd e"] - C5_DAVIDSON["ccsd_e"]) < 1e-9 and
                                     np.abs(np.asarray(ee) - np.asarray(C5_DAVIDSON["ee"])).max() < 1e-7 and
                                     eom.iterations == C5_DAVIDSON["passes"])
        out["c5_davidson_golden"] = "regression check: this engine's round-4 values (no reference run at this size)"
        # PARITY at this size: the sigma build is pinned to the reference's own output (c5_ok above), so the converged roots
        # are certified by it — Ritz vectors rebuilt, a fresh sigma applied once, |sigma r - e r| / |r| per root (outside the
        # timed region; EOM_CCSD.ritz_residuals).  The driver stops at |dE| < 1e-8 (eom_ccsd.py:150), which leaves residuals
        # of this size; a wrong root would give O(1).
        rr = eom.ritz_residuals(fdd, Vd, res["t2"])
        out["c5_davidson_ritz_residuals"] = rr
        # ... and the same solve carried on to |dE| < 1e-12 (outside the timed region): its roots must have residuals below 1e-6
        # under the fresh sigma AND agree with the roots of the timed solve — the timed roots are then eigenvalues of the
        # reference-pinned operator to that accuracy, whatever the round-4 literals say
        tight = EOM_CCSD(no, n_excit=3, device=device)
        tight.e_epsilon, tight.max_iter = C5_TIGHT_DE, 200
        ee_t = _quiet(tight.solve, fdd, Vd, res["t2"])
        rr_t = tight.ritz_residuals(fdd, Vd, res["t2"])
        out["c5_davidson_tight"] = {"dE": C5_TIGHT_DE, "passes": tight.iterations, "ritz_residuals": rr_t,
                                    "max_root_difference": float(np.abs(np.asarray(ee_t) - np.asarray(ee)).max())}
        out["c5_davidson_certified"] = bool(max(rr) < C5_RITZ_TOL and max(rr_t) < C5_TIGHT_RITZ_TOL and
                                            out["c5_davidson_tight"]["max_root_difference"] < 1e-7)
        ctx.close()
    finally:
        gc.enable()
    out["wall_s"] = time.perf_counter() - t_all
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nocc", type=int, default=50)
    ap.add_argument("--nvirt", type=int, default=200)
    ap.add_argument("--dcsd", action="store_true")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-diis", action="store_true")
    ap.add_argument("--events", choices=("auto", "timed", "separate"), default="auto",
                    help="where the per-GEMM HIP events of `roofline` are taken: inside the timed steps (eager launches), "
                         "or in a separate pass after them so that the timed steps run as the solver runs them (launch graph, "
                         "phase launches); auto = separate")
    ap.add_argument("--backend", default=os.environ.get("PYMES_DIST_BACKEND", "nccl"),
                    help="torch.distributed backend; 'gloo' lets several ranks share one GPU in test rigs")
    ap.add_argument("--collective-timeout-s", type=float, default=300.0,
                    help="process-group timeout: a collective that does not complete ends the rank with an error")
    ap.add_argument("--run-timeout-s", type=float, default=1500.0,
                    help="overall limit of a self-launched multi-rank run (the launcher stops the ranks and exits non-zero)")
    ap.add_argument("--owner-tiles", action="store_true",
                    help="N > 1: all-to-all of the ring-product tiles each pair owner reads instead of two all-gathers")
    ap.add_argument("--stub-collectives", action="store_true",
                    help="ONE GPU: run rank --as-rank of a world of --of ranks with every collective a no-op — that rank's "
                         "compute time (timings valid, energies not); the compute-only leg of the 1/2/4/8 curve")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the C2 / C4 / C5 legs that follow the headline measurement on a single GPU (~10 s)")
    ap.add_argument("--as-rank", type=int, default=0)
    ap.add_argument("--of", type=int, default=8)
    args = ap.parse_args()
    if args.owner_tiles:
        os.environ["PYMES_OWNER_TILES"] = "1"
    if args.stub_collectives and args.gpus != 1:
        raise SystemExit("--stub-collectives rehearses one rank on ONE GPU: use --gpus 1 --as-rank r --of N")

    if args.gpus > 1 and "RANK" not in os.environ and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        return launch_ranks(args.gpus, sys.argv[1:], timeout_s=args.run_timeout_s)   # no launcher around us: be the launcher

    # Native libraries print to the process's stdout on their own (RCCL: a five-line version banner at communicator set-up;
    # gloo logs there too): file descriptor 1 goes to stderr for the whole run and the ONE JSON line is written to the saved
    # descriptor at the end, so that stdout of rank 0 holds exactly that line whoever launched the ranks.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    ndev = torch.cuda.device_count()
    if world > 1 and args.backend == "nccl" and world > ndev:
        # RCCL with two ranks on one device fails or hangs in communicator set-up: refuse before touching the GPU
        raise SystemExit(f"bench.py: --backend nccl needs one GPU per rank, but WORLD_SIZE={world} > {ndev} visible "
                         f"device(s); use --backend gloo to let ranks share a GPU (test rigs only)")
    local = local % max(1, ndev)          # several ranks per device: gloo test rigs only (refused above for RCCL)
    torch.cuda.set_device(local)
    # PYMES_FORCE_SHARDED=1 with one rank: rehearse the one-process-per-GPU path (RCCL communicator of one rank)
    if world > 1 or os.environ.get("PYMES_FORCE_SHARDED"):
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # a collective that never completes (a dead peer, a wedged link) aborts the rank after the timeout instead of
        # hanging the run: RCCL's watchdog tears the process down, gloo raises; the launcher then stops the other ranks
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
        limit = datetime.timedelta(seconds=args.collective_timeout_s)
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local),
                                    timeout=limit)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world, timeout=limit)
    from pymes_amd import dist as pdist
    if args.stub_collectives:
        pdist.stub(args.as_rank, args.of)

    from pymes_amd.integral.device import DeviceIntegrals
    from pymes_amd.model import synthetic
    from pymes_amd.solver.ccsd import CCSD

    no, nv = args.nocc, args.nvirt
    B, eps = synthetic.factors(no, nv, seed=args.seed)
    t0 = time.time()
    # the engine runs on a stream of its own (launch-graph capture needs one); under torch.distributed the solver binds
    # it to torch's current stream instead, so that RCCL collectives and kernels are ordered on the device (dist.py)
    ints = DeviceIntegrals.from_factors(no, B, device=local)
    if args.stub_collectives and not ints.ctx.pairs_supported():
        raise SystemExit("--stub-collectives: the pair-sharded tail is not available for this nocc")
    ctx = ints.ctx
    ctx.sync()
    t_build = time.time() - t0
    if not pdist.sharded():
        # single rank: the engine works on a non-default stream that torch knows too, so that the per-phase events of
        # pymes_amd.dist.trace (torch.cuda.Event) are recorded on the stream the kernels run on
        engine_stream = torch.cuda.Stream(device=local)
        ctx.set_stream(engine_stream.cuda_stream)
        pdist.trace.stream = engine_stream
    del B
    f = np.diag(eps)
    solver = CCSD(no, is_dcsd=args.dcsd, is_diis=not args.no_diis, device=local)
    import contextlib, io
    t0 = time.time()
    st = solver.setup(f, ints)          # orbital energies, exchange-symmetry test of all 16 blocks, MP2 amplitudes, buffers
    ctx.sync()
    t_setup = time.time() - t0
    energies = []

    shortcuts = []       # T1 = 0 shortcut taken by a step (only the very first step from MP2 may: the workload has T1 != 0)

    def step():
        shortcuts.append(bool(st.get("t1_zero")))
        with contextlib.redirect_stdout(io.StringIO()):
            e = solver.iterate(st)
        energies.append(e[0] + e[1] + e[2])

    # the interpreter's cyclic collector stays off from here to the end of the timed region (as CCSD.solve does for its own
    # loop, and as timeit does): a generation-2 collection of this process takes ~38 ms and would land in some step
    import gc
    gc.collect()
    gc.disable()
    t0 = time.time()
    for _ in range(args.warmup):
        step()
    ctx.sync()
    t_warm = time.time() - t0           # includes the per-solve statics the first passes build (packed V+-, permutations)

    def fence():
        ctx.sync()
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
            torch.cuda.synchronize()

    sharded_run = pdist.sharded()
    # auto: a single rank times the loop body as the solver runs it (residual part replayed as a launch graph: one launch
    # instead of ~130, so the figure does not depend on how fast this box's host happens to enqueue) and takes the per-GEMM
    # events in a second, eager pass over the same number of steps
    # (round 6: one process per GPU as well — events inside the timed steps switch the phase launches off, DESIGN 6f, and cost
    # 0.1 ms of a 27-ms rank; "--events timed" keeps the old form)
    separate = args.events != "timed"
    if separate:          # the launch graph of a variant is recorded on its second pass: keep that out of the timed steps
        for _ in range(max(0, 3 - args.warmup)):
            step()
    n_before_timed = len(shortcuts)
    pdist.trace.enable(True)
    ctx.stats(reset=True)
    ctx.prof_enable(not separate)
    ctx.prof_reset()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    timed_energy = energies[-1]
    # no timed step may have run the T1 = 0 shortcut (residuals from undressed f and V: a third of the dressing work less):
    # with --warmup 0 the very first step from MP2 legitimately does, anything else means the step has silently shrunk
    if any(shortcuts[max(n_before_timed, 1):]):
        raise SystemExit("bench.py: a timed step took the T1 = 0 shortcut - the measured step is not the full CCSD iteration")
    phases, collectives = pdist.trace.summary(args.steps)      # of the timed steps only
    pdist.trace.on = False
    replayed = bool(st.get("graph") is not None and separate)
    if separate:        # the same number of steps again, eagerly, with one event pair per GEMM call
        # (the last timed step has enqueued the residual segment of the step after it — the pipelined iteration, DESIGN 6d;
        # the counting pass must build every residual itself, or its first step would count none)
        ctx.sync()
        st.pop("residuals_in_flight", None)
        ctx.stats(reset=True)
        ctx.prof_enable(True)
        ctx.prof_reset()
        for _ in range(args.steps):
            step()
        fence()
    gc.enable()
    prof = ctx.prof_query()
    prof_dma = ctx.prof_query(kernel_class=1)
    stats = ctx.stats()
    ctx.prof_enable(False)
    if dist.is_initialized():
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if not all(np.isfinite(energies)) and not args.stub_collectives:
        raise SystemExit("non-finite energy in the timed region")

    if rank == 0:
        s_per_step = elapsed / args.steps
        def tf(p):
            return p["flops"] / (p["ms"] * 1e-3) / 1e12 if p["ms"] > 0 else 0.0
        # dominant kernel: the LDS-DMA 128x128 MFMA GEMM (o^3v^3 ring products, packed ladders); small workloads
        # never reach it, then the figures are those of all GEMM launches
        dom = prof_dma if prof_dma["launches"] else prof
        achieved = tf(dom)
        ref_fl = reference_flops(no, nv, args.dcsd)
        cap, high = ctx.workspace()
        traffic, traffic_source = pmc_traffic_per_launch(no, nv, world, "void dgemm_glds_kernel" if
                                                         prof_dma["launches"] else "void dgemm_")
        line = {
            "metric": "ccsd_iteration_time", "value": s_per_step, "unit": "s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * s_per_step,
            "higher_is_better": False, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{'DCSD' if args.dcsd else 'CCSD'} iteration, synthetic density-fitted "
                                   f"integrals (nocc={no}, nvirt={nv}), seed {args.seed}",
                       "no": no, "nv": nv, "diis": not args.no_diis,
                       "parallelism": "single GPU" if world == 1 else
                       f"ring-product column slabs + packed-ladder / Q_kb rows + pair-sharded update and DIIS over {world} ranks; "
                       f"all-gathers of ETd, ETx, Q_kb (overlapped) and T2 per iteration "
                       f"({args.backend})"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_MFMA_PEAK_TFLOPS,
                         "traffic": traffic, "traffic_source": traffic_source,
                         "events": "separate eager pass after the timed steps" if separate else "inside the timed steps",
                         "kernel": ("dgemm_glds_kernel + dgemm_glds_group_kernel" if prof_dma["launches"] else "dgemm_kernel") +
                                   " (v_mfma_f64_16x16x4_f64)",
                         "launches_per_step": dom["kernel_launches"] / args.steps,
                         "avg_launch_ms": dom["ms"] / max(1, dom["kernel_launches"]),
                         "flops_per_launch": dom["flops"] / max(1, dom["kernel_launches"]),
                         "ms_per_step": dom["ms"] / args.steps,
                         "all_gemm": {"achieved": tf(prof), "frac": tf(prof) / FP64_MFMA_PEAK_TFLOPS,
                                      "calls_per_step": prof["launches"] / args.steps,
                                      "ms_per_step": prof["ms"] / args.steps,
                                      "executed_flops_per_step": prof["flops"] / args.steps}},
            "iteration": {"reference_algorithmic_flops": ref_fl,
                          "algorithmic_tflops": ref_fl / s_per_step / 1e12,
                          "algorithmic_frac_of_fp64_peak": ref_fl / s_per_step / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                          "executed_gemm_tflops_over_step": stats["gemm_flops"] / args.steps / s_per_step / 1e12,
                          "permute_gbytes_per_step": stats["permute_bytes"] / args.steps / 1e9,
                          "integral_build_s": t_build, "setup_s": t_setup,
                          "first_passes_extra_s": max(0.0, t_warm - args.warmup * s_per_step) if args.warmup else None,
                          "workspace_high_water_gb": high / 1e9,
                          "launch_graph_replay": replayed, "last_energy": timed_energy,
                          # device time between phase marks (HIP events on the engine's stream) and host wall time between
                          # the same marks: a phase whose host time is close to its device time waited for the host
                          "phases_ms": phases, "phases_host_ms": getattr(pdist.trace, "host_ms", {})},
        }
        if sharded_run:
            # rank 0's view of the sharded iteration: device time per phase (HIP events on the stream the kernels and the
            # collectives are ordered on), per collective the time the stream stood still in its wait() (= exposed
            # communication) and the bytes the rank put on the wire; the roofline block above is this rank's GEMMs
            prank, pworld, _ = pdist.world()
            line["multi_gpu"] = {
                "rccl_world": dist.get_world_size() if dist.is_initialized() else 0,
                "backend": args.backend if dist.is_initialized() else "none (collectives stubbed)",
                "rank": prank, "of": pworld, "collectives_stubbed": bool(args.stub_collectives),
                "owner_tiles": bool(os.environ.get("PYMES_OWNER_TILES")),
                "phases_ms": phases, "phases_host_ms": getattr(pdist.trace, "host_ms", {}), "collectives": collectives,
                "exposed_wait_ms": sum(c["exposed_wait_ms"] for c in collectives.values()),
                "wire_gbytes_per_step": sum(c["wire_bytes"] for c in collectives.values()) / 1e9,
                "devices_visible": ndev}
            if args.stub_collectives:
                line["config"]["parallelism"] = (f"rank {prank} of {pworld}, collectives stubbed (compute only, one GPU): "
                                                 "timings valid, energies not")
        if world == 1 and not args.no_cpu_baseline and not args.stub_collectives:
            from oracle.baseline import sample, algorithmic_fma
            cpu = sample(no, nv)
            # the sample times the doubles residual (93 % of the reference's flops); dressing + singles + Fock are
            # priced at the same rate through the SURVEY 8(d) flop table so that the value is the bench metric
            to_iter = ref_fl / (2.0 * algorithmic_fma(no, nv, args.dcsd))
            line["cpu_baseline"] = {
                "value": cpu["faithful"]["seconds_per_doubles_residual"] * to_iter, "unit": "s", "cores": 1,
                "kind": "port",
                "sample": "oracle contraction forms (plain np.einsum, as the reference's T2 residual): ladder on a "
                          "(1 x 25) (a,b)-slab and one o^3v^3 ring term on an a-slab of 1, extrapolated linearly to "
                          f"one doubles residual with the SURVEY 8(d) flop table, x {to_iter:.3f} (flop ratio "
                          "iteration / doubles residual) for one CCSD iteration",
                "doubles_residual_s": cpu["faithful"]["seconds_per_doubles_residual"],
                # the same contractions through tensordot / dgemm on every host core (SURVEY 8(d) mode B)
                "blas_value": cpu["blas"]["seconds_per_doubles_residual"] * to_iter, "blas_cores": cpu["cores"]["blas"],
                "blas_all_cores": {"value": cpu["blas"]["seconds_per_doubles_residual"] * to_iter, "unit": "s",
                                   "cores": cpu["cores"]["blas"]}}
        if (world == 1 and not sharded_run and not args.no_other_configs and (no, nv) == (50, 200) and not args.dcsd):
            # BASELINE configs 2, 4, 5 under the same clock, AFTER the headline region: the (50,200) context goes first
            del st, solver
            ctx.close()
            try:
                line["other_configs"] = other_configs(device=local)
            except Exception as exc:          # the headline line must survive a failure of a side leg
                line["other_configs"] = {"error": f"{type(exc).__name__}: {exc}"}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    os.close(json_fd)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
