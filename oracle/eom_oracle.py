"""numpy oracle of the closed-shell EOM-CCSD sigma build and Davidson-like driver
(pymes/solver/eom_ccsd.py).  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Table-driven restatement: every row is (coefficient, subscripts, operands) of one term of
the reference; "f??" are blocks of the (T1-dressed) Fock matrix, four-letter names are
(T1-dressed) V blocks, "t" = ground-state T2, "u1"/"u2" = trial vector.  Pinned against the
imported reference by oracle/make_golden_eom.py -> tests/golden/eom_*.{npz,json}.
"""
import numpy as np

# eom_ccsd.py:288-308
SINGLES_TERMS = (
    (+2.0, "jb,baji->ai", ("fov", "u2")),
    (-1.0, "ji,aj->ai", ("foo", "u1")),
    (-1.0, "jb,abji->ai", ("fov", "u2")),
    (+1.0, "ab,bi->ai", ("fvv", "u1")),
    (+2.0, "jabi,bj->ai", ("iabj", "u1")),
    (-1.0, "jaib,bj->ai", ("iajb", "u1")),
    (-2.0, "jkib,abjk->ai", ("ijka", "u2")),
    (+2.0, "jabc,bcji->ai", ("iabc", "u2")),
    (+1.0, "jkib,bajk->ai", ("ijka", "u2")),
    (-1.0, "jacb,bcji->ai", ("iabc", "u2")),
    (+4.0, "jkbc,baji,ck->ai", ("ijab", "t", "u1")),
    (-2.0, "jkbc,bajk,ci->ai", ("ijab", "t", "u1")),
    (-2.0, "jkbc,bcji,ak->ai", ("ijab", "t", "u1")),
    (-2.0, "jkbc,abji,ck->ai", ("ijab", "t", "u1")),
    (-2.0, "jkcb,baji,ck->ai", ("ijab", "t", "u1")),
    (+1.0, "jkbc,abjk,ci->ai", ("ijab", "t", "u1")),
    (+1.0, "jkcb,bcji,ak->ai", ("ijab", "t", "u1")),
    (+1.0, "jkcb,abji,ck->ai", ("ijab", "t", "u1")),
)

# eom_ccsd.py:332-373: terms under the permutation P(ijab, jiba)
DOUBLES_TERMS_P = (
    (-2.0, "klid,abkj,dl->abij", ("ijka", "t", "u1")),
    (-2.0, "klci,cbkj,al->abij", ("ijak", "t", "u1")),
    (+2.0, "kacd,cbkj,di->abij", ("iabc", "t", "u1")),
    (+2.0, "ladc,cbij,dl->abij", ("iabc", "t", "u1")),
    (-1.0, "kd,abkj,di->abij", ("fov", "t", "u1")),
    (-1.0, "lc,cbij,al->abij", ("fov", "t", "u1")),
    (+1.0, "klid,abkl,dj->abij", ("ijka", "t", "u1")),
    (+1.0, "klic,cbkj,al->abij", ("ijka", "t", "u1")),
    (+1.0, "klid,adkj,bl->abij", ("ijka", "t", "u1")),
    (-1.0, "kbij,ak->abij", ("iajk", "u1")),
    (+1.0, "kldi,bdkj,al->abij", ("ijak", "t", "u1")),
    (-1.0, "kacd,bckj,di->abij", ("iabc", "t", "u1")),
    (+1.0, "kldi,abkj,dl->abij", ("ijak", "t", "u1")),
    (-1.0, "kadc,cbkj,di->abij", ("iabc", "t", "u1")),
    (-1.0, "kadc,bcki,dj->abij", ("iabc", "t", "u1")),
    (-1.0, "lacd,cdji,bl->abij", ("iabc", "t", "u1")),
    (-1.0, "lacd,cbij,dl->abij", ("iabc", "t", "u1")),
    (+1.0, "abic,cj->abij", ("abic", "u1")),
    (+4.0, "klcd,caki,dblj->abij", ("ijab", "t", "u2")),
    (-2.0, "klcd,cakl,dbij->abij", ("ijab", "t", "u2")),
    (-2.0, "klcd,cdki,ablj->abij", ("ijab", "t", "u2")),
    (-2.0, "klcd,caki,bdlj->abij", ("ijab", "t", "u2")),
    (+2.0, "kaci,cbkj->abij", ("iabj", "u2")),
    (-2.0, "klcd,acki,dblj->abij", ("ijab", "t", "u2")),
    (-2.0, "kldc,caki,dblj->abij", ("ijab", "t", "u2")),
    (-2.0, "kldc,abkj,dcil->abij", ("ijab", "t", "u2")),
    (-2.0, "lkcd,cbij,adlk->abij", ("ijab", "t", "u2")),
    (-1.0, "ki,abkj->abij", ("foo", "u2")),
    (+1.0, "ac,cbij->abij", ("fvv", "u2")),
    (-1.0, "kaic,cbkj->abij", ("iajb", "u2")),
    (-1.0, "kbic,ackj->abij", ("iajb", "u2")),
    (+1.0, "klcd,ackl,dbij->abij", ("ijab", "t", "u2")),
    (+1.0, "kldc,cdki,ablj->abij", ("ijab", "t", "u2")),
    (+1.0, "klcd,acki,bdlj->abij", ("ijab", "t", "u2")),
    (-1.0, "kaci,bckj->abij", ("iabj", "u2")),
    (+1.0, "kldc,acki,dblj->abij", ("ijab", "t", "u2")),
    (+1.0, "kldc,abkj,dcli->abij", ("ijab", "t", "u2")),
    (+1.0, "kldc,caki,dbjl->abij", ("ijab", "t", "u2")),
    (+1.0, "kldc,ackj,dbil->abij", ("ijab", "t", "u2")),
    (+1.0, "lkcd,cbij,dalk->abij", ("ijab", "t", "u2")),
)
# eom_ccsd.py:380-383: not permuted
DOUBLES_TERMS_N = (
    (+1.0, "klij,abkl->abij", ("klij", "u2")),
    (+1.0, "kldc,abkl,dcij->abij", ("ijab", "t", "u2")),
    (+1.0, "lkcd,cdij,ablk->abij", ("ijab", "t", "u2")),
    (+1.0, "abcd,cdij->abij", ("abcd", "u2")),
)


# eom_ccsd.py:169-198: (coefficient, subscripts, operands); results are broadcast onto [a,i]
DIAG_SINGLES_TERMS = (
    (+2.0, "iaai->ai", ("iabj",)),
    (-1.0, "iaia->ai", ("iajb",)),
    (+4.0, "jiba,baji->ai", ("ijab", "t")),
    (-2.0, "jkba,abjk->a", ("ijab", "t")),
    (-2.0, "jicb,bcji->i", ("ijab", "t")),
    (-2.0, "jiba,abji->ai", ("ijab", "t")),
    (-2.0, "jiab,baji->ai", ("ijab", "t")),
    (+1.0, "jkab,abjk->a", ("ijab", "t")),
    (+1.0, "jicb,bcji->i", ("ijab", "t")),
    (+1.0, "jiab,abji->ai", ("ijab", "t")),
)
# eom_ccsd.py:200-252: the terms under P(ijab, jiba); results are broadcast onto [a,b,i,j].  The reference places
# "ibib->bi" on the axes (0, 2) like "iaia->ai" (:232), which makes the two terms identical: kept as written.
DIAG_DOUBLES_TERMS_P = (
    (+4.0, "kica,caki->ai", ("ijab", "t")),
    (-2.0, "klca,cakl->a", ("ijab", "t")),
    (-2.0, "kicd,cdki->i", ("ijab", "t")),
    (-2.0, "kica,caki->ai", ("ijab", "t")),
    (+2.0, "iaai->ai", ("iabj",)),
    (-2.0, "kica,acki->ai", ("ijab", "t")),
    (-2.0, "kiac,caki->ai", ("ijab", "t")),
    (-2.0, "kjab,abkj->abj", ("ijab", "t")),
    (-2.0, "ijcb,cbij->ij", ("ijab", "t")),
    (-1.0, "iaia->ai", ("iajb",)),
    (-1.0, "iaia->ai", ("iajb",)),
    (+1.0, "klca,ackl->a", ("ijab", "t")),
    (+1.0, "kidc,cdki->i", ("ijab", "t")),
    (+1.0, "kicb,acki->ai", ("ijab", "t")),
    (-1.0, "iaai->ai", ("iabj",)),
    (+1.0, "kiac,acki->ai", ("ijab", "t")),
    (+1.0, "kiab,abkj->abij", ("ijab", "t")),
    (+1.0, "kjac,caki->aij", ("ijab", "t")),
    (+1.0, "kjac,ackj->aj", ("ijab", "t")),
    (+1.0, "ijca,cbij->abij", ("ijab", "t")),
)
# eom_ccsd.py:255-265: not permuted
DIAG_DOUBLES_TERMS_N = (
    (+1.0, "ijij->ij", ("klij",)),
    (+1.0, "klab,abkl->ab", ("ijab", "t")),
    (+1.0, "ijcd,cdij->ij", ("ijab", "t")),
    (+1.0, "abab->ab", ("abcd",)),
)


def _broadcast(x, labels, target):
    """Place the result of an einsum with output ``labels`` on the axes of ``target`` (e.g. "abij")."""
    idx = tuple(slice(None) if c in labels else None for c in target)
    order = [labels.index(c) for c in target if c in labels]
    return np.transpose(x, order)[idx]


def _diag_sum(terms, env, target, shape):
    out = np.zeros(shape)
    for c, spec, names in terms:
        spec = spec.replace(" ", "")
        out = out + c * _broadcast(np.einsum(spec, *[env[n] for n in names]), spec.split("->")[1], target)
    return out


def diag_singles(no, f, Vd, t2):
    """EOM_CCSD.get_diag_singles, eom_ccsd.py:169-198."""
    nv = f.shape[0] - no
    env = dict(Vd, t=t2)
    out = -f.diagonal()[:no][None, :] + f.diagonal()[no:][:, None]
    return out + _diag_sum(DIAG_SINGLES_TERMS, env, "ai", (nv, no))


def diag_doubles(no, f, Vd, t2):
    """EOM_CCSD.get_diag_doubles, eom_ccsd.py:200-266."""
    nv = f.shape[0] - no
    env = dict(Vd, t=t2)
    shape = (nv, nv, no, no)
    out = _diag_sum(DIAG_DOUBLES_TERMS_P, env, "abij", shape)
    out = out - f.diagonal()[:no][None, None, :, None] + f.diagonal()[no:][:, None, None, None]      # :228
    out = out + out.transpose(1, 0, 3, 2)                                                               # :253
    return out + _diag_sum(DIAG_DOUBLES_TERMS_N, env, "abij", shape)


def _env(no, f, Vd, t2, u1, u2):
    env = dict(Vd)
    env.update(foo=f[:no, :no], fov=f[:no, no:], fvv=f[no:, no:], t=t2, u1=u1, u2=u2)
    return env


def sigma_singles(no, f, Vd, u1, u2, t2):
    """EOM_CCSD.update_singles, eom_ccsd.py:268-310."""
    env = _env(no, f, Vd, t2, u1, u2)
    out = np.zeros_like(u1)                      # complex u (FEAST / real-time callers) flows through
    for c, spec, names in SINGLES_TERMS:
        out += c * np.einsum(spec, *[env[n] for n in names], optimize=True)
    return out


def sigma_doubles(no, f, Vd, u1, u2, t2):
    """EOM_CCSD.update_doubles, eom_ccsd.py:312-385."""
    env = _env(no, f, Vd, t2, u1, u2)
    out = np.zeros_like(u2)
    for c, spec, names in DOUBLES_TERMS_P:
        out += c * np.einsum(spec, *[env[n] for n in names], optimize=True)
    out = out + out.transpose(1, 0, 3, 2)                               # :377
    for c, spec, names in DOUBLES_TERMS_N:
        out += c * np.einsum(spec, *[env[n] for n in names], optimize=True)
    return out


def orthonormalise(us1, us2):
    """EOM_CCSD.QR, eom_ccsd.py:512-541: thin QR of the stacked [singles; doubles] columns."""
    n1 = us1[0].size
    mat = np.stack([np.concatenate([a.ravel(), b.ravel()]) for a, b in zip(us1, us2)], axis=1)
    Q, _ = np.linalg.qr(mat)
    return ([Q[:n1, i].reshape(us1[0].shape) for i in range(Q.shape[1])],
            [Q[n1:, i].reshape(us2[0].shape) for i in range(Q.shape[1])])


def eom_solve(no, f, Vd, t2, n_excit=3, max_iter=500, e_epsilon=1e-8, sigma=None):
    """EOM_CCSD.solve, eom_ccsd.py:46-167 (same subspace bookkeeping, including the collapse at
    4*n_excit vectors and the (e - D_ai[guess] + 1e-5) preconditioner)."""
    nv = f.shape[0] - no
    eps_o, eps_v = f.diagonal()[:no], f.diagonal()[no:]
    D_ai = -(eps_o[None, :] - eps_v[:, None]).ravel()
    guess = np.argsort(D_ai)[:n_excit]
    us1, us2 = [], []
    for g in guess:
        a = np.zeros(nv * no)
        a[g] = 1.0
        us1.append(a.reshape(nv, no))
        us2.append(np.zeros((nv, nv, no, no)))
    sigma = sigma or (lambda u1, u2: (sigma_singles(no, f, Vd, u1, u2, t2), sigma_doubles(no, f, Vd, u1, u2, t2)))
    e_excit = np.zeros(n_excit)
    e_old = e_excit
    max_dim = 4 * n_excit
    history = []
    e = e_excit
    for it in range(max_iter):
        dim = len(us1)
        us1, us2 = orthonormalise(us1, us2)
        ws = [sigma(us1[l], us2[l]) for l in range(dim)]
        B = np.zeros((dim, dim))
        for l in range(dim):
            for j in range(dim):
                B[j, l] = np.vdot(us1[j], ws[l][0]) + np.vdot(us2[j], ws[l][1])
        e_old = e_excit                                                 # :110, every pass
        lam, vec = np.linalg.eig(B)
        pick = lam.argsort()[:n_excit]
        e = np.real(lam[pick])
        v = np.real(vec[:, pick])
        if dim >= max_dim:                                              # collapse :122-133
            n1 = [sum(us1[l] * v[l, n] for l in range(dim)) for n in range(n_excit)]
            n2 = [sum(us2[l] * v[l, n] for l in range(dim)) for n in range(n_excit)]
            us1, us2 = n1, n2
            e_excit = e_old
            diff = history[-1][1] if history else np.inf
        else:                                                           # expand :135-147
            for n in range(n_excit):
                y1 = sum((ws[l][0] - e[n] * us1[l]) * v[l, n] for l in range(dim))
                y2 = sum((ws[l][1] - e[n] * us2[l]) * v[l, n] for l in range(dim))
                den = e[n] - D_ai[guess[n]] + 1e-5
                us1.append(y1 / den)
                us2.append(y2 / den)
            e_old = e_excit
            diff = float(np.linalg.norm(e_excit - e))
            e_excit = e
        history.append((e.copy(), diff))
        if diff < e_epsilon:
            break
    return {"e": e_excit, "iterations": len(history), "history": history}
