"""snp_autoSVD / bed_autoSVD — host mirror of R/autoSVD.R:67-186,226-339.

Everything that touches genotypes (MAF/MAC counts, clumping, the partial SVD) runs on the
GPU through the C ABI.  The outer loop also needs three functions of the external package
bigutilsr (>= 0.3.3, not in the reference tree and without golden data there):
``dist_ogk`` (robust Mahalanobis distance with the orthogonalised Gnanadesikan-Kettenring
estimator of Maronna & Zamar 2002, as implemented by robustbase::covOGK with scaleTau2 and
hard rejection at beta = 0.9), ``rollmean`` (Gaussian-weighted rolling mean with renormalised
edge windows) and ``tukey_mc_up`` (upper Tukey fence adjusted for skewness with the medcouple,
Hubert & Vandervieren 2008, with the coefficient chosen for a family-wise type-I error
``alpha`` under normality).  They are restated below from their published descriptions.

PARITY STATUS of this module: **unpinned** — neither the reference tree nor this image holds
bigutilsr or outputs of it, so `attr(, "subset")` cannot be compared with the reference here;
the tests assert the behavioural properties of tests/testthat/test-2-autoSVD.R instead.
"""
import numpy as np

from .bed import ERROR_DIM, assert_lengths, bed_MAF, bed_scaleBinom, cols_along, rows_along
from .ld import _ind, bed_clumping, big_randomSVD, chr_groups, snp_clumping, snp_MAF, snp_scaleBinom
from .svd import bed_randomSVD


# ---- bigutilsr restatements (host, O(m k^2) / O(m log m)) ---------------------------------
def _erho(b):
    from scipy.stats import norm
    return 2 * ((1 - b * b) * norm.cdf(b) - b * norm.pdf(b) + b * b) - 1


def scale_tau2(x, c1=4.5, c2=3.0, mu_too=False):
    """robustbase::scaleTau2 (Maronna & Zamar 2002), consistency = TRUE."""
    x = np.asarray(x, dtype=np.float64)
    n = x.size
    med = np.median(x)
    xa = np.abs(x - med)
    sigma0 = np.median(xa)
    if sigma0 <= 0:
        return (med, 0.0) if mu_too else 0.0
    w = np.maximum(0.0, 1.0 - (xa / (sigma0 * c1)) ** 2) ** 2
    mu = np.sum(x * w) / np.sum(w)
    rho = np.minimum(((x - mu) / sigma0) ** 2, c2 * c2)
    from scipy.stats import norm
    q = norm.ppf(0.75)                      # sigma0 is the raw MAD: Es2(c2) = Erho(c2 * qnorm(3/4))
    s = sigma0 * np.sqrt(np.sum(rho) / (n * _erho(c2 * q)))
    return (mu, s) if mu_too else s


def _map_threads(f, items):
    import os
    from concurrent.futures import ThreadPoolExecutor
    items = list(items)
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    with ThreadPoolExecutor(max_workers=max(1, min(16, ncpu, len(items)))) as ex:
        return list(ex.map(f, items))


def covrob_ogk(U, niter=2, beta=0.9):
    """Orthogonalised Gnanadesikan-Kettenring estimate with reweighting (hard rejection)."""
    from scipy.stats import chi2
    U = np.asarray(U, dtype=np.float64)
    n, p = U.shape
    Z = U.copy()
    A = []
    # the p + p (p - 1) robust scales of a round are independent of each other and each is a few passes over n
    # values (two medians, two weighted sums): on a million variants they are the whole cost of the outlier step, so
    # they are spread over a few threads (numpy releases the interpreter lock inside them; same values, any order)
    run = _map_threads if n * p >= 200000 else (lambda f, items: [f(t) for t in items])
    for _ in range(niter):
        d = np.array(run(lambda j: scale_tau2(Z[:, j]), range(p)))
        d[d <= 0] = 1.0
        Z = Z / d
        R = np.eye(p)
        pairs = [(i, j) for i in range(p) for j in range(i)]
        vals = run(lambda ij: (scale_tau2(Z[:, ij[0]] + Z[:, ij[1]]) ** 2 -
                               scale_tau2(Z[:, ij[0]] - Z[:, ij[1]]) ** 2) / 4, pairs)
        for (i, j), v in zip(pairs, vals):
            R[i, j] = R[j, i] = v
        _, E = np.linalg.eigh(R)
        E = E[:, ::-1]
        A.append(d[:, None] * E)
        Z = Z @ E
    ms = run(lambda j: scale_tau2(Z[:, j], mu_too=True), range(p))
    mu = np.array([t[0] for t in ms]); sig = np.array([t[1] for t in ms])
    sig[sig <= 0] = 1.0
    wdist = np.sum(((Z - mu) / sig) ** 2, axis=1)
    d0 = np.median(wdist) * chi2.ppf(beta, p) / chi2.ppf(0.5, p)
    keep = wdist <= d0
    center = U[keep].mean(0)
    cov = np.cov(U[keep], rowvar=False).reshape(p, p)
    return dict(center=center, cov=cov, weights=keep)


def covrob_ogk_device(U, niter=2, beta=0.9, c1=4.5, c2=3.0):
    """covrob_ogk with its robust scales on the device (csrc/robust.hip, bsn_robust_*): the same loop, statement for
    statement — the p + p (p - 1) scales of a round are one batch each (medians by radix select), the m x p working
    matrix lives in HBM, the host keeps the p x p matrices and the final moments.  Values equal to the host path's to
    rounding (the sums are taken in another order): tests/test_gpu_autosvd.py."""
    import ctypes as C
    from scipy.stats import chi2
    from . import _lib
    from ._lib import check, f64p, ptr
    L = _lib.load()
    U = np.asarray(U, dtype=np.float64)
    n, p = U.shape
    if p > 64:
        raise ValueError("the device path of dist_ogk holds at most 64 columns")
    Z = _lib.DeviceArray.from_numpy(U)
    npair = p * (p - 1) // 2
    try:
        for _ in range(niter):
            d = np.empty(p)
            check(L.bsn_robust_scale_tau2(Z.ptr, n, n, p, c1, c2, None, ptr(d, f64p)))
            d[d <= 0] = 1.0
            check(L.bsn_robust_scale_cols(Z.ptr, n, n, p, ptr(d, f64p)))
            R = np.eye(p)
            if npair:
                ss, sd = np.empty(npair), np.empty(npair)
                check(L.bsn_robust_pair_scales(Z.ptr, n, n, p, c1, c2, ptr(ss, f64p), ptr(sd, f64p)))
                t = 0
                for i in range(p):
                    for j in range(i):
                        R[i, j] = R[j, i] = (ss[t] ** 2 - sd[t] ** 2) / 4
                        t += 1
            _, E = np.linalg.eigh(R)
            E = np.asfortranarray(E[:, ::-1])
            check(L.bsn_robust_rotate(Z.ptr, n, n, p, ptr(E, f64p)))
        mu, sig = np.empty(p), np.empty(p)
        check(L.bsn_robust_scale_tau2(Z.ptr, n, n, p, c1, c2, ptr(mu, f64p), ptr(sig, f64p)))
        sig[sig <= 0] = 1.0
        wdist = np.empty(n)
        check(L.bsn_robust_wdist(Z.ptr, n, n, p, ptr(mu, f64p), ptr(sig, f64p), ptr(wdist, f64p)))
    finally:
        Z.free()
    d0 = np.median(wdist) * chi2.ppf(beta, p) / chi2.ppf(0.5, p)
    keep = wdist <= d0
    center = U[keep].mean(0)
    cov = np.cov(U[keep], rowvar=False).reshape(p, p)
    return dict(center=center, cov=cov, weights=keep)


def dist_ogk_device(U, niter=2, beta=0.9, c1=4.5, c2=3.0):
    """dist_ogk end to end on the device (bsn_robust_dist_ogk): covrob_ogk_device's rounds, the hard rejection, centre and
    covariance of the kept rows and the distances in one call — only p x p matrices visit the host, where at a million
    variants the host tail of the step-by-step version (a fancy-index copy of the kept rows, np.cov, the m x p products of
    the distances) cost more than the rounds.  Same values to rounding: tests/test_gpu_autosvd.py."""
    import ctypes as C
    from scipy.stats import chi2
    from . import _lib
    from ._lib import check, f64p, ptr
    L = _lib.load()
    U = np.asarray(U, dtype=np.float64)
    n, p = U.shape
    if p > 64:
        raise ValueError("the device path of dist_ogk holds at most 64 columns")
    Z = _lib.DeviceArray.from_numpy(U)
    out = np.empty(n)
    try:
        check(L.bsn_robust_dist_ogk(Z.ptr, n, n, p, int(niter), float(chi2.ppf(beta, p) / chi2.ppf(0.5, p)), c1, c2,
                                    ptr(out, f64p), None))
    finally:
        Z.free()
    return out


def dist_ogk(U, niter=2, beta=0.9, device=False):
    """squared robust Mahalanobis distances (bigutilsr::dist_ogk).  device=True: on the GPU (dist_ogk_device; what
    snp_autoSVD / bed_autoSVD use — their loadings come from a solve on that GPU); device="steps": the robust scales
    on the GPU and the loop around them here (covrob_ogk_device, round 5's path, kept as the comparator)."""
    U = np.asarray(U, dtype=np.float64)
    if device is True:
        return dist_ogk_device(U, niter, beta)
    est = covrob_ogk_device(U, niter, beta) if device else covrob_ogk(U, niter, beta)
    Xc = U - est["center"]
    return np.einsum("ij,ij->i", Xc @ np.linalg.pinv(est["cov"]), Xc)


def _rollmean_weights(size):
    from scipy.stats import norm
    half = int(np.floor(size))
    length = 2 * half + 1
    a = 3.0 / 8 if length <= 10 else 0.5                       # stats::ppoints
    pp = (np.arange(1, length + 1) - a) / (length + 1 - 2 * a)
    lims = norm.ppf([pp[0], pp[-1]])
    return norm.pdf(np.linspace(lims[0], lims[1], length)), half


def rollmean(x, size):
    """bigutilsr::rollmean: Gaussian weights over 2*floor(size)+1 points, edge windows
    renormalised by the weights they contain."""
    x = np.asarray(x, dtype=np.float64)
    if size == 0:
        return x
    w, half = _rollmean_weights(size)
    if w.size >= x.size:
        raise ValueError("Parameter 'size' is too large.")
    num = np.convolve(x, w[::-1], mode="full")[half:half + x.size]
    # the weights a window contains: all of them except within `half` of an end (partial sums from either side)
    den = np.full(x.size, w.sum())
    cs = np.cumsum(w)
    den[:half] = cs[half:2 * half]
    den[x.size - half:] = np.cumsum(w[::-1])[half:2 * half][::-1]
    return num / den


def rollmean_groups(x, size, groups, device=False):
    """rollmean inside every group of indices (R/autoSVD.R:143-144: per chromosome).  `groups` as chr_groups returns
    them.  device=True and the groups consecutive index ranges (a genotype file sorted by chromosome): one launch for
    all of them (bsn_robust_rollmean)."""
    x = np.asarray(x, dtype=np.float64)
    out = np.full(x.size, np.nan)
    if size == 0:
        for _c, idx in groups:
            out[idx] = x[idx]
        return out
    w, half = _rollmean_weights(size)
    for _c, idx in groups:
        if w.size >= idx.size:
            raise ValueError("Parameter 'size' is too large.")
    starts = np.array([idx[0] for _c, idx in groups], dtype=np.int64)
    stops = np.array([idx[-1] + 1 for _c, idx in groups], dtype=np.int64)
    sizes = np.array([idx.size for _c, idx in groups], dtype=np.int64)
    by_start = np.argsort(starts)
    runs = (np.array_equal(stops - starts, sizes) and starts[by_start[0]] == 0 and stops[by_start[-1]] == x.size
            and np.array_equal(starts[by_start][1:], stops[by_start][:-1]))
    if device and runs and len(groups) <= 1 << 20:
        import ctypes as C
        from . import _lib
        from ._lib import check, f64p, ptr
        off = np.ascontiguousarray(np.r_[starts[by_start], x.size].astype(np.int64))
        xc, wc = np.ascontiguousarray(x), np.ascontiguousarray(w)
        check(_lib.load().bsn_robust_rollmean(ptr(xc, f64p), x.size, ptr(wc, f64p), int(w.size),
                                              off.ctypes.data_as(C.POINTER(C.c_int64)), int(off.size - 1), ptr(out, f64p)))
        return out
    for _c, idx in groups:
        out[idx] = rollmean(x[idx], size)
    return out


def medcouple(x, device=False, assume_sorted=False):
    """Medcouple (Brys, Hubert & Struyf 2004): the median of the kernel
        h(xi, xj) = ((xi - med) - (med - xj)) / (xi - xj),   xi >= med >= xj,
    over all such pairs; for pairs tied AT the median (xi = xj = med) the kernel is defined through
    their order among the m ties, h = sign(m - 1 - i - j) with i, j the positions among the tied values
    on each side (the convention of robustbase::mc and of the original paper).  The kernel is monotone
    in both arguments, so #{h <= t} is a sum of searchsorted counts and the median is found by bisection
    on t, followed by a snap to the nearest attained kernel value (the median IS a kernel value).
    device=True: the counts of the bisection — one binary search per value above the median — and the window of kernel
    values that ends it on the GPU (bsn_robust_mc_count, bsn_robust_mc_window; what snp_autoSVD / bed_autoSVD use);
    same integers, same kernel values, same result.  assume_sorted: x is ascending already."""
    x = np.asarray(x, dtype=np.float64)
    if not assume_sorted:
        x = np.sort(x)
    n = x.size
    if n < 3:
        return 0.0
    med = np.median(x)
    up = x[x >= med] - med          # zi+ >= 0, ascending
    lo = (med - x[x <= med])[::-1]  # zj- >= 0 ... as positive distances, ascending
    if up[-1] == 0 and lo[-1] == 0:      # all values equal
        return 0.0
    upp, lop = up[up > 0], lo[lo > 0]
    k0 = int(np.sum(up == 0))       # number of values tied at the median (the same on both sides)
    # kernel values of the pairs that involve a tie, as explicit lists (few):
    #   (u > 0, l = 0): h = +1;  (u = 0, l > 0): h = -1;  (u = 0, l = 0): sign(k0 - 1 - i - j)
    n_plus = upp.size * k0
    n_minus = k0 * lop.size
    ii, jj = np.meshgrid(np.arange(k0), np.arange(k0), indexing="ij")
    tie = np.sign(k0 - 1 - ii - jj).ravel() if k0 else np.zeros(0)
    total = upp.size * lop.size + n_plus + n_minus + tie.size
    dev = None
    if device and upp.size * lop.size >= 1 << 24:
        import ctypes as C
        from . import _lib
        dev = (_lib.DeviceArray.from_numpy(upp), _lib.DeviceArray.from_numpy(np.ascontiguousarray(lop)), _lib.load(), C.c_int64())

    def count_le(t):  # number of pairs with h <= t
        if t >= 1:
            return total
        if t < -1:
            return 0
        # regular pairs: (u - l) / (u + l) <= t  <=>  l >= u (1 - t) / (1 + t)
        c = 0
        if t > -1 and dev is not None:
            import ctypes as C
            from . import _lib
            _lib.check(dev[2].bsn_robust_mc_count(dev[0].ptr, upp.size, dev[1].ptr, lop.size, float(t), C.byref(dev[3])))
            c = int(dev[3].value)
        elif t > -1:
            thr = upp * (1 - t) / (1 + t)
            c = int(np.sum(lop.size - np.searchsorted(lop, thr, side="left")))
        return c + n_minus + int(np.sum(tie <= t))       # the +1 pairs only at t >= 1

    def window(a, b, expect):  # every kernel value in (a, b], -1 < a < b < 1 (regular pairs and the ties among themselves)
        t_in = tie[(tie > a) & (tie <= b)]
        if dev is not None:
            import ctypes as C
            from . import _lib
            cap = max(int(expect) - int(t_in.size), 0)
            buf, cnt = np.empty(max(cap, 1)), C.c_int64()
            _lib.check(dev[2].bsn_robust_mc_window(dev[0].ptr, upp.size, dev[1].ptr, lop.size, float(a), float(b), cap,
                                                   _lib.ptr(buf, _lib.f64p), C.byref(cnt)))
            if cnt.value != cap:                                              # (rounding moved a pair across a bound)
                return np.zeros(0)
            return np.concatenate([buf[:cap], t_in])
        lo_i = np.searchsorted(lop, upp * (1 - b) / (1 + b), side="left")     # h <= b  <=>  l >= this
        hi_i = np.searchsorted(lop, upp * (1 - a) / (1 + a), side="left")     # h >  a  <=>  l <  this
        cnt = np.maximum(hi_i - lo_i, 0)
        tot = int(cnt.sum())
        u_rep = np.repeat(upp, cnt)
        first = np.repeat(lo_i - (np.cumsum(cnt) - cnt), cnt)                 # l index = first + running position
        l = lop[first + np.arange(tot)]
        return np.concatenate([(u_rep - l) / (u_rep + l), t_in])

    def kth(k, pair=False):  # k-th smallest kernel value (1-based); pair: the k-th and the (k + 1)-th
        if pair:
            first = kth(k, pair=None)
            return first if isinstance(first, tuple) else (first, kth(k + 1))
        a, b = -1.0, 1.0
        ca = count_le(-1.0)
        if ca >= k:
            return -1.0
        cb = total
        # bisection on t until few kernel values are left between the bounds; those are then written out and the
        # wanted one picked by its rank (exact, and a fifth of the count evaluations of a bisection down to 1e-15)
        for _ in range(200):
            if -1.0 < a and b < 1.0 and cb - ca <= 400000:
                vals = window(a, b, cb - ca)
                if vals.size == cb - ca:                                      # (always, unless rounding moved a pair across a bound)
                    if pair is None and k + 1 <= cb:                          # both middle values from the one window
                        part = np.partition(vals, [k - ca - 1, k - ca])
                        return float(part[k - ca - 1]), float(part[k - ca])
                    return float(np.partition(vals, k - ca - 1)[k - ca - 1])
            mid = 0.5 * (a + b)
            cm = count_le(mid)
            if cm >= k:
                b, cb = mid, cm
            else:
                a, ca = mid, cm
            if b - a < 1e-15:
                break
        # attained values near b: ties give exactly -1, 0, +1; regular pairs (u - l) / (u + l)
        cand = [v for v in (-1.0, 0.0, 1.0) if abs(v - b) < 1e-9]
        if upp.size and lop.size:
            # for each u the l that brings the kernel closest to b
            l_star = upp * (1 - b) / (1 + b) if b > -1 else np.full(upp.size, np.inf)
            idx = np.clip(np.searchsorted(lop, l_star), 0, lop.size - 1)
            for d in (-1, 0):
                l = lop[np.clip(idx + d, 0, lop.size - 1)]
                cand.extend(((upp - l) / (upp + l)).tolist())
        cand = np.asarray(cand)
        return float(cand[np.argmin(np.abs(cand - b))])

    if total % 2 == 1:
        return kth((total + 1) // 2)
    m1, m2 = kth(total // 2, pair=True)
    return 0.5 * (m1 + m2)


def _quantile_sorted(xs, q):
    """np.quantile(xs, q) (the default method, R's type 7) of an ASCENDING vector, bit for bit: numpy's virtual index
    and its two-sided interpolation restated."""
    n = xs.size
    v = (n - 1) * q
    lo = int(np.floor(v))
    if lo >= n - 1:
        return float(xs[n - 1])
    if lo < 0:
        return float(xs[0])
    g = v - lo
    a, b = float(xs[lo]), float(xs[lo + 1])
    d = b - a
    return b - d * (1.0 - g) if g >= 0.5 else a + d * g


def tukey_mc_up(x, coef=None, alpha=0.05, a=-4.0, b=3.0, device=False):
    """Upper fence Q3 + coef * exp(b MC | -a MC) * IQR.  With coef = NULL the coefficient is
    the one for which a sample of m = length(x) normal values has probability `alpha` of
    containing at least one value above the fence: per-value tail p = 1 - (1 - alpha)^(1/m),
    coef = (qnorm(1 - p) - q75) / (2 q75).  (This family-wise form is what makes
    alpha.tukey -> 1 flag outliers at every iteration, tests/testthat/test-2-autoSVD.R:60-62.)"""
    from scipy.stats import norm
    x = np.asarray(x, dtype=np.float64)
    x = x[~np.isnan(x)]
    if device and x.size >= 1 << 16:
        # one device sort serves the quartiles and the medcouple (np.sort of a million values: 70 ms of host time)
        from . import _lib
        x = np.ascontiguousarray(x)
        if not np.all(np.isfinite(x)):
            raise ValueError("tukey_mc_up: infinite values")
        _lib.check(_lib.load().bsn_robust_sort(_lib.ptr(x, _lib.f64p), x.size))
        q1, q3 = _quantile_sorted(x, 0.25), _quantile_sorted(x, 0.75)
        is_sorted = True
    else:
        q1, q3 = np.quantile(x, [0.25, 0.75])   # R's default quantile type 7 == numpy's default
        is_sorted = False
    iqr = q3 - q1
    if coef is None:
        q75 = norm.ppf(0.75)
        p = -np.expm1(np.log1p(-alpha) / x.size)
        coef = (norm.isf(p) - q75) / (2 * q75)
    mc = medcouple(x, device=device, assume_sorted=is_sorted)
    return q3 + coef * iqr * (np.exp(b * mc) if mc >= 0 else np.exp(-a * mc))


def getIntervals(x, n=2):
    """R/autoSVD.R:4-12: regroup consecutive integers into [start, stop] intervals of
    length >= n"""
    x = np.asarray(x)
    if x.size < 2:
        return np.zeros((0, 2), dtype=x.dtype)
    dx = np.diff(x)
    # rle(diff(x))
    change = np.r_[True, dx[1:] != dx[:-1]]
    starts = np.nonzero(change)[0]
    lengths = np.diff(np.r_[starts, dx.size])
    values = dx[starts]
    ind = np.nonzero((values == 1) & (lengths >= (n - 1)))[0]
    pos = np.cumsum(lengths)            # 0-based index of the last element of each run in x
    first = np.r_[0, pos][ind]
    last = pos[ind]
    return np.column_stack([x[first], x[last]])


# ---- the outer loop ---------------------------------------------------------------------
def _auto_svd(svd_fun, clump_fun, maf_nok, ind_col, infos_chr, infos_pos, thr_r2, k, roll_size,
              int_min_size, alpha_tukey, max_iter, verbose, n_all_cols):
    def printf2(fmt, *a):
        if verbose:
            print(fmt % a, end="")

    if maf_nok is None:
        raise ValueError("You cannot use variants with no variation; set min.mac > 0 and min.maf > 0.")
    ns = int(maf_nok[0].sum())
    printf2("Discarding %d variant%s with MAC < %s or MAF < %s.\n", ns, "s" if ns > 1 else "",
            maf_nok[1], maf_nok[2])
    ind_keep = ind_col[~maf_nok[0]]
    if thr_r2 is None or (isinstance(thr_r2, float) and np.isnan(thr_r2)):
        printf2("\nSkipping clumping.\n")
    else:
        printf2("\nPhase of clumping (on %s) at r^2 > %s.. ", maf_nok[3], thr_r2)
        gone = np.ones(n_all_cols, dtype=bool)      # setdiff(seq_len(ncol), ind.keep) without sorting a million indices
        gone[ind_keep] = False
        excl = np.nonzero(gone)[0]
        ind_keep = clump_fun(excl)
        printf2("keep %d variants.\n", ind_keep.size)

    it = 0
    lrldr = dict(Chr=[], Start=[], Stop=[], Iter=[])
    while True:
        it += 1
        printf2("\nIteration %d:\n", it)
        printf2("Computing SVD..\n")
        obj = svd_fun(ind_keep)
        if it > max_iter:
            printf2("Maximum number of iterations reached.\n")
            break
        S = np.sqrt(dist_ogk(obj["v"], device=obj["v"].shape[1] <= 64))
        chr_keep = infos_chr[ind_keep]
        S2 = rollmean_groups(S, roll_size, chr_groups(chr_keep), device=True)
        thr = tukey_mc_up(S2, alpha=alpha_tukey, device=True)
        excl = np.nonzero(S2 > thr)[0]
        printf2("%d outlier variant%s detected..\n", excl.size, "s" if excl.size > 1 else "")
        if excl.size > 0:
            if infos_pos is not None:
                rng = getIntervals(excl, n=int_min_size)
                printf2("%d long-range LD region%s detected..\n", len(rng), "s" if len(rng) > 1 else "")
                for lo, hi in rng:
                    seq = np.arange(lo, hi + 1)
                    seq_chr = infos_chr[ind_keep[seq]]
                    vals, cnts = np.unique(seq_chr, return_counts=True)
                    mode = vals[np.argmax(cnts)]     # sort(table(.), decreasing = TRUE)[1]
                    in_chr = seq_chr == mode
                    p = infos_pos[ind_keep[seq[in_chr]]]
                    lrldr["Chr"].append(mode); lrldr["Start"].append(p.min())
                    lrldr["Stop"].append(p.max()); lrldr["Iter"].append(it)
            ind_keep = np.delete(ind_keep, excl)
        else:
            printf2("\nConverged!\n")
            break
    order = np.lexsort((lrldr["Stop"], lrldr["Start"], lrldr["Chr"])) if lrldr["Chr"] else []
    obj = dict(obj)
    obj["subset"] = ind_keep
    obj["lrldr"] = {key: np.asarray(val)[order] if len(order) else np.asarray(val)
                    for key, val in lrldr.items()}
    return obj


def snp_autoSVD(G, infos_chr, infos_pos=None, ind_row=None, ind_col=None, fun_scaling=None,
                thr_r2=0.2, size=None, k=10, roll_size=50, int_min_size=20, alpha_tukey=0.05,
                min_mac=10, min_maf=0.02, max_iter=5, ncores=1, verbose=True):
    """R/autoSVD.R:67-186.  Returns the last big_SVD dict with keys `subset` (kept column
    indices, 0-based) and `lrldr`."""
    im, ir, ic = _ind(G, ind_row, ind_col)
    infos_chr = np.asarray(infos_chr)
    if infos_chr.size != im.ncol:
        raise ValueError(ERROR_DIM)
    if infos_pos is not None:
        infos_pos = np.asarray(infos_pos)
        if infos_pos.size != im.ncol:
            raise ValueError(ERROR_DIM)
    # fun_scaling = None stands for the default snp_scaleBinom(): big_randomSVD then evaluates it inside
    # the solve (no separate statistics pass per outlier-removal round)
    size = (100.0 / thr_r2 if thr_r2 is not None and not np.isnan(thr_r2) else 500.0) if size is None else size
    maf_nok = None
    if min_mac > 0 and min_maf > 0:
        maf = snp_MAF(G, ir, ic)
        maf_nok = (maf < max(min_maf, min_mac / (2.0 * ir.size)), min_mac, min_maf, "MAF")
    return _auto_svd(
        lambda keep: big_randomSVD(G, fun_scaling, ind_row=ir, ind_col=keep, k=k),
        lambda excl: snp_clumping(G, infos_chr, ind_row=ir, exclude=excl, thr_r2=thr_r2, size=size,
                                  infos_pos=infos_pos),
        maf_nok, ic, infos_chr, infos_pos, thr_r2, k, roll_size, int_min_size, alpha_tukey,
        max_iter, verbose, im.ncol)


def bed_autoSVD(obj_bed, ind_row=None, ind_col=None, fun_scaling=bed_scaleBinom, thr_r2=0.2,
                size=None, k=10, roll_size=50, int_min_size=20, alpha_tukey=0.05, min_mac=10,
                min_maf=0.02, max_iter=5, ncores=1, verbose=True):
    """R/autoSVD.R:226-339"""
    im, ir, ic = _ind(obj_bed, ind_row, ind_col)
    infos_chr = np.asarray(obj_bed.map["chromosome"])
    infos_pos = np.asarray(obj_bed.map["physical_pos"])
    size = (100.0 / thr_r2 if thr_r2 is not None and not np.isnan(thr_r2) else 500.0) if size is None else size
    maf_nok = None
    if min_mac > 0 and min_maf > 0:
        info = bed_MAF(obj_bed, ir, ic)
        maf_nok = ((info["mac"] < min_mac) | (info["maf"] < min_maf), min_mac, min_maf, "MAC")
    return _auto_svd(
        lambda keep: bed_randomSVD(obj_bed, fun_scaling=fun_scaling, ind_row=ir, ind_col=keep, k=k),
        lambda excl: bed_clumping(obj_bed, ind_row=ir, exclude=excl, thr_r2=thr_r2, size=size),
        maf_nok, ic, infos_chr, infos_pos, thr_r2, k, roll_size, int_min_size, alpha_tukey,
        max_iter, verbose, im.ncol)
