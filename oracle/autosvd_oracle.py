"""TEST INFRASTRUCTURE — CPU restatement of the outer loop of snp_autoSVD / bed_autoSVD, written from the
reference's R code line by line and INDEPENDENTLY of bigsnpr_amd/autosvd.py (it shares no code with the
product; the GPU test runs the product's loop on GPU pieces against this loop on the oracle's pieces).

    auto_svd_loop   R/autoSVD.R:95-186   (the FBM method; the bed method, :249-339, is the same loop)
    get_intervals   R/autoSVD.R:4-12

The three bigutilsr functions the loop calls are NOT in the reference tree (package bigutilsr >= 0.3.3, external);
they are restated here a second time, from their published definitions and with deliberately plain algorithms
(all-pairs medcouple, direct-sum rolling mean, loop-by-loop OGK), so that the product's faster implementations
have an independent implementation to agree with.  Against bigutilsr itself both remain "parity unpinned".
Only tests/ may import this module.
"""
import math

import numpy as np
from scipy.stats import chi2, norm


# ---- R/autoSVD.R:4-12 -----------------------------------------------------------------------------------
def get_intervals(x, n=2):
    """regroup consecutive integers in intervals [start, stop] of at least n members (0-based values in,
    0-based values out; the R code works on 1-based positions, the arithmetic is on differences only)"""
    x = [int(v) for v in x]
    out = []
    if len(x) < 2:
        return out
    d = [x[i + 1] - x[i] for i in range(len(x) - 1)]
    # rle(diff(x))
    values, lengths = [], []
    for v in d:
        if values and values[-1] == v:
            lengths[-1] += 1
        else:
            values.append(v)
            lengths.append(1)
    pos = np.cumsum(lengths)                 # R: cumsum(le$lengths) + 1 (1-based) == 0-based index of the run's last x
    starts = [0] + list(pos)                 # R: c(1, pos)
    for r, (v, le) in enumerate(zip(values, lengths)):
        if v == 1 and le >= n - 1:
            out.append((x[starts[r]], x[pos[r]]))
    return out


# ---- bigutilsr::rollmean (external): Gaussian-weighted rolling mean, edges renormalised ----------------------
def rollmean(x, size):
    x = [float(v) for v in x]
    if size == 0:
        return np.asarray(x)
    half = int(math.floor(size))
    length = 2 * half + 1
    if length >= len(x):
        raise ValueError("Parameter 'size' is too large.")
    a = 3.0 / 8.0 if length <= 10 else 0.5                     # stats::ppoints(length)
    p_first, p_last = (1 - a) / (length + 1 - 2 * a), (length - a) / (length + 1 - 2 * a)
    lo, hi = norm.ppf(p_first), norm.ppf(p_last)
    w = [norm.pdf(lo + (hi - lo) * t / (length - 1)) for t in range(length)]
    out = []
    for i in range(len(x)):
        num = den = 0.0
        for t in range(length):
            j = i - half + t
            if 0 <= j < len(x):
                num += w[t] * x[j]
                den += w[t]
        out.append(num / den)
    return np.asarray(out)


# ---- medcouple (Brys, Hubert & Struyf 2004), all pairs --------------------------------------------------
def medcouple(x):
    x = sorted(float(v) for v in x)
    n = len(x)
    if n < 3:
        return 0.0
    med = float(np.median(x))
    plus = [v for v in x if v >= med]            # x_i >= med
    minus = [v for v in x if v <= med]           # x_j <= med
    ties_p = [i for i, v in enumerate(plus) if v == med]
    ties_m = [j for j, v in enumerate(minus) if v == med]
    k = len(ties_p)
    h = []
    for i, xi in enumerate(plus):
        for j, xj in enumerate(minus):
            if xi == med and xj == med:
                # both at the median: the sign of their order among the k tied values
                # (i runs over the ties from the median upwards, j from the median downwards)
                ii = ties_p.index(i)
                jj = len(ties_m) - 1 - ties_m.index(j)
                s = k - 1 - ii - jj
                h.append(0.0 if s == 0 else (1.0 if s > 0 else -1.0))
            else:
                h.append(((xi - med) - (med - xj)) / (xi - xj))
    return float(np.median(h))


# ---- bigutilsr::tukey_mc_up (external) -----------------------------------------------------------------
def tukey_mc_up(x, alpha=0.05):
    x = np.asarray([v for v in x if not math.isnan(v)], dtype=np.float64)
    q1, q3 = np.quantile(x, 0.25), np.quantile(x, 0.75)
    q75 = norm.ppf(0.75)
    p = 1.0 - (1.0 - alpha) ** (1.0 / x.size)                   # family-wise alpha over length(x) values
    coef = (norm.ppf(1.0 - p) - q75) / (2.0 * q75)
    mc = medcouple(x)
    return q3 + coef * (q3 - q1) * (math.exp(3.0 * mc) if mc >= 0 else math.exp(4.0 * mc))


# ---- bigutilsr::dist_ogk (external): robust Mahalanobis distance, OGK of Maronna & Zamar (2002) --------
def _tau_scale(x, c1=4.5, c2=3.0):
    x = np.asarray(x, dtype=np.float64)
    med = np.median(x)
    mad = np.median(np.abs(x - med))
    if mad <= 0:
        return med, 0.0
    w = np.array([max(0.0, 1.0 - ((v - med) / (c1 * mad)) ** 2) ** 2 for v in x])
    mu = float(np.sum(w * x) / np.sum(w))
    rho = np.array([min(((v - mu) / mad) ** 2, c2 * c2) for v in x])
    b = c2 * norm.ppf(0.75)
    e_rho = 2.0 * ((1.0 - b * b) * norm.cdf(b) - b * norm.pdf(b) + b * b) - 1.0   # consistency at the normal
    return mu, float(mad * math.sqrt(np.sum(rho) / (x.size * e_rho)))


def dist_ogk(U, niter=2, beta=0.9):
    U = np.asarray(U, dtype=np.float64)
    n, p = U.shape
    Z = U.copy()
    for _ in range(niter):
        s = np.array([_tau_scale(Z[:, j])[1] for j in range(p)])
        s[s <= 0] = 1.0
        Z = Z / s
        R = np.eye(p)
        for i in range(p):
            for j in range(i + 1, p):
                sp = _tau_scale(Z[:, i] + Z[:, j])[1]
                sm = _tau_scale(Z[:, i] - Z[:, j])[1]
                R[i, j] = R[j, i] = 0.25 * (sp * sp - sm * sm)
        lam, E = np.linalg.eigh(R)
        E = E[:, np.argsort(-lam)]
        Z = Z @ E
    loc = np.empty(p)
    sc = np.empty(p)
    for j in range(p):
        loc[j], sc[j] = _tau_scale(Z[:, j])
    sc[sc <= 0] = 1.0
    d = np.sum(((Z - loc) / sc) ** 2, axis=1)
    cut = np.median(d) * chi2.ppf(beta, p) / chi2.ppf(0.5, p)
    good = d <= cut                                                # hard rejection
    center = U[good].mean(axis=0)
    cov = np.atleast_2d(np.cov(U[good], rowvar=False))
    Xc = U - center
    return np.sum((Xc @ np.linalg.pinv(cov)) * Xc, axis=1)


# ---- R/autoSVD.R:95-186 ------------------------------------------------------------------------------------
def auto_svd_loop(svd_fun, clump_fun, maf, n_rows, ind_col, infos_chr, infos_pos=None, thr_r2=0.2, k=10,
                  roll_size=50, int_min_size=20, alpha_tukey=0.05, min_mac=10, min_maf=0.02, max_iter=5,
                  n_all_cols=None):
    """svd_fun(ind_keep) -> dict with "v" (and whatever else the caller wants back); clump_fun(exclude) -> kept
    column indices; maf: MAF of every column of ind_col over the n_rows selected samples (snp_MAF).  Indices
    are 0-based.  Returns (last svd, subset, lrldr rows sorted by (Chr, Start, Stop))."""
    if not (min_mac > 0 and min_maf > 0):                                             # :104-112
        raise ValueError("You cannot use variants with no variation; set min.mac > 0 and min.maf > 0.")
    maf_nok = np.asarray(maf) < max(min_maf, min_mac / (2.0 * n_rows))
    ind_keep = np.asarray(ind_col)[~maf_nok]
    if not (thr_r2 is None or (isinstance(thr_r2, float) and math.isnan(thr_r2))):    # :115-127
        everything = np.arange(n_all_cols if n_all_cols is not None else len(infos_chr))
        ind_keep = np.asarray(clump_fun(np.setdiff1d(everything, ind_keep)))
    it = 0                                                                            # :129-130
    lrldr = []
    while True:                                                                       # :131 repeat
        it += 1
        obj = svd_fun(ind_keep)                                                       # :136-141
        if it > max_iter:                                                             # :143-146
            break
        s_col = np.sqrt(dist_ogk(obj["v"]))                                           # :149
        chr_keep = np.asarray(infos_chr)[ind_keep]
        s2 = np.full(s_col.size, np.nan)                                              # :151-154
        for c in sorted(set(chr_keep.tolist())):
            idx = np.nonzero(chr_keep == c)[0]
            s2[idx] = rollmean(s_col[idx], roll_size)
        thr = tukey_mc_up(s2, alpha=alpha_tukey)                                      # :155
        excl = np.nonzero(s2 > thr)[0]                                                # :156
        if excl.size == 0:                                                            # :178-181
            break
        if infos_pos is not None:                                                     # :164-176
            for lo, hi in get_intervals(excl, n=int_min_size):
                seq = np.arange(lo, hi + 1)
                seq_chr = chr_keep[seq]
                # names(sort(table(seq.range.chr), decreasing = TRUE)[1]): the most frequent chromosome, the
                # smallest label first among equally frequent ones (table() sorts its names, sort() is stable)
                labels = sorted(set(seq_chr.tolist()))
                counts = [int(np.sum(seq_chr == lab)) for lab in labels]
                mode = labels[int(np.argmax(counts))]
                in_chr = seq_chr == mode
                pos_in = np.asarray(infos_pos)[ind_keep[seq[in_chr]]]
                lrldr.append((mode, pos_in.min(), pos_in.max(), it))
        ind_keep = np.delete(ind_keep, excl)                                          # :178
    lrldr.sort(key=lambda r: (r[0], r[1], r[2]))                                      # :186 order(Chr, Start, Stop)
    return obj, ind_keep, lrldr
