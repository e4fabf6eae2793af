"""snp_readBed / snp_writeBed — host mirror of R/read-plink.R:27-111 and R/write-plink.R:15-44
(the data formats either side of the path).  The decode (2 bits -> byte) and
the packing (byte -> 2 bits, with row / column subsets) run on the GPU; only the small text
files are handled on the host.  No .bk/.rds persistence: a bigSNP here is an in-memory dict.
"""
import os

import numpy as np

from . import _lib
from ._lib import check, i64p, ptr, u8p
from .bed import _check_ind, bed, cols_along, rows_along
from .ld import FBM_code256, _image

NAMES_FAM = ["family.ID", "sample.ID", "paternal.ID", "maternal.ID", "sex", "affection"]   # R/utils.R:51-53
NAMES_MAP = ["chromosome", "marker.ID", "genetic.dist", "physical.pos", "allele1", "allele2"]


def _read_table(path, ncol):
    rows = [line.split() for line in open(path)]
    return [[r[c] for r in rows] for c in range(ncol)]


def bed_to_bytes(obj_bed, ind_row=None, ind_col=None):
    """readbina2 (src/read-plink.cpp:61-80): n x m uint8, values 0/1/2 and 3 for missing"""
    ir = rows_along(obj_bed) if ind_row is None else _check_ind("ind.row", ind_row, obj_bed.nrow)
    ic = cols_along(obj_bed) if ind_col is None else _check_ind("ind.col", ind_col, obj_bed.ncol)
    out = np.empty((ic.size, ir.size), dtype=np.uint8)
    check(_lib.load().bsn_bed_to_fbm(obj_bed.handle, ptr(ir, i64p), ir.size, ptr(ic, i64p), ic.size,
                                     ptr(out, u8p)))
    return out.T


def get_code(na_val=3):
    """getCode() (R/utils.R:21-31): the 4 x 256 raw table readbina decodes with."""
    t4 = np.array([2, na_val, 1, 0], dtype=np.uint8)          # .bed bit pairs 00, 01, 10, 11
    return np.ascontiguousarray(t4[(np.arange(256)[None, :] >> (2 * np.arange(4)[:, None])) & 3])


def readbina(bedfile, tab=None, n=None, m=None):
    """readbina (src/read-plink.cpp:13-56) as snp_readBed calls it (R/read-plink.R:42-55: dimensions from the
    .fam / .bim files unless given): (n x m uint8 FBM bytes of the whole file decoded through the 4 x 256 table `tab`
    on the GPU, reached-EOF flag).  Like the reference it does not go through the `bed` class: a file longer than
    n x m genotypes is not an error, the flag says so."""
    from .bed import _count_lines
    tab = get_code() if tab is None else np.ascontiguousarray(np.asarray(tab, dtype=np.uint8).reshape(4, 256))
    bedfile = os.path.expanduser(str(bedfile))
    n = _count_lines(bedfile[:-4] + ".fam") if n is None else int(n)
    m = _count_lines(bedfile[:-4] + ".bim") if m is None else int(m)
    raw = np.fromfile(bedfile, dtype=np.uint8)
    if raw.size < 3 or not (raw[0] == 108 and raw[1] == 27):
        raise ValueError("Wrong magic number. Aborting..")
    nb = (n + 3) // 4
    if raw.size < 3 + m * nb:
        raise ValueError("readbina: '%s' holds fewer than %d x %d genotypes" % (bedfile, n, m))
    b = bed.from_payload(raw[3:3 + m * nb], n, m)
    out = np.empty((m, n), dtype=np.uint8)
    # column `byte` of the R matrix is contiguous: tab[4 * byte + e]
    check(_lib.load().bsn_bed_readbina(b.handle, ptr(np.ascontiguousarray(tab.T), u8p), ptr(out, u8p)))
    return out.T, bool(raw.size <= 3 + m * nb)


def snp_readBed(bedfile, ind_row=None, ind_col=None):
    """R/read-plink.R:27-111 (snp_readBed / snp_readBed2): returns a bigSNP-like dict
    {genotypes: FBM_code256, fam, map}."""
    b = bed(bedfile)
    ir = rows_along(b) if ind_row is None else _check_ind("ind.row", ind_row, b.nrow)
    ic = cols_along(b) if ind_col is None else _check_ind("ind.col", ind_col, b.ncol)
    g = bed_to_bytes(b, ir, ic)
    fam = _read_table(b.famfile, 6)
    bim = _read_table(b.bimfile, 6)
    return dict(genotypes=FBM_code256(g), bytes=g,
                fam={k: [v[i] for i in ir] for k, v in zip(NAMES_FAM, fam)},
                map={k: [v[j] for j in ic] for k, v in zip(NAMES_MAP, bim)})


def snp_writeBed(x, bedfile, ind_row=None, ind_col=None):
    """R/write-plink.R:15-44: writes .bed (+ .bim, .fam) of x[ind.row, ind.col]; `x` is a dict
    with genotypes (FBM_code256 or bed), fam and map.  Refuses to overwrite, like the reference."""
    bedfile = os.path.expanduser(str(bedfile))
    if not bedfile.endswith(".bed"):
        raise ValueError("Path '%s' must have 'bed' extension." % bedfile)
    bim, fam = bedfile[:-4] + ".bim", bedfile[:-4] + ".fam"
    for f in (bedfile, bim, fam):
        if os.path.exists(f):
            raise FileExistsError("File '%s' already exists." % f)
    im = _image(x["genotypes"])
    ir = rows_along(im) if ind_row is None else _check_ind("ind.row", ind_row, im.nrow)
    ic = cols_along(im) if ind_col is None else _check_ind("ind.col", ind_col, im.ncol)
    payload = np.empty(((ir.size + 3) // 4) * ic.size, dtype=np.uint8)
    check(_lib.load().bsn_bed_subset_payload(im.handle, ptr(ir, i64p), ir.size, ptr(ic, i64p),
                                             ic.size, ptr(payload, u8p)))
    with open(bedfile, "wb") as f:
        f.write(bytes([0x6C, 0x1B, 0x01]))     # src/write-plink.cpp:30-31
        f.write(payload.tobytes())
    with open(fam, "w") as f:
        for i in ir:
            f.write("\t".join(str(x["fam"][k][i]) for k in NAMES_FAM) + "\n")
    with open(bim, "w") as f:
        for j in ic:
            f.write("\t".join(str(x["map"][k][j]) for k in NAMES_MAP) + "\n")
    return bedfile
