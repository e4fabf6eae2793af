"""bigsnpr_amd — MI355X (gfx950) implementation of bigsnpr's genotype-matrix hot path.

The product is libbigsnpr_hip.so (hand-written HIP kernels behind the C ABI of
include/bigsnpr_hip.h); this package is the host-side mirror of the reference's R API
used by the tests and the benchmark.  No CPU fallback exists anywhere in this package.
"""
from ._lib import BsnError, DeviceArray, load  # noqa: F401
from .bed import (ERROR_DIM, ScaledOp, bed, bed_colstats, bed_counts, bed_cprodVec,  # noqa: F401
                  bed_MAF, bed_prodVec, bed_scaleBinom, cols_along, read_bed,
                  read_bed_scaled, rows_along, sub_bed)
from .svd import bed_randomSVD  # noqa: F401,E402
from .comm import Comm, negotiate as negotiate_exchange, set_exchange_mode  # noqa: F401,E402
from .ld import (CODE_012, CODE_DOSAGE, CODE_IMPUTE_PRED, FBM_code256, bed_clumping, bed_cor,  # noqa: F401,E402
                 bed_ld_scores, big_cprodVec, big_prodVec, big_randomSVD, snp_clumping, snp_fake, snp_colstats, snp_cor, snp_ld_scores, snp_MAF, snp_scaleAlpha,
                 snp_scaleBinom)
from .prs import (bed_projectSelfPCA, bed_tcrossprodSelf, prod_and_rowSumsSq,  # noqa: F401,E402
                  prod_and_rowSumsSq2, prodVecRev, snp_PRS, snp_projectSelfPCA)
from .autosvd import bed_autoSVD, snp_autoSVD  # noqa: F401,E402
from .plink_io import bed_to_bytes, snp_readBed, snp_writeBed  # noqa: F401,E402
from .pcadapt import bed_pcadapt, multLinReg, snp_pcadapt  # noqa: F401,E402
from .sct import seq_log, snp_grid_clumping, snp_grid_PRS  # noqa: F401,E402


def selftest():
    from ._lib import check
    check(load().bsn_selftest())


def device_count():
    import ctypes
    n = ctypes.c_int(0)
    load().bsn_device_count(ctypes.byref(n))
    return n.value
