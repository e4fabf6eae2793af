################################################################################
# bigsnprhip: R-side glue between bigsnpr and libbigsnpr_hip (original code; nothing here is taken from bigsnpr's R files).
#
# bigsnpr reaches native code through objects such as `_bigsnpr_bed_pMatVec4` that useDynLib(bigsnpr, .registration = TRUE)
# puts into ITS namespace (NAMESPACE:124 there); every R function of the hot path — bed_prodVec, bed_cprodVec, bed_counts,
# bed_scaleBinom, snp_cor, snp_ld_scores, snp_clumping, ... — ends in .Call(<such an object>, ...).  This package registers
# routines of the SAME names and arities (src/bigsnpr_hip_shim.c); hip_enable() swaps the objects, so the R functions of
# bigsnpr run unchanged on the GPU, and hip_disable() swaps them back.
################################################################################

# the reference's .Call targets on the hot path (the table of src/bigsnpr_hip_shim.c without its _hip additions)
hip_symbols <- function() {
  c("_bigsnpr_bedXPtr", "_bigsnpr_bed_colstats", "_bigsnpr_bed_col_counts_cpp", "_bigsnpr_bed_row_counts_cpp",
    "_bigsnpr_read_bed", "_bigsnpr_read_bed_scaled", "_bigsnpr_bed_pMatVec4", "_bigsnpr_bed_cpMatVec4",
    "_bigsnpr_prod_and_rowSumsSq", "_bigsnpr_prod_and_rowSumsSq2", "_bigsnpr_snp_colstats", "_bigsnpr_multLinReg",
    "_bigsnpr_readbina", "_bigsnpr_readbina2", "_bigsnpr_writebina", "_bigsnpr_corMat", "_bigsnpr_ld_scores",
    "_bigsnpr_clumping_chr", "_bigsnpr_bed_clumping_chr", "_bigsnpr_clumping_chr_cached")
}

.state <- new.env(parent = emptyenv())
.state$saved <- list()     # what bigsnpr's namespace held before hip_enable()
.state$on <- FALSE

.swap <- function(ns, name, value) {
  locked <- bindingIsLocked(name, ns)
  if (locked) unlockBinding(name, ns)
  assign(name, value, envir = ns)
  if (locked) lockBinding(name, ns)
}

# The whole solve on the device (replaces the big_randomSVD(..., bed_prodVec, bed_cprodVec) call inside bigsnpr's
# bed_randomSVD).  fun.scaling = bed_scaleBinom is evaluated INSIDE the solve — its code counts ride along the first
# crossproduct pass — which center = scale = NULL selects; the values used come back as $center / $scale either way.
# `tol` may carry up to three more numbers: c(tol, slices, block, vec.floor), see the shim.
bed_randomSVD <- function(obj.bed, fun.scaling = bigsnpr::bed_scaleBinom, ind.row = rows_along(obj.bed),
                          ind.col = cols_along(obj.bed), k = 10, tol = 1e-4, verbose = FALSE, ncores = 1) {
  ms <- if (identical(fun.scaling, bigsnpr::bed_scaleBinom)) NULL else
    fun.scaling(obj.bed, ind.row = ind.row, ind.col = ind.col, ncores = ncores)
  structure(.Call(`_bigsnpr_bed_randomSVD_hip`, obj.bed$light, ind.row, ind.col,
                  ms$center, ms$scale, k, tol, verbose), class = "big_SVD")
}

# the same for an FBM.code256 (what snp_autoSVD hands to bigstatsr::big_randomSVD); fun.scaling is called like there
big_randomSVD_hip <- function(X, fun.scaling, ind.row = rows_along(X), ind.col = cols_along(X), k = 10, tol = 1e-4,
                              verbose = FALSE, ncores = 1) {
  ms <- fun.scaling(X, ind.row = ind.row, ind.col = ind.col, ncores = ncores)
  structure(.Call(`_bigsnpr_big_randomSVD_hip`, X, ind.row, ind.col, ms$center, ms$scale, k, tol, verbose),
            class = "big_SVD")
}

# FBM.code256 mat-vec operators with the argument list of the fun.prod / fun.cprod closures of bigstatsr::big_randomSVD
# (and of big_prodVec / big_cprodVec: callers in bigsnpr are snp_PRS and snp_autoSVD)
big_prodVec_hip <- function(X, y.col, ind.row = rows_along(X), ind.col = cols_along(X), center = NULL, scale = NULL) {
  .Call(`_bigsnpr_big_prodVec_hip`, X, y.col, ind.row, ind.col, center, scale)
}
big_cprodVec_hip <- function(X, y.row, ind.row = rows_along(X), ind.col = cols_along(X), center = NULL, scale = NULL) {
  .Call(`_bigsnpr_big_cprodVec_hip`, X, y.row, ind.row, ind.col, center, scale)
}

# device images of FBMs are cached by backing file and decode table: after WRITING to an FBM, drop them
hip_fbm_cache_clear <- function() invisible(.Call(`_bigsnpr_fbm_cache_clear_hip`))

hip_enabled <- function() .state$on

hip_enable <- function(svd = TRUE) {
  ns <- asNamespace("bigsnpr")
  mine <- asNamespace("bigsnprhip")
  for (s in hip_symbols()) {
    if (!exists(s, envir = ns, inherits = FALSE)) next          # an older bigsnpr without this routine
    if (is.null(.state$saved[[s]])) .state$saved[[s]] <- get(s, envir = ns, inherits = FALSE)
    .swap(ns, s, get(s, envir = mine, inherits = FALSE))
  }
  if (svd) {
    if (is.null(.state$saved[["bed_randomSVD"]])) .state$saved[["bed_randomSVD"]] <- get("bed_randomSVD", envir = ns)
    .swap(ns, "bed_randomSVD", bed_randomSVD)                    # callers inside bigsnpr (bed_autoSVD) see it too
  }
  .state$on <- TRUE
  invisible(TRUE)
}

hip_disable <- function() {
  ns <- asNamespace("bigsnpr")
  for (s in names(.state$saved)) .swap(ns, s, .state$saved[[s]])
  .state$saved <- list()
  .state$on <- FALSE
  invisible(TRUE)
}

.onUnload <- function(libpath) {
  if (.state$on) hip_disable()
  library.dynam.unload("bigsnprhip", libpath)
}
