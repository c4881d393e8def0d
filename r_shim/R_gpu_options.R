# Optional helper added to R/ (additive; no existing signature changes).  The shim reads these options when a
# routine starts:  snpgpu.device  HIP device ordinal (default 0),
#                  snpgpu.block.snps  SNPs per block handed to the device (default 16384 GRM / PCA, 65536 IBS / KING).
snpgdsGPUOptions <- function(device=NULL, block.snps=NULL)
{
    if (!is.null(device))
    {
        stopifnot(is.numeric(device), length(device)==1L, device >= 0)
        options(snpgpu.device=as.integer(device))
    }
    if (!is.null(block.snps))
    {
        stopifnot(is.numeric(block.snps), length(block.snps)==1L, block.snps >= 64)
        options(snpgpu.block.snps=as.integer(block.snps))
    }
    invisible(list(device=getOption("snpgpu.device", 0L), block.snps=getOption("snpgpu.block.snps")))
}
