# Optional helper added to R/ (additive; no existing signature changes).  The shim reads these options when a
# routine starts:  snpgpu.device  HIP device ordinal (default 0),
#                  snpgpu.block.snps  SNPs per block handed to the device (default 32768 GRM / PCA, 65536 IBS / KING),
#                  snpgpu.devices  several ordinals: ONE R process drives all of them (snpgpu_multi: row panels of the
#                                  output triangle per GPU, blocks forwarded over xGMI, results gathered),
#                  snpgpu.panels.per.device (default -1: the fewest that fit the devices' free memory), snpgpu.passes (KING-robust: walks over the SNPs, default 1).
snpgdsGPUOptions <- function(device=NULL, block.snps=NULL, devices=NULL, panels.per.device=NULL, passes=NULL)
{
    if (!is.null(devices))
    {
        stopifnot(is.numeric(devices), length(devices) >= 1L, all(devices >= 0))
        options(snpgpu.devices=as.integer(devices))
    }
    if (!is.null(panels.per.device))
        options(snpgpu.panels.per.device=as.integer(panels.per.device))
    if (!is.null(passes))
        options(snpgpu.passes=as.integer(passes))
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
