"""
ml_super_resolution_amd -- MI355X (gfx950) native engine for the data-parallel hot path of
imironhead/ml_super_resolution: the stacked conv+bias+activation forward / backward of SRCNN,
ESPCN, VDSR (and the EnhanceNet generator), the ESPCN sub-pixel map, loss and optimizer step.

The compute lives in libsrx.so (hand-written HIP, C ABI in include/srx.h).  This package is the
host side: a ctypes binding (`_lib`), thin op wrappers (`ops`), the layer-stack engine
(`engine`), and mirrors of the reference's model-build / experiment entry points
(`vdsr`, `espcn`, `srcnn`, `enet`).  PyTorch is used only for device memory, streams and
torch.distributed (RCCL).
"""
__version__ = '0.1.0'
