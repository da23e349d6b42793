"""imagetranslate_amd -- MI355X-native (gfx950) implementation of ImageTranslate's transformer
encoder-decoder train-step hot path behind the reference's Seq2Seq / MassSeq2Seq / ImageMassSeq2Seq /
ImageCaptioning / SmoothedNLLLoss API.  Python host code calls hand-written HIP kernels through the C ABI in
``include/imt_hip.h`` (``libimt_hip.so``); PyTorch-ROCm is used for device memory, streams and
``torch.distributed`` (RCCL) only.
"""
__version__ = "0.1.0"
