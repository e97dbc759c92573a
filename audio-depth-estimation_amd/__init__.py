"""MI355X-native hot path of Kang-ChangWoo/audio-depth-estimation.

Python mirror of the reference's interface for the depth-regression path
(models.*, utils_loss, utils_criterion, config_loader, dataloader front-end, train/test
entry points) on top of libadn.so, a C-ABI library of hand-written gfx950 HIP kernels.
Importable as ``audio_depth_estimation_amd`` (the directory name carries a hyphen).
"""
__version__ = '0.1.0'
