"""audioset-convnext-inf_amd -- MI355X-native (gfx950) audio-tagging inference path.

Drop-in for the hot path of topel/audioset-convnext-inf: the `ConvNeXt.forward /
forward_scene_embeddings / forward_frame_embeddings` surface
(reference src/audioset_convnext_inf/pytorch/convnext.py:287-402), computed by hand-written
HIP kernels behind a C ABI (`include/acx.h`, `csrc/`).  The directory name carries a hyphen
(repo convention); import it as `audioset_convnext_inf_amd` (alias package at the repo root).
"""
__version__ = "0.1.0"
