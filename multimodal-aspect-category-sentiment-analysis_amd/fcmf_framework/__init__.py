"""fcmf_framework -- MI355X (gfx950) implementation of the FCMF training hot path.

Drop-in for the reference package of the same name: `fcmf_multimodal.FCMF`,
`fcmf_pretraining.{FCMFEncoder,FCMFSeq2Seq}`, `mm_modeling`, `roi_modeling`, `resnet_utils`,
`optimization` keep their import paths, signatures and state-dict keys; every operator runs in
libfcmf_hip.so (include/fcmf_hip.h).  There is no CPU or eager-PyTorch fallback.
"""
from .ops import compute_dtype, manual_seed, set_compute_dtype  # noqa: F401
