"""g2vlm_amd — MI355X-native engine for the G2VLM inference hot path (recon + chat decode).

HIP kernels live in csrc/ behind the C ABI of include/g2vlm_hip.h (lib/libg2vlm_hip.so);
`hip.py` binds them, `engine.py` sequences them, `modeling/` and `g2vlm_utils.py` mirror the
reference's Python surface so its scripts run against this package unchanged.
"""
__version__ = "0.1.0"
