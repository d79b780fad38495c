from .dinov3_model import DINOv3ViTConfig, DINOv3ViTModel  # noqa: F401
