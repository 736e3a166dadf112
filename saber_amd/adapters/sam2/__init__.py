from saber_amd.adapters.sam2.predictor import SAM2Adapter

__all__ = ["SAM2Adapter"]
