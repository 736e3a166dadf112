"""get_predictor (reference: saber/classifier/models/common.py:22-47)."""
import os


def get_predictor(model_weights, model_config, deviceID: int = 0):
    from saber_amd.classifier.models.predictor import Predictor
    if model_weights is None or model_config is None:
        return None
    if not os.path.exists(model_weights):
        raise FileNotFoundError(f"Model weights file {model_weights} does not exist.")
    if not os.path.exists(model_config):
        raise FileNotFoundError(f"Model config file {model_config} does not exist.")
    return Predictor(model_config, model_weights, deviceID=deviceID)
