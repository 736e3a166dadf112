"""Device selection with the semantics of the reference's saber/utils/io.py:93-149 (get_available_devices /
determine_device), restricted to what the hot path needs.  There is no CPU execution path for the engine:
selecting a CPU device is reported here and fails when an Engine is created."""
import torch


def determine_device(deviceID: int = 0):
    if torch.cuda.is_available():
        n = torch.cuda.device_count()
        if deviceID >= n:
            print(f"Warning: Requested device {deviceID} but only {n} devices available")
            print("Falling back to device 0")
            deviceID = 0
        return torch.device(f"cuda:{deviceID}")
    print("No ROCm device available (the MI355X engine cannot run on CPU)")
    return torch.device("cpu")


def get_available_devices(deviceID: int = None):
    if deviceID is None:
        return torch.device("cuda" if torch.cuda.is_available() else "cpu")
    return determine_device(deviceID)
