"""time_sync / select_device of reference skyeye/utils/torch_utils.py:70-118 (the timing convention of the CLIs)."""
import time

import torch


def time_sync():
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    return time.time()


def select_device(device=""):
    if str(device).lower() == "cpu" or not torch.cuda.is_available():
        raise RuntimeError("the SkyEye HIP engine needs an MI355X: there is no CPU path (reference select_device, torch_utils.py:70-106)")
    idx = 0 if device in ("", "cuda") else int(str(device).replace("cuda:", "").split(",")[0])
    return torch.device("cuda", idx)
