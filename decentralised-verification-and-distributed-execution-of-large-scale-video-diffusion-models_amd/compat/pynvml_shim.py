"""`pynvml` as the reference uses it (`fsdp_chunked_coherent.py:41-45,262`): device-wide used memory of GPU i.
Backed by `torch.cuda.mem_get_info` (HIP's hipMemGetInfo: total - free of the whole device, which is what
`nvmlDeviceGetMemoryInfo(h).used` reports — not just this process's allocations)."""
from types import SimpleNamespace

import torch


class NVMLError(RuntimeError):
    pass


def nvmlInit():
    if not torch.cuda.is_available():
        raise NVMLError("no GPU visible")


def nvmlShutdown():
    pass


def nvmlDeviceGetCount():
    return torch.cuda.device_count()


def nvmlDeviceGetHandleByIndex(i: int):
    if i < 0 or i >= torch.cuda.device_count():
        raise NVMLError(f"invalid device index {i}")
    return int(i)


def nvmlDeviceGetMemoryInfo(handle):
    free, total = torch.cuda.mem_get_info(int(handle))
    return SimpleNamespace(total=total, free=free, used=total - free)


def nvmlDeviceGetName(handle):
    return torch.cuda.get_device_name(int(handle))
