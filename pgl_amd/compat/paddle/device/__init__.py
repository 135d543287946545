"""paddle.device: the accelerator is the MI355X torch sees as "cuda"."""
import torch as _t


def is_compiled_with_cuda():
    return _t.cuda.is_available()


def set_device(device):
    if isinstance(device, str) and ":" in device and _t.cuda.is_available():
        _t.cuda.set_device(int(device.split(":")[1]))
    return device


def get_device():
    return "gpu:%d" % _t.cuda.current_device() if _t.cuda.is_available() else "cpu"
