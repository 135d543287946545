import torch


class _Opt:
    def step(self):
        self._o.step()

    def clear_grad(self):
        self._o.zero_grad(set_to_none=True)

    clear_gradients = clear_grad

    def minimize(self, loss):
        self._o.step()

    def get_lr(self):
        return self._o.param_groups[0]["lr"]


class Adam(_Opt):
    def __init__(self, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8, parameters=None, weight_decay=None, **kw):
        wd = float(weight_decay) if isinstance(weight_decay, (int, float)) else 0.0
        self._o = torch.optim.Adam(list(parameters), lr=learning_rate, betas=(beta1, beta2), eps=epsilon, weight_decay=wd)


class SGD(_Opt):
    def __init__(self, learning_rate=0.001, parameters=None, weight_decay=None, **kw):
        wd = float(weight_decay) if isinstance(weight_decay, (int, float)) else 0.0
        self._o = torch.optim.SGD(list(parameters), lr=learning_rate, weight_decay=wd)
