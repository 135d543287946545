"""paddle.optimizer.Adam(learning_rate, parameters, weight_decay) with .step() / .clear_grad().  Paddle's float
`weight_decay` is L2 regularisation added to the gradient -- torch.optim.Adam's `weight_decay` means the same.
On the accelerator the step counter lives on the DEVICE (torch's `capturable=True`): the same arithmetic, no host-side bookkeeping
per step (measured on the GCN example model at |E| = 20 M: 6.96 -> 6.55 ms per training step, profiles/r06/example_models_training_step.txt)
and the whole step can be captured into a HIP graph."""
import torch as _t


class Adam(_t.optim.Adam):
    def __init__(self, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8, parameters=None, weight_decay=None,
                 grad_clip=None, lazy_mode=False, multi_precision=False, name=None):
        params = list(parameters)
        on_gpu = bool(params) and all(p.is_cuda for p in params)
        _t.optim.Adam.__init__(self, params, lr=float(learning_rate), betas=(beta1, beta2), eps=epsilon,
                               weight_decay=float(weight_decay or 0.0), capturable=on_gpu)

    def clear_grad(self, set_to_zero=True):
        self.zero_grad(set_to_none=True)

    def get_lr(self):
        return self.param_groups[0]["lr"]


class SGD(_t.optim.SGD):
    def __init__(self, learning_rate=0.001, parameters=None, weight_decay=None, grad_clip=None, name=None):
        _t.optim.SGD.__init__(self, list(parameters), lr=float(learning_rate), weight_decay=float(weight_decay or 0.0))

    def clear_grad(self, set_to_zero=True):
        self.zero_grad(set_to_none=True)
