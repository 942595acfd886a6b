"""SharedMLP and its building blocks -- the subset of the reference's vendored
pvn3d/lib/utils/etw_pytorch_utils/pytorch_utils.py (:25-50 SharedMLP, :53-77 BatchNorm*,
:80-134 _ConvBase, Conv2d) that the set-abstraction / feature-propagation modules use.

The module tree and therefore the ``state_dict`` keys are the reference's
(``layer<i>.conv.weight``, ``layer<i>.normlayer.bn.{weight,bias,running_mean,running_var}``),
so reference checkpoints load unchanged.  A layer is 1x1 Conv2d (bias only without BN) ->
BatchNorm2d -> ReLU.
"""
import torch.nn as nn


class _Norm2d(nn.Sequential):
    def __init__(self, channels):
        super(_Norm2d, self).__init__()
        self.add_module("bn", nn.BatchNorm2d(channels))
        nn.init.constant_(self.bn.weight, 1.0)
        nn.init.constant_(self.bn.bias, 0.0)


BatchNorm2d = _Norm2d


class Conv2d(nn.Sequential):
    def __init__(self, in_size, out_size, kernel_size=(1, 1), stride=(1, 1), padding=(0, 0),
                 activation=None, bn=False, init=nn.init.kaiming_normal_, bias=True,
                 preact=False, name=""):
        super(Conv2d, self).__init__()
        if activation is None:
            activation = nn.ReLU(inplace=True)
        # (evaluating the 1x1 convolution as a batched torch.matmul instead was measured on MI355X:
        # 95.8 vs 65.0 ms per bf16 training step of 24 frames -- MIOpen's tuned solvers win once found)
        conv = nn.Conv2d(in_size, out_size, kernel_size=kernel_size, stride=stride,
                         padding=padding, bias=bias and not bn)
        init(conv.weight)
        if conv.bias is not None:
            nn.init.constant_(conv.bias, 0.0)
        norm = _Norm2d(in_size if preact else out_size) if bn else None
        if preact:
            if norm is not None:
                self.add_module(name + "normlayer", norm)
            if activation is not False:
                self.add_module(name + "activation", activation)
        self.add_module(name + "conv", conv)
        if not preact:
            if norm is not None:
                self.add_module(name + "normlayer", norm)
            if activation is not False:
                self.add_module(name + "activation", activation)


class SharedMLP(nn.Sequential):
    """Per-point MLP: a chain of 1x1 Conv2d(+BN)+ReLU over a (B,C,npoint,nsample) tensor."""

    def __init__(self, args, bn=False, activation=None, preact=False, first=False, name=""):
        super(SharedMLP, self).__init__()
        for i in range(len(args) - 1):
            plain = first and preact and i == 0
            self.add_module(
                name + "layer{}".format(i),
                Conv2d(args[i], args[i + 1], bn=bn and not plain,
                       activation=False if plain else activation, preact=preact))
