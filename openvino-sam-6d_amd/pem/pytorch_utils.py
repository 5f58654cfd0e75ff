"""Parameter containers with the reference's SharedMLP / Conv1d / Conv2d state_dict layout
(PEM/model/pointnet2/pytorch_utils.py:25-206).  The arithmetic runs in libsam6d_hip.so (see fine_point_matching.py)."""
import torch.nn as nn


class _BN(nn.Sequential):
    def __init__(self, n, dims):
        super().__init__()
        self.add_module("bn", (nn.BatchNorm1d if dims == 1 else nn.BatchNorm2d)(n))


class _ConvBlock(nn.Sequential):
    def __init__(self, cin, cout, dims, bn, bias=True):
        super().__init__()
        conv = (nn.Conv1d if dims == 1 else nn.Conv2d)(cin, cout, 1, bias=bias and not bn)
        nn.init.kaiming_normal_(conv.weight)
        if conv.bias is not None:
            nn.init.constant_(conv.bias, 0)
        self.add_module("conv", conv)
        if bn:
            self.add_module("normlayer", _BN(cout, dims))


class Conv1d(_ConvBlock):
    def __init__(self, in_size, out_size, kernel_size=1, activation=None, bn=False, bias=True, **kw):
        super().__init__(in_size, out_size, 1, bool(bn), bias)


class Conv2d(_ConvBlock):
    def __init__(self, in_size, out_size, kernel_size=(1, 1), activation=None, bn=False, bias=True, **kw):
        super().__init__(in_size, out_size, 2, bool(bn), bias)


class SharedMLP(nn.Sequential):
    def __init__(self, args, bn=False, **kw):
        super().__init__()
        for i in range(len(args) - 1):
            self.add_module("layer%d" % i, Conv2d(args[i], args[i + 1], bn=bn))
