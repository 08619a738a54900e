"""Plain AdaCoF network -- mirror of reference src/adacof/models/adacofnet.py (eval mode: returns frame1)."""
import torch

from ...fusion_net import fusion_adacofnet as _f


def make_model(args):
    return AdaCoFNet(args).to(torch.device("cuda:{}".format(args.gpu_id)))


KernelEstimation = _f.KernelEstimation


class AdaCoFNet(_f.AdaCoFNet):
    def forward(self, frame0, frame2):
        return super().forward(frame0, frame2)[2]      # adacofnet.py:216-219 (eval branch)
