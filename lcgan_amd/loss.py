"""Losses of the LC-GAN step on HIP kernels -- same functions as the reference's loss.py:9-34 plus the two reductions
worker.py applies inline (BCE with logits worker.py:156-157,191; L1 sparsity worker.py:207-209)."""
from __future__ import annotations

import torch
from torch import autograd

from . import ops


def contrastive_loss(anchor, p_sample, n_sample, tau):
    """reference loss.py:9-15 == mean softplus((a.n - a.p) / tau)"""
    return ops.ContrastiveFn.apply(anchor, p_sample, n_sample, float(tau))


def cal_derivative(inputs, outputs, device=None):
    """reference loss.py:27-34 (create_graph=True: the backward graph itself runs on the HIP kernels).  only_inputs=True: the
    parameter gradients of this pass are never formed (ops.inputs_only)."""
    with ops.inputs_only():
        return autograd.grad(outputs=outputs, inputs=inputs, grad_outputs=torch.ones_like(outputs),
                             create_graph=True, retain_graph=True, only_inputs=True)[0]


def cal_r1_reg(adv_output, images, device=None):
    """reference loss.py:18-24: 0.5 * mean_b sum (d sum(logit) / d image)^2.  (`+ images[:,0,0,0].mean()*0` adds zero.)"""
    batch_size = images.size(0)
    grad_dout = cal_derivative(inputs=images, outputs=adv_output.sum(), device=device)
    assert grad_dout.size() == images.size()
    return ops.PowSumFn.apply(grad_dout, 2, 0.5 / batch_size)


def bce_with_logits(logit, target_is_one: bool):
    """F.binary_cross_entropy_with_logits(logit, ones|zeros)  (worker.py:156-157, 168-169, 191, 203)"""
    return ops.BCELogitsFn.apply(logit, bool(target_is_one))


def l1_sparsity(params, weight: float):
    """torch.norm(torch.cat(params), p=1) * weight  (worker.py:207-209)"""
    total = None
    for p in params:
        t = ops.PowSumFn.apply(p.view(-1), 1, float(weight))
        total = t if total is None else total + t
    return total
