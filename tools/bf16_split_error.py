#!/usr/bin/env python
"""Numerical backing for DESIGN.md's round-2 lever (1): error of a GEMM whose fp32 operands are split into 2 or 3 bf16
terms and multiplied on the bf16 matrix cores with fp32 accumulation, against the exact-fp32 MFMA path used today.
CPU-only emulation: products of bf16 values are exact in fp32, the accumulation is done in fp32 like the MFMA does."""
import torch

torch.manual_seed(0)


def split(x, n):
    parts, r = [], x.clone()
    for _ in range(n):
        p = r.to(torch.bfloat16).to(torch.float32)
        parts.append(p)
        r = r - p
    return parts


def gemm_split(a, b, n, terms):
    pa, pb = split(a, n), split(b, n)
    acc = torch.zeros(a.shape[0], b.shape[1], dtype=torch.float32)
    for (i, j) in terms:
        acc = acc + pa[i] @ pb[j]
    return acc


for K in (224, 384, 672):
    a = torch.randn(2048, K) * torch.rand(2048, 1) * 3
    b = torch.randn(K, 128) / K ** 0.5
    ref = a.double() @ b.double()
    scale = ref.abs().max()
    f32 = (a @ b).double()
    x3 = gemm_split(a, b, 2, [(0, 0), (0, 1), (1, 0)]).double()
    x6 = gemm_split(a, b, 3, [(0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (1, 1)]).double()
    print("K=%4d  max|err|/max|ref|:  fp32 %.2e   bf16x3 %.2e   bf16x6 %.2e"
          % (K, (f32 - ref).abs().max() / scale, (x3 - ref).abs().max() / scale, (x6 - ref).abs().max() / scale))
