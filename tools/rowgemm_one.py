#!/usr/bin/env python3
"""A few eager launches of the C2 LayerNorm -> qkv shape on both paths, for rocprofv3 --pmc runs.   python tools/rowgemm_one.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppt_amd import ops

g = torch.Generator().manual_seed(0)
M = 32 * 513
x = torch.randn(M, 384, generator=g).cuda()
gam, bet = torch.ones(384).cuda(), torch.zeros(384).cuda()
for N, act in ((1152, ops.ACT_NONE), (1536, ops.ACT_GELU)):
    w = (torch.randn(N, 384, generator=g) * 0.05).cuda().to(torch.bfloat16)
    b = torch.randn(N, generator=g).cuda()
    for _ in range(6):
        h, _, _ = ops.layernorm_fwd(x, gam, bet, torch.bfloat16)
        ops.gemm(h, w, out_dtype=torch.bfloat16, bias=b, act=act)
        ops.rowgemm(x, w, ln=(gam, bet), bias=b, act=act)
        ops.rowgemm(h, w, bias=b, act=act)
        torch.cuda.synchronize()
a16 = torch.randn(M, 384, generator=g).cuda().to(torch.bfloat16)
w = (torch.randn(384, 384, generator=g) * 0.05).cuda().to(torch.bfloat16)
dp = torch.ones(32).cuda()
for _ in range(6):
    ops.rowgemm(a16, w, bias=gam, residual=x, out=x, row_scale=dp, row_scale_rows=513)
    ops.gemm(a16, w, out=x, bias=gam, row_scale=dp, row_scale_rows=513, residual=x)
    torch.cuda.synchronize()
