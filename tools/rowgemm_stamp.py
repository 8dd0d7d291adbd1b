#!/usr/bin/env python3
"""In-kernel phase timing of csrc/rowgemm.hip (diagnostic build with s_memtime stamps; see RG_STAMP in that file).

    hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -fno-slp-vectorize -DPPT_RG_STAMP -shared ppt_amd/csrc/rowgemm.hip -o tools/_build/librg_stamp.so
    python tools/rowgemm_stamp.py

Prints, for the C2 LayerNorm -> qkv launch, the median cycles per phase and iteration over all waves of each half."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from ppt_amd import _lib

L = ctypes.CDLL(os.path.join(ROOT, "tools", "_build", "librg_stamp.so"))
L.ppt_rowgemm_bf16.restype = ctypes.c_int
L.ppt_rowgemm_bf16.argtypes = [ctypes.POINTER(_lib.RowGemmParams), ctypes.c_void_p]


def run(ln, N, act, M=32 * 513, K=384):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(M, K, generator=g).cuda()
    a16 = x.to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * 0.05).cuda().to(torch.bfloat16)
    b = torch.randn(N).cuda()
    gam, bet = torch.ones(K).cuda(), torch.zeros(K).cuda()
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    stamps = torch.zeros(512 * 8 * 8 * 8, dtype=torch.int64, device="cuda")
    p = _lib.RowGemmParams()
    p.A, p.W, p.C, p.M, p.N, p.K = (x if ln else a16).data_ptr(), w.data_ptr(), out.data_ptr(), M, N, K
    if ln:
        p.a_ln, p.ln_w, p.ln_b, p.ln_eps = 1, gam.data_ptr(), bet.data_ptr(), 1e-5
    p.bias, p.act = b.data_ptr(), act
    p.residual2 = stamps.data_ptr()
    for _ in range(3):
        stamps.zero_()
        rc = L.ppt_rowgemm_bf16(ctypes.byref(p), None)
        assert rc == 0, rc
        torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(512, 8, 8, 8)
    used = s[:, 0, 0, 0] > 0
    s = s[used]
    entry = s[:, :, 7, 7]
    print(f"ln={ln} N={N} act={act}: {used.sum()} workgroups")
    t0 = entry.min()
    print("  kernel entry spread (cycles, all waves): median %d max %d" % (np.median(entry - t0), (entry - t0).max()))
    for half, name in ((slice(0, 4), "waves 0-3 (MFMA | stage | load || epilogue)"), (slice(4, 8), "waves 4-7 (stage | load, MFMA || epilogue)")):
        h = s[:, half]
        print("  " + name)
        print("    it   start-entry  phase1   phase2   (->3)  epilogue  barrier   total")
        for it in range(7):
            v = h[:, :, it, :]
            ok = v[:, :, 5] > 0
            if not ok.any():
                continue
            st = v[:, :, 0] - entry[:, half]
            d = [v[:, :, k + 1] - v[:, :, k] for k in range(5)]
            med = lambda a: int(np.median(a[ok]))
            print("    %d   %10d  %7d  %7d  %6d  %7d  %7d  %7d" % (it, med(st), med(d[0]), med(d[1]), med(d[2]), med(d[3]), med(d[4]),
                                                                 med(v[:, :, 5] - v[:, :, 0])))
        last = np.where(h[:, :, :, 5] > 0, h[:, :, :, 5], 0).max(axis=2) - entry[:, half]
        print("    lifetime (entry -> last barrier): median %d max %d cycles" % (np.median(last), last.max()))


if __name__ == "__main__":
    run(True, 1152, 0)
    run(False, 1152, 0)
    run(True, 1536, 2)
