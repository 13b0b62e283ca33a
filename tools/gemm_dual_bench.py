"""Micro-benchmark of the dual (value + tangent) MFMA GEMM (developer tool)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "torchmd-net_amd"))
import torch
from torchmdnet_amd import _C
L = _C.lib()
P = 193710
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
SHAPES = [(P, 384, 256, 2)] if os.environ.get("TMDNET_SB_DBG") else None
for (M, N, K, kind) in SHAPES or [(P // 4, 384, 1024, 2), (P // 8, 384, 2048, 2), (P, 128, 256, 2), (P, 384, 256, 2), (P, 256, 128, 1), (P, 128, 32, 1), (P, 384, 32, 0), (671, 384, 256, 2)]:
    A = torch.randn(M, K, device="cuda"); A2 = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.05
    b = torch.randn(N, device="cuda"); rs = torch.rand(M, device="cuda"); rs2 = torch.randn(M, device="cuda")
    C1 = torch.empty(M, N, device="cuda"); C2 = torch.empty(M, N, device="cuda")
    Wh = W.cpu().contiguous()
    nsb = L.tmdnet_debug_split_weight(C.c_void_p(Wh.data_ptr()), N, K, None)
    Wsb_h = torch.empty(nsb, dtype=torch.int16)
    L.tmdnet_debug_split_weight(C.c_void_p(Wh.data_ptr()), N, K, C.c_void_p(Wsb_h.data_ptr()))
    Wsb = Wsb_h.cuda()
    use_sb = os.environ.get("SB", "1") == "1"
    args = (s, p(A), p(A2), p(W), p(b), p(C1), p(C2), M, N, K, kind, p(rs), p(rs2), p(Wsb) if use_sb else None)
    for _ in range(3): L.tmdnet_debug_gemm_dual(*args)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): L.tmdnet_debug_gemm_dual(*args)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    gf = 4.0 * M * N * K / 1e9
    gb = 4.0 * (2 * M * K + N * K + 2 * M * N) / 1e9
    n = min(M, 2048)
    tail = slice(M - 300, M)  # last (partial) row tile
    for sl in (slice(0, n), tail):
        pass
    e = A[:n].double() @ W.double().t() + b.double(); r = A2[:n].double() @ W.double().t()
    sg = torch.sigmoid(e); f = e * sg; df = sg * (1 + e * (1 - sg))
    if kind == 0: r1, r2 = e, r
    elif kind == 1: r1, r2 = f, df * r
    else: r1, r2 = f * rs[:n, None].double(), df * r * rs[:n, None].double() + f * rs2[:n, None].double()
    err = max((C1[:n].double() - r1).abs().max().item() / r1.abs().max().item(), (C2[:n].double() - r2).abs().max().item() / r2.abs().max().item())
    print(f"M={M:7d} N={N:4d} K={K:4d} kind={kind}: {us:8.1f} us  {gf/us*1e3:6.1f} TF/s  {gb/us*1e3:6.2f} TB/s  relerr {err:.1e}", flush=True)
